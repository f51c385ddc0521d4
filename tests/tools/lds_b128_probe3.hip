// ds_read_b128 conflicts of conv3_bf16_k's B-fragment reads, model-free: for every k-step of the 8- and 16-channel kernels
// and every ring row length RW (mod 16 voxels) / slot length (mod 16 voxels), time the read pattern
//   address = (lane & 15) * A + goff[lane >> 4]        (A = 16 / 32 bytes per voxel, goff = tap and channel-half offset)
// and print the relative cost summed over the k-steps.  Picks the row / slot padding of the kernel.
//   hipcc -O3 --offload-arch=gfx950 -w -o transfer_em_amd/lib/lds_b128_probe3 tests/tools/lds_b128_probe3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct G { int a, g[4]; };
__global__ __launch_bounds__(512) void probe(uint32_t *out, int iters, G q) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t a = (lane & 15) * q.a + q.g[lane >> 4] + wave * 1024;
  a &= 0xfff0u; if (a > 65536 - 256) a -= 32768;
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds + a;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      u32x4 v;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(k * 256));
      asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
      acc.x ^= v.x;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  out[blockIdx.x * 512 + threadIdx.x] = acc.x;
}
static uint32_t *out;
double run(G q) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  probe<<<256, 512, 65536>>>(out, iters, q);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms;
}
int main() {
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipMalloc(&out, 256 * 512 * 4);
  G base{16, {0, 256, 512, 768}};
  run(base);
  const double t0 = run(base);
  printf("reference (contiguous 1 KB): %.3f ms\n", t0);
  for (int CI = 8; CI <= 16; CI += 8) {
    const int NSTEP = (27 * CI + 31) / 32;
    printf("CI = %d: relative read cost summed over the %d k-steps (1.0 = conflict-free); rows: RW mod 16, columns: slot voxels mod 16 = 0, 4, 8, 12\n", CI, NSTEP);
    for (int r = 0; r < 16; ++r) {
      printf("  RW%%16=%2d:", r);
      for (int sm = 0; sm < 16; sm += 4) {
        const int RW = 64 + r, SBv = 1024 + sm;
        double tot = 0;
        for (int s = 0; s < NSTEP; ++s) {
          G q; q.a = CI * 2;
          for (int kq = 0; kq < 4; ++kq) {
            const int e0 = 32 * s + 8 * kq;
            int tap = e0 / CI; if (tap > 26) tap = 26;
            const int h = (e0 % CI) >> 3;
            const int dz = tap / 9, dy = (tap - 9 * dz) / 3, dx = tap - 9 * dz - 3 * dy;
            q.g[kq] = ((dz * SBv + dy * RW + dx) * (CI / 8) + h) * 16;
          }
          tot += run(q) / t0;
        }
        printf(" %5.2f", tot / NSTEP);
      }
      printf("\n");
    }
  }
  return 0;
}
