// ds_read_b128 bank-conflict rules, second probe: address = (lane & 15) * A + (lane >> 4) * B  (+ optional swizzle), runtime A / B.
//   hipcc -O3 --offload-arch=gfx950 -w -o transfer_em_amd/lib/lds_b128_probe2 tests/tools/lds_b128_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void probe(uint32_t *out, int iters, int A, int B, int mode) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += 512) ((uint32_t *)lds)[i] = i;
  __syncthreads();
  uint32_t a;
  if (mode == 0) a = (lane & 15) * A + (lane >> 4) * B;
  else if (mode == 1) {            // 16-channel layout: voxel v = m + off(kq>>1), chunk h = kq & 1, swizzled; B = voxel offset of the second tap
    const int v = (lane & 15) + ((lane >> 5) ? B : 0), h = (lane >> 4) & 1;
    a = (2 * v + (h ^ ((v >> 3) & 1))) * 16;
  } else if (mode == 2) {          // 8-channel layout: voxel v = m + off(kq), offsets 0, 1, 2, B
    const int kq = lane >> 4, off = kq == 3 ? B : kq;
    a = ((lane & 15) + off) * 16;
  } else {                         // 32-channel layout: voxel m, chunk kq, swizzled (one tap per k-step)
    const int v = (lane & 15) + B, h = lane >> 4;
    a = (4 * v + (h ^ ((v >> 2) & 3))) * 16;
  }
  a += wave * 7168;
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds + a;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      u32x4 v;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(k * 16));
      asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
      acc.x ^= v.x;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  out[blockIdx.x * 512 + threadIdx.x] = acc.x;
}
double run(uint32_t *out, int A, int B, int mode) {
  const int iters = 10000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<<<256, 512, 65536>>>(out, 10, A, B, mode);
  hipEventRecord(e0);
  probe<<<256, 512, 65536>>>(out, iters, A, B, mode);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("mode %d A=%4d B=%5d: %.3f ms (1.0 = %.3f)\n", mode, A, B, ms, ms / 1.13);
  return ms;
}
int main() {
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  uint32_t *out; hipMalloc(&out, 256 * 512 * 4);
  int Bs[] = {0, 16, 32, 48, 64, 128, 256, 272, 288, 320, 1024, 1040, 1056, 1088, 1152};
  for (int b : Bs) run(out, 16, b, 0);
  int Bs2[] = {0, 16, 32, 64, 128, 256, 512, 528, 1040};
  for (int b : Bs2) run(out, 32, b, 0);
  for (int b : {1, 2, 3, 7, 8, 9, 50, 51, 52, 100, 101, 102, 103}) run(out, 0, b, 1);
  for (int b : {3, 50, 51, 52, 100, 126, 127, 128, 129, 130}) run(out, 0, b, 2);
  for (int b : {0, 1, 2, 3, 4, 5}) run(out, 0, b, 3);
  return 0;
}
