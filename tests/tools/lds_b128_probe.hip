// LDS read throughput / bank-conflict probe for ds_read_b128 and ds_read_b64 access patterns (gfx950).
//   hipcc -O3 --offload-arch=gfx950 -o transfer_em_amd/lib/lds_b128_probe tests/tools/lds_b128_probe.hip
// Every wave of a 512-thread workgroup (one per CU) issues ITER x 8 reads with a per-lane address pattern:
//   0: b128, lane*16 (contiguous)          1: b128, lane*32 (stride 32 B)       2: b128, stride 32 B, halves swapped per 8 lanes
//   3: b128, lane*64 (stride 64 B)         4: b128, stride 64 B, chunk ^ (lane>>2)&3
//   5: b64, lane*8 (contiguous)            6: b64, lane*16                      7: b128, (lane&15)*16 + (lane>>4)*1040
//   8: b128 (lane&15)*32+(lane>>4)*16 [4 k-groups interleaved]  9: b128 stride 48 B
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(uint32_t *out, int iters) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += 512) ((uint32_t *)lds)[i] = i;
  __syncthreads();
  uint32_t a;
  if (MODE == 0) a = lane * 16;
  else if (MODE == 1) a = lane * 32;
  else if (MODE == 2) a = (lane * 2 + ((lane >> 3) & 1)) * 16;
  else if (MODE == 3) a = lane * 64;
  else if (MODE == 4) a = (lane * 4 + ((lane >> 2) & 3)) * 16;
  else if (MODE == 5) a = lane * 8;
  else if (MODE == 6) a = lane * 16;
  else if (MODE == 7) a = (lane & 15) * 16 + (lane >> 4) * 1040;
  else if (MODE == 8) a = (lane & 15) * 32 + (lane >> 4) * 16;
  else a = lane * 48;
  a += wave * 4096;
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds + a;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (MODE == 5 || MODE == 6) {
        u32x2 v;
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(k * 8));
        asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        acc.x ^= v.x; 
      } else {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(k * 16));
        asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        acc.x ^= v.x;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  out[blockIdx.x * 512 + threadIdx.x] = acc.x;
}

template <int MODE> double run(uint32_t *out, int iters) {
  hipFuncSetAttribute((const void *)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<256, 512, 65536>>>(out, 10);
  hipEventRecord(e0);
  probe<MODE><<<256, 512, 65536>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (MODE == 5 || MODE == 6 ? 8.0 : 16.0) * 64 * 8 * 8 * iters;     // per CU
  const double cycles = ms * 1e-3 * 2.4e9;
  printf("mode %d: %.3f ms  %.1f B/clk/CU (at 2.4 GHz)\n", MODE, ms, bytes / cycles);
  return ms;
}

int main() {
  uint32_t *out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  run<0>(out, iters); run<1>(out, iters); run<2>(out, iters); run<3>(out, iters); run<4>(out, iters);
  run<5>(out, iters); run<6>(out, iters); run<7>(out, iters); run<8>(out, iters); run<9>(out, iters);
  return 0;
}
