// Throughput of ds_read_b64_tr_b16 (the transposing LDS read) beside ds_read_b64 / ds_read_b128, 8 waves per CU, for the access
// pattern of bww_bf16_k's fragments: lane = (voxel 4 g4 + q, channel quad pq), voxel pitch PITCH bytes.
//   hipcc -O3 --offload-arch=gfx950 -w -o transfer_em_amd/lib/lds_tr_probe tests/tools/lds_tr_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
template <int mode>
__global__ __launch_bounds__(512) void probe(uint32_t *out, int iters, int pitch) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += 512) ((uint32_t *)lds)[i] = i;
  __syncthreads();
  const int m = lane & 15, g4 = lane >> 4, q = m >> 2, pq = m & 3;
  uint32_t a = (uint32_t)((4 * g4 + q) * pitch + pq * 8 + wave * 4096);
  if (mode == 2) a = lane * 16 + wave * 4096;                 // b128 contiguous reference
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds + a;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    if (mode == 0) {                                           // (builtin: 8 reads in flight, then consumed)
      s16x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(uintptr_t)(base + k * 512));
#pragma unroll
      for (int k = 0; k < 8; ++k) acc ^= (uint32_t)v[k].x;
      continue;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (mode == 1) {
        u32x2 v;
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(k * 512));
        asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        acc ^= v.x;
      } else {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(k * 16));
        asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        acc ^= v.x;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  out[blockIdx.x * 512 + threadIdx.x] = acc;
}
template <int mode> void run(uint32_t *out, int pitch) {
  const int iters = 10000;
  hipFuncSetAttribute((const void *)probe<mode>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<mode><<<256, 512, 65536>>>(out, 10, pitch);
  hipEventRecord(e0);
  probe<mode><<<256, 512, 65536>>>(out, iters, pitch);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (mode == 2 ? 16.0 : 8.0) * 64 * 8 * 8 * iters;
  printf("%s voxel pitch %2d B: %.3f ms  %.1f B/clk/CU (at 2.4 GHz)\n", mode == 0 ? "ds_read_b64_tr_b16" : mode == 1 ? "ds_read_b64       " : "ds_read_b128      ", pitch, ms, bytes / (ms * 1e-3 * 2.4e9));
}
int main() {
  uint32_t *out; hipMalloc(&out, 256 * 512 * 4);
  for (int pitch : {16, 24, 32, 40, 64, 72}) run<0>(out, pitch);
  for (int pitch : {16, 24, 32, 40, 64, 72}) run<1>(out, pitch);
  run<2>(out, 16);
  return 0;
}
