// Probe 2 (perf triage): [4 ds_read2_b32 -> s_waitcnt lgkmcnt(0) -> NM MFMAs on 3 accumulators] per round, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int NM, bool WAITEACH>
__global__ __launch_bounds__(256) void probe_k(float *out, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 1e-3f;
  __syncthreads();
  f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const unsigned base = (threadIdx.x & 63) * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f2 a, b, c, d;
      asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:64" : "=v"(a) : "v"(base + 256 * r));
      asm volatile("ds_read2_b32 %0, %1 offset0:1 offset1:65" : "=v"(b) : "v"(base + 256 * r));
      asm volatile("ds_read2_b32 %0, %1 offset0:2 offset1:66" : "=v"(c) : "v"(base + 256 * r));
      asm volatile("ds_read2_b32 %0, %1 offset0:3 offset1:67" : "=v"(d) : "v"(base + 256 * r));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const float x = (m & 1) ? ((m & 2) ? a.x : b.x) : ((m & 2) ? c.x : a.y), y = (m % 3 == 0) ? d.x : d.y;
        acc[m % 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[m % 3], 0, 0, 0);
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2];
}

template <int NM> static void run(float *out, int wps) {
  const int iters = 4000, nb = 256 * wps;
  hipLaunchKernelGGL((probe_k<NM, false>), dim3(nb), dim3(256), 0, 0, out, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe_k<NM, false>), dim3(nb), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ns_per_mfma_simd = ms * 1e6 / ((double)iters * 4 * NM * wps);
  printf("NM %2d per group, %d waves/SIMD: %6.2f ns per MFMA per SIMD (32 cycles at 2.1 GHz = 15.2 ns)\n", NM, wps, ns_per_mfma_simd);
}

int main() {
  float *out; hipMalloc(&out, 8 << 20);
  for (int w = 1; w <= 4; ++w) { run<6>(out, w); run<4>(out, w); run<12>(out, w); }
  return 0;
}
