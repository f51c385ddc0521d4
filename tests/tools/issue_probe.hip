// Probe (perf triage): do VALU instructions hide under v_mfma_f32_16x16x4_f32 on one SIMD?  One wave per SIMD (256 threads
// per CU-sized block), loop of NM MFMAs interleaved with NV independent VALU ops of a kind; cycles per iteration by s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NM, int NV, int KIND>   // KIND 0: v_add_u32, 1: v_fma_f32, 2: v_pk_fma_f32, 3: v_pk_add_f32, 4: ds_read_b32
__global__ __launch_bounds__(256) void probe_k(float *out, unsigned long long *cyc, int iters) {
  __shared__ float lds[1024];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  float v[8]; unsigned u[8];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p2[8];
  for (int i = 0; i < 8; ++i) { v[i] = i + a; u[i] = i + threadIdx.x; p2[i] = f2{v[i], v[i] + 1}; }
  const f2 c2 = {1.0001f, 0.9999f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int m = 0; m < NM; ++m) acc[(r * NM + m) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[(r * NM + m) & 3], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int j = (r * NV + k) & 7;
        if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
        if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(b), "v"(a));
        if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p2[j]) : "v"(c2));
        if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p2[j]) : "v"(c2));
        if (KIND == 4) asm volatile("ds_read_b32 %0, %1" : "=v"(v[j]) : "v"((unsigned)((threadIdx.x * 4 + 64 * j) & 4095)));
      }
    }
    if (KIND == 4) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1];
  for (int i = 0; i < 8; ++i) s += v[i] + (float)u[i] + p2[i].x + p2[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NM, int NV, int KIND> static void run(const char *what, float *out, unsigned long long *cyc, int waves_per_simd) {
  const int iters = 2000, nb = 256 * waves_per_simd;
  hipLaunchKernelGGL((probe_k<NM, NV, KIND>), dim3(nb), dim3(256), 0, 0, out, cyc, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe_k<NM, NV, KIND>), dim3(nb), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  // per "round" = NM MFMAs + NV VALU ops (8 rounds per iteration)
  printf("%-34s waves/SIMD %d: %7.1f ticks/round (s_memtime), %8.1f ns/round wall  [MFMA alone would be %d x 32 cycles]\n", what,
         waves_per_simd, (double)h[0] / iters / 8, ms * 1e6 / iters / 8 / waves_per_simd, NM);
}

int main() {
  float *out; unsigned long long *cyc;
  hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 1 << 16);
  for (int w = 1; w <= 2; ++w) {
    run<4, 0, 0>("4 mfma", out, cyc, w);
    run<0, 16, 0>("16 v_add_u32", out, cyc, w);
    run<4, 16, 0>("4 mfma + 16 v_add_u32", out, cyc, w);
    run<4, 32, 0>("4 mfma + 32 v_add_u32", out, cyc, w);
    run<0, 16, 1>("16 v_fma_f32", out, cyc, w);
    run<4, 16, 1>("4 mfma + 16 v_fma_f32", out, cyc, w);
    run<0, 16, 2>("16 v_pk_fma_f32", out, cyc, w);
    run<4, 16, 2>("4 mfma + 16 v_pk_fma_f32", out, cyc, w);
    run<0, 16, 3>("16 v_pk_add_f32", out, cyc, w);
    run<4, 16, 3>("4 mfma + 16 v_pk_add_f32", out, cyc, w);
    run<0, 16, 4>("16 ds_read_b32", out, cyc, w);
    run<4, 16, 4>("4 mfma + 16 ds_read_b32", out, cyc, w);
  }
  return 0;
}
