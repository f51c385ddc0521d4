// Probe: does LDS-DMA (buffer_load_dwordx4 ... lds, 16 bytes per lane) work from global addresses that are only 4-byte
// aligned?  Each lane fetches 16 bytes from src + shift floats + 4 lane floats; the result is compared with a plain copy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float *src, float *dst, int shift, int nbytes) {
  __shared__ __attribute__((aligned(16))) float lds[256];
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, nbytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds, 16, (shift + 4 * (int)threadIdx.x) * 4, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) dst[i] = lds[i];
}
int main() {
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  float *s, *d; hipMalloc(&s, 4096); hipMalloc(&d, 1024);
  hipMemcpy(s, h.data(), 4096, hipMemcpyHostToDevice);
  for (int shift = 0; shift < 6; ++shift) {
    hipMemset(d, 0, 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, d, shift, 4096);
    std::vector<float> o(256);
    hipMemcpy(o.data(), d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += o[i] != (float)(shift + i);
    printf("shift %d floats (%2d-byte alignment): %s (%d wrong; first values %g %g %g %g %g)\n", shift, (shift % 4) ? 4 * (shift & -shift) : 16,
           bad ? "MISMATCH" : "ok", bad, o[0], o[1], o[2], o[3], o[4]);
  }
  return 0;
}
