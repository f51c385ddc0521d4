// Probe (perf triage, not part of the library): achievable HBM read rate of the z-marching access pattern of the
// one-channel kernels -- a workgroup owns a patch of ROWS x 512 bytes of every plane of a [D][H][W][8] float tensor and
// walks along z with DEPTH 16-byte loads per lane in flight -- against a linear stream of the same bytes.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/stream_probe tests/tools/stream_probe.hip && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int DEPTH>
__global__ __launch_bounds__(256) void march_k(const float4 *g, float *out, int D, int H, int W, int zper, int nxb, int nyb) {
  int b = blockIdx.x;
  const int zseg = b % ((D + zper - 1) / zper); b /= ((D + zper - 1) / zper);
  const int bx = b % nxb, by = b / nxb;
  const int tid = threadIdx.x, cq = tid & 1, vox = tid >> 1, ly = vox >> 4, lx = vox & 15;
  const int oy = by * 8 + ly, ox = bx * 16 + lx;
  const bool ok = oy < H && ox < W;
  const size_t plane = (size_t)H * W * 2;
  const float4 *p = g + ((size_t)(ok ? oy : 0) * W + (ok ? ox : 0)) * 2 + cq;
  const int z0 = zseg * zper, z1 = min(D, z0 + zper);
  float4 r[DEPTH];
  float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
  for (int k = 0; k < DEPTH; ++k) r[k] = (z0 + k < z1) ? p[(size_t)(z0 + k) * plane] : make_float4(0, 0, 0, 0);
  for (int z = z0; z < z1; z += DEPTH) {
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
      const float4 v = r[k];
      r[k] = (z + k + DEPTH < z1) ? p[(size_t)(z + k + DEPTH) * plane] : make_float4(0, 0, 0, 0);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[blockIdx.x] = acc.x;
}

__global__ __launch_bounds__(256) void linear_k(const float4 *g, float *out, size_t n4, int per) {
  size_t i = ((size_t)blockIdx.x * per) * 256 + threadIdx.x;
  float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll 8
  for (int k = 0; k < per; ++k, i += 256)
    if (i < n4) { const float4 v = g[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[blockIdx.x] = acc.x;
}

template <typename F> static float time_us(F f, int n = 20) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < n; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / n;
}

int main() {
  const int D = 130, H = 130, W = 130;
  const size_t n4 = (size_t)D * H * W * 2;
  float4 *g; float *out;
  hipMalloc(&g, n4 * 16); hipMalloc(&out, 1 << 20);
  hipMemset(g, 0, n4 * 16);
  const double mb = n4 * 16 / 1e6;
  for (int per : {4, 16, 64}) {
    const int nb = (int)((n4 + (size_t)256 * per - 1) / ((size_t)256 * per));
    const float us = time_us([&] { hipLaunchKernelGGL(linear_k, dim3(nb), dim3(256), 0, 0, g, out, n4, per); });
    printf("linear  per-thread %3d x 16 B, %6d blocks: %7.1f us  %7.1f GB/s\n", per, nb, us, mb / us * 1e3);
  }
  const int nxb = (W + 15) / 16, nyb = (H + 7) / 8;
  for (int zper : {130, 44, 22, 11}) {
    const int zsegs = (D + zper - 1) / zper, nb = nxb * nyb * zsegs;
    float us = time_us([&] { hipLaunchKernelGGL(march_k<4>, dim3(nb), dim3(256), 0, 0, g, out, D, H, W, zper, nxb, nyb); });
    printf("march depth  4 zper %3d, %5d blocks: %7.1f us  %7.1f GB/s\n", zper, nb, us, mb / us * 1e3);
    us = time_us([&] { hipLaunchKernelGGL(march_k<8>, dim3(nb), dim3(256), 0, 0, g, out, D, H, W, zper, nxb, nyb); });
    printf("march depth  8 zper %3d, %5d blocks: %7.1f us  %7.1f GB/s\n", zper, nb, us, mb / us * 1e3);
    us = time_us([&] { hipLaunchKernelGGL(march_k<16>, dim3(nb), dim3(256), 0, 0, g, out, D, H, W, zper, nxb, nyb); });
    printf("march depth 16 zper %3d, %5d blocks: %7.1f us  %7.1f GB/s\n", zper, nb, us, mb / us * 1e3);
  }
  return 0;
}
