// datapipe.hip -- the input pipeline's per-sample transforms on the device: random
// augmentation (datasets.py:123-155: axis permutation, flips, intensity jitter) and the self-supervision warp
// (debug.py:7-63: 3-wide box blur with zero SAME padding, random holes dilated by a 4-wide box, holes := image
// mean).  Single-channel float32 volumes (D, H, W); 2-D data is D == 1.  All HBM-streaming kernels.
#include "tem_common.h"

namespace {

inline unsigned grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// dst = flip(transpose(src, perm)) * scale + shift.   tf.transpose: dst dim k has the extent of src dim perm[k];
// tf.reverse along every flagged dst dim.
__global__ __launch_bounds__(256) void augment_k(const float *src, int sD, int sH, int sW, int p0, int p1, int p2, int f0,
                                                 int f1, int f2, float scale, float shift, float *dst, int64_t total) {
  const int sdim[3] = {sD, sH, sW};
  const int64_t sstr[3] = {(int64_t)sH * sW, sW, 1};
  const int d0 = sdim[p0], d1 = sdim[p1], d2 = sdim[p2];
  (void)d0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int i2 = (int)(i % d2); int64_t r = i / d2;
    int i1 = (int)(r % d1); int i0 = (int)(r / d1);
    if (f0) i0 = sdim[p0] - 1 - i0;
    if (f1) i1 = d1 - 1 - i1;
    if (f2) i2 = d2 - 1 - i2;
    const int64_t so = i0 * sstr[p0] + i1 * sstr[p1] + i2 * sstr[p2];
    float v = src[so] * scale;                                      // tensor *= var_adj
    asm volatile("" : "+v"(v));                                     // two roundings, as the reference: no fma contraction
    dst[i] = v + shift;                                             // tensor += mean_adj
  }
}

// 3-wide box blur per axis (1 along a unit axis), zero padding, /27 (or /9); block sums of the result -> *sum
__global__ __launch_bounds__(256) void warp_blur_k(const float *src, int D, int H, int W, float *dst, double *sum,
                                                   int64_t total) {
  const int rz = D > 1 ? 1 : 0;
  const float inv = 1.f / (float)((2 * rz + 1) * 9);
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % W); const int64_t r = i / W;
    const int y = (int)(r % H), z = (int)(r / H);
    float acc = 0.f;
    for (int dz = -rz; dz <= rz; ++dz)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int zz = z + dz, yy = y + dy, xx = x + dx;
          const bool ok = (unsigned)zz < (unsigned)D && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
          acc += ok ? src[((int64_t)zz * H + yy) * W + xx] : 0.f;
        }
    const float v = acc * inv;
    dst[i] = v;
    s += (double)v;
  }
  __shared__ double part[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sum, part[0] + part[1] + part[2] + part[3]);
}

// hole seeds: Bernoulli(rate) per voxel from the Philox stream (seed, site 0xD0, step 0): uniform = word / 2^32
__global__ __launch_bounds__(256) void warp_seeds_k(uint8_t *seeds, int64_t total, float rate, uint32_t k0, uint32_t k1) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const Philox128 p = philox4x32_10((uint32_t)(i >> 2), (uint32_t)(i >> 34), 0xD0u, 0u, k0, k1);
    const uint32_t j = (uint32_t)(i & 3);
    const uint32_t w = j == 0 ? p.r[0] : (j == 1 ? p.r[1] : (j == 2 ? p.r[2] : p.r[3]));
    seeds[i] = ((float)w * 2.3283064365386963e-10f) < rate ? 1 : 0;
  }
}

// grown[i] = any seed in [i-1, i+2] per axis (4-wide box, TensorFlow SAME: 1 before, 2 after); holes := mean
__global__ __launch_bounds__(256) void warp_holes_k(float *img, const uint8_t *seeds, int D, int H, int W, const double *sum,
                                                    int64_t total) {
  const float mean = (float)(*sum / (double)total);
  const int z0 = D > 1 ? -1 : 0, z1 = D > 1 ? 2 : 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % W); const int64_t r = i / W;
    const int y = (int)(r % H), z = (int)(r / H);
    bool hole = false;
    for (int dz = z0; dz <= z1; ++dz)
      for (int dy = -1; dy <= 2; ++dy)
        for (int dx = -1; dx <= 2; ++dx) {
          const int zz = z + dz, yy = y + dy, xx = x + dx;
          if ((unsigned)zz < (unsigned)D && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
            hole = hole || seeds[((int64_t)zz * H + yy) * W + xx];
        }
    if (hole) img[i] = mean;
  }
}

}  // namespace

extern "C" int tem_augment_f32(const float *src, int32_t D, int32_t H, int32_t W, int32_t p0, int32_t p1, int32_t p2,
                               int32_t f0, int32_t f1, int32_t f2, float scale, float shift, float *dst,
                               tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!src || !dst || D < 1 || H < 1 || W < 1) return TEM_EINVAL;
  const int seen = (1 << p0) | (1 << p1) | (1 << p2);
  if (p0 < 0 || p0 > 2 || p1 < 0 || p1 > 2 || p2 < 0 || p2 > 2 || seen != 7) return TEM_EINVAL;   // a permutation of (0,1,2)
  if (src == dst) return TEM_EINVAL;
  const int64_t total = (int64_t)D * H * W;
  hipLaunchKernelGGL(augment_k, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, D, H, W, p0, p1, p2, f0,
                     f1, f2, scale, shift, dst, total);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_warp_f32(const float *src, int32_t D, int32_t H, int32_t W, float rate, uint64_t seed, float *dst,
                            uint8_t *seeds, double *sum, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!src || !dst || !seeds || !sum || D < 1 || H < 1 || W < 1 || src == dst) return TEM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)D * H * W;
  hipError_t e = hipMemsetAsync(sum, 0, sizeof(double), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(warp_blur_k, dim3(grid_for(total)), dim3(256), 0, st, src, D, H, W, dst, sum, total);
  TEM_CHECK_LAUNCH();
  hipLaunchKernelGGL(warp_seeds_k, dim3(grid_for(total)), dim3(256), 0, st, seeds, total, rate, (uint32_t)seed,
                     (uint32_t)(seed >> 32));
  TEM_CHECK_LAUNCH();
  hipLaunchKernelGGL(warp_holes_k, dim3(grid_for(total)), dim3(256), 0, st, dst, seeds, D, H, W, sum, total);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}
