// elementwise_bf16.hip -- the non-convolution kernels of the bf16 mixed-precision mode (BASELINE config 5): casts,
// the per-step bf16 kernel copies, losses on bf16 activations (fp32 arithmetic, double loss sums, bf16 gradients),
// view copies / adds and the bias gradient.  All HBM-streaming or tiny.
#include "tem_common.h"

namespace {

typedef unsigned short u16;
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }   // round to nearest even

struct V5h {  // device copy of a tem_view over bf16 elements
  u16 *ptr; int32_t N, D, H, W, C; int64_t sN, sD, sH, sW;
};
inline V5h dvh(const tem_view &v) { return V5h{reinterpret_cast<u16 *>(v.ptr), v.N, v.D, v.H, v.W, v.C, v.sN, v.sD, v.sH, v.sW}; }

__device__ __forceinline__ int64_t voff(const V5h &v, int64_t i) {
  int c = (int)(i % v.C); int64_t r = i / v.C;
  int x = (int)(r % v.W); r /= v.W;
  int y = (int)(r % v.H); r /= v.H;
  int z = (int)(r % v.D); int n = (int)(r / v.D);
  return n * v.sN + z * v.sD + y * v.sH + x * v.sW + c;
}

// the same for tensors of fewer than 2^31 elements: 32-bit divisions (the loss kernels sit on the step's critical chain)
__device__ __forceinline__ int64_t voff32(const V5h &v, uint32_t i) {
  uint32_t r = i, c = 0;
  if (v.C != 1) { c = r % (uint32_t)v.C; r /= (uint32_t)v.C; }
  const uint32_t x = r % (uint32_t)v.W; r /= (uint32_t)v.W;
  const uint32_t y = r % (uint32_t)v.H; r /= (uint32_t)v.H;
  const uint32_t z = r % (uint32_t)v.D, n = r / (uint32_t)v.D;
  return (int64_t)n * v.sN + (int64_t)z * v.sD + (int64_t)y * v.sH + (int64_t)x * v.sW + c;
}

__device__ __forceinline__ void block_accumulate(double s, double *losses, uint32_t mask, double scale) {
  __shared__ double part[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0 && losses) {
    double t = (part[0] + part[1] + part[2] + part[3]) * scale;
    for (int k = 0; k < 8; ++k)
      if (mask & (1u << k)) atomicAdd(&losses[k], t);
  }
}

__device__ __forceinline__ float pow_gamma(float base, float gamma, float &dpow) {
  if (gamma == 2.f) { dpow = 2.f * base; return base * base; }
  float v = powf(base, gamma);
  float d = gamma * powf(base, gamma - 1.f);
  dpow = isfinite(d) ? d : 0.f;
  return v;
}

__global__ __launch_bounds__(256) void cast_k(const float *src, u16 *dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = f2bf(src[i]);
}

// theta_h = bf16(theta) (same layout); theta_ht: every listed kernel with its last two axes transposed,
//   theta_ht[off + (t*B + b)*A + a] = bf16(theta[off + (t*A + a)*B + b])      (stored dims [ntap][A][B])
__global__ __launch_bounds__(256) void pack_weights_k(const float *theta, u16 *th, u16 *tht, const tem_wlayer *layers,
                                                      int nlayers, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const u16 v = f2bf(theta[i]);
    th[i] = v;
    int64_t dst = i;
    for (int l = 0; l < nlayers; ++l) {
      const tem_wlayer L = layers[l];
      const int64_t sz = (int64_t)L.ntap * L.ci * L.co;
      if (i >= L.offset && i < L.offset + sz) {
        const int64_t e = i - L.offset;
        const int b = (int)(e % L.co); const int64_t r = e / L.co;
        const int a = (int)(r % L.ci); const int t = (int)(r / L.ci);
        dst = L.offset + ((int64_t)t * L.co + b) * L.ci + a;
        break;
      }
    }
    tht[dst] = v;
  }
}

__global__ __launch_bounds__(256) void focal_logits_h_k(V5h z, int target, float gamma, double *losses, uint32_t mask,
                                                        double loss_scale, V5h dz, float grad_scale, int64_t total) {
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    float zz = bf2f(z.ptr[voff(z, i)]);
    float ce = fmaxf(zz, 0.f) - zz * (float)target + log1pf(expf(-fabsf(zz)));
    float pr = 1.f / (1.f + expf(-zz));
    float dce = pr - (float)target;
    float base = target ? 1.f - pr : pr;
    float dbase = target ? -pr * (1.f - pr) : pr * (1.f - pr);
    float dmod, mod = pow_gamma(base, gamma, dmod);
    s += (double)(0.5f * mod * ce);
    if (dz.ptr) dz.ptr[voff(dz, i)] = f2bf(grad_scale * 0.5f * (dmod * dbase * ce + mod * dce));
  }
  block_accumulate(s, losses, mask, loss_scale);
}

__global__ __launch_bounds__(256) void focal_match_h_k(V5h a, V5h b, float gamma, double *losses, uint32_t mask,
                                                       double loss_scale, V5h db, float grad_scale, int64_t total) {
  const float eps = 1e-7f, hi = 1.0f - 1e-7f;
  double s = 0.0;
  const bool small = total < ((int64_t)1 << 31);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    float av = bf2f(a.ptr[small ? voff32(a, (uint32_t)i) : voff(a, i)]), bv = bf2f(b.ptr[small ? voff32(b, (uint32_t)i) : voff(b, i)]);
    float diff = av - bv;
    float t = 1.f - fabsf(diff) * 0.5f;
    float tc = fminf(fmaxf(t, eps), hi);
    float ce = -logf(tc + eps);
    bool inside = t >= eps && t <= hi;
    float dce = inside ? -1.f / (tc + eps) : 0.f;
    float dmod, mod = pow_gamma(1.f - t, gamma, dmod);
    s += (double)(0.5f * mod * ce);
    if (db.ptr) {
      float dper = 0.5f * (-dmod * ce + mod * dce);
      float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
      db.ptr[small ? voff32(db, (uint32_t)i) : voff(db, i)] = f2bf(grad_scale * dper * 0.5f * sg);
    }
  }
  block_accumulate(s, losses, mask, loss_scale);
}

template <bool ADD>
__global__ __launch_bounds__(256) void copy_view_h_k(V5h s, V5h d, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t so = voff(s, i), dof = voff(d, i);
    d.ptr[dof] = ADD ? f2bf(bf2f(d.ptr[dof]) + bf2f(s.ptr[so])) : s.ptr[so];
  }
}

__global__ __launch_bounds__(256) void channel_sum_h_k(V5h g, float *out, int accumulate) {
  const int c = blockIdx.x;
  const int64_t total = (int64_t)g.N * g.D * g.H * g.W;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < total; i += 256) {
    int x = (int)(i % g.W); int64_t r = i / g.W;
    int y = (int)(r % g.H); r /= g.H;
    int z = (int)(r % g.D); int n = (int)(r / g.D);
    s += (double)bf2f(g.ptr[n * g.sN + z * g.sD + y * g.sH + x * g.sW + c]);
  }
  __shared__ double part[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = (float)(part[0] + part[1] + part[2] + part[3]);
    out[c] = accumulate ? out[c] + v : v;
  }
}

inline unsigned grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
inline int64_t vtotal(const tem_view &v) { return (int64_t)v.N * v.D * v.H * v.W * v.C; }
inline bool same_extents(const tem_view &a, const tem_view &b) {
  return a.N == b.N && a.D == b.D && a.H == b.H && a.W == b.W && a.C == b.C;
}

}  // namespace

extern "C" int tem_cast_f32_to_bf16(const float *src, void *dst, int64_t n, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!src || !dst || n < 0) return TEM_EINVAL;
  if (n == 0) return TEM_OK;
  hipLaunchKernelGGL(cast_k, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src, (u16 *)dst, n);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_pack_weights_bf16(const float *theta, void *theta_h, void *theta_ht, const tem_wlayer *layers_dev,
                                     int32_t nlayers, int64_t total, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!theta || !theta_h || !theta_ht || (nlayers > 0 && !layers_dev) || total < 0) return TEM_EINVAL;
  if (total == 0) return TEM_OK;
  hipLaunchKernelGGL(pack_weights_k, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, theta, (u16 *)theta_h,
                     (u16 *)theta_ht, layers_dev, nlayers, total);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_focal_logits_bf16(const tem_view *z, int32_t target, float gamma, double *losses, uint32_t slot_mask,
                                     float loss_scale, const tem_view *dz, float grad_scale, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!z || !tem_view_ok(*z) || (target != 0 && target != 1)) return TEM_EINVAL;
  V5h d{};
  if (dz && dz->ptr) { if (!same_extents(*z, *dz)) return TEM_ESHAPE; d = dvh(*dz); }
  const int64_t total = vtotal(*z);
  unsigned g = grid_for(total); if (g > 256) g = 256;
  hipLaunchKernelGGL(focal_logits_h_k, dim3(g), dim3(256), 0, (hipStream_t)stream, dvh(*z), target, gamma, losses, slot_mask,
                     (double)loss_scale / (double)total, d, grad_scale / (float)total, total);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_focal_match_bf16(const tem_view *a, const tem_view *b, float gamma, double *losses, uint32_t slot_mask,
                                    float loss_scale, const tem_view *db, float grad_scale, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!a || !b || !tem_view_ok(*a) || !tem_view_ok(*b)) return TEM_EINVAL;
  if (!same_extents(*a, *b)) return TEM_ESHAPE;
  V5h d{};
  if (db && db->ptr) { if (!same_extents(*b, *db)) return TEM_ESHAPE; d = dvh(*db); }
  const int64_t total = vtotal(*a);
  // (grid balanced between the serialized fp64 atomics of the workgroups' ends and the load latency per loop iteration, as in
  // tem_focal_match: 1024 workgroups took 36 us for the 60^3 cycle loss)
  unsigned g = grid_for(total);
  {
    const int slots = __builtin_popcount(slot_mask) > 0 ? __builtin_popcount(slot_mask) : 1;
    const double best = sqrt((double)total / 256.0 * 1300.0 / (12.0 * slots));
    const unsigned cap = best < 64 ? 64u : (best > 1024 ? 1024u : (unsigned)best);
    if (g > cap) g = cap;
  }
  hipLaunchKernelGGL(focal_match_h_k, dim3(g), dim3(256), 0, (hipStream_t)stream, dvh(*a), dvh(*b), gamma, losses, slot_mask,
                     (double)loss_scale / (double)total, d, grad_scale / (float)total, total);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_copy_view_bf16(const tem_view *src, const tem_view *dst, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!src || !dst || !tem_view_ok(*src) || !tem_view_ok(*dst)) return TEM_EINVAL;
  if (!same_extents(*src, *dst)) return TEM_ESHAPE;
  const int64_t total = vtotal(*src);
  hipLaunchKernelGGL(copy_view_h_k<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dvh(*src), dvh(*dst), total);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_add_view_bf16(const tem_view *src, const tem_view *dst, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!src || !dst || !tem_view_ok(*src) || !tem_view_ok(*dst)) return TEM_EINVAL;
  if (!same_extents(*src, *dst)) return TEM_ESHAPE;
  const int64_t total = vtotal(*src);
  hipLaunchKernelGGL(copy_view_h_k<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dvh(*src), dvh(*dst), total);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_channel_sum_bf16(const tem_view *g, float *out, int32_t accumulate, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!g || !tem_view_ok(*g) || !out) return TEM_EINVAL;
  hipLaunchKernelGGL(channel_sum_h_k, dim3(g->C), dim3(256), 0, (hipStream_t)stream, dvh(*g), out, accumulate);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}
