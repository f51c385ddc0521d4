// elementwise.hip -- losses, optimizer and data-format kernels of the CycleGAN step.
// All are HBM-streaming kernels (or tiny); float32 storage, float/double accumulation as noted.
#include "tem_common.h"
#include <cmath>

namespace {

struct V5 {  // device copy of a tem_view
  float *ptr; int32_t N, D, H, W, C; int64_t sN, sD, sH, sW;
};
inline V5 dv(const tem_view &v) { return V5{v.ptr, v.N, v.D, v.H, v.W, v.C, v.sN, v.sD, v.sH, v.sW}; }

__device__ __forceinline__ int64_t voff(const V5 &v, int64_t i, int &c) {
  // i = dense index over (N,D,H,W,C)
  c = (int)(i % v.C); int64_t r = i / v.C;
  int x = (int)(r % v.W); r /= v.W;
  int y = (int)(r % v.H); r /= v.H;
  int z = (int)(r % v.D); int n = (int)(r / v.D);
  return n * v.sN + z * v.sD + y * v.sH + x * v.sW + c;
}

// the same for tensors of fewer than 2^31 elements: 32-bit divisions (a 64-bit division is ~5x the instructions; the loss
// kernels sit on the step's critical chain between the forward and the backward sweeps)
__device__ __forceinline__ int64_t voff32(const V5 &v, uint32_t i, int &c) {
  uint32_t r = i;
  if (v.C == 1) c = 0; else { c = (int)(r % (uint32_t)v.C); r /= (uint32_t)v.C; }
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
  // (base^gamma, d/dbase) with tf.pow's gradient gamma*base^(gamma-1)
  if (gamma == 2.f) { dpow = 2.f * base; return base * base; }
  float v = powf(base, gamma);
  float d = gamma * powf(base, gamma - 1.f);
  dpow = isfinite(d) ? d : 0.f;
  return v;
}

// tfa sigmoid_focal_crossentropy(from_logits=True), alpha = 0.5 (cgan.py:78-79)
__global__ __launch_bounds__(256) void focal_logits_k(V5 z, int target, float gamma, double *losses, uint32_t mask,
                                                      double loss_scale, V5 dz, float grad_scale, int64_t total) {
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int c;
    int64_t o = voff(z, i, c);
    float zz = z.ptr[o];
    float ce = fmaxf(zz, 0.f) - zz * (float)target + log1pf(expf(-fabsf(zz)));
    float pr = 1.f / (1.f + expf(-zz));
    float dce = pr - (float)target;
    float base = target ? 1.f - pr : pr;
    float dbase = target ? -pr * (1.f - pr) : pr * (1.f - pr);
    float dmod, mod = pow_gamma(base, gamma, dmod);
    s += (double)(0.5f * mod * ce);
    if (dz.ptr) {
      int c2;
      dz.ptr[voff(dz, i, c2)] = grad_scale * 0.5f * (dmod * dbase * ce + mod * dce);
    }
  }
  block_accumulate(s, losses, mask, loss_scale);
}

// cgan.py:129-130 / 140-141 with loss_obj_nl (from_logits=False): t = 1 - |a-b|/2
__global__ __launch_bounds__(256) void focal_match_k(V5 a, V5 b, float gamma, double *losses, uint32_t mask,
                                                     double loss_scale, V5 db, float grad_scale, int64_t total) {
  const float eps = 1e-7f, hi = 1.0f - 1e-7f;
  double s = 0.0;
  const bool small = total < ((int64_t)1 << 31);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int c;
    const int64_t oa = small ? voff32(a, (uint32_t)i, c) : voff(a, i, c), ob = small ? voff32(b, (uint32_t)i, c) : voff(b, i, c);
    float av = a.ptr[oa], bv = b.ptr[ob];
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
      db.ptr[small ? voff32(db, (uint32_t)i, c) : voff(db, i, c)] = grad_scale * dper * 0.5f * sg;
    }
  }
  block_accumulate(s, losses, mask, loss_scale);
}

__global__ __launch_bounds__(256) void adam_keras_k(float *theta, const float *grad, float *m, float *v, int64_t n,
                                                    float lr, float b1, float b2, float eps, float gscale,
                                                    const uint32_t *step_dev) {
  // TF's ApplyAdam functor, op for op in float32 (tensorflow/core/kernels/training_ops.cc):
  //   alpha = lr * sqrt(1 - beta2^t) / (1 - beta1^t);  m += (g - m)*(1 - beta1);
  //   v += (g*g - v)*(1 - beta2);  var -= (m*alpha) / (sqrt(v) + epsilon)
  float t = (float)(*step_dev) + 1.f;
  float alpha = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
  float omb1 = 1.f - b1, omb2 = 1.f - b2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float g = gscale * grad[i];
    float mm = m[i] + (g - m[i]) * omb1;
    float vv = v[i] + (g * g - v[i]) * omb2;
    m[i] = mm; v[i] = vv;
    theta[i] = theta[i] - (mm * alpha) / (sqrtf(vv) + eps);
  }
}

__global__ void step_tick_k(uint32_t *s) { *s += 1u; }

__global__ __launch_bounds__(256) void u8_to_f32_std_k(const uint8_t *in, float *out, int64_t n, float mean, float std) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    // explicit _rn ops: no FMA contraction, so the result is bit-identical to the
    // reference's op-by-op float32 arithmetic
    float x = (float)in[i];
    x = __fsub_rn(__fdiv_rn(x, 127.5f), 1.f);          // datasets.py:200
    out[i] = __fdiv_rn(__fsub_rn(x, mean), std);       // datasets.py:161-162
  }
}

__global__ __launch_bounds__(256) void f32_unstd_to_u8_k(V5 y, uint8_t *out, int64_t oD, int64_t oH, int64_t oW,
                                                         float mean, float std, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int x = (int)(i % y.W); int64_t r = i / y.W;
    int yy = (int)(r % y.H); int z = (int)(r / y.H);
    float v = y.ptr[z * y.sD + yy * y.sH + x * y.sW];
    v = __fmul_rn(__fadd_rn(__fadd_rn(__fmul_rn(v, std), mean), 1.f), 127.5f);   // utils.py:109, op by op
    int q = (int)rintf(v);                               // np.around: half to even
    out[z * oD + yy * oH + x * oW] = (uint8_t)(q & 0xFF);  // astype(uint8) wraps
  }
}

__global__ __launch_bounds__(256) void fill_k(float *dst, int64_t n, float value) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = value;
}

template <bool ADD>
__global__ __launch_bounds__(256) void copy_view_k(V5 s, V5 d, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int c;
    int64_t so = voff(s, i, c), dof = voff(d, i, c);
    d.ptr[dof] = ADD ? d.ptr[dof] + s.ptr[so] : s.ptr[so];
  }
}

// g(view) = saved > 0 ? g : slope * g  (LeakyReLU gradient gated on the saved OUTPUT, in place)
__global__ __launch_bounds__(256) void leaky_gate_view_k(V5 g, V5 sv, float slope, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int c;
    int64_t go = voff(g, i, c), so = voff(sv, i, c);
    float v = g.ptr[go];
    g.ptr[go] = sv.ptr[so] > 0.f ? v : slope * v;
  }
}

// ---------------------------------------------------------------- InstanceNormalization (models/utils.py:10-38)
// One workgroup per (n, c): two passes over the channel's voxels (mean, then the mean squared deviation, as
// tf.nn.moments does), double accumulation, fixed summation order.  Channels-last makes a channel a stride-C
// gather, but all C workgroups of a sample walk the same cache lines at the same time, so HBM sees one pass.
__device__ __forceinline__ double block_sum_1024(double s, double *part) {
  s = wave_sum(s);
  __syncthreads();                                        // part[] may still be read from a previous call
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += part[w];
  return t;
}

__device__ __forceinline__ int64_t vox_off(const V5 &v, int64_t i) {   // i over (D,H,W)
  int x = (int)(i % v.W); int64_t r = i / v.W;
  int y = (int)(r % v.H); int z = (int)(r / v.H);
  return z * v.sD + y * v.sH + x * v.sW;
}

__global__ __launch_bounds__(1024) void inorm_stats_k(V5 x, float eps, float *mean_out, float *rstd_out) {
  __shared__ double part[16];
  const int n = blockIdx.x / x.C, c = blockIdx.x % x.C;
  const int64_t vox = (int64_t)x.D * x.H * x.W;
  const float *xp = x.ptr + n * x.sN + c;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < vox; i += blockDim.x) s += (double)xp[vox_off(x, i)];
  const float mean = (float)(block_sum_1024(s, part) / (double)vox);
  s = 0.0;
  for (int64_t i = threadIdx.x; i < vox; i += blockDim.x) {
    const float d = xp[vox_off(x, i)] - mean;
    s += (double)(d * d);
  }
  const float var = (float)(block_sum_1024(s, part) / (double)vox);
  if (threadIdx.x == 0) { mean_out[blockIdx.x] = mean; rstd_out[blockIdx.x] = rsqrtf(var + eps); }
}

// y = scale[c] * ((x - mean[n,c]) * rstd[n,c]) + offset[c]
__global__ __launch_bounds__(256) void inorm_apply_k(V5 x, V5 y, const float *scale, const float *offset, const float *mean,
                                                     const float *rstd, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int c, c2;
    const int64_t xo = voff(x, i, c), yo = voff(y, i, c2);
    const int n = (int)(i / ((int64_t)x.D * x.H * x.W * x.C));
    const float nrm = (x.ptr[xo] - mean[n * x.C + c]) * rstd[n * x.C + c];
    y.ptr[yo] = scale[c] * nrm + offset[c];
  }
}

// per (n, c): s1 = sum dy, s2 = sum dy * xhat  (double), written to sums[(n*C + c)*2 + {0,1}]
__global__ __launch_bounds__(1024) void inorm_bwd_sums_k(V5 x, V5 dy, const float *mean, const float *rstd, double *sums) {
  __shared__ double part[16];
  const int n = blockIdx.x / x.C, c = blockIdx.x % x.C;
  const int64_t vox = (int64_t)x.D * x.H * x.W;
  const float *xp = x.ptr + n * x.sN + c, *gp = dy.ptr + n * dy.sN + c;
  const float mu = mean[blockIdx.x], rs = rstd[blockIdx.x];
  double s1 = 0.0, s2 = 0.0;
  for (int64_t i = threadIdx.x; i < vox; i += blockDim.x) {
    const float g = gp[vox_off(dy, i)], xh = (xp[vox_off(x, i)] - mu) * rs;
    s1 += (double)g; s2 += (double)(g * xh);
  }
  s1 = block_sum_1024(s1, part);
  s2 = block_sum_1024(s2, part);
  if (threadIdx.x == 0) { sums[blockIdx.x * 2] = s1; sums[blockIdx.x * 2 + 1] = s2; }
}

// dx = scale*rstd * (dy - mean(dy) - xhat * mean(dy*xhat));  block 0 also folds the sums over n into dscale / doffset
__global__ __launch_bounds__(256) void inorm_bwd_apply_k(V5 x, V5 dy, V5 dx, const float *scale, const float *mean,
                                                         const float *rstd, const double *sums, float *dscale,
                                                         float *doffset, int64_t total) {
  const double inv = 1.0 / (double)((int64_t)x.D * x.H * x.W);
  if (blockIdx.x == 0 && (int)threadIdx.x < x.C) {
    double a = 0.0, b = 0.0;
    for (int n = 0; n < x.N; ++n) { a += sums[(n * x.C + threadIdx.x) * 2]; b += sums[(n * x.C + threadIdx.x) * 2 + 1]; }
    if (doffset) doffset[threadIdx.x] = (float)a;
    if (dscale) dscale[threadIdx.x] = (float)b;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int c, c2, c3;
    const int64_t xo = voff(x, i, c), go = voff(dy, i, c2), doo = voff(dx, i, c3);
    const int nc = (int)(i / ((int64_t)x.D * x.H * x.W * x.C)) * x.C + c;
    const float xh = (x.ptr[xo] - mean[nc]) * rstd[nc];
    const float m1 = (float)(sums[nc * 2] * inv), m2 = (float)(sums[nc * 2 + 1] * inv);
    dx.ptr[doo] = scale[c] * rstd[nc] * (dy.ptr[go] - m1 - xh * m2);
  }
}

// ---------------------------------------------------------------- tiled inference boundaries (utils.py:77-126)
// gather: tile t = the edge^3 window of the uint8 volume at origins[t] (zeros outside), scaled + standardized
__global__ __launch_bounds__(256) void u8_tiles_to_f32_std_k(const uint8_t *vol, int Z, int Y, int X, const int32_t *origins,
                                                             int edge, float *out, float mean, float std, int64_t per_tile) {
  const int t = blockIdx.y;
  const int oz = origins[3 * t], oy = origins[3 * t + 1], ox = origins[3 * t + 2];
  float *o = out + (int64_t)t * per_tile;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_tile; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % edge); const int64_t r = i / edge;
    const int y = (int)(r % edge), z = (int)(r / edge);
    const int gz = oz + z, gy = oy + y, gx = ox + x;
    const bool in = (unsigned)gz < (unsigned)Z && (unsigned)gy < (unsigned)Y && (unsigned)gx < (unsigned)X;
    float v = in ? (float)vol[((int64_t)gz * Y + gy) * X + gx] : 0.f;
    v = __fsub_rn(__fdiv_rn(v, 127.5f), 1.f);          // datasets.py:200
    o[i] = __fdiv_rn(__fsub_rn(v, mean), std);         // datasets.py:161-162
  }
}

// scatter: the interior (tpad stripped, utils.py:113-116) of tile t of y -> uint8 block of the output volume at index[t]
__global__ __launch_bounds__(256) void f32_tiles_unstd_to_u8_k(const float *y, int yedge, int tpad, const int32_t *index,
                                                               uint8_t *out, int OY, int OX, float mean, float std) {
  const int t = blockIdx.y, od = yedge - 2 * tpad;
  const int iz = index[3 * t], iy = index[3 * t + 1], ix = index[3 * t + 2];
  const float *src = y + (int64_t)t * yedge * yedge * yedge;
  const int64_t per_tile = (int64_t)od * od * od;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_tile; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % od); const int64_t r = i / od;
    const int yy = (int)(r % od), z = (int)(r / od);
    float v = src[((int64_t)(z + tpad) * yedge + (yy + tpad)) * yedge + (x + tpad)];
    v = __fmul_rn(__fadd_rn(__fadd_rn(__fmul_rn(v, std), mean), 1.f), 127.5f);   // utils.py:109, op by op
    const int q = (int)rintf(v);                                                  // np.around: half to even
    out[((int64_t)(iz + z) * OY + (iy + yy)) * OX + (ix + x)] = (uint8_t)(q & 0xFF);   // astype(uint8) wraps
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

extern "C" int tem_focal_logits(const tem_view *z, int32_t target, float gamma, double *losses, uint32_t slot_mask,
                                float loss_scale, const tem_view *dz, float grad_scale, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!z || !tem_view_ok(*z) || (target != 0 && target != 1)) return TEM_EINVAL;
  V5 d{};
  if (dz && dz->ptr) { if (!same_extents(*z, *dz)) return TEM_ESHAPE; d = dv(*dz); }
  int64_t total = vtotal(*z);
  unsigned g = grid_for(total); if (g > 256) g = 256;
  hipLaunchKernelGGL(focal_logits_k, dim3(g), dim3(256), 0, (hipStream_t)stream, dv(*z), target, gamma, losses,
                     slot_mask, (double)loss_scale / (double)total, d, grad_scale / (float)total, total);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_focal_match(const tem_view *a, const tem_view *b, float gamma, double *losses, uint32_t slot_mask,
                               float loss_scale, const tem_view *db, float grad_scale, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!a || !b || !tem_view_ok(*a) || !tem_view_ok(*b)) return TEM_EINVAL;
  if (!same_extents(*a, *b)) return TEM_ESHAPE;
  V5 d{};
  if (db && db->ptr) { if (!same_extents(*b, *db)) return TEM_ESHAPE; d = dv(*db); }
  int64_t total = vtotal(*a);
  // Every workgroup ends in one fp64 atomic per loss slot on the same few addresses (~12 ns each, back to back), every
  // loop iteration of a thread costs a load latency (~1.3 us): the grid that balances the two (was 1024 workgroups:
  // loss.cyc_x 38 us, loss.id_x 22 us)
  unsigned g = grid_for(total);
  {
    const int slots = __builtin_popcount(slot_mask) > 0 ? __builtin_popcount(slot_mask) : 1;
    const double best = sqrt((double)total / 256.0 * 1300.0 / (12.0 * slots));
    const unsigned cap = best < 64 ? 64u : (best > 1024 ? 1024u : (unsigned)best);
    if (g > cap) g = cap;
  }
  hipLaunchKernelGGL(focal_match_k, dim3(g), dim3(256), 0, (hipStream_t)stream, dv(*a), dv(*b), gamma, losses,
                     slot_mask, (double)loss_scale / (double)total, d, grad_scale / (float)total, total);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_instance_norm(const tem_view *x, const float *scale, const float *offset, float eps,
                                 const tem_view *y, float *mean, float *rstd, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!x || !y || !tem_view_ok(*x) || !tem_view_ok(*y) || !scale || !offset || !mean || !rstd) return TEM_EINVAL;
  if (!same_extents(*x, *y)) return TEM_ESHAPE;
  if (x->C > 256) return TEM_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(inorm_stats_k, dim3(x->N * x->C), dim3(1024), 0, st, dv(*x), eps, mean, rstd);
  TEM_CHECK_LAUNCH();
  const int64_t total = vtotal(*x);
  hipLaunchKernelGGL(inorm_apply_k, dim3(grid_for(total)), dim3(256), 0, st, dv(*x), dv(*y), scale, offset, mean, rstd, total);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_instance_norm_bwd(const tem_view *x, const tem_view *dy, const float *scale, const float *mean,
                                     const float *rstd, const tem_view *dx, float *dscale, float *doffset,
                                     double *workspace, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!x || !dy || !dx || !tem_view_ok(*x) || !tem_view_ok(*dy) || !tem_view_ok(*dx) || !scale || !mean || !rstd ||
      !workspace)
    return TEM_EINVAL;
  if (!same_extents(*x, *dy) || !same_extents(*x, *dx)) return TEM_ESHAPE;
  if (x->C > 256) return TEM_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(inorm_bwd_sums_k, dim3(x->N * x->C), dim3(1024), 0, st, dv(*x), dv(*dy), mean, rstd, workspace);
  TEM_CHECK_LAUNCH();
  const int64_t total = vtotal(*x);
  hipLaunchKernelGGL(inorm_bwd_apply_k, dim3(grid_for(total)), dim3(256), 0, st, dv(*x), dv(*dy), dv(*dx), scale, mean,
                     rstd, workspace, dscale, doffset, total);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_u8_tiles_to_f32_std(const uint8_t *vol, int32_t Z, int32_t Y, int32_t X, const int32_t *origins_dev,
                                       int32_t ntile, int32_t edge, float *out, float mean, float std,
                                       tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!vol || !origins_dev || !out || Z < 1 || Y < 1 || X < 1 || ntile < 0 || edge < 1) return TEM_EINVAL;
  if (ntile == 0) return TEM_OK;
  if (ntile > 65535) return TEM_EUNSUPPORTED;
  const int64_t per_tile = (int64_t)edge * edge * edge;
  unsigned gx = grid_for(per_tile); if (gx > 512) gx = 512;
  hipLaunchKernelGGL(u8_tiles_to_f32_std_k, dim3(gx, (unsigned)ntile), dim3(256), 0, (hipStream_t)stream, vol, Z, Y, X,
                     origins_dev, edge, out, mean, std, per_tile);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_f32_tiles_unstd_to_u8(const float *y, int32_t ntile, int32_t yedge, int32_t tpad,
                                         const int32_t *index_dev, uint8_t *out, int32_t OZ, int32_t OY, int32_t OX,
                                         float mean, float std, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!y || !index_dev || !out || ntile < 0 || yedge < 1 || tpad < 0 || 2 * tpad >= yedge || OZ < 1 || OY < 1 || OX < 1)
    return TEM_EINVAL;
  if (ntile == 0) return TEM_OK;
  if (ntile > 65535) return TEM_EUNSUPPORTED;
  const int od = yedge - 2 * tpad;
  unsigned gx = grid_for((int64_t)od * od * od); if (gx > 512) gx = 512;
  hipLaunchKernelGGL(f32_tiles_unstd_to_u8_k, dim3(gx, (unsigned)ntile), dim3(256), 0, (hipStream_t)stream, y, yedge, tpad,
                     index_dev, out, OY, OX, mean, std);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_adam_keras(float *theta, const float *grad, float *m, float *v, int64_t n, float lr, float beta1,
                              float beta2, float eps, float grad_scale, const uint32_t *step_dev,
                              tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!theta || !grad || !m || !v || !step_dev || n < 0) return TEM_EINVAL;
  if (n == 0) return TEM_OK;
  hipLaunchKernelGGL(adam_keras_k, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, theta, grad, m, v, n, lr,
                     beta1, beta2, eps, grad_scale, step_dev);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

// Keep bits of up to two Dropout layers for one step: mask bit e = DropoutStream::bit of dense element e, i.e. the 16
// bytes of 128-element block b ARE the four Philox words of that block (little-endian) -- one 16-byte store per thread.
__global__ __launch_bounds__(256) void dropout_masks_k(uint4 *m0, int64_t nblk0, uint32_t site0, uint4 *m1, int64_t nblk1,
                                                       uint32_t site1, uint32_t k0, uint32_t k1, const uint32_t *step_dev,
                                                       uint32_t step) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nblk0 + nblk1) return;
  const bool second = i >= nblk0;
  const int64_t b = second ? i - nblk0 : i;
  DropoutStream ds{k0, k1, second ? site1 : site0, step_dev ? *step_dev : step};
  const Philox128 ph = ds.block((uint64_t)b);
  (second ? m1 : m0)[b] = make_uint4(ph.r[0], ph.r[1], ph.r[2], ph.r[3]);
}

extern "C" int tem_dropout_masks(uint8_t *mask0, int64_t nbytes0, uint32_t site0, uint8_t *mask1, int64_t nbytes1,
                                 uint32_t site1, uint64_t seed, const uint32_t *step_dev, uint32_t step,
                                 tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!mask0 || nbytes0 <= 0 || nbytes0 % 16 || ((uintptr_t)mask0 & 15)) return TEM_EINVAL;
  if (mask1 && (nbytes1 <= 0 || nbytes1 % 16 || ((uintptr_t)mask1 & 15))) return TEM_EINVAL;
  const int64_t n0 = nbytes0 / 16, n1 = mask1 ? nbytes1 / 16 : 0;
  hipLaunchKernelGGL(dropout_masks_k, dim3((unsigned)((n0 + n1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (uint4 *)mask0, n0, site0, (uint4 *)mask1, n1, site1, (uint32_t)seed, (uint32_t)(seed >> 32), step_dev, step);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_step_tick(uint32_t *step_dev, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!step_dev) return TEM_EINVAL;
  hipLaunchKernelGGL(step_tick_k, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_u8_to_f32_std(const uint8_t *in, float *out, int64_t n, float mean, float std,
                                 tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!in || !out || n < 0) return TEM_EINVAL;
  if (n == 0) return TEM_OK;
  hipLaunchKernelGGL(u8_to_f32_std_k, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, in, out, n, mean, std);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_f32_unstd_to_u8(const tem_view *y, uint8_t *out, int64_t oD, int64_t oH, int64_t oW, float mean,
                                   float std, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!y || !tem_view_ok(*y) || !out) return TEM_EINVAL;
  int64_t total = (int64_t)y->D * y->H * y->W;
  hipLaunchKernelGGL(f32_unstd_to_u8_k, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dv(*y), out, oD, oH,
                     oW, mean, std, total);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_fill_f32(float *dst, int64_t n, float value, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!dst || n < 0) return TEM_EINVAL;
  if (n == 0) return TEM_OK;
  hipLaunchKernelGGL(fill_k, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dst, n, value);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_copy_view(const tem_view *src, const tem_view *dst, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!src || !dst || !tem_view_ok(*src) || !tem_view_ok(*dst)) return TEM_EINVAL;
  if (!same_extents(*src, *dst)) return TEM_ESHAPE;
  int64_t total = vtotal(*src);
  hipLaunchKernelGGL(copy_view_k<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dv(*src), dv(*dst),
                     total);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_add_view(const tem_view *src, const tem_view *dst, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!src || !dst || !tem_view_ok(*src) || !tem_view_ok(*dst)) return TEM_EINVAL;
  if (!same_extents(*src, *dst)) return TEM_ESHAPE;
  int64_t total = vtotal(*src);
  hipLaunchKernelGGL(copy_view_k<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dv(*src), dv(*dst),
                     total);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

// theta_t = every conv kernel of a flat parameter vector with its taps reversed and (ci, co) transposed:
//   theta_t[off + ((ntap-1-t)*co + b)*ci + a] = theta[off + (t*ci + a)*co + b];  elements outside the table are copied.
__global__ __launch_bounds__(256) void flip_transpose_k(const float *__restrict__ theta, float *__restrict__ theta_t,
                                                        const tem_wlayer *__restrict__ layers, int nlayers, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t dst = i;
    for (int l = 0; l < nlayers; ++l) {
      const tem_wlayer L = layers[l];
      const int64_t rel = i - L.offset, n = (int64_t)L.ntap * L.ci * L.co;
      if (rel >= 0 && rel < n) {
        const int b = (int)(rel % L.co);
        const int64_t ta = rel / L.co;
        const int a = (int)(ta % L.ci), t = (int)(ta / L.ci);
        dst = L.offset + ((int64_t)(L.ntap - 1 - t) * L.co + b) * L.ci + a;
        break;
      }
    }
    theta_t[dst] = theta[i];
  }
}

extern "C" int tem_flip_transpose(const float *theta, float *theta_t, const tem_wlayer *layers_dev, int32_t nlayers,
                                  int64_t total, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!theta || !theta_t || theta == theta_t || nlayers < 0 || (nlayers > 0 && !layers_dev) || total < 0) return TEM_EINVAL;
  if (total == 0) return TEM_OK;
  hipLaunchKernelGGL(flip_transpose_k, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, theta, theta_t, layers_dev,
                     nlayers, total);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_leaky_gate_view(const tem_view *g, const tem_view *saved, float slope, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!g || !saved || !tem_view_ok(*g) || !tem_view_ok(*saved)) return TEM_EINVAL;
  if (!same_extents(*g, *saved)) return TEM_ESHAPE;
  int64_t total = vtotal(*g);
  hipLaunchKernelGGL(leaky_gate_view_k, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dv(*g), dv(*saved), slope,
                     total);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_abi_version(const char **arch) {
  TEM_CLEAR_ERR();
  if (arch) *arch = "gfx950";
  return TEM_ABI_VERSION;
}
