// elementwise.hip -- losses, optimizer and data-format kernels of the CycleGAN step.
// All are HBM-streaming kernels (or tiny); float32 storage, float/double accumulation as noted.
#include "tem_common.h"

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
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int c;
    float av = a.ptr[voff(a, i, c)], bv = b.ptr[voff(b, i, c)];
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
      db.ptr[voff(db, i, c)] = grad_scale * dper * 0.5f * sg;
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
  unsigned g = grid_for(total); if (g > 1024) g = 1024;
  hipLaunchKernelGGL(focal_match_k, dim3(g), dim3(256), 0, (hipStream_t)stream, dv(*a), dv(*b), gamma, losses,
                     slot_mask, (double)loss_scale / (double)total, d, grad_scale / (float)total, total);
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
