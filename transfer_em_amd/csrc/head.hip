// head.hip -- the discriminator's 1x1x1 head as ONE kernel per direction (reference discriminator.py:78-99):
//
//   p1 = LeakyReLU(e6 . W1)   (Conv 1x1x1, 32 -> 32, no bias; e6 = the doubly rectified output of Downsample_3)
//   z  = p1 . w2 + b          (Conv 1x1x1, 32 -> 1, with bias)
//
// and its adjoint: g_p1 = LeakyReLU'(p1) (dz w2), g_e6 = gate(e6) (g_p1 . W1^T), dW2 = sum_v p1 dz, db = sum_v dz,
// dW1 = sum_v e6 (x) g_p1.  The logits map of a 96^3 patch has 8^3 = 512 voxels: as separate convolution / kernel-gradient /
// bias launches (2 forward, 5 backward; 20 us each for a microsecond of work) the head was ~0.3 ms of pure launch latency
// per step on the discriminators' dependent chain.  Here a workgroup owns groups of 8 voxels (thread = (voxel, channel)),
// keeps its W1 column (forward) / row (backward) in registers and writes one kernel-gradient slab per workgroup for the
// ordinary slab reduction (fixed order: bit-reproducible).
#include "tem_common.h"

namespace head {

constexpr int C = 32, VG = 8;          // channels; voxels per group (256 threads)

__global__ __launch_bounds__(256) void head_fwd_k(const float *__restrict__ e6, const float *__restrict__ w1, const float *__restrict__ w2,
                                                  const float *__restrict__ bias, float *__restrict__ p1, float *__restrict__ z,
                                                  int64_t nvox, float slope) {
  const int co = threadIdx.x & 31, vl = threadIdx.x >> 5;
  float w[C];
#pragma unroll
  for (int ci = 0; ci < C; ++ci) w[ci] = w1[ci * C + co];
  const float w2c = w2[co], b = bias ? bias[0] : 0.f;
  for (int64_t v0 = (int64_t)blockIdx.x * VG; v0 < nvox; v0 += (int64_t)gridDim.x * VG) {
    const int64_t v = v0 + vl;
    const bool ok = v < nvox;
    const float *x = e6 + (ok ? v : 0) * C;
    float s = 0.f;
#pragma unroll
    for (int ci = 0; ci < C; ++ci) s = fmaf(x[ci], w[ci], s);      // k-ordered chain, as the convolution kernels
    s = s > 0.f ? s : slope * s;
    if (ok) p1[v * C + co] = s;
    float t = s * w2c;                                                // sum over the voxel's 32 lanes (one half-wave)
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 32);
    if (ok && co == 0) z[v] = t + b;
  }
}

// slab_* may be null (adversarial path: input gradient only).  Slab row blockIdx.x of each layer's slab set.
__global__ __launch_bounds__(256) void head_bwd_k(const float *__restrict__ dz, const float *__restrict__ e6, const float *__restrict__ p1,
                                                  const float *__restrict__ w1, const float *__restrict__ w2, float *__restrict__ g_e6,
                                                  float slope_p1, float slope_e6, float *slab_w1, float *slab_w2, float *slab_b,
                                                  int64_t nvox) {
  __shared__ float sx[VG][C + 1], sg[VG][C + 1], sdz[VG];
  const int c = threadIdx.x & 31, vl = threadIdx.x >> 5;
  float wrow[C];                                                      // W1[ci = c][co]: the lane's row
#pragma unroll
  for (int co = 0; co < C; ++co) wrow[co] = w1[c * C + co];
  const float w2c = w2[c];
  // kernel-gradient partial sums of this workgroup: dW1 entries 4 per thread (ci = t / 8, co = 4 (t % 8) ..), dW2 / db per lane
  const int eci = threadIdx.x >> 3, eco = (threadIdx.x & 7) * 4;
  float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2 = 0.f, ab = 0.f;
  for (int64_t v0 = (int64_t)blockIdx.x * VG; v0 < nvox; v0 += (int64_t)gridDim.x * VG) {
    const int64_t v = v0 + vl;
    const bool ok = v < nvox;
    const float d = ok ? dz[v] : 0.f;
    const float pv = ok ? p1[v * C + c] : 0.f, xv = ok ? e6[v * C + c] : 0.f;
    const float gp = (pv > 0.f ? 1.f : slope_p1) * (d * w2c);       // g_p1[v][co = c]
    a2 = fmaf(pv, d, a2);
    if (c == 0) ab += d;
    __syncthreads();                                                  // the previous group's LDS reads are done
    sx[vl][c] = xv; sg[vl][c] = gp;
    if (c == 0) sdz[vl] = d;
    __syncthreads();
    float s = 0.f;                                                    // g_e6[v][ci = c] = sum_co g_p1[v][co] W1[ci][co]
#pragma unroll
    for (int co = 0; co < C; ++co) s = fmaf(sg[vl][co], wrow[co], s);
    if (ok && g_e6) g_e6[v * C + c] = (xv > 0.f ? 1.f : slope_e6) * s;
    if (slab_w1) {
#pragma unroll
      for (int u = 0; u < VG; ++u) {
        const float xe = sx[u][eci];
#pragma unroll
        for (int k = 0; k < 4; ++k) a1[k] = fmaf(xe, sg[u][eco + k], a1[k]);
      }
    }
  }
  if (!slab_w1) return;
  float *s1 = slab_w1 + (size_t)blockIdx.x * C * C;
#pragma unroll
  for (int k = 0; k < 4; ++k) s1[eci * C + eco + k] = a1[k];
  // dW2[c] and db: sum over the 8 voxel rows of the workgroup, fixed order
  __syncthreads();
  sx[vl][c] = a2;
  if (c == 0) sdz[vl] = ab;
  __syncthreads();
  if (vl == 0) {
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < VG; ++u) t += sx[u][c];
    slab_w2[(size_t)blockIdx.x * C + c] = t;
    if (c == 0) {
      float tb = 0.f;
#pragma unroll
      for (int u = 0; u < VG; ++u) tb += sdz[u];
      slab_b[blockIdx.x] = tb;
    }
  }
}

static int nblocks_for(int64_t nvox) {
  const int64_t groups = (nvox + VG - 1) / VG;
  return (int)(groups < 64 ? (groups < 1 ? 1 : groups) : 64);       // 8 voxel groups per workgroup at 8^3; at most 64 slabs
}

}  // namespace head

extern "C" int tem_disc_head_nslab(int64_t nvox) { return nvox > 0 ? head::nblocks_for(nvox) : TEM_EINVAL; }

extern "C" int tem_disc_head_fwd(const float *e6, const float *w1, const float *w2, const float *bias, float *p1, float *z,
                                 int64_t nvox, float slope, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!e6 || !w1 || !w2 || !p1 || !z || nvox <= 0) return TEM_EINVAL;
  hipLaunchKernelGGL(head::head_fwd_k, dim3(head::nblocks_for(nvox)), dim3(256), 0, (hipStream_t)stream, e6, w1, w2, bias, p1, z, nvox, slope);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_disc_head_bwd(const tem_head_bwd_args *a, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!a || !a->dz || !a->e6 || !a->p1 || !a->w1 || !a->w2 || a->nvox <= 0) return TEM_EINVAL;
  const bool dw = a->slab_w1 != nullptr;
  if (dw && (!a->slab_w2 || !a->slab_b || a->nslab != head::nblocks_for(a->nvox))) return TEM_EINVAL;
  if (!dw && !a->g_e6) return TEM_EINVAL;
  hipLaunchKernelGGL(head::head_bwd_k, dim3(head::nblocks_for(a->nvox)), dim3(256), 0, (hipStream_t)stream, a->dz, a->e6, a->p1, a->w1, a->w2,
                     a->g_e6, a->slope_p1, a->slope_e6, a->slab_w1, a->slab_w2, a->slab_b, a->nvox);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}
