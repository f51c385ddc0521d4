// conv_bww.hip -- kernel-gradient ("backprop-filter") contraction on the fp32 matrix cores.
//
//   dW[(tap,ci)][co] = sum over output voxels o of  X[o*s + tap - p][ci] * G[o][co]
//
// is a GEMM with M = ntap*C_in rows, N = C_out columns and K = all output voxels.  With
// v_mfma_f32_16x16x4_f32 the A fragment of one k-step is a [4 voxels][16 rows] patch and the
// B fragment a [4 voxels][16 co] patch -- both are exactly the channels-last layout the
// activations already have in HBM, so lanes read them with plain coalesced dword loads (16
// consecutive channels = 64 B per 16-lane group), re-used across the overlapping taps through
// L1/L2.  fp32 MFMA is an exact k-ordered fmaf chain, so the result is plain fp32 arithmetic.
//
// Decomposition: blockIdx.y picks MT*16 consecutive rows of the (tap,ci) space, blockIdx.x a
// share of the output rows (n,oz,oy); the 4 waves of a block interleave over that share and
// are summed through LDS, so each block writes ONE deterministic partial slab.
// tem_reduce_slabs finishes the K reduction (no float atomics: bitwise reproducible).
#include "tem_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct BwwDev {
  const float *in0, *in1;
  int64_t i0N, i0D, i0H, i0W, i1N, i1D, i1H, i1W;
  int32_t C0, CI, N, D, H, W;
  const float *g;
  int64_t gN, gD, gH, gW;
  int32_t CO, OD, OH, OW;
  int32_t kd, kh, kw, sd, sh, sw, pd, ph, pw;
  int32_t rows;            // ntap * CI
  float *slabs;
  int64_t slab_stride;
  int32_t nslab, accumulate;
  int64_t nrows_out;       // N*OD*OH
};

template <int MT, int NT>
__global__ __launch_bounds__(256) void bww_mfma_k(BwwDev p) {
  __shared__ float red[4][MT * NT * 4 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, kq = lane >> 4;
  const int slab = blockIdx.x, grp = blockIdx.y;

  // per-lane description of the MT rows this lane feeds (row = (tap, ci))
  int a_dz[MT], a_dy[MT], a_dx[MT], a_ci[MT];
  bool a_ok[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int row = (grp * MT + mt) * 16 + m;
    a_ok[mt] = row < p.rows;
    int tap = row / p.CI;
    a_ci[mt] = row - tap * p.CI;
    a_dx[mt] = tap % p.kw; tap /= p.kw;
    a_dy[mt] = tap % p.kh;
    a_dz[mt] = tap / p.kh;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this block's share of output rows, interleaved over the 4 waves
  int64_t per = (p.nrows_out + p.nslab - 1) / p.nslab;
  int64_t r0 = (int64_t)slab * per, r1 = r0 + per < p.nrows_out ? r0 + per : p.nrows_out;

  for (int64_t r = r0 + wave; r < r1; r += 4) {
    int oy = (int)(r % p.OH); int64_t t = r / p.OH;
    int oz = (int)(t % p.OD); int n = (int)(t / p.OD);
    const float *aptr[MT];
    bool okzy[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      int iz = oz * p.sd + a_dz[mt] - p.pd, iy = oy * p.sh + a_dy[mt] - p.ph;
      okzy[mt] = a_ok[mt] && iz >= 0 && iz < p.D && iy >= 0 && iy < p.H;
      bool first = a_ci[mt] < p.C0;
      const float *src = first ? p.in0 + a_ci[mt] : p.in1 + (a_ci[mt] - p.C0);
      int64_t sN = first ? p.i0N : p.i1N, sD = first ? p.i0D : p.i1D, sH = first ? p.i0H : p.i1H;
      aptr[mt] = src + n * sN + iz * sD + iy * sH;
    }
    const float *gptr = p.g + n * p.gN + oz * p.gD + oy * p.gH + m;

#pragma unroll 2
    for (int x0 = 0; x0 < p.OW; x0 += 4) {
      int ox = x0 + kq;
      bool okx = ox < p.OW;
      float b[NT], a[MT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        bool ok = okx && (16 * nt + m) < p.CO;
        b[nt] = ok ? gptr[ox * p.gW + 16 * nt] : 0.f;
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        int ix = ox * p.sw + a_dx[mt] - p.pw;
        bool ok = okzy[mt] && okx && ix >= 0 && ix < p.W;
        int64_t sW = a_ci[mt] < p.C0 ? p.i0W : p.i1W;
        a[mt] = ok ? aptr[mt][ix * sW] : 0.f;
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
    }
  }

  // sum the 4 waves through LDS; thread t then owns element (tile, reg, lane) = t-strided
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wave][((mt * NT + nt) * 4 + j) * 64 + lane] = acc[mt][nt][j];
  __syncthreads();
  float *slabp = p.slabs + (int64_t)slab * p.slab_stride;
  for (int e = threadIdx.x; e < MT * NT * 256; e += 256) {
    float v = red[0][e] + red[1][e] + red[2][e] + red[3][e];
    int l = e & 63, j = (e >> 6) & 3, tile = e >> 8;
    int nt = tile % NT, mt = tile / NT;
    int row = (grp * MT + mt) * 16 + (l >> 4) * 4 + j;     // C/D map: row = 4*(lane>>4)+reg
    int co = nt * 16 + (l & 15);                            //          col = lane&15
    if (row < p.rows && co < p.CO) {
      float *dst = slabp + (int64_t)row * p.CO + co;
      *dst = p.accumulate ? *dst + v : v;
    }
  }
}

__global__ __launch_bounds__(256) void reduce_slabs_k(const float *slabs, int nslab, int64_t n, int64_t stride,
                                                      float *out, int accumulate, float scale) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < nslab; ++k) s += slabs[k * stride + i];
  s *= scale;
  out[i] = accumulate ? out[i] + s : s;
}

// One launch finishes the split-K sums of ALL layers of a network: block b reduces items[b]
// (<= 64 consecutive kernel entries of one layer) over that layer's slabs.  Thread = (entry QUAD tid & 15, slab lane
// tid >> 4): a lane reads 16 bytes of a slab row, 16 lanes stride the slabs with 8 loads in flight each, then the lanes are
// summed through LDS in a fixed order (bitwise reproducible).  [Round 2: 4-byte loads, 4 slab lanes -- with the 750 slabs
// a generator's 32-channel layers collect over three calls a lane walked 24 dependent round trips: 76 us per network
// and 0.3 ms per step for 170 MB that stream in 35 us.]  Items whose rows are not 16-byte aligned (the bias: one float)
// take the scalar form.
__global__ __launch_bounds__(256) void reduce_multi_k(const tem_reduce_item *items, float scale) {
  const tem_reduce_item it = items[blockIdx.x];
  __shared__ float part[16][64];
  const bool vec = (it.count & 3) == 0 && (it.stride & 3) == 0 && (((uintptr_t)it.slabs) & 15) == 0;
  if (vec) {
    const int q = threadIdx.x & 15, sl = threadIdx.x >> 4;
    float4 a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (4 * q < it.count) {
      const float *p = it.slabs + 4 * q;
      int s = sl;
      for (; s + 112 < it.nslab; s += 128) {                 // 8 loads in flight per lane
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float4 v = *reinterpret_cast<const float4 *>(p + (int64_t)(s + 16 * k) * it.stride);
          a[k].x += v.x; a[k].y += v.y; a[k].z += v.z; a[k].w += v.w;
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (s + 16 * k < it.nslab) {
          const float4 v = *reinterpret_cast<const float4 *>(p + (int64_t)(s + 16 * k) * it.stride);
          a[k].x += v.x; a[k].y += v.y; a[k].z += v.z; a[k].w += v.w;
        }
      }
    }
#pragma unroll
    for (int k = 4; k > 0; k >>= 1)
#pragma unroll
      for (int j = 0; j < k; ++j) { a[j].x += a[j + k].x; a[j].y += a[j + k].y; a[j].z += a[j + k].z; a[j].w += a[j + k].w; }
    *reinterpret_cast<float4 *>(&part[sl][4 * q]) = a[0];
    __syncthreads();
    if (threadIdx.x < it.count) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) v += part[k][threadIdx.x];
      it.out[threadIdx.x] = v * scale;
    }
    return;
  }
  const int pi = threadIdx.x & 63, sl = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (pi < it.count) {
    const float *p = it.slabs + pi;
    int s = sl;
    for (; s + 12 < it.nslab; s += 16) {
      s0 += p[(int64_t)s * it.stride];
      s1 += p[(int64_t)(s + 4) * it.stride];
      s2 += p[(int64_t)(s + 8) * it.stride];
      s3 += p[(int64_t)(s + 12) * it.stride];
    }
    for (; s < it.nslab; s += 4) s0 += p[(int64_t)s * it.stride];
  }
  part[sl][pi] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && pi < it.count) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) v += part[k][pi];
    it.out[pi] = v * scale;
  }
}

__global__ __launch_bounds__(256) void channel_sum_k(const float *g, int64_t sN, int64_t sD, int64_t sH, int64_t sW,
                                                     int N, int D, int H, int W, int C, float *out, int accumulate) {
  // one block per channel; tensors here are tiny (discriminator logits)
  int c = blockIdx.x;
  int64_t total = (int64_t)N * D * H * W;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < total; i += 256) {
    int x = (int)(i % W); int64_t r = i / W;
    int y = (int)(r % H); r /= H;
    int z = (int)(r % D); int n = (int)(r / D);
    s += (double)g[n * sN + z * sD + y * sH + x * sW + c];
  }
  __shared__ double part[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = (float)(part[0] + part[1] + part[2] + part[3]);
    out[c] = accumulate ? out[c] + v : v;
  }
}

template <int MT, int NT>
int launch_bww(const BwwDev &p, hipStream_t st) {
  int ngrp = (p.rows + MT * 16 - 1) / (MT * 16);
  hipLaunchKernelGGL((bww_mfma_k<MT, NT>), dim3(p.nslab, ngrp), dim3(256), 0, st, p);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

}  // namespace

int tem_bww_lds_try(const tem_bww_args *a, hipStream_t st, bool dry, int *nslab_out);   // bww_lds.hip
int tem_bww_c1_try(const tem_bww_args *a, hipStream_t st, bool dry, int *nslab_out);    // bww_c1.hip (C_in = 1 layers)
int tem_bww_s2_try(const tem_bww_args *a, hipStream_t st, bool dry, int *nslab_out);    // bww_s2.hip (k4 s2 layers)

extern "C" int tem_conv_bwd_weight_nslab(const tem_bww_args *a) {
  if (!a || !tem_view_ok(a->in0) || !tem_view_ok(a->dout) || a->nslab < 1) return TEM_EINVAL;
  int n = 0;
  if (tem_bww_c1_try(a, nullptr, true, &n) == TEM_OK) return n;
  if (tem_bww_s2_try(a, nullptr, true, &n) == TEM_OK) return n;
  if (tem_bww_lds_try(a, nullptr, true, &n) == TEM_OK) return n;
  return a->nslab < 32 ? a->nslab : 32;          // global-load kernel: any split works, 32 is plenty
}

extern "C" int tem_conv_bwd_weight(const tem_bww_args *a, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!a || !tem_view_ok(a->in0) || !tem_view_ok(a->dout) || !a->slabs || a->nslab < 1) return TEM_EINVAL;
  if (a->in0.N != a->dout.N) return TEM_ESHAPE;
  if (!a->accumulate) {
    // the tiled kernel writes exactly tem_conv_bwd_weight_nslab(a) slabs; take it only when the
    // caller sized the workspace with that query (nslab equal), otherwise slabs would be left stale
    int n = 0;
    {
      // (the C_in = 1 kernel sizes its z runs by the slab budget: ask with the caller's own count as the budget)
      if (tem_bww_c1_try(a, nullptr, true, &n) == TEM_OK && n == a->nslab) {
        const int rc = tem_bww_c1_try(a, (hipStream_t)stream, false, nullptr);
        if (rc != TEM_EUNSUPPORTED) return rc;
      }
    }
    // a tiled kernel that declines at launch time (a check only the launch can make, e.g. the slab alignment) hands the
    // call on to the next kernel instead of failing it
    if (tem_bww_s2_try(a, nullptr, true, &n) == TEM_OK && n == a->nslab) {
      const int rc = tem_bww_s2_try(a, (hipStream_t)stream, false, nullptr);
      if (rc != TEM_EUNSUPPORTED) return rc;
    }
    if (tem_bww_lds_try(a, nullptr, true, &n) == TEM_OK && n == a->nslab) {
      const int rc = tem_bww_lds_try(a, (hipStream_t)stream, false, nullptr);
      if (rc != TEM_EUNSUPPORTED) return rc;
    }
  }
  BwwDev p{};
  const tem_view &i0 = a->in0, &g = a->dout;
  p.in0 = i0.ptr; p.i0N = i0.sN; p.i0D = i0.sD; p.i0H = i0.sH; p.i0W = i0.sW;
  p.C0 = i0.C; p.CI = i0.C; p.N = i0.N; p.D = i0.D; p.H = i0.H; p.W = i0.W;
  p.in1 = i0.ptr; p.i1N = i0.sN; p.i1D = i0.sD; p.i1H = i0.sH; p.i1W = i0.sW;
  if (a->in1.ptr) {
    const tem_view &i1 = a->in1;
    if (i1.N != i0.N || i1.D != i0.D || i1.H != i0.H || i1.W != i0.W) return TEM_ESHAPE;
    p.in1 = i1.ptr; p.i1N = i1.sN; p.i1D = i1.sD; p.i1H = i1.sH; p.i1W = i1.sW;
    p.CI += i1.C;
  }
  if (g.N != i0.N) return TEM_ESHAPE;
  p.g = g.ptr; p.gN = g.sN; p.gD = g.sD; p.gH = g.sH; p.gW = g.sW;
  p.CO = g.C; p.OD = g.D; p.OH = g.H; p.OW = g.W;
  p.kd = a->kd; p.kh = a->kh; p.kw = a->kw; p.sd = a->sd; p.sh = a->sh; p.sw = a->sw;
  p.pd = a->pd; p.ph = a->ph; p.pw = a->pw;
  p.rows = a->kd * a->kh * a->kw * p.CI;
  p.slabs = a->slabs; p.nslab = a->nslab; p.accumulate = a->accumulate;
  p.slab_stride = a->slab_stride ? a->slab_stride : (int64_t)p.rows * p.CO;
  p.nrows_out = (int64_t)g.N * g.D * g.H;
  if (p.CO > 32) return TEM_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const bool wide = p.CO > 16;
  // rows per block-y: keep whole taps together where C_in allows (better L1 re-use of X)
  if (p.CI >= 32) return wide ? launch_bww<6, 2>(p, st) : launch_bww<6, 1>(p, st);
  if (p.CI >= 16) return wide ? launch_bww<3, 2>(p, st) : launch_bww<3, 1>(p, st);
  return wide ? launch_bww<2, 2>(p, st) : launch_bww<2, 1>(p, st);
}

extern "C" int tem_reduce_slabs(const float *slabs, int32_t nslab, int64_t n, int64_t slab_stride, float *out,
                                int32_t accumulate, float scale, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!slabs || !out || nslab < 1 || n < 0) return TEM_EINVAL;
  if (n == 0) return TEM_OK;
  hipLaunchKernelGGL(reduce_slabs_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, slabs,
                     nslab, n, slab_stride ? slab_stride : n, out, accumulate, scale);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_channel_sum(const tem_view *g, float *out, int32_t accumulate, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!g || !tem_view_ok(*g) || !out) return TEM_EINVAL;
  hipLaunchKernelGGL(channel_sum_k, dim3(g->C), dim3(256), 0, (hipStream_t)stream, g->ptr, g->sN, g->sD, g->sH,
                     g->sW, g->N, g->D, g->H, g->W, g->C, out, accumulate);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

extern "C" int tem_reduce_slabs_multi(const tem_reduce_item *items_dev, int32_t nitems, float scale,
                                      tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!items_dev || nitems < 0) return TEM_EINVAL;
  if (nitems == 0) return TEM_OK;
  hipLaunchKernelGGL(reduce_multi_k, dim3((unsigned)nitems), dim3(256), 0, (hipStream_t)stream, items_dev, scale);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}
