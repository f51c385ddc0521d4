// conv_direct.hip -- shape-generic direct convolution kernels (VALU, one output voxel per
// lane, weights through the scalar path).  They cover every (C_in, C_out, k, stride, pad)
// the hot path uses and are the fallback for geometries the LDS/MFMA-tiled kernels in
// conv_mfma.hip do not take.  NDHWC float32; see include/tem_hip.h for the contract.
//
// Data movement: lane i of a wave owns output voxel x0+i, so the C_in floats a tap needs
// are contiguous per lane and the wave reads one contiguous stride-C_in span of the input
// row (coalesced dwordx4 loads, re-used through L1/L2 by the k^3 taps); the kernel taps are
// wave-uniform and come through s_load.  Bounds (padding / crop) are handled by predicated
// loads, never by divergent control flow.
#include "tem_common.h"
#include <cstdio>
#include <cstdlib>

namespace {

struct ConvDev {
  const float *in0, *in1;
  int64_t i0N, i0D, i0H, i0W, i1N, i1D, i1H, i1W;
  int32_t N, D, H, W;              // input extents
  const float *w;
  int32_t kd, kh, kw, sd, sh, sw, pd, ph, pw;
  float *out0, *out1;
  int64_t o0N, o0D, o0H, o0W, o1N, o1D, o1H, o1W;
  int32_t OD, OH, OW;
  int64_t total;                   // N*OD*OH*OW (conv) or per-class count (convT)
  EpilogueDev ep;
};

template <int C>
__device__ __forceinline__ void load_vec(float (&v)[C], const float *p, bool ok) {
  if constexpr (C % 4 == 0) {
#pragma unroll
    for (int i = 0; i < C; i += 4) {
      float4 t = ok ? *reinterpret_cast<const float4 *>(p + i) : make_float4(0.f, 0.f, 0.f, 0.f);
      v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < C; ++i) v[i] = ok ? p[i] : 0.f;
  }
}

template <int C>
__device__ __forceinline__ void store_vec(float *p, const float (&v)[C]) {
  if constexpr (C % 4 == 0) {
#pragma unroll
    for (int i = 0; i < C; i += 4) *reinterpret_cast<float4 *>(p + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < C; ++i) p[i] = v[i];
  }
}

// acc[co] += sum_ci xv[ci] * W(ci, co); wt points at the tap's C_in x C_out block.
// COCI: block is stored [co][ci] (TEM_W_FLIP_CO_CI / transposed-conv layout), else [ci][co].
// The accumulator is kept as PAIRS so that the compiler can issue v_pk_fma_f32 (two fp32 FMAs per
// lane-instruction) with a pair of ADJACENT scalar-loaded weights as one operand:
//   [ci][co] layout: the pair is (co, co+1) of one ci            -> acc pair = two output channels;
//   [co][ci] layout: the pair is (ci, ci+1) of one co            -> acc pair = even-ci / odd-ci partial
//                                                                   sums of ONE output channel (ACC2).
template <int CIP, int CI, int CO, bool COCI>
__device__ __forceinline__ void fma_block(float (&acc)[CO], const float (&xv)[CIP], const float *wt, int ci_base) {
#pragma unroll
  for (int ci = 0; ci < CIP; ++ci) {
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      float wv = COCI ? wt[co * CI + ci_base + ci] : wt[(ci_base + ci) * CO + co];
      acc[co] = fmaf(xv[ci], wv, acc[co]);
    }
  }
}

// [co][ci] layout with split accumulators: acc2[2*co] sums the even input channels, acc2[2*co+1] the odd ones
template <int CIP, int CI, int CO>
__device__ __forceinline__ void fma_block_split(float (&acc2)[2 * CO], const float (&xv)[CIP], const float *wt, int ci_base) {
  static_assert(CIP % 2 == 0, "pairs of input channels");
#pragma unroll
  for (int co = 0; co < CO; ++co) {
#pragma unroll
    for (int ci = 0; ci < CIP; ci += 2) {
      acc2[2 * co] = fmaf(xv[ci], wt[co * CI + ci_base + ci], acc2[2 * co]);
      acc2[2 * co + 1] = fmaf(xv[ci + 1], wt[co * CI + ci_base + ci + 1], acc2[2 * co + 1]);
    }
  }
}

template <int CO0, int CO1>
__device__ __forceinline__ void finish(const ConvDev &p, float (&acc)[CO0 + CO1], int n, int z, int y, int x) {
  float v0[CO0];
#pragma unroll
  for (int i = 0; i < CO0; ++i) v0[i] = acc[i];
  apply_epilogue<CO0>(p.ep, v0, n, z, y, x, 0, p.OD, p.OH, p.OW, CO0);
  store_vec<CO0>(p.out0 + n * p.o0N + z * p.o0D + y * p.o0H + x * p.o0W, v0);
  if constexpr (CO1 > 0) {
    float v1[CO1];
#pragma unroll
    for (int i = 0; i < CO1; ++i) v1[i] = acc[CO0 + i];
    store_vec<CO1>(p.out1 + n * p.o1N + z * p.o1D + y * p.o1H + x * p.o1W, v1);
  }
}

// ------------------------------------------------------------------ conv (gather form)
template <int CI0, int CI1, int CO0, int CO1, bool FLIP>
__global__ __launch_bounds__(256) void conv_direct_k(ConvDev p) {
  constexpr int CI = CI0 + CI1, CO = CO0 + CO1;
  constexpr bool SPLIT = FLIP && CI0 % 2 == 0 && CI1 % 2 == 0;     // see fma_block_split
  constexpr int NACC = SPLIT ? 2 * CO : CO;
  int64_t idx = (int64_t)xcd_contiguous_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
  if (idx >= p.total) return;
  int x = (int)(idx % p.OW); int64_t r = idx / p.OW;
  int y = (int)(r % p.OH); r /= p.OH;
  int z = (int)(r % p.OD); int n = (int)(r / p.OD);

  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.f;

  const int ntap = p.kd * p.kh * p.kw;
  int tap = 0;
  for (int dz = 0; dz < p.kd; ++dz) {
    int iz = z * p.sd + dz - p.pd;
    bool okz = iz >= 0 && iz < p.D;
    for (int dy = 0; dy < p.kh; ++dy) {
      int iy = y * p.sh + dy - p.ph;
      bool oky = okz && iy >= 0 && iy < p.H;
      for (int dx = 0; dx < p.kw; ++dx, ++tap) {
        int ix = x * p.sw + dx - p.pw;
        bool ok = oky && ix >= 0 && ix < p.W;
        const float *wt = p.w + (int64_t)(FLIP ? ntap - 1 - tap : tap) * (CI * CO);
        float xv[CI0];
        load_vec<CI0>(xv, p.in0 + n * p.i0N + iz * p.i0D + iy * p.i0H + ix * p.i0W, ok);
        if constexpr (SPLIT) fma_block_split<CI0, CI, CO>(acc, xv, wt, 0);
        else fma_block<CI0, CI, CO, FLIP>(acc, xv, wt, 0);
        if constexpr (CI1 > 0) {
          float xw[CI1];
          load_vec<CI1>(xw, p.in1 + n * p.i1N + iz * p.i1D + iy * p.i1H + ix * p.i1W, ok);
          if constexpr (SPLIT) fma_block_split<CI1, CI, CO>(acc, xw, wt, CI0);
          else fma_block<CI1, CI, CO, FLIP>(acc, xw, wt, CI0);
        }
      }
    }
  }
  if constexpr (SPLIT) {
    float out[CO];
#pragma unroll
    for (int i = 0; i < CO; ++i) out[i] = acc[2 * i] + acc[2 * i + 1];
    finish<CO0, CO1>(p, out, n, z, y, x);
  } else {
    finish<CO0, CO1>(p, acc, n, z, y, x);
  }
}

// ------------------------------------------------------------------ conv, NY output rows per lane
// The gather kernel above is bound by L1 bandwidth, not by FMA issue: per tap a wave pulls 64 x C_in
// floats through the texture path for 64 x C_in x C_out FMAs.  Here a lane owns NY vertically adjacent
// output voxels (same x, rows y0 .. y0+NY-1): an input row is loaded ONCE and feeds every output row
// whose kernel window covers it, so the loads per output drop from KH to (KH + (NY-1)*SH) / NY rows
// (3 -> 2 for a 3-tap kernel at NY = 2, 3 -> 1.5 at NY = 4) while lanes still read dense runs along x.
template <int CI0, int CI1, int CO0, int CO1, bool FLIP, int KH, int SH, int NY>
__global__ __launch_bounds__(256) void conv_rows_k(ConvDev p) {
  constexpr int CI = CI0 + CI1, CO = CO0 + CO1, NR = KH + (NY - 1) * SH;
  constexpr bool SPLIT = FLIP && CI0 % 2 == 0 && CI1 % 2 == 0;
  constexpr int NACC = SPLIT ? 2 * CO : CO;
  const int QH = (p.OH + NY - 1) / NY;
  int64_t idx = (int64_t)xcd_contiguous_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
  if (idx >= (int64_t)p.N * p.OD * QH * p.OW) return;
  int x = (int)(idx % p.OW); int64_t r_ = idx / p.OW;
  int yq = (int)(r_ % QH); r_ /= QH;
  int z = (int)(r_ % p.OD); int n = (int)(r_ / p.OD);
  const int y0 = yq * NY;

  float acc[NY][NACC];
#pragma unroll
  for (int v = 0; v < NY; ++v)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[v][i] = 0.f;

  const int ntap = p.kd * KH * p.kw;
  const int iy_first = y0 * SH - p.ph;
  for (int dz = 0; dz < p.kd; ++dz) {
    int iz = z * p.sd + dz - p.pd;
    bool okz = iz >= 0 && iz < p.D;
    const float *pl0 = p.in0 + n * p.i0N + iz * p.i0D;
    const float *pl1 = CI1 > 0 ? p.in1 + n * p.i1N + iz * p.i1D : nullptr;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      int iy = iy_first + r;
      bool oky = okz && iy >= 0 && iy < p.H;
      for (int dx = 0; dx < p.kw; ++dx) {
        int ix = x * p.sw + dx - p.pw;
        bool ok = oky && ix >= 0 && ix < p.W;
        float xv[CI0];
        load_vec<CI0>(xv, pl0 + iy * p.i0H + ix * p.i0W, ok);
        float xw[CI1 > 0 ? CI1 : 1];
        if constexpr (CI1 > 0) load_vec<CI1>(xw, pl1 + iy * p.i1H + ix * p.i1W, ok);
#pragma unroll
        for (int v = 0; v < NY; ++v) {
          constexpr int dummy = 0; (void)dummy;
          const int dy = r - v * SH;                       // compile-time after unrolling
          if (dy >= 0 && dy < KH) {
            const int tap = (dz * KH + dy) * p.kw + dx;
            const float *wt = p.w + (int64_t)(FLIP ? ntap - 1 - tap : tap) * (CI * CO);
            if constexpr (SPLIT) fma_block_split<CI0, CI, CO>(acc[v], xv, wt, 0);
            else fma_block<CI0, CI, CO, FLIP>(acc[v], xv, wt, 0);
            if constexpr (CI1 > 0) {
              if constexpr (SPLIT) fma_block_split<CI1, CI, CO>(acc[v], xw, wt, CI0);
              else fma_block<CI1, CI, CO, FLIP>(acc[v], xw, wt, CI0);
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < NY; ++v) {
    if (y0 + v < p.OH) {
      if constexpr (SPLIT) {
        float out[CO];
#pragma unroll
        for (int i = 0; i < CO; ++i) out[i] = acc[v][2 * i] + acc[v][2 * i + 1];
        finish<CO0, CO1>(p, out, n, z, y0 + v, x);
      } else {
        finish<CO0, CO1>(p, acc[v], n, z, y0 + v, x);
      }
    }
  }
}

// ------------------------------------------------------------------ transposed conv
// One launch covers all output residue classes (blockIdx.y); inside a class every lane uses
// the same taps, so the weights stay wave-uniform.  out[o] = sum_{j,t: o = j*s + t - p}.
// CPT = output channels per thread: CO (one thread per voxel) or a slice of it (blockIdx.z picks the
// slice).  The 32-channel layers of the discriminators are tiny (8^3 .. 42^3 voxels): one thread per
// voxel leaves most CUs idle behind a serial chain of 8 x 32 x 32 FMAs fed by scalar loads, so those
// run with 8 channels per thread and four times the threads.
template <int CI0, int CO0, int CO1, int CPT>
__global__ __launch_bounds__(256) void convT_direct_k(ConvDev p) {
  constexpr int CI = CI0, CO = CO0 + CO1;
  static_assert(CPT == CO || (CO1 == 0 && CO % CPT == 0), "channel slices only without a split output");
  const int cbase = CPT == CO ? 0 : blockIdx.z * CPT;
  int cls = blockIdx.y;
  int rx = cls % p.sw, ry = (cls / p.sw) % p.sh, rz = cls / (p.sw * p.sh);
  int QW = (p.OW - rx + p.sw - 1) / p.sw, QH = (p.OH - ry + p.sh - 1) / p.sh, QD = (p.OD - rz + p.sd - 1) / p.sd;
  int64_t cnt = (int64_t)p.N * QD * QH * QW;
  int64_t idx = (int64_t)xcd_contiguous_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
  if (idx >= cnt) return;
  int qx = (int)(idx % QW); int64_t r = idx / QW;
  int qy = (int)(r % QH); r /= QH;
  int qz = (int)(r % QD); int n = (int)(r / QD);
  int x = qx * p.sw + rx, y = qy * p.sh + ry, z = qz * p.sd + rz;

  constexpr bool SPLIT = CI0 % 2 == 0;                 // see fma_block_split
  constexpr int NACC = SPLIT ? 2 * CPT : CPT;
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.f;

  // taps with t == (o + p) mod s; j = (o + p - t) / s
  int tz0 = (rz + p.pd) % p.sd, ty0 = (ry + p.ph) % p.sh, tx0 = (rx + p.pw) % p.sw;
  for (int dz = tz0; dz < p.kd; dz += p.sd) {
    int jz = (z + p.pd - dz) / p.sd;
    bool okz = (z + p.pd - dz) >= 0 && jz < p.D;
    for (int dy = ty0; dy < p.kh; dy += p.sh) {
      int jy = (y + p.ph - dy) / p.sh;
      bool oky = okz && (y + p.ph - dy) >= 0 && jy < p.H;
      for (int dx = tx0; dx < p.kw; dx += p.sw) {
        int jx = (x + p.pw - dx) / p.sw;
        bool ok = oky && (x + p.pw - dx) >= 0 && jx < p.W;
        int tap = (dz * p.kh + dy) * p.kw + dx;
        const float *wt = p.w + (int64_t)tap * (CI * CO) + cbase * CI;       // [co][ci] block, this thread's rows
        float xv[CI0];
        load_vec<CI0>(xv, p.in0 + n * p.i0N + jz * p.i0D + jy * p.i0H + jx * p.i0W, ok);
        if constexpr (SPLIT) fma_block_split<CI0, CI, CPT>(acc, xv, wt, 0);
        else fma_block<CI0, CI, CPT, true>(acc, xv, wt, 0);
      }
    }
  }
  float out[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) out[i] = SPLIT ? acc[2 * i] + acc[2 * i + 1] : acc[i];
  if constexpr (CPT == CO) {
    finish<CO0, CO1>(p, out, n, z, y, x);
  } else {
    apply_epilogue<CPT>(p.ep, out, n, z, y, x, cbase, p.OD, p.OH, p.OW, CO);
    store_vec<CPT>(p.out0 + n * p.o0N + z * p.o0D + y * p.o0H + x * p.o0W + cbase, out);
  }
}

int fill_dev(const tem_conv_args *a, ConvDev &p, bool transposed) {
  if (!a || !tem_view_ok(a->in0) || !tem_view_ok(a->out0) || !a->w) return TEM_EINVAL;
  if (a->kd < 1 || a->kh < 1 || a->kw < 1 || a->sd < 1 || a->sh < 1 || a->sw < 1) return TEM_EINVAL;
  const tem_view &i0 = a->in0, &o0 = a->out0;
  p.in0 = i0.ptr; p.i0N = i0.sN; p.i0D = i0.sD; p.i0H = i0.sH; p.i0W = i0.sW;
  p.N = i0.N; p.D = i0.D; p.H = i0.H; p.W = i0.W;
  p.in1 = nullptr;
  if (a->in1.ptr) {
    const tem_view &i1 = a->in1;
    if (i1.N != i0.N || i1.D != i0.D || i1.H != i0.H || i1.W != i0.W) return TEM_ESHAPE;
    p.in1 = i1.ptr; p.i1N = i1.sN; p.i1D = i1.sD; p.i1H = i1.sH; p.i1W = i1.sW;
  }
  p.w = a->w;
  p.kd = a->kd; p.kh = a->kh; p.kw = a->kw; p.sd = a->sd; p.sh = a->sh; p.sw = a->sw;
  p.pd = a->pd; p.ph = a->ph; p.pw = a->pw;
  p.out0 = o0.ptr; p.o0N = o0.sN; p.o0D = o0.sD; p.o0H = o0.sH; p.o0W = o0.sW;
  p.OD = o0.D; p.OH = o0.H; p.OW = o0.W;
  p.out1 = nullptr;
  if (a->out1.ptr) {
    const tem_view &o1 = a->out1;
    if (o1.N != o0.N || o1.D != o0.D || o1.H != o0.H || o1.W != o0.W) return TEM_ESHAPE;
    p.out1 = o1.ptr; p.o1N = o1.sN; p.o1D = o1.sD; p.o1H = o1.sH; p.o1W = o1.sW;
  }
  if (o0.N != i0.N) return TEM_ESHAPE;
  if (!transposed) {
    // every output voxel must be defined by the geometry: o*s - p + (k-1) may exceed the
    // input only through zero padding, which predicated loads provide -- no constraint.
  }
  p.total = (int64_t)o0.N * o0.D * o0.H * o0.W;
  p.ep = make_epilogue(a->ep);
  if (a->ep.gate.ptr) {
    const tem_view &g = a->ep.gate;
    if (g.N != o0.N || g.D != o0.D || g.H != o0.H || g.W != o0.W || g.C < o0.C) return TEM_ESHAPE;
  }
  if (a->ep.add.ptr && (a->ep.add.C < o0.C || a->ep.add.N != o0.N)) return TEM_ESHAPE;
  // keep_mode 1 (write the dropout keep mask) needs whole bytes per thread: C_out a multiple of 8
  if (a->ep.dropout && a->ep.keep_mask && a->ep.keep_mode == 1 && o0.C % 8 != 0) return TEM_EUNSUPPORTED;
  return TEM_OK;
}

}  // namespace

#define CONV_CASE(ci0, ci1, co0, co1)                                                         \
  if (CI0 == ci0 && CI1 == ci1 && CO0 == co0 && CO1 == co1) {                                 \
    if (flip)                                                                                 \
      hipLaunchKernelGGL((conv_direct_k<ci0, ci1, co0, co1, true>), grid, dim3(256), 0, st, p); \
    else                                                                                      \
      hipLaunchKernelGGL((conv_direct_k<ci0, ci1, co0, co1, false>), grid, dim3(256), 0, st, p); \
    TEM_CHECK_LAUNCH();                                                                       \
    return TEM_OK;                                                                            \
  }

// ------------------------------------------------------------------ any channel count
// One thread per (output voxel, output channel), every extent a runtime value.  Not a fast path:
// it exists so that geometries outside the hot path's channel table (a frozen prior network fed to
// the discriminator, discriminator.py:62-66) run on the device instead of being refused.
__global__ __launch_bounds__(256) void conv_generic_k(ConvDev p, int CI0, int CI1, int CO0, int CO1, int flip,
                                                      int transposed) {
  const int CI = CI0 + CI1, CO = CO0 + CO1;
  int64_t idx = (int64_t)xcd_contiguous_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
  if (idx >= p.total * CO) return;
  const int co = (int)(idx % CO); int64_t r = idx / CO;
  int x = (int)(r % p.OW); r /= p.OW;
  int y = (int)(r % p.OH); r /= p.OH;
  int z = (int)(r % p.OD); int n = (int)(r / p.OD);
  const int ntap = p.kd * p.kh * p.kw;
  float acc = 0.f;
  for (int dz = 0; dz < p.kd; ++dz)
    for (int dy = 0; dy < p.kh; ++dy)
      for (int dx = 0; dx < p.kw; ++dx) {
        int iz, iy, ix;
        bool ok;
        if (!transposed) {
          iz = z * p.sd + dz - p.pd; iy = y * p.sh + dy - p.ph; ix = x * p.sw + dx - p.pw;
          ok = iz >= 0 && iz < p.D && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        } else {                                          // out[o] += in[j] * w[t] for o = j*s + t - p
          int tz = z + p.pd - dz, ty = y + p.ph - dy, tx = x + p.pw - dx;
          ok = tz >= 0 && ty >= 0 && tx >= 0 && tz % p.sd == 0 && ty % p.sh == 0 && tx % p.sw == 0;
          iz = tz / p.sd; iy = ty / p.sh; ix = tx / p.sw;
          ok = ok && iz < p.D && iy < p.H && ix < p.W;
        }
        if (!ok) continue;
        const int tap = (dz * p.kh + dy) * p.kw + dx;
        const float *wt = p.w + (int64_t)((flip && !transposed) ? ntap - 1 - tap : tap) * (CI * CO);
        const bool coci = flip || transposed;
        const float *x0 = p.in0 + n * p.i0N + iz * p.i0D + iy * p.i0H + ix * p.i0W;
        for (int ci = 0; ci < CI0; ++ci) acc = fmaf(x0[ci], coci ? wt[co * CI + ci] : wt[ci * CO + co], acc);
        if (CI1 > 0) {
          const float *x1 = p.in1 + n * p.i1N + iz * p.i1D + iy * p.i1H + ix * p.i1W;
          for (int ci = 0; ci < CI1; ++ci)
            acc = fmaf(x1[ci], coci ? wt[co * CI + CI0 + ci] : wt[(CI0 + ci) * CO + co], acc);
        }
      }
  if (co < CO0) {
    float v[1] = {acc};
    apply_epilogue<1>(p.ep, v, n, z, y, x, co, p.OD, p.OH, p.OW, CO0);
    p.out0[n * p.o0N + z * p.o0D + y * p.o0H + x * p.o0W + co] = v[0];
  } else {
    p.out1[n * p.o1N + z * p.o1D + y * p.o1H + x * p.o1W + (co - CO0)] = acc;
  }
}

static int launch_generic(const ConvDev &p, const tem_conv_args *a, hipStream_t st, bool transposed) {
  if (a->ep.dropout && a->ep.keep_mask && a->ep.keep_mode == 1) return TEM_EUNSUPPORTED;   // one channel per thread
  const int CI1 = a->in1.ptr ? a->in1.C : 0, CO1 = a->out1.ptr ? a->out1.C : 0;
  const int64_t cnt = p.total * (a->out0.C + CO1);
  if (cnt <= 0 || cnt > 0x7fffffffll * 256) return TEM_EUNSUPPORTED;
  hipLaunchKernelGGL(conv_generic_k, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, p, a->in0.C, CI1, a->out0.C, CO1,
                     a->w_layout == TEM_W_FLIP_CO_CI ? 1 : 0, transposed ? 1 : 0);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

// row-blocked kernel: the full-resolution 8-channel / 1-channel layers of both networks
#define ROWS_CASE(ci0, co0, fl, kh_, sh_, ny_)                                                            \
  if (CI0 == ci0 && CI1 == 0 && CO0 == co0 && CO1 == 0 && flip == fl && a->kh == kh_ && a->sh == sh_) {  \
    if (name) { snprintf(name, name_len, "conv_rows_k<%d, 0, %d, 0, %s, %d, %d, %d>", ci0, co0, fl ? "true" : "false", kh_, sh_, ny_); return TEM_OK; } \
    const int64_t cnt = (int64_t)p.N * p.OD * ((p.OH + ny_ - 1) / ny_) * p.OW;                           \
    hipLaunchKernelGGL((conv_rows_k<ci0, 0, co0, 0, fl, kh_, sh_, ny_>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, p); \
    TEM_CHECK_LAUNCH();                                                                                   \
    return TEM_OK;                                                                                        \
  }

static int conv_direct_impl(const tem_conv_args *a, hipStream_t st, char *name, int name_len) {
  ConvDev p{};
  int rc = fill_dev(a, p, false);
  if (rc) return rc;
  const int CI0 = a->in0.C, CI1 = a->in1.ptr ? a->in1.C : 0;
  const int CO0 = a->out0.C, CO1 = a->out1.ptr ? a->out1.C : 0;
  const bool flip = a->w_layout == TEM_W_FLIP_CO_CI;
  dim3 grid((unsigned)((p.total + 255) / 256));
  {
    // measured on MI355X (tests/microbench.py, 132^3 layer sizes): the row-blocked kernel wins where C_in or
    // C_out is 1 and for the forward 8->8 layer; the [co][ci]-layout 8/16-channel cases and the strided
    // layer stay on conv_direct_k (TEM_ROWS=0 disables it for A/B runs)
    static int rows = -1;
    if (rows < 0) rows = tem_env_int("TEM_ROWS", 1);
    if (rows && p.OH >= 16) {
      ROWS_CASE(8, 8, false, 3, 1, 4)
      ROWS_CASE(1, 8, false, 3, 1, 4) ROWS_CASE(8, 1, true, 3, 1, 4)
      ROWS_CASE(16, 1, false, 3, 1, 4) ROWS_CASE(1, 16, true, 3, 1, 4)
    }
    // TEM_ROWS2 (bit mask, perf triage): row-blocked form for the k4 s2 layers (bit 0), 8->16 / 16->8 k3 (bit 1)
    static int rows2 = -1;
    if (rows2 < 0) rows2 = tem_env_int("TEM_ROWS2", 0);
    if (p.OH >= 16 && a->kw == a->kh && a->kd == a->kh) {
      if (rows2 & 1) { ROWS_CASE(8, 8, false, 4, 2, 2) ROWS_CASE(8, 16, false, 4, 2, 2) }
      if (rows2 & 4) { ROWS_CASE(8, 8, false, 4, 2, 4) ROWS_CASE(8, 16, false, 4, 2, 4) }
      if (rows2 & 2) { ROWS_CASE(8, 16, false, 3, 1, 4) ROWS_CASE(16, 8, false, 3, 1, 4) }
    }
  }
  if (name) {
    static const int table[][4] = {{1, 0, 8, 0}, {1, 0, 16, 0}, {1, 0, 32, 0}, {8, 0, 8, 0}, {8, 0, 16, 0}, {8, 0, 1, 0},
                                   {16, 0, 16, 0}, {16, 0, 32, 0}, {16, 0, 8, 0}, {16, 0, 1, 0}, {32, 0, 32, 0}, {32, 0, 16, 0},
                                   {32, 0, 1, 0}, {8, 8, 16, 0}, {16, 16, 32, 0}, {16, 0, 8, 8}, {32, 0, 16, 16}};
    bool hit = false;
    for (auto &t : table) hit = hit || (t[0] == CI0 && t[1] == CI1 && t[2] == CO0 && t[3] == CO1);
    if (hit) snprintf(name, name_len, "conv_direct_k<%d, %d, %d, %d, %s>", CI0, CI1, CO0, CO1, flip ? "true" : "false");
    else snprintf(name, name_len, "conv_generic_k");
    return TEM_OK;
  }
  // forward shapes
  CONV_CASE(1, 0, 8, 0) CONV_CASE(1, 0, 16, 0) CONV_CASE(1, 0, 32, 0)
  CONV_CASE(8, 0, 8, 0) CONV_CASE(8, 0, 16, 0) CONV_CASE(8, 0, 1, 0)
  CONV_CASE(16, 0, 16, 0) CONV_CASE(16, 0, 32, 0) CONV_CASE(16, 0, 8, 0) CONV_CASE(16, 0, 1, 0)
  CONV_CASE(32, 0, 32, 0) CONV_CASE(32, 0, 16, 0) CONV_CASE(32, 0, 1, 0)
  CONV_CASE(8, 8, 16, 0) CONV_CASE(16, 16, 32, 0)
  // split outputs (input-gradient through a concat)
  CONV_CASE(16, 0, 8, 8) CONV_CASE(32, 0, 16, 16)
  return launch_generic(p, a, st, false);
}

extern "C" int tem_conv_direct(const tem_conv_args *a, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  return conv_direct_impl(a, (hipStream_t)stream, nullptr, 0);
}

int tem_conv_direct_describe(const tem_conv_args *a, char *name, int len) {
  return conv_direct_impl(a, nullptr, name, len);
}

#define CONVT_CASE(ci0, co0, co1, cpt)                                                                  \
  if (CI0 == ci0 && CO0 == co0 && CO1 == co1) {                                                         \
    dim3 g(grid.x, grid.y, (unsigned)((co0 + co1) / cpt));                                              \
    hipLaunchKernelGGL((convT_direct_k<ci0, co0, co1, cpt>), g, dim3(256), 0, st, p);                   \
    TEM_CHECK_LAUNCH();                                                                                 \
    return TEM_OK;                                                                                      \
  }

extern "C" int tem_conv_transpose_direct(const tem_conv_args *a, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  ConvDev p{};
  int rc = fill_dev(a, p, true);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (a->in1.ptr) return launch_generic(p, a, st, true);
  const int CI0 = a->in0.C;
  const int CO0 = a->out0.C, CO1 = a->out1.ptr ? a->out1.C : 0;
  int ncls = a->sd * a->sh * a->sw;
  int64_t qmax = (int64_t)p.N * ((p.OD + a->sd - 1) / a->sd) * ((p.OH + a->sh - 1) / a->sh) * ((p.OW + a->sw - 1) / a->sw);
  dim3 grid((unsigned)((qmax + 255) / 256), (unsigned)ncls);
  CONVT_CASE(32, 16, 0, 16) CONVT_CASE(16, 8, 0, 8)                           // Conv3DTranspose forward
  CONVT_CASE(8, 8, 0, 8) CONVT_CASE(16, 16, 0, 16) CONVT_CASE(32, 32, 0, 8)   // input-grad of the k4 s2 convs
  return launch_generic(p, a, st, true);
}
