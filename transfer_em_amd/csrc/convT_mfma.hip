// convT_mfma.hip -- transposed convolution k4 s2 (Conv3DTranspose forward, models/utils.py:129-130, and the
// input-gradient of the k4 s2 VALID convolutions, models/utils.py:80) on the fp32 matrix cores.
//
//   out[o][co] = sum over (j, t) with o = 2 j + t - p of in[j][ci] * w[t][co][ci]            (per axis)
//
// Per axis write o + p = 2 Q + r (r = parity class, Q = floor((o + p) / 2)): the taps that reach o are
// t = r + 2 c, c in {0, 1}, from input voxel j = Q - c.  ALL eight parity classes of one Q read the same 2x2x2
// input neighbourhood; only the kernel taps differ.  So for a fixed (r_z, r_y) class pair the operator is a GEMM
//
//   D[Q voxels][(r_x, co)] = sum_{(c_z, c_y, c_x, ci)} X[Q - c][ci] * B[(c, ci)][(r_x, co)],   K = 8 C_in,
//
// with the two x-classes side by side in the N dimension: the columns (r_x, co) of one Q voxel are the channels
// of two ADJACENT output voxels, i.e. one contiguous 2*C_out run in memory -- C_out = 8 fills a full 16-wide
// MFMA tile, and the A fragments are shared by all n-tiles.
//
// A workgroup owns (n, r_z, r_y, Q_z, a band of Q_y rows): it loads the 2-plane input patch once into LDS
// (channels-last, voxel pitch C_in + 2: conflict-free ds_read_b64 of a k-step PAIR), every wave keeps the B
// fragments of its n-tile in registers for the whole run (2 C_in VGPRs), gathers A per 16-voxel tile (tiles run
// across row ends: v = q_y * nQx + q_x), and leaves through the fused epilogue (skip-gradient add, LeakyReLU
// gradient gate, Philox dropout incl. writing / reading the keep mask, LeakyReLU) as 16-byte channel runs.
#include "tem_common.h"
#include <cstdio>
#include <cstdlib>

namespace convt_mfma {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Ep {
  float slope;
  const float *gate; int32_t gN, gD, gH, gW; float gate_slope;
  const float *add;  int32_t aN, aD, aH, aW, aoz, aoy, aox, aDd, aHh, aWw;
  int32_t dropout;
  DropoutStream ds;
  const uint32_t *step_dev;
  int32_t doz, doy, dox, dD, dH, dW;
  uint8_t *keep_mask;
  int32_t keep_mode;
  int32_t gbytes, abytes, mbytes;                // extents (bytes) of the gate / add views and of the keep mask: buffer ranges
};

struct Dev {
  const float *in;
  int32_t iN, iD, iH, iW, D, H, W;
  float *out;
  int32_t oN, oD, oH, oW, OD, OH, OW;
  int32_t P;
  int32_t Qlo_x, nQx, Qlo_y, nQy, Qlo_z, nQz;   // Q ranges (union over the parity classes)
  int32_t TY, nband;                           // Q_y rows per workgroup, bands
  int32_t cols, rows;                          // patch extents (voxels): nQx + 1, TY + 1
  uint32_t magicQx, magicCols;
  Ep ep;
};

// NCLS: (r_z, r_y) classes handled per workgroup on ONE loaded patch (their B fragments all stay in registers:
// NCLS * 2 C_in VGPRs) -- 4 for C_in 8, 2 (both r_y of one r_z) for C_in 16, 1 for C_in 32
// EPM: compiled epilogue -- 0: everything by run-time flags; 1: Conv3DTranspose forward of the train step (no gate, no
// skip-gradient; Dropout by the keep bits drawn ahead of the launch; LeakyReLU); 2: input-gradient (gate, optional
// skip-gradient add, no Dropout).  The epilogue runs once per 4 output floats of a lane on the vector pipe, which at
// C_in = 8 has fewer MFMA cycles to hide under than it takes: what a launch does not use is not compiled into it.
template <int CI, int CO, int PF, int NCLS, int EPM>
__global__ __launch_bounds__(256) void convT_mfma_k(Dev p, const float *__restrict__ wgt) {
  constexpr int CIP = CI + 2;
  constexpr int NT = 2 * CO / 16;                 // n-tiles over the columns (r_x, co)
  constexpr int WPN = 4 / NT;                     // waves per n-tile (tile subsets)
  constexpr int NPAIR = 8 * (CI / 8);             // k-step pairs: 8 taps x C_in/8 channel blocks
  constexpr int CPV = CI / 4;                     // 16-byte chunks per voxel
  constexpr int TPITCH = 20;
  static_assert(NT == 1 || NT == 2 || NT == 4, "C_out in {8, 16, 32}");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const int plane = p.rows * p.cols;

  int b = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);     // the 4 classes and z-neighbours of a patch share input: one L2
  constexpr int NGRP = 4 / NCLS;                               // class groups per patch
  const int cls0 = (b % NGRP) * NCLS; b /= NGRP;
  const int zq = b % p.nQz; b /= p.nQz;
  const int band = b % p.nband;
  const int n = b / p.nband;
  const int Qz = p.Qlo_z + zq, Qy0 = p.Qlo_y + band * p.TY;
  const int nrow = min(p.TY, p.nQy - band * p.TY);             // Q_y rows of this band
  {
    bool any = false;                                          // block-uniform: no class of this group has a plane here
#pragma unroll
    for (int c = 0; c < NCLS; ++c) { const int o = 2 * Qz + ((cls0 + c) >> 1) - p.P; any = any || (o >= 0 && o < p.OD); }
    if (!any) return;
  }

  // ---- B fragments of this wave's n-tile: pair pp = tap8 * (CI/8) + cb; k-steps (2pp, 2pp+1) multiply channels
  // ci = 8 cb + 2 kq + {0, 1} of input voxel Q - (c_z, c_y, c_x), tap8 = (c_z, c_y, c_x)
  const int nt = wave % NT;
  const int ncol = nt * 16 + m;                   // column (r_x, co)
  const int rx = ncol / CO, co = ncol - rx * CO;
  float B[NCLS][2 * NPAIR];
#pragma unroll
  for (int c = 0; c < NCLS; ++c) {
    const int rz = (cls0 + c) >> 1, ry = (cls0 + c) & 1;
#pragma unroll
    for (int pp = 0; pp < NPAIR; ++pp) {
      const int tap8 = pp / (CI / 8), cb = pp - tap8 * (CI / 8);
      const int cz = tap8 >> 2, cy = (tap8 >> 1) & 1, cx = tap8 & 1;
      const int tap = ((rz + 2 * cz) * 4 + (ry + 2 * cy)) * 4 + (rx + 2 * cx);
      const float2 w2 = *reinterpret_cast<const float2 *>(wgt + ((tap * CO + co) * CI + 8 * cb + 2 * kq));
      B[c][2 * pp] = w2.x;
      B[c][2 * pp + 1] = w2.y;
    }
  }

  // ---- input patch: planes j_z = Qz-1, Qz; rows j_y = Qy0-1 .. Qy0+nrow-1; cols j_x = Qlo_x-1 .. Qlo_x+nQx-1
  {
    const int total = 2 * plane * CPV;
    float4 pf[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int id = tid + i * 256;
      const int vox = id / CPV, c = (id - vox * CPV) * 4;
      const int pl = vox >= plane ? 1 : 0, r2 = vox - pl * plane;
      const int r = (int)__umulhi((uint32_t)r2, p.magicCols), cx = r2 - r * p.cols;
      const int jz = Qz - 1 + pl, jy = Qy0 - 1 + r, jx = p.Qlo_x - 1 + cx;
      const bool ok = id < total && (unsigned)jz < (unsigned)p.D && (unsigned)jy < (unsigned)p.H && (unsigned)jx < (unsigned)p.W;
      pf[i] = ok ? *reinterpret_cast<const float4 *>(p.in + (n * p.iN + jz * p.iD + jy * p.iH + jx * p.iW + c))
                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int id = tid + i * 256;
      if (id < total) {
        const int vox = id / CPV, c = (id - vox * CPV) * 4;
        float *d = lds + vox * CIP + c;                      // 8-byte aligned (CIP even)
        *reinterpret_cast<float2 *>(d) = make_float2(pf[i].x, pf[i].y);
        *reinterpret_cast<float2 *>(d + 2) = make_float2(pf[i].z, pf[i].w);
      }
    }
  }
  __syncthreads();

  // A gather: voxel (plane 1 - cz, row qy + 1 - cy, col qx + 1 - cx) of the patch, channel pair 8 cb + 2 kq
  int aoff[8];
#pragma unroll
  for (int t8 = 0; t8 < 8; ++t8) {
    const int cz = t8 >> 2, cy = (t8 >> 1) & 1, cx = t8 & 1;
    aoff[t8] = ((1 - cz) * plane + (1 - cy) * p.cols + (1 - cx)) * CIP + 2 * kq;
  }
  const int padded = (2 * plane * CIP + 3) & ~3;
  float *tp = lds + padded + wave * (16 * TPITCH);
  const int ti = lane >> 2, tcq = lane & 3;                   // transposed role: Q voxel of the tile, column quad
  const int ecol = nt * 16 + tcq * 4;                         // first of this lane's 4 columns
  const int erx = ecol / CO, eco = ecol - erx * CO;
  const int L = nrow * p.nQx;                                 // linearised Q voxels of the band
  const int ntiles = (L + 15) >> 4;
  const Ep &ep = p.ep;
  DropoutStream ds = ep.ds;
  if (ep.dropout && ep.step_dev) ds.step = *ep.step_dev;

  auto a_base = [&](int t) -> const float * {
    const int v = min(t * 16 + m, L - 1);                     // lanes past the band recompute its last voxel, never stored
    const int qy = p.nQx == 1 ? v : (int)__umulhi((uint32_t)v, p.magicQx), qx = v - qy * p.nQx;
    return lds + (qy * p.cols + qx) * CIP;
  };
  // Epilogue in two halves: `prep` (before the tile's MFMA chain) computes the lane's output voxel and ISSUES the
  // gate / skip-gradient loads, `finish` (after it) consumes them -- their HBM/L2 latency hides under the matrix work.
  // (Their loads go through buffer descriptors: a lane without an output voxel, a voxel outside the skip-gradient window or
  // an absent tensor is an out-of-range offset = zeros.  As conditional plain loads (`if (valid) g4 = *...`) the merge of
  // loaded value and default sat right behind the load, with s_waitcnt vmcnt(0) IN FRONT of the MFMA chain.)
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)ep.gate, 0, ep.gate ? ep.gbytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void *)ep.add, 0, ep.add ? ep.abytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void *)ep.keep_mask, 0, ep.keep_mode == 2 ? ep.mbytes : 0, 0x00020000);
  struct Prep { int oy, ox; bool valid; float4 g4, a4; uint32_t kb; };
  auto prep = [&](int t, int ry, int oz) -> Prep {
    Prep q;
    const int v = t * 16 + ti;
    const int qy = p.nQx == 1 ? v : (int)__umulhi((uint32_t)v, p.magicQx), qx = v - qy * p.nQx;
    q.oy = 2 * (Qy0 + qy) + ry - p.P; q.ox = 2 * (p.Qlo_x + qx) + erx - p.P;
    q.valid = v < L && (unsigned)q.oy < (unsigned)p.OH && (unsigned)q.ox < (unsigned)p.OW;
    q.g4 = make_float4(0.f, 0.f, 0.f, 0.f); q.a4 = q.g4; q.kb = 0;
    if (EPM != 1) {
      int goff = q.valid ? (n * ep.gN + oz * ep.gD + q.oy * ep.gH + q.ox * ep.gW + eco) * 4 : (int)0x80000000;
      asm volatile("" : "+v"(goff));
      const u32x4 g = __builtin_amdgcn_raw_buffer_load_b128(grs, goff, 0, 0);
      q.g4 = make_float4(__uint_as_float(g.x), __uint_as_float(g.y), __uint_as_float(g.z), __uint_as_float(g.w));
      const int az = oz - ep.aoz, ay = q.oy - ep.aoy, ax = q.ox - ep.aox;
      const bool ain = q.valid && (unsigned)az < (unsigned)ep.aDd && (unsigned)ay < (unsigned)ep.aHh && (unsigned)ax < (unsigned)ep.aWw;
      int aoff = ain ? (n * ep.aN + az * ep.aD + ay * ep.aH + ax * ep.aW + eco) * 4 : (int)0x80000000;
      asm volatile("" : "+v"(aoff));
      const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(ars, aoff, 0, 0);
      q.a4 = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
    }
    if (EPM != 2) {
      // keep bits drawn ahead of this launch (keep_mode 2): the byte of the lane's channel quad = voxel * C_out/8 + eco/8
      // (C_out is a multiple of 8 whenever there is a mask; the element count is below 2^32: host)
      const uint32_t vox = (((uint32_t)n * ep.dD + (oz + ep.doz)) * ep.dH + (q.oy + ep.doy)) * ep.dW + (q.ox + ep.dox);
      int moff = q.valid ? (int)(vox * (uint32_t)(CO >> 3) + (uint32_t)(eco >> 3)) : (int)0x80000000;
      asm volatile("" : "+v"(moff));
      q.kb = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(mrs, moff, 0, 0);
    }
    return q;
  };
  auto finish = [&](const f32x4 &acc, const Prep &q, int oz) {
#pragma unroll
    for (int r = 0; r < 4; ++r) tp[(kq * 4 + r) * TPITCH + m] = acc[r];
    __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): this wave's own LDS writes have landed
    const float4 v4 = *reinterpret_cast<const float4 *>(tp + ti * TPITCH + tcq * 4);
    const int oy = q.oy, ox = q.ox;
    const bool valid = q.valid;
    float vv[4] = {v4.x + q.a4.x, v4.y + q.a4.y, v4.z + q.a4.z, v4.w + q.a4.w};
    if (EPM == 2 || (EPM == 0 && ep.gate)) {
      vv[0] = q.g4.x > 0.f ? vv[0] : ep.gate_slope * vv[0];
      vv[1] = q.g4.y > 0.f ? vv[1] : ep.gate_slope * vv[1];
      vv[2] = q.g4.z > 0.f ? vv[2] : ep.gate_slope * vv[2];
      vv[3] = q.g4.w > 0.f ? vv[3] : ep.gate_slope * vv[3];
    }
    if (EPM == 1 || (EPM == 0 && ep.dropout)) {                // kernel-uniform
      uint32_t bits;
      if (EPM == 1 || ep.keep_mode == 2) {
        bits = (q.kb >> (uint32_t)(eco & 4)) & 15u;             // (fetched by prep; zero for lanes without a voxel)
      } else {
        const uint64_t e = ((((uint64_t)n * ep.dD + (oz + ep.doz)) * ep.dH + (oy + ep.doy)) * ep.dW + (ox + ep.dox)) * (uint64_t)CO + eco;
        const Philox128 ph = ds.block(e >> 7);
        const uint32_t eb = (uint32_t)(e & 127);
        bits = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) bits |= (DropoutStream::bit(ph, eb + c) ? 1u : 0u) << c;
        if (ep.keep_mode == 1) {
          // a byte of the mask = the 8 channels eco&~7 .. +7 of one voxel = this lane's nibble and its neighbour's
          const uint32_t other = (uint32_t)__shfl_xor((int)bits, 1, 64);
          if (valid && !(tcq & 1)) ep.keep_mask[e >> 3] = (uint8_t)(bits | (other << 4));
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) vv[c] = ((bits >> c) & 1u) ? 2.f * vv[c] : 0.f;
    }
    if (valid) {
      if (ep.slope != 1.f) {
#pragma unroll
        for (int c = 0; c < 4; ++c) vv[c] = vv[c] > 0.f ? vv[c] : ep.slope * vv[c];
      }
      *reinterpret_cast<float4 *>(p.out + (n * p.oN + oz * p.oD + oy * p.oH + ox * p.oW + eco)) =
          make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
  };

  // two tiles per iteration (independent accumulator chains interleave)
#pragma unroll
  for (int c = 0; c < NCLS; ++c) {
    const int rz = (cls0 + c) >> 1, ry = (cls0 + c) & 1;
    const int oz = 2 * Qz + rz - p.P;
    if (oz < 0 || oz >= p.OD) continue;                        // block-uniform
    for (int t = wave / NT; t < ntiles; t += 2 * WPN) {       // wave-uniform
      const int t2 = t + WPN;
      const bool two = t2 < ntiles;
      const float *s0 = a_base(t), *s1 = a_base(two ? t2 : t);
      const Prep q0 = prep(t, ry, oz), q1 = prep(two ? t2 : t, ry, oz);
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t8 = 0; t8 < 8; ++t8) {
        float2 a0[CI / 8], a1[CI / 8];
#pragma unroll
        for (int cb = 0; cb < CI / 8; ++cb) {
          a0[cb] = *reinterpret_cast<const float2 *>(s0 + aoff[t8] + 8 * cb);
          a1[cb] = *reinterpret_cast<const float2 *>(s1 + aoff[t8] + 8 * cb);
        }
#pragma unroll
        for (int cb = 0; cb < CI / 8; ++cb) {
          const int pp = t8 * (CI / 8) + cb;
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[cb].x, B[c][2 * pp], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[cb].x, B[c][2 * pp], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[cb].y, B[c][2 * pp + 1], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[cb].y, B[c][2 * pp + 1], acc1, 0, 0, 0);
        }
      }
      finish(acc0, q0, oz);
      if (two) finish(acc1, q1, oz);
    }
  }
}

// ------------------------------------------------------------------------------------------ host
static uint32_t magic_for(int d) { return (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d); }

static bool fits32(const tem_view &v) {
  int64_t span = (int64_t)(v.N - 1) * v.sN + (int64_t)(v.D - 1) * v.sD + (int64_t)(v.H - 1) * v.sH +
                 (int64_t)(v.W - 1) * v.sW + v.C;
  return span < ((int64_t)1 << 31);
}

static thread_local char *g_name = nullptr;
static thread_local int g_name_len = 0;

static int floordiv2(int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); }

template <int CI, int CO, int PF, int NCLS>
int run(Dev p, int N, const float *w, hipStream_t st, bool dry, int epm) {
  constexpr int CIP = CI + 2, CPV = CI / 4;
  // o + P = 2Q + r  =>  Q in [floor(P/2), floor((O-1+P)/2)]
  p.Qlo_x = floordiv2(p.P); p.nQx = floordiv2(p.OW - 1 + p.P) - p.Qlo_x + 1;
  p.Qlo_y = floordiv2(p.P); p.nQy = floordiv2(p.OH - 1 + p.P) - p.Qlo_y + 1;
  p.Qlo_z = floordiv2(p.P); p.nQz = floordiv2(p.OD - 1 + p.P) - p.Qlo_z + 1;
  p.cols = p.nQx + 1;
  // rows per band: as many as the loader's registers and ~48 KB of LDS allow, but at least ~8 tiles per workgroup
  int TY = 0;
  for (int ty = 1; ty <= p.nQy && ty <= 32; ++ty) {
    const size_t chunks = (size_t)2 * (ty + 1) * p.cols * CPV;
    const size_t bytes = ((size_t)2 * (ty + 1) * p.cols * CIP + 4 * 16 * 20 + 4) * 4;
    if (chunks > (size_t)PF * 256 || bytes > 56 * 1024) break;
    TY = ty;
    if ((ty * p.nQx + 15) / 16 >= 16) break;
  }
  if (TY < 1) return TEM_EUNSUPPORTED;
  p.TY = TY; p.rows = TY + 1;
  p.nband = (p.nQy + TY - 1) / TY;
  p.magicQx = magic_for(p.nQx);
  p.magicCols = magic_for(p.cols);
  if (dry) {
    if (g_name) snprintf(g_name, g_name_len, "convT_mfma_k<%d, %d, %d, %d, %d>", CI, CO, PF, NCLS, epm);
    return TEM_OK;
  }
  static int dbg = -1;
  if (dbg < 0) dbg = tem_env_int("TEM_DEBUG_FLAGS", 0);
  const size_t lds_bytes = ((((size_t)2 * p.rows * p.cols * CIP + 3) & ~(size_t)3) + 4 * 16 * 20) * 4;
  const int nblocks = N * p.nband * p.nQz * (4 / NCLS);
  if (dbg & 8)
    fprintf(stderr, "convT_mfma<%d,%d> O=%dx%dx%d P=%d: nQ=%dx%dx%d TY=%d bands=%d blocks=%d lds=%zu\n", CI, CO, p.OD, p.OH,
            p.OW, p.P, p.nQz, p.nQy, p.nQx, p.TY, p.nband, nblocks, lds_bytes);
  if (epm == 1) hipLaunchKernelGGL((convT_mfma_k<CI, CO, PF, NCLS, 1>), dim3((unsigned)nblocks), dim3(256), lds_bytes, st, p, w);
  else if (epm == 2) hipLaunchKernelGGL((convT_mfma_k<CI, CO, PF, NCLS, 2>), dim3((unsigned)nblocks), dim3(256), lds_bytes, st, p, w);
  else hipLaunchKernelGGL((convT_mfma_k<CI, CO, PF, NCLS, 0>), dim3((unsigned)nblocks), dim3(256), lds_bytes, st, p, w);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

int dispatch(const tem_conv_args *a, hipStream_t st, bool dry) {
  const tem_view &i0 = a->in0, &o0 = a->out0;
  if (a->in1.ptr || a->out1.ptr || a->ep.bias) return TEM_EUNSUPPORTED;
  if (a->kd != 4 || a->kh != 4 || a->kw != 4 || a->sd != 2 || a->sh != 2 || a->sw != 2) return TEM_EUNSUPPORTED;
  if (a->pd != a->ph || a->ph != a->pw) return TEM_EUNSUPPORTED;
  if (o0.N != i0.N) return TEM_ESHAPE;
  if (!fits32(i0) || !fits32(o0)) return TEM_EUNSUPPORTED;
  static int enabled = -1;
  if (enabled < 0) enabled = tem_env_int("TEM_CONVT_MFMA", 1);
  if (!enabled) return TEM_EUNSUPPORTED;
  auto aligned = [](const tem_view &v) {
    return ((uintptr_t)v.ptr & 15) == 0 && v.sW % 4 == 0 && v.sH % 4 == 0 && v.sD % 4 == 0 && v.sN % 4 == 0;
  };
  if (!aligned(i0) || !aligned(o0)) return TEM_EUNSUPPORTED;
  Dev p{};
  p.in = i0.ptr; p.iN = (int)i0.sN; p.iD = (int)i0.sD; p.iH = (int)i0.sH; p.iW = (int)i0.sW;
  p.D = i0.D; p.H = i0.H; p.W = i0.W;
  p.out = o0.ptr; p.oN = (int)o0.sN; p.oD = (int)o0.sD; p.oH = (int)o0.sH; p.oW = (int)o0.sW;
  p.OD = o0.D; p.OH = o0.H; p.OW = o0.W;
  p.P = a->pd;
  const tem_epilogue &e = a->ep;
  Ep &q = p.ep;
  q.slope = e.slope; q.gate_slope = e.gate_slope;
  if (e.gate.ptr) {
    const tem_view &g = e.gate;
    if (g.N != o0.N || g.D != o0.D || g.H != o0.H || g.W != o0.W || g.C < o0.C) return TEM_ESHAPE;
    if (!fits32(g) || !aligned(g)) return TEM_EUNSUPPORTED;
    q.gate = g.ptr; q.gN = (int)g.sN; q.gD = (int)g.sD; q.gH = (int)g.sH; q.gW = (int)g.sW;
    q.gbytes = (int)(((int64_t)(g.N - 1) * g.sN + (int64_t)(g.D - 1) * g.sD + (int64_t)(g.H - 1) * g.sH + (int64_t)(g.W - 1) * g.sW + g.C) * 4);
  }
  if (e.add.ptr) {
    const tem_view &ad = e.add;
    if (ad.C < o0.C || ad.N != o0.N) return TEM_ESHAPE;
    if (!fits32(ad) || !aligned(ad)) return TEM_EUNSUPPORTED;
    q.add = ad.ptr; q.aN = (int)ad.sN; q.aD = (int)ad.sD; q.aH = (int)ad.sH; q.aW = (int)ad.sW;
    q.abytes = (int)(((int64_t)(ad.N - 1) * ad.sN + (int64_t)(ad.D - 1) * ad.sD + (int64_t)(ad.H - 1) * ad.sH + (int64_t)(ad.W - 1) * ad.sW + ad.C) * 4);
    q.aoz = e.add_off[0]; q.aoy = e.add_off[1]; q.aox = e.add_off[2];
    q.aDd = ad.D; q.aHh = ad.H; q.aWw = ad.W;
  }
  q.dropout = e.dropout;
  q.ds.k0 = (uint32_t)e.seed; q.ds.k1 = (uint32_t)(e.seed >> 32); q.ds.site = e.site; q.ds.step = e.step;
  q.step_dev = e.step_dev;
  q.keep_mask = (e.dropout && e.keep_mask) ? e.keep_mask : nullptr;
  q.keep_mode = q.keep_mask ? e.keep_mode : 0;
  if (q.keep_mode && o0.C % 8 != 0) return TEM_EUNSUPPORTED;
  q.doz = e.drop_org[0]; q.doy = e.drop_org[1]; q.dox = e.drop_org[2];
  q.dD = e.drop_dims[0] ? e.drop_dims[0] : o0.D; q.dH = e.drop_dims[0] ? e.drop_dims[1] : o0.H;
  q.dW = e.drop_dims[0] ? e.drop_dims[2] : o0.W;
  {
    const int64_t melems = (int64_t)o0.N * q.dD * q.dH * q.dW * o0.C;
    if (melems >= ((int64_t)1 << 32)) return TEM_EUNSUPPORTED;
    q.mbytes = (int)((melems + 7) / 8);
    auto span = [](const tem_view &v) {
      return (int64_t)(v.N - 1) * v.sN + (int64_t)(v.D - 1) * v.sD + (int64_t)(v.H - 1) * v.sH + (int64_t)(v.W - 1) * v.sW + v.C;
    };
    if ((e.gate.ptr && span(e.gate) >= ((int64_t)1 << 29)) || (e.add.ptr && span(e.add) >= ((int64_t)1 << 29)))
      return TEM_EUNSUPPORTED;                     // byte offsets of the epilogue's buffer loads stay below 2^31
  }
  const int CI = i0.C, CO = o0.C, N = i0.N;
  static int epm_on = -1;
  if (epm_on < 0) epm_on = tem_env_int("TEM_CONVT_EPM", 1);
  int epm = 0;
  if (epm_on) {
    if (q.dropout && q.keep_mode == 2 && !q.gate && !q.add) epm = 1;
    else if (!q.dropout && q.gate) epm = 2;
  }
#define CT_CASE(ci, co, pf, ncls) if (CI == ci && CO == co) return run<ci, co, pf, ncls>(p, N, a->w, st, dry, epm);
  CT_CASE(16, 8, 12, 1)     // g.u1b forward (Conv3DTranspose 16 -> 8)
  CT_CASE(32, 16, 12, 1)    // g.u2b forward
  CT_CASE(8, 8, 12, 1)      // input-gradient of g.d1b / d.d1b    (more classes per patch measured no faster)
  CT_CASE(16, 16, 12, 1)    // input-gradient of g.d2b
  CT_CASE(32, 32, 12, 1)    // input-gradient of d.d2b / d.d3b
#undef CT_CASE
  return TEM_EUNSUPPORTED;
}

}  // namespace convt_mfma

// Called by tem_conv_transpose (dispatch.hip) before it falls back to the direct kernel.
int tem_convT_mfma_try(const tem_conv_args *a, hipStream_t st, bool dry) { return convt_mfma::dispatch(a, st, dry); }

int tem_convT_mfma_describe(const tem_conv_args *a, char *buf, int len) {
  convt_mfma::g_name = buf; convt_mfma::g_name_len = len;
  int rc = convt_mfma::dispatch(a, nullptr, true);
  convt_mfma::g_name = nullptr;
  return rc;
}
