// convT_bf16.hip -- the bf16 mixed-precision form (BASELINE config 5) of convT_mfma.hip: transposed convolution
// k4 s2 (Conv3DTranspose forward, models/utils.py:129-130, and the input-gradient of the k4 s2 VALID convolutions,
// models/utils.py:80) with bf16 activations / kernel copy, fp32 accumulation on v_mfma_f32_16x16x32_bf16, bf16 stores.
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

namespace convt_bf16 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned short u16;
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }   // round to nearest even


typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Ep {
  float slope;
  const u16 *gate; int32_t gN, gD, gH, gW; float gate_slope;
  const u16 *add;  int32_t aN, aD, aH, aW, aoz, aoy, aox, aDd, aHh, aWw;
  int32_t dropout;
  DropoutStream ds;
  const uint32_t *step_dev;
  int32_t doz, doy, dox, dD, dH, dW;
  uint8_t *keep_mask;
  int32_t keep_mode;
  int32_t gbytes, abytes, mbytes;                // extents (bytes) of the gate / add views and of the keep mask: buffer ranges
};

struct Dev {
  const u16 *in;
  int32_t iN, iD, iH, iW, D, H, W;
  u16 *out;
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
// EPM: compiled epilogue as in convT_mfma_k -- 0: run-time flags; 1: forward of the train step (keep bits drawn ahead, LeakyReLU; no
// gate / skip-gradient); 2: input-gradient (gate, optional skip-gradient add, no Dropout).  [With run-time flags a tile iteration
// was ~800 instructions for 4..16 MFMAs.]
template <int CI, int CO, int PF, int NCLS, int EPM>
__global__ __launch_bounds__(256) void convT_bf16_k(Dev p, const u16 *__restrict__ wgt) {
  constexpr int CIP = CI;                         // LDS voxel pitch (bf16 elements): the plain channels-last image (b128 fragment reads are conflict-free, lds_b128_probe; the 16-byte pad of round 2 cost ~8 %)
  constexpr int NT = 2 * CO / 16;                 // n-tiles over the columns (r_x, co)
  constexpr int WPN = 4 / NT;                     // waves per n-tile (tile subsets)
  constexpr int NSTEP = 8 * CI / 32;              // k-steps of 32: k = (tap8, ci), a lane's 8 k-values = 8 channels of one tap
  constexpr int CPV = CI / 8;                     // 16-byte chunks per voxel
  constexpr int TPITCH = 20;
  static_assert(NT == 1 || NT == 2 || NT == 4, "C_out in {8, 16, 32}");
  extern __shared__ __attribute__((aligned(16))) u16 lds[];
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
  bf16x8 B[NCLS][NSTEP];
#pragma unroll
  for (int c = 0; c < NCLS; ++c) {
    const int rz = (cls0 + c) >> 1, ry = (cls0 + c) & 1;
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const int e0 = 32 * st + 8 * kq, tap8 = e0 / CI, c0 = e0 - tap8 * CI;
      const int cz = tap8 >> 2, cy = (tap8 >> 1) & 1, cx = tap8 & 1;
      const int tap = ((rz + 2 * cz) * 4 + (ry + 2 * cy)) * 4 + (rx + 2 * cx);
      B[c][st] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(wgt + ((tap * CO + co) * CI + c0)));
    }
  }

  // ---- input patch: planes j_z = Qz-1, Qz; rows j_y = Qy0-1 .. Qy0+nrow-1; cols j_x = Qlo_x-1 .. Qlo_x+nQx-1
  {
    const int total = 2 * plane * CPV;
    uint4 pf[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int id = tid + i * 256;
      const int vox = id / CPV, c = (id - vox * CPV) * 8;
      const int pl = vox >= plane ? 1 : 0, r2 = vox - pl * plane;
      const int r = (int)__umulhi((uint32_t)r2, p.magicCols), cx = r2 - r * p.cols;
      const int jz = Qz - 1 + pl, jy = Qy0 - 1 + r, jx = p.Qlo_x - 1 + cx;
      const bool ok = id < total && (unsigned)jz < (unsigned)p.D && (unsigned)jy < (unsigned)p.H && (unsigned)jx < (unsigned)p.W;
      pf[i] = ok ? *reinterpret_cast<const uint4 *>(p.in + (n * p.iN + jz * p.iD + jy * p.iH + jx * p.iW + c))
                 : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int id = tid + i * 256;
      if (id < total) {
        const int vox = id / CPV, c = (id - vox * CPV) * 8;
        *reinterpret_cast<uint4 *>(lds + vox * CIP + c) = pf[i];   // 16-byte aligned: CIP and c are multiples of 8
      }
    }
  }
  __syncthreads();

  // A gather: k-step st, lane group kq -> tap8 = (c_z, c_y, c_x), first channel c0: voxel (plane 1 - cz, row qy + 1 - cy,
  // col qx + 1 - cx) of the patch
  int aoff[NSTEP];
#pragma unroll
  for (int st = 0; st < NSTEP; ++st) {
    const int e0 = 32 * st + 8 * kq, tap8 = e0 / CI, c0 = e0 - tap8 * CI;
    const int cz = tap8 >> 2, cy = (tap8 >> 1) & 1, cx = tap8 & 1;
    aoff[st] = ((1 - cz) * plane + (1 - cy) * p.cols + (1 - cx)) * CIP + c0;
  }
  const int padded = (2 * plane * CIP + 7) & ~7;             // bf16 elements; 16-byte aligned
  float *tp = reinterpret_cast<float *>(lds + padded) + wave * (16 * TPITCH);
  const int ti = lane >> 2, tcq = lane & 3;                   // transposed role: Q voxel of the tile, column quad
  const int ecol = nt * 16 + tcq * 4;                         // first of this lane's 4 columns
  const int erx = ecol / CO, eco = ecol - erx * CO;
  const int L = nrow * p.nQx;                                 // linearised Q voxels of the band
  const int ntiles = (L + 15) >> 4;
  const Ep &ep = p.ep;
  DropoutStream ds = ep.ds;
  if (ep.dropout && ep.step_dev) ds.step = *ep.step_dev;

  auto a_base = [&](int t) -> const u16 * {
    const int v = min(t * 16 + m, L - 1);                     // lanes past the band recompute its last voxel, never stored
    const int qy = p.nQx == 1 ? v : (int)__umulhi((uint32_t)v, p.magicQx), qx = v - qy * p.nQx;
    return lds + (qy * p.cols + qx) * CIP;
  };
  // Epilogue in two halves: `prep` (before the tile's MFMA chain) computes the lane's output voxel and ISSUES the
  // gate / skip-gradient loads, `finish` (after it) consumes them -- their HBM/L2 latency hides under the matrix work.
  // (through buffer descriptors, out-of-range = zeros: as conditional plain loads the value merge put s_waitcnt vmcnt(0) in
  // front of the MFMA chain -- see conv_bf16.hip)
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)ep.gate, 0, ep.gate ? ep.gbytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void *)ep.add, 0, ep.add ? ep.abytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void *)ep.keep_mask, 0, ep.keep_mode == 2 ? ep.mbytes : 0, 0x00020000);
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  struct Prep { int oy, ox; bool valid; uint2 g4, a4; uint32_t kb; };
  auto prep = [&](int t, int ry, int oz) -> Prep {
    Prep q;
    const int v = t * 16 + ti;
    const int qy = p.nQx == 1 ? v : (int)__umulhi((uint32_t)v, p.magicQx), qx = v - qy * p.nQx;
    q.oy = 2 * (Qy0 + qy) + ry - p.P; q.ox = 2 * (p.Qlo_x + qx) + erx - p.P;
    q.valid = v < L && (unsigned)q.oy < (unsigned)p.OH && (unsigned)q.ox < (unsigned)p.OW;
    q.g4 = make_uint2(0u, 0u); q.a4 = q.g4; q.kb = 0;
    if (EPM != 1) {
      int goff = q.valid ? (n * ep.gN + oz * ep.gD + q.oy * ep.gH + q.ox * ep.gW + eco) * 2 : (int)0x80000000;
      asm volatile("" : "+v"(goff));
      const u32x2 g = __builtin_amdgcn_raw_buffer_load_b64(grs, goff, 0, 0);
      q.g4 = make_uint2(g.x, g.y);
      const int az = oz - ep.aoz, ay = q.oy - ep.aoy, ax = q.ox - ep.aox;
      const bool ain = q.valid && (unsigned)az < (unsigned)ep.aDd && (unsigned)ay < (unsigned)ep.aHh && (unsigned)ax < (unsigned)ep.aWw;
      int aoff = ain ? (n * ep.aN + az * ep.aD + ay * ep.aH + ax * ep.aW + eco) * 2 : (int)0x80000000;
      asm volatile("" : "+v"(aoff));
      const u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(ars, aoff, 0, 0);
      q.a4 = make_uint2(a.x, a.y);
    }
    if (EPM != 2) {
      // (the element count is below 2^32: host)
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
    float vv[4] = {v4.x + bf2f((u16)(q.a4.x & 0xffffu)), v4.y + bf2f((u16)(q.a4.x >> 16)),
                   v4.z + bf2f((u16)(q.a4.y & 0xffffu)), v4.w + bf2f((u16)(q.a4.y >> 16))};
    if (EPM == 2 || (EPM == 0 && ep.gate)) {
      vv[0] = bf2f((u16)(q.g4.x & 0xffffu)) > 0.f ? vv[0] : ep.gate_slope * vv[0];
      vv[1] = bf2f((u16)(q.g4.x >> 16)) > 0.f ? vv[1] : ep.gate_slope * vv[1];
      vv[2] = bf2f((u16)(q.g4.y & 0xffffu)) > 0.f ? vv[2] : ep.gate_slope * vv[2];
      vv[3] = bf2f((u16)(q.g4.y >> 16)) > 0.f ? vv[3] : ep.gate_slope * vv[3];
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
      *reinterpret_cast<uint2 *>(p.out + (n * p.oN + oz * p.oD + oy * p.oH + ox * p.oW + eco)) =
          make_uint2(f2bf(vv[0]) | ((uint32_t)f2bf(vv[1]) << 16), f2bf(vv[2]) | ((uint32_t)f2bf(vv[3]) << 16));
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
      const u16 *s0 = a_base(t), *s1 = a_base(two ? t2 : t);
      const Prep q0 = prep(t, ry, oz), q1 = prep(two ? t2 : t, ry, oz);
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(s0 + aoff[st]));
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(s1 + aoff[st]));
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, B[c][st], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, B[c][st], acc1, 0, 0, 0);
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
int run(Dev p, int N, const u16 *w, hipStream_t st, bool dry, int epm) {
  constexpr int CIP = CI, CPV = CI / 8;
  // o + P = 2Q + r  =>  Q in [floor(P/2), floor((O-1+P)/2)]
  p.Qlo_x = floordiv2(p.P); p.nQx = floordiv2(p.OW - 1 + p.P) - p.Qlo_x + 1;
  p.Qlo_y = floordiv2(p.P); p.nQy = floordiv2(p.OH - 1 + p.P) - p.Qlo_y + 1;
  p.Qlo_z = floordiv2(p.P); p.nQz = floordiv2(p.OD - 1 + p.P) - p.Qlo_z + 1;
  p.cols = p.nQx + 1;
  // rows per band: as many as the loader's registers and ~48 KB of LDS allow, but at least ~8 tiles per workgroup
  int TY = 0;
  for (int ty = 1; ty <= p.nQy && ty <= 32; ++ty) {
    const size_t chunks = (size_t)2 * (ty + 1) * p.cols * CPV;
    const size_t bytes = (size_t)2 * (ty + 1) * p.cols * CIP * 2 + 16 + 4 * 16 * 20 * 4;
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
    if (g_name) snprintf(g_name, g_name_len, "convT_bf16_k<%d, %d, %d, %d, %d>", CI, CO, PF, NCLS, epm);
    return TEM_OK;
  }
  static int dbg = -1;
  if (dbg < 0) dbg = tem_env_int("TEM_DEBUG_FLAGS", 0);
  const size_t lds_bytes = (((size_t)2 * p.rows * p.cols * CIP + 7) & ~(size_t)7) * 2 + 4 * 16 * 20 * 4;
  const int nblocks = N * p.nband * p.nQz * (4 / NCLS);
  if (dbg & 8)
    fprintf(stderr, "convT_bf16<%d,%d> O=%dx%dx%d P=%d: nQ=%dx%dx%d TY=%d bands=%d blocks=%d lds=%zu\n", CI, CO, p.OD, p.OH,
            p.OW, p.P, p.nQz, p.nQy, p.nQx, p.TY, p.nband, nblocks, lds_bytes);
  if (epm == 1) hipLaunchKernelGGL((convT_bf16_k<CI, CO, PF, NCLS, 1>), dim3((unsigned)nblocks), dim3(256), lds_bytes, st, p, w);
  else if (epm == 2) hipLaunchKernelGGL((convT_bf16_k<CI, CO, PF, NCLS, 2>), dim3((unsigned)nblocks), dim3(256), lds_bytes, st, p, w);
  else hipLaunchKernelGGL((convT_bf16_k<CI, CO, PF, NCLS, 0>), dim3((unsigned)nblocks), dim3(256), lds_bytes, st, p, w);
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
  auto U = [](const float *q) { return reinterpret_cast<const u16 *>(q); };
  auto al16 = [](const tem_view &v) {       // 16-byte chunks of 8 bf16
    return ((uintptr_t)v.ptr & 15) == 0 && v.sW % 8 == 0 && v.sH % 8 == 0 && v.sD % 8 == 0 && v.sN % 8 == 0;
  };
  auto aligned = [](const tem_view &v) {    // 8-byte accesses of 4 bf16
    return ((uintptr_t)v.ptr & 7) == 0 && v.sW % 4 == 0 && v.sH % 4 == 0 && v.sD % 4 == 0 && v.sN % 4 == 0;
  };
  if (!al16(i0) || !aligned(o0)) return TEM_EUNSUPPORTED;
  Dev p{};
  p.in = U(i0.ptr); p.iN = (int)i0.sN; p.iD = (int)i0.sD; p.iH = (int)i0.sH; p.iW = (int)i0.sW;
  p.D = i0.D; p.H = i0.H; p.W = i0.W;
  p.out = const_cast<u16 *>(U(o0.ptr)); p.oN = (int)o0.sN; p.oD = (int)o0.sD; p.oH = (int)o0.sH; p.oW = (int)o0.sW;
  p.OD = o0.D; p.OH = o0.H; p.OW = o0.W;
  p.P = a->pd;
  const tem_epilogue &e = a->ep;
  Ep &q = p.ep;
  q.slope = e.slope; q.gate_slope = e.gate_slope;
  if (e.gate.ptr) {
    const tem_view &g = e.gate;
    if (g.N != o0.N || g.D != o0.D || g.H != o0.H || g.W != o0.W || g.C < o0.C) return TEM_ESHAPE;
    if (!fits32(g) || !aligned(g)) return TEM_EUNSUPPORTED;
    q.gate = U(g.ptr); q.gN = (int)g.sN; q.gD = (int)g.sD; q.gH = (int)g.sH; q.gW = (int)g.sW;
  }
  if (e.add.ptr) {
    const tem_view &ad = e.add;
    if (ad.C < o0.C || ad.N != o0.N) return TEM_ESHAPE;
    if (!fits32(ad) || !aligned(ad)) return TEM_EUNSUPPORTED;
    q.add = U(ad.ptr); q.aN = (int)ad.sN; q.aD = (int)ad.sD; q.aH = (int)ad.sH; q.aW = (int)ad.sW;
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
    auto span = [](const tem_view &v) {
      return (int64_t)(v.N - 1) * v.sN + (int64_t)(v.D - 1) * v.sD + (int64_t)(v.H - 1) * v.sH + (int64_t)(v.W - 1) * v.sW + v.C;
    };
    const int64_t melems = (int64_t)o0.N * q.dD * q.dH * q.dW * o0.C;
    if (melems >= ((int64_t)1 << 33)) return TEM_EUNSUPPORTED;
    q.mbytes = (int)((melems + 7) / 8);
    if ((e.gate.ptr && span(e.gate) >= ((int64_t)1 << 30)) || (e.add.ptr && span(e.add) >= ((int64_t)1 << 30)))
      return TEM_EUNSUPPORTED;                     // byte offsets of the epilogue's buffer loads stay below 2^31
    q.gbytes = e.gate.ptr ? (int)(span(e.gate) * 2) : 0;
    q.abytes = e.add.ptr ? (int)(span(e.add) * 2) : 0;
  }
  const int CI = i0.C, CO = o0.C, N = i0.N;
  int epm = 0;
  if (p.ep.dropout && p.ep.keep_mode == 2 && !p.ep.gate && !p.ep.add) epm = 1;
  else if (!p.ep.dropout && p.ep.gate) epm = 2;
#define CT_CASE(ci, co, pf, ncls) if (CI == ci && CO == co) return run<ci, co, pf, ncls>(p, N, U(a->w), st, dry, epm);
  CT_CASE(16, 8, 12, 1)     // g.u1b forward (Conv3DTranspose 16 -> 8)
  CT_CASE(32, 16, 12, 1)    // g.u2b forward
  CT_CASE(8, 8, 12, 1)      // input-gradient of g.d1b / d.d1b    (more classes per patch measured no faster)
  CT_CASE(16, 16, 12, 1)    // input-gradient of g.d2b
  CT_CASE(32, 32, 12, 1)    // input-gradient of d.d2b / d.d3b
#undef CT_CASE
  return TEM_EUNSUPPORTED;
}

}  // namespace convt_bf16

// bf16 mode of tem_conv_transpose (see tem_conv_bf16): `w` is the bf16 kernel [tap][co][ci].
extern "C" int tem_conv_transpose_bf16(const tem_conv_args *a, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!a || !tem_view_ok(a->in0) || !tem_view_ok(a->out0) || !a->w) return TEM_EINVAL;
  return convt_bf16::dispatch(a, (hipStream_t)stream, false);
}

extern "C" int tem_conv_transpose_bf16_describe(const tem_conv_args *a, char *buf, int32_t len) {
  if (!a || !tem_view_ok(a->in0) || !tem_view_ok(a->out0) || !a->w) return TEM_EINVAL;
  convt_bf16::g_name = buf; convt_bf16::g_name_len = len;
  int rc = convt_bf16::dispatch(a, nullptr, true);
  convt_bf16::g_name = nullptr;
  return rc;
}
