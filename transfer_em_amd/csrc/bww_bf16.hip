// bww_bf16.hip -- kernel gradient for the bf16 mixed-precision mode (BASELINE config 5): bf16 activations and
// gradients in, fp32 accumulation on the matrix cores, fp32 partial slabs out (the split-K finish, the gradient
// vector and Adam stay fp32 -- "fp32 master weights and accumulators").
//
//   dW[(tap,ci)][co] = sum over output voxels o of  X[o*S + tap - P][ci] * G[o][co]
//
// GEMM with M = (tap,ci) rows, N = co and K = voxels.  Both operands are "K-major" (a lane needs consecutive VOXELS
// of one channel) while memory and LDS are channels-last: ds_read_b64_tr_b16 -- the transposing LDS read of gfx950 --
// delivers exactly that: per 16-lane group a [4 voxels][16 channels] block, column-major, so one read per operand
// feeds v_mfma_f32_16x16x16_bf16 (k = 4 voxels per lane group, 16 per instruction) straight from the channels-last
// image.  A workgroup owns (n, a run of output planes, a band of output rows) and ALL accumulator tiles of its row
// group; they live in registers for the whole run, so each workgroup writes one deterministic fp32 slab
// (tem_reduce_slabs_multi finishes the sum, as in fp32 mode).  C_in == 1 (first layers, and the swapped form of the
// C_out == 1 layer) takes its A fragments with plain 2-byte reads: 4 consecutive voxels of one tap are contiguous.
#include "tem_common.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

int tem_bww_c1_bf16_try(const tem_bww_args *a, hipStream_t st, bool dry, int *nslab_out, char *name, int name_len);

namespace bww_bf16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// x / d for 0 <= x < 2^31 with magic = ceil(2^32 / d) (d == 1: the magic does not fit 32 bits)
__device__ __forceinline__ int fdiv(int x, int d, uint32_t magic) { return d == 1 ? x : (int)__umulhi((uint32_t)x, magic); }

struct Dev {
  const u16 *in0, *in1;
  int32_t i0N, i0D, i0H, i0W, i1N, i1D, i1H, i1W, C0;
  int32_t D, H, W;
  const u16 *g;
  int32_t gN, gD, gH, gW, OD, OH, OW, P;
  float *slabs;
  int64_t slab_stride;
  int32_t TY, nband, zsegs, zper;
  int32_t rows, colsR, colsA, OWp;            // X patch rows, loaded / allocated columns; G row length padded to 16
  uint32_t magicColsR, magicPlaneR, magicOW;
};

template <int CI, int CO, int K, int S, int PFX, int PFG, int MTG>
__global__ __launch_bounds__(256) void bww_bf16_k(Dev p) {
  constexpr int NTAP = K * K * K, ROWS = NTAP * CI, MT = (ROWS + 15) / 16, NT = (CO + 15) / 16;
  // C_out = 32 (two n-tiles): a wave multiplies each of its A fragments with BOTH n-tiles' B fragments -- 2 + TPW LDS reads per
  // 2 TPW MFMAs instead of 1 + 2 TPW (the transposing reads, not the matrix cores, set the kernel's pace); the four waves split the
  // m-tiles.  C_out <= 16: one n-tile, WPN = 4 waves on it.
  constexpr int NB = NT;                                   // n-tiles per wave
  constexpr int WPN = 4;                                   // waves per m-tile group
  constexpr int TPW = (MTG + WPN - 1) / WPN;               // m-tiles per wave (x NB accumulator tiles)
  // voxel pitches (bf16 elements): measured (tests/tools/lds_tr_probe.hip) the transposing read streams 171 B/clk/CU at voxel
  // pitches up to 32 bytes and 117 beyond -- the 8-byte pad of round 2 (40 bytes at 16 channels) cost a third of the rate
  constexpr int PITCH = CI >= 8 ? (CI <= 16 ? CI : CI + 4) : 1;
  constexpr int GP = CO <= 16 ? CO : CO + 4;
  constexpr int CPX = CI >= 8 ? CI / 8 : 1, CPG = CO / 8;  // 16-byte chunks per voxel
  static_assert(CO % 8 == 0 && (CI == 1 || CI % 8 == 0) && NT <= 2, "channel counts");
  extern __shared__ __attribute__((aligned(16))) u16 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g4 = lane >> 4, q = m >> 2, pq = m & 3;
  const int planeA = p.rows * p.colsA;
  u16 *Xs = lds;
  const int x_elems = (K * planeA * PITCH + 7) & ~7;
  u16 *Gs = lds + x_elems;
  const int g_elems = p.TY * p.OWp * GP + 16;

  int b = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int zseg = b % p.zsegs; b /= p.zsegs;
  const int band = b % p.nband;
  const int n = b / p.nband;
  const int grp = blockIdx.y;
  const int oy0 = band * p.TY, TYr = min(p.TY, p.OH - oy0);
  const int oz0 = zseg * p.zper, oz1 = min(p.OD, oz0 + p.zper);

  // zero the whole image once: padded columns / voxels are never written again and must stay zero (G) / finite (X)
  for (int i = tid; i < (x_elems + g_elems + 7) / 8; i += 256) reinterpret_cast<uint4 *>(lds)[i] = make_uint4(0u, 0u, 0u, 0u);

  // ---- this wave's accumulator tiles: m-tiles grp*MTG + wave + j*WPN, all n-tiles
  int aconst[TPW], adz[TPW];                               // fragment offset inside a plane, z tap of the lane's rows
  f32x4 acc[TPW][NB];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int mt = min(grp * MTG + wave + j * WPN, MT - 1);            // surplus tiles recompute the last one, never stored
    if constexpr (CI >= 8) {
      const int m0 = min(16 * mt + 4 * pq, ROWS - 4);                   // this lane addresses rows m0..m0+3 (4 channels of one tap)
      const int tap = m0 / CI, ci0 = m0 - tap * CI;
      const int dz = tap / (K * K), rem = tap - dz * (K * K), dy = rem / K, dx = rem - dy * K;
      adz[j] = dz;
      aconst[j] = (dy * p.colsA + dx) * PITCH + ci0 + (4 * g4 + q) * S * PITCH;
    } else {
      const int tap = min(16 * mt + m, NTAP - 1);                       // row m of the tile = one tap
      const int dz = tap / (K * K), rem = tap - dz * (K * K), dy = rem / K, dx = rem - dy * K;
      adz[j] = dz;
      aconst[j] = dy * p.colsA + dx + 4 * g4;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[j][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  int bconst[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) bconst[nb] = (4 * g4 + q) * GP + min(nb * 16 + 4 * pq, CO + 4 - 4);

  const int iy0 = oy0 * S - p.P;
  const int nk = p.OWp >> 4;
  // The K input planes of an output plane live in a RING of K LDS slots (plane counter c -> slot c % K): only the S planes that
  // are new to an output plane are fetched (round 2 re-loaded all K every time: 3-4x the bytes through L2 and the loader, which
  // -- not MFMAs or LDS reads -- was the kernel's time).  One pass of the loader = one plane; all of a pass's loads are issued
  // before its LDS writes.
  constexpr int PF1 = (PFX + 2) / 3;                         // loader chunks per thread and PLANE
  typedef typename std::conditional<(CI >= 8), uint4, u16>::type xchunk;
  auto issue_x = [&](int c, xchunk (&pf)[PF1]) {             // global loads of plane counter c: input plane (oz0 * S - P) + c
    const int iz = oz0 * S - p.P + c;
#pragma unroll
    for (int i = 0; i < PF1; ++i) {
      const int id = tid + i * 256;
      if constexpr (CI >= 8) {
        const int totalX = p.rows * p.colsR * CPX;
        const int vox = id / CPX, c8 = (id - vox * CPX) * 8;
        const int r = fdiv(vox, p.colsR, p.magicColsR), cx = vox - r * p.colsR;
        const int iy = iy0 + r, ix = cx - p.P;
        const bool ok = id < totalX && (unsigned)iz < (unsigned)p.D && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const u16 *src = c8 < p.C0 ? p.in0 + (n * p.i0N + iz * p.i0D + iy * p.i0H + ix * p.i0W + c8)
                                   : p.in1 + (n * p.i1N + iz * p.i1D + iy * p.i1H + ix * p.i1W + (c8 - p.C0));
        pf[i] = ok ? *reinterpret_cast<const uint4 *>(src) : make_uint4(0u, 0u, 0u, 0u);
      } else {
        const int totalX = p.rows * p.colsR;
        const int r = fdiv(id, p.colsR, p.magicColsR), cx = id - r * p.colsR;
        const int iy = iy0 + r, ix = cx - p.P;
        const bool ok = id < totalX && (unsigned)iz < (unsigned)p.D && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        pf[i] = ok ? p.in0[n * p.i0N + iz * p.i0D + iy * p.i0H + ix * p.i0W] : (u16)0;
      }
    }
  };
  auto commit_x = [&](int c, const xchunk (&pf)[PF1]) {      // ... into ring slot c % K
    u16 *dstp = Xs + (c % K) * planeA * PITCH;
#pragma unroll
    for (int i = 0; i < PF1; ++i) {
      const int id = tid + i * 256;
      if constexpr (CI >= 8) {
        if (id < p.rows * p.colsR * CPX) {
          const int vox = id / CPX, c8 = (id - vox * CPX) * 8;
          const int r = fdiv(vox, p.colsR, p.magicColsR), cx = vox - r * p.colsR;
          u16 *d = dstp + (r * p.colsA + cx) * PITCH + c8;       // 8-byte aligned
          *reinterpret_cast<uint2 *>(d) = make_uint2(pf[i].x, pf[i].y);
          *reinterpret_cast<uint2 *>(d + 4) = make_uint2(pf[i].z, pf[i].w);
        }
      } else {
        if (id < p.rows * p.colsR) {
          const int r = fdiv(id, p.colsR, p.magicColsR), cx = id - r * p.colsR;
          dstp[r * p.colsA + cx] = pf[i];
        }
      }
    }
  };
  auto issue_g = [&](int oz, uint4 (&pg)[PFG]) {
    const int totalG = TYr * p.OW * CPG;
#pragma unroll
    for (int i = 0; i < PFG; ++i) {
      const int id = tid + i * 256;
      const int vox = id / CPG, c = (id - vox * CPG) * 8;
      const int r = fdiv(vox, p.OW, p.magicOW), x = vox - r * p.OW;
      pg[i] = id < totalG ? *reinterpret_cast<const uint4 *>(p.g + (n * p.gN + oz * p.gD + (oy0 + r) * p.gH + x * p.gW + c))
                          : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto commit_g = [&](const uint4 (&pg)[PFG]) {
    const int totalG = TYr * p.OW * CPG;
#pragma unroll
    for (int i = 0; i < PFG; ++i) {
      const int id = tid + i * 256;
      if (id < totalG) {
        const int vox = id / CPG, c = (id - vox * CPG) * 8;
        const int r = fdiv(vox, p.OW, p.magicOW), x = vox - r * p.OW;
        u16 *d = Gs + (r * p.OWp + x) * GP + c;
        *reinterpret_cast<uint2 *>(d) = make_uint2(pg[i].x, pg[i].y);
        *reinterpret_cast<uint2 *>(d + 4) = make_uint2(pg[i].z, pg[i].w);
      }
    }
  };
  // prologue: the K planes and the G rows of the first output plane
  {
    // (all K planes and the G rows requested before the first LDS write: plane by plane -- issue, wait, write -- the prologue of
    // every workgroup was K + 1 serialized memory round trips, tests/tools/isa_serial_loads.py)
    xchunk pf[K][PF1];
    uint4 pg[PFG];
    __syncthreads();                                         // zero fill done
#pragma unroll
    for (int c = 0; c < K; ++c) issue_x(c, pf[c]);
    issue_g(oz0, pg);
#pragma unroll
    for (int c = 0; c < K; ++c) commit_x(c, pf[c]);
    commit_g(pg);
  }
  for (int oz = oz0; oz < oz1; ++oz) {
    __syncthreads();                                         // this plane's image is complete
    // the S new planes and the G rows of the NEXT output plane fly into registers under this plane's matrix work
    const bool more = oz + 1 < oz1;
    xchunk nf[S][PF1];
    uint4 ng[PFG];
    const int cn = (oz + 1 - oz0) * S + K - S;               // first new plane counter of the next output plane
    if (more) {
#pragma unroll
      for (int i = 0; i < S; ++i) issue_x(cn + i, nf[i]);
      issue_g(oz + 1, ng);
    }
    // ---- k-blocks of 16 voxels along x, row by row
    int xsl[TPW];                                            // + the ring slot of the lane's z tap
#pragma unroll
    for (int j = 0; j < TPW; ++j) xsl[j] = aconst[j] + (((oz - oz0) * S + adz[j]) % K) * planeA * PITCH;
    // (measured and rejected: all fragments of a k-block requested before its first MFMA plus the next k-block's ahead of them, two
    // register sets in turn -- 40 -> 43 us at 16 -> 16, 25 -> 35 at 16 -> 32 k4 s2; with a branch around the second MFMA group hipcc
    // shuttles every accumulator between AGPRs and VGPRs per MFMA: 53 us)
    for (int r = 0; r < TYr; ++r) {
      const u16 *xr = Xs + r * S * p.colsA * PITCH;
      const u16 *gr = Gs + r * p.OWp * GP;
      for (int kb = 0; kb < nk; ++kb) {
        s16x4 bfrag[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bfrag[nb] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(gr + bconst[nb] + kb * 16 * GP));
        s16x4 afrag[TPW];
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          if constexpr (CI >= 8) {
            afrag[j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(xr + xsl[j] + kb * 16 * S * PITCH));
          } else {
            const u16 *s = xr + xsl[j] + kb * 16;
            afrag[j] = s16x4{(short)s[0], (short)s[1], (short)s[2], (short)s[3]};
          }
        }
#pragma unroll
        for (int j = 0; j < TPW; ++j)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[j][nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(afrag[j], bfrag[nb], acc[j][nb], 0, 0, 0);
      }
    }
    if (more) {
      __syncthreads();                                       // every wave is done with the S oldest planes and the G rows
#pragma unroll
      for (int i = 0; i < S; ++i) commit_x(cn + i, nf[i]);
      commit_g(ng);
    }
  }

  // ---- one partial slab per workgroup column (grid.x); row groups (grid.y) write disjoint rows of it
  float *slab = p.slabs + (int64_t)blockIdx.x * p.slab_stride;
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int mtl = wave + j * WPN, mt = grp * MTG + mtl;
    if (mtl < MTG && mt < MT) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int co = nb * 16 + m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = mt * 16 + g4 * 4 + r;            // C/D map: row = 4*(lane>>4)+reg, col = lane&15
          if (row < ROWS && co < CO) slab[(int64_t)row * CO + co] = acc[j][nb][r];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ host
static uint32_t magic_for(int d) { return d <= 1 ? 0u : (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d); }
static thread_local char *g_name = nullptr;
static thread_local int g_name_len = 0;

template <int CI, int CO, int K, int S, int PFX, int PFG, int MTG>
int run(Dev p, int N, int max_slabs, hipStream_t st, bool dry, int *nslab_out) {
  constexpr int NTAP = K * K * K, MT = (NTAP * CI + 15) / 16;
  constexpr int PITCH = CI >= 8 ? (CI <= 16 ? CI : CI + 4) : 1, GP = CO <= 16 ? CO : CO + 4, CPX = CI >= 8 ? CI / 8 : 1, CPG = CO / 8;
  p.OWp = (p.OW + 15) & ~15;
  p.colsR = (p.OW - 1) * S + K;
  p.colsA = (p.OWp - 1) * S + K + 4;                      // the last k-block reads up to OWp voxels (+ C_in == 1: 4-voxel reads)
  int TY = 0;
  size_t lds_bytes = 0;
  for (int ty = 1; ty <= 8 && ty <= p.OH; ++ty) {
    const int rows = (ty - 1) * S + K;
    const size_t xel = (((size_t)K * rows * p.colsA * PITCH) + 7) & ~(size_t)7, gel = (size_t)ty * p.OWp * GP + 16;
    const size_t bytes = ((xel + gel) * 2 + 15) & ~(size_t)15;
    // two workgroups per CU (<= 72 KB each) where the patch allows; a single row band may take up to 120 KB
    if ((size_t)rows * p.colsR * CPX > (size_t)((PFX + 2) / 3) * 256 || (size_t)ty * p.OW * CPG > (size_t)PFG * 256 ||
        bytes > (ty == 1 ? 120 : 72) * 1024) break;
    TY = ty; lds_bytes = bytes;
  }
  if (TY < 1) return TEM_EUNSUPPORTED;
  p.TY = TY; p.rows = (TY - 1) * S + K;
  p.nband = (p.OH + TY - 1) / TY;
  const int cols = N * p.nband;
  // Workgroup budget.  With the plane ring and the register prefetch a workgroup overlaps its own fetches, and every workgroup costs a
  // slab (written here, read again by reduce_multi_k); the budget counts WORKGROUPS -- a layer whose rows are split over NGRP row
  // groups gets budget / NGRP slabs (the 32 -> 32 4x4x4 layer of the discriminators wrote 200 slabs of 262 KB per call, 52 MB
  // against 4 MB of operands).  bf16 step by budget, kernel gradients still in the dependent chains: 512: 4.64 ms, 384: 4.47, 256:
  // 4.38, 192: 4.34, 128: 4.49; since they run beside the chains (cgan.py) their CU-time and slab bytes count, not their latency:
  // 128 for slabs above 32 KB, 256 below: 4.03 -> 3.97 ms (96: 4.09; 128 for all: 4.12; the 32 KB line at 16 / 64 KB: 4.02 / 4.20).
  constexpr int NGRP = (MT + MTG - 1) / MTG;
  const bool small_slab = NTAP * CI * CO * 4 <= tem_env_int("TEM_BWWH_SMALLKB", 32) * 1024;
  const int want_knob = (small_slab ? tem_env_int("TEM_BWWH_SMALL", 256) : tem_env_int("TEM_BWWH_WANT", 128)) / NGRP;
  int want = max_slabs < want_knob ? max_slabs : want_knob;
  int zsegs = want / cols;
  if (zsegs < 1) zsegs = 1;
  if (zsegs > p.OD) zsegs = p.OD;
  p.zper = (p.OD + zsegs - 1) / zsegs;
  p.zsegs = (p.OD + p.zper - 1) / p.zper;
  const int nblocks = cols * p.zsegs;
  if (nblocks > max_slabs) return TEM_EUNSUPPORTED;
  if (nslab_out) *nslab_out = nblocks;
  p.magicColsR = magic_for(p.colsR);
  p.magicPlaneR = magic_for(p.rows * p.colsR);
  p.magicOW = magic_for(p.OW);
  if (dry) {
    if (g_name) snprintf(g_name, g_name_len, "bww_bf16_k<%d, %d, %d, %d, %d, %d, %d>", CI, CO, K, S, PFX, PFG, MTG);
    return TEM_OK;
  }
  auto kern = bww_bf16_k<CI, CO, K, S, PFX, PFG, MTG>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks, NGRP), dim3(256), lds_bytes, st, p);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

int dispatch(const tem_bww_args *a, hipStream_t st, bool dry, int *nslab_out) {
  const tem_view &i0 = a->in0, &g = a->dout;
  {                                                          // one input channel, 3x3x3: the matrix-core march of bww_c1.hip
    const int rc = tem_bww_c1_bf16_try(a, st, dry, nslab_out, g_name, g_name_len);
    if (rc != TEM_EUNSUPPORTED) return rc;
  }
  if (!(a->kd == a->kh && a->kh == a->kw && a->sd == a->sh && a->sh == a->sw && a->pd == a->ph && a->ph == a->pw))
    return TEM_EUNSUPPORTED;
  if (g.N != i0.N) return TEM_ESHAPE;
  auto U = [](const float *q) { return reinterpret_cast<const u16 *>(q); };
  auto span_ok = [](const tem_view &v) {
    int64_t span = (int64_t)(v.N - 1) * v.sN + (int64_t)(v.D - 1) * v.sD + (int64_t)(v.H - 1) * v.sH + (int64_t)(v.W - 1) * v.sW + v.C;
    return span < ((int64_t)1 << 31);
  };
  auto al16 = [](const tem_view &v) {
    return v.C % 8 != 0 || (((uintptr_t)v.ptr & 15) == 0 && v.sW % 8 == 0 && v.sH % 8 == 0 && v.sD % 8 == 0 && v.sN % 8 == 0);
  };
  if (!span_ok(i0) || !span_ok(g) || !al16(i0) || !al16(g) || g.C % 8) return TEM_EUNSUPPORTED;
  Dev p{};
  p.in0 = U(i0.ptr); p.i0N = (int)i0.sN; p.i0D = (int)i0.sD; p.i0H = (int)i0.sH; p.i0W = (int)i0.sW; p.C0 = i0.C;
  p.in1 = p.in0; p.i1N = p.i0N; p.i1D = p.i0D; p.i1H = p.i0H; p.i1W = p.i0W;
  int CI = i0.C;
  if (a->in1.ptr) {
    const tem_view &i1 = a->in1;
    if (i1.N != i0.N || i1.D != i0.D || i1.H != i0.H || i1.W != i0.W) return TEM_ESHAPE;
    if (!span_ok(i1) || !al16(i1) || i0.C % 8 || i1.C % 8) return TEM_EUNSUPPORTED;
    p.in1 = U(i1.ptr); p.i1N = (int)i1.sN; p.i1D = (int)i1.sD; p.i1H = (int)i1.sH; p.i1W = (int)i1.sW;
    CI += i1.C;
  }
  p.D = i0.D; p.H = i0.H; p.W = i0.W;
  p.g = U(g.ptr); p.gN = (int)g.sN; p.gD = (int)g.sD; p.gH = (int)g.sH; p.gW = (int)g.sW;
  p.OD = g.D; p.OH = g.H; p.OW = g.W; p.P = a->pd;
  p.slabs = a->slabs;
  const int CO = g.C, K = a->kd, S = a->sd, N = i0.N;
  p.slab_stride = a->slab_stride ? a->slab_stride : (int64_t)K * K * K * CI * CO;
  const int max_slabs = a->nslab;
#define BW(ci, co, k, s, pfx, pfg, mtg) \
  if (CI == ci && CO == co && K == k && S == s) return run<ci, co, k, s, pfx, pfg, mtg>(p, N, max_slabs, st, dry, nslab_out);
  //  CI  CO  K  S  X-chunks  G-chunks  m-tiles per row group (all of them unless the accumulators would not fit)
  BW(1, 8, 3, 1, 12, 4, 2)  BW(1, 16, 3, 1, 12, 4, 2)
  BW(8, 8, 3, 1, 12, 4, 14) BW(8, 16, 3, 1, 12, 4, 14) BW(16, 8, 3, 1, 12, 4, 27) BW(16, 16, 3, 1, 12, 4, 27)
  BW(16, 32, 3, 1, 12, 4, 27) BW(32, 16, 3, 1, 12, 4, 54) BW(32, 32, 3, 1, 12, 4, 27)
  BW(8, 8, 4, 2, 12, 4, 32) BW(16, 16, 4, 2, 12, 4, 64) BW(8, 16, 4, 2, 12, 4, 32) BW(16, 32, 4, 2, 12, 4, 32)
  BW(32, 32, 4, 2, 12, 4, 32)
  BW(32, 32, 1, 1, 12, 4, 2) BW(1, 32, 1, 1, 12, 4, 1)      // 1x1 head; C_out == 1 in the swapped form
#undef BW
  return TEM_EUNSUPPORTED;
}

}  // namespace bww_bf16

// bf16 mode of tem_conv_bwd_weight: in0 / in1 / dout are bf16 views, the partial slabs stay float32.
// The launch writes exactly tem_conv_bwd_weight_bf16_nslab(a) slabs.
extern "C" int tem_conv_bwd_weight_bf16(const tem_bww_args *a, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!a || !tem_view_ok(a->in0) || !tem_view_ok(a->dout) || !a->slabs || a->nslab < 1 || a->accumulate) return TEM_EINVAL;
  int n = 0;
  int rc = bww_bf16::dispatch(a, nullptr, true, &n);
  if (rc != TEM_OK) return rc;
  if (n != a->nslab) return TEM_EINVAL;                    // size the workspace with tem_conv_bwd_weight_bf16_nslab
  return bww_bf16::dispatch(a, (hipStream_t)stream, false, nullptr);
}

extern "C" int tem_conv_bwd_weight_bf16_nslab(const tem_bww_args *a, char *name, int32_t name_len) {
  if (!a || !tem_view_ok(a->in0) || !tem_view_ok(a->dout) || a->nslab < 1) return TEM_EINVAL;
  int n = 0;
  bww_bf16::g_name = name; bww_bf16::g_name_len = name_len;
  int rc = bww_bf16::dispatch(a, nullptr, true, &n);
  bww_bf16::g_name = nullptr;
  return rc == TEM_OK ? n : rc;
}
