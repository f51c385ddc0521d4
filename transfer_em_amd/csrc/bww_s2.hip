// bww_s2.hip -- kernel gradient of the k4 s2 convolutions (Conv3DBackpropFilter of models/utils.py:80 and, with input and
// gradient swapped, of the Conv3DTranspose layers, models/utils.py:129-130) on the fp32 matrix cores, fragments
// straight from HBM/L2:
//
//   dW[(kz,ky,kx)][ci][co] = sum_v X[2 v + k - P][ci] * dY[v][co]
//
// as D[co][(tap, ci)] += A[co][v] * B[v][(tap, ci)] with the OUTPUT VOXELS as the reduction index of
// v_mfma_f32_16x16x4_f32 (a k-step = 4 x-consecutive voxels):
//   A: lane (co, voxel kq) reads dY[v][co] -- the 16 lanes of a voxel are one contiguous run;
//   B: for a row tap (kz, ky) the 4 x-taps of a voxel are ONE contiguous run of 4 C_in floats of X.  The assignment of
//      matrix columns to lanes is free, so lane n takes floats 4 n .. 4 n + 3 of that run (C_in 8: 2 n, 2 n + 1) as the
//      columns of four (two) n-tiles: ONE 16-byte (8-byte) load per lane feeds four (two) MFMAs per m-tile -- no LDS
//      image, no transposition, 5-6 loads per 8-32 MFMAs.
// A workgroup owns (a range of output rows, one kz): its 4 waves take the rows of the range in turn (all four ky taps
// each, C_in x ceil(C_out / 16) accumulator tiles per wave; a ring of 8 fragment sets keeps 7 k-steps of loads in flight), add
// their accumulators through LDS in wave order at
// the end and write the kz slice of ONE ordinary slab [tap][ci][co] per row range (deterministic; the four kz
// workgroups of a range fill its slab) -- 64-128 slabs per launch.
// Measured stand-alone (132^3 step shapes): g.bww.u2b 16 -> 32: 43.6 -> 22.8 us, g.bww.d2b 16 -> 16: 24.1 -> 19.2 us,
// g.bww.u1b 8 -> 16: 44.2 -> 42.5 us (cone 24.9 -> 20.8) against bww_lds_k.
#include "tem_common.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace bwws2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct Dev {
  const float *in;
  int32_t iN, iD, iH, iW, D, H, W, in_bytes;
  const float *g;
  int32_t gN, gD, gH, gW, OD, OH, OW, g_bytes;
  int32_t P;
  int32_t rows, R;                                  // output rows (n, oz, oy), row ranges (= slabs)

  uint32_t magicOH, magicOD;
  float *slabs;
  int64_t slab_stride;
};

constexpr int OOB = (int)0x80000000;

__device__ __forceinline__ uint32_t fdiv(uint32_t v, uint32_t d, uint32_t magic) { return d == 1 ? v : __umulhi(v, magic); }

// KS, S: kernel extent and stride -- 4, 2 (the layers this file was written for) or 3, 1 (the 3x3x3 layers under ~30^3 voxels,
// where the Winograd-domain kernel's prologue outweighs its gain: the run of a row tap is still 4 voxels = 4 C_in floats,
// the columns of the fourth voxel are computed and dropped).
template <int CI, int CO, int KS, int S>
__global__ __launch_bounds__(256) void bww_s2_k(Dev p) {
  constexpr int MT = (CO + 15) / 16;                // m-tiles (16 output channels each)
  constexpr int J = CI / 4;                         // n-tiles per row tap = floats per lane of its load (16 -> 4, 8 -> 2)
  constexpr bool WKY = CI == 32;                    // 32 input channels: 8 n-tiles per row tap -- a wave takes ONE ky (of every
                                                    // row of the range) instead of every fourth row with all four: its 16
                                                    // accumulator tiles are its own taps, no sum over the waves
  constexpr int NKY = WKY ? 1 : KS;                 // ky taps per wave
  constexpr int NACC = NKY * J * MT;                // accumulator tiles per wave: (ky, j, mt)
  constexpr int NBUF = 8;                           // fragment sets in flight
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int kz = blockIdx.y;
  if (WKY && wave >= KS) return;                    // (k 3: the fourth wave has no ky; no barrier on this path)
  const int ra = (int)(((long long)blockIdx.x * p.rows) / p.R), rb = (int)(((long long)(blockIdx.x + 1) * p.rows) / p.R);

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)p.g, 0, p.g_bytes, 0x00020000);

  f32x4 acc[NKY][J][MT];
#pragma unroll
  for (int ky = 0; ky < NKY; ++ky)
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[ky][j][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the lane's floats of a row tap's run: floats J n .. J n + J - 1 = input voxel 2 ox - P + (J n) / C_in, channels (J n) % C_in ..
  const int bvox = (J * n) / CI, bci = (J * n) & (CI - 1);

  struct Frag { float a[MT]; float b[NKY][J]; };
  int irow = WKY ? ra : ra + wave, iox0 = 0;                   // issue position: (row, first voxel of the k-step), wave-uniform
  // A single wave issues one instruction every 4 cycles while an MFMA holds the matrix pipe for 32: the bookkeeping of a
  // k-step must stay well under 7 instructions per MFMA or it sets the pace.  The row arithmetic (two divisions, plane / row
  // offsets and their range checks: ~100 instructions) is therefore kept as wave-uniform state and redone only when the
  // issue position enters a new row; a k-step adds the voxel's x terms.
  // Every load is issued unconditionally (one schedule for the compiler's vmcnt bookkeeping); lanes / steps with nothing to
  // read send an out-of-range offset and receive zeros, which add nothing.
  bool live = false, zyok[NKY];
  int rg = 0, ry[NKY];
  auto row_setup = [&]() {
    live = irow < rb;
    const int rowc = live ? irow : ra;
    const int t = (int)fdiv((uint32_t)rowc, (uint32_t)p.OH, p.magicOH), oy = rowc - t * p.OH;   // t = nb * OD + oz
    const int nb = (int)fdiv((uint32_t)t, (uint32_t)p.OD, p.magicOD), oz = t - nb * p.OD;
    rg = (nb * p.gN + oz * p.gD + oy * p.gH) * 4;
    const int iz = S * oz + kz - p.P;
    const bool zok = live && (unsigned)iz < (unsigned)p.D;
#pragma unroll
    for (int kyi = 0; kyi < NKY; ++kyi) {
      const int iy = S * oy + (WKY ? wave : kyi) - p.P;
      zyok[kyi] = zok && (unsigned)iy < (unsigned)p.H;
      ry[kyi] = (nb * p.iN + iz * p.iD + iy * p.iH) * 4;
    }
  };
  row_setup();
  const int Lg = (q * p.gW + n) * 4;                          // lane terms: gradient voxel q of the k-step, channel n (+ 16 mt)
  const int Lx = S * q - p.P + bvox, LxB = (Lx * p.iW + bci) * 4;   // input x of the lane's floats: S iox0 + Lx
  auto issue = [&](Frag &f) {
    const bool okx = live && iox0 + q < p.OW;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      int goff = (okx && 16 * mt + n < CO) ? rg + Lg + iox0 * p.gW * 4 + 64 * mt : OOB;
      asm volatile("" : "+v"(goff));                  // (opaque: the compiler would otherwise split the load into two exec-masked ones)
      f.a[mt] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(grs, goff, 0, 0));
    }
    const int x = S * iox0 + Lx;
    const bool okb = okx && (unsigned)x < (unsigned)p.W;
    const int xb = LxB + iox0 * (S * p.iW * 4);
#pragma unroll
    for (int kyi = 0; kyi < NKY; ++kyi) {
      int off = (okb && zyok[kyi]) ? xb + ry[kyi] : OOB;
      asm volatile("" : "+v"(off));
#pragma unroll
      for (int c = 0; c < J / 4 + (J < 4 ? 1 : 0); ++c) {          // 16-byte (C_in 8: 8-byte) pieces of the lane's J floats
        if constexpr (J >= 4) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrs, off + 16 * c, 0, 0);
          f.b[kyi][4 * c + 0] = __uint_as_float(v.x); f.b[kyi][4 * c + 1] = __uint_as_float(v.y);
          f.b[kyi][4 * c + 2] = __uint_as_float(v.z); f.b[kyi][4 * c + 3] = __uint_as_float(v.w);
        } else {
          const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xrs, off, 0, 0);
          f.b[kyi][0] = __uint_as_float(v.x); f.b[kyi][1] = __uint_as_float(v.y);
        }
      }
    }
    iox0 += 4;
    if (iox0 >= p.OW) { iox0 = 0; irow += WKY ? 1 : 4; row_setup(); }
  };
  auto consume = [&](const Frag &f) {
#pragma unroll
    for (int ky = 0; ky < NKY; ++ky)
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[ky][j][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[mt], f.b[ky][j], acc[ky][j][mt], 0, 0, 0);
  };

  // k-steps of this wave: its rows (ra + wave, + 4, ...) x ceil(OW / 4).  A k-step is only 8-32 MFMAs (0.1-0.4 us) while a
  // load takes 1-2 us to arrive: a ring of NBUF fragment sets keeps NBUF - 1 k-steps of loads in flight (5-6 VGPR-light
  // loads each; static ring indices, no copies).
  const int nrow_w = WKY ? rb - ra : (rb > ra + wave ? (rb - ra - wave + 3) >> 2 : 0);
  const int total = nrow_w * ((p.OW + 3) >> 2);
  // One k-step in the steady state: the set's MFMAs with the loads of its next use between them -- per ky the MFMAs that
  // read f.b[ky], then that register's reload; the gradient registers behind the last MFMA (sched_barrier pins it: in one
  // clump each, MFMAs and bookkeeping add up instead of overlapping).
  auto step = [&](Frag &f) {
    const bool okx = live && iox0 + q < p.OW;
    const int x = S * iox0 + Lx;
    const bool okb = okx && (unsigned)x < (unsigned)p.W;
    const int xb = LxB + iox0 * (S * p.iW * 4);
#pragma unroll
    for (int kyi = 0; kyi < NKY; ++kyi) {
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[kyi][j][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[mt], f.b[kyi][j], acc[kyi][j][mt], 0, 0, 0);
      int off = (okb && zyok[kyi]) ? xb + ry[kyi] : OOB;
      asm volatile("" : "+v"(off));
#pragma unroll
      for (int c = 0; c < J / 4 + (J < 4 ? 1 : 0); ++c) {
        if constexpr (J >= 4) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrs, off + 16 * c, 0, 0);
          f.b[kyi][4 * c + 0] = __uint_as_float(v.x); f.b[kyi][4 * c + 1] = __uint_as_float(v.y);
          f.b[kyi][4 * c + 2] = __uint_as_float(v.z); f.b[kyi][4 * c + 3] = __uint_as_float(v.w);
        } else {
          const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xrs, off, 0, 0);
          f.b[kyi][0] = __uint_as_float(v.x); f.b[kyi][1] = __uint_as_float(v.y);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      int goff = (okx && 16 * mt + n < CO) ? rg + Lg + iox0 * p.gW * 4 + 64 * mt : OOB;
      asm volatile("" : "+v"(goff));
      f.a[mt] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(grs, goff, 0, 0));
    }
    iox0 += 4;
    if (iox0 >= p.OW) { iox0 = 0; irow += WKY ? 1 : 4; row_setup(); }
    __builtin_amdgcn_sched_barrier(0);
  };
  Frag f[NBUF];
#pragma unroll
  for (int u = 0; u < NBUF; ++u) issue(f[u]);
  int t = 0;
  for (; t + NBUF <= total; t += NBUF) {            // full rounds of the ring: straight-line code
#pragma unroll
    for (int u = 0; u < NBUF; ++u) step(f[u]);
  }
#pragma unroll
  for (int u = 0; u < NBUF - 1; ++u)                // the last total % NBUF sets (already loaded; wave-uniform)
    if (t + u < total) consume(f[u]);

  // ---- sum over the waves through LDS (fixed order), then the kz slice of the range's slab:
  //   part[wave][tile (ky, j, mt)][lane] (16 bytes each)
  float *const slab0 = p.slabs + (size_t)blockIdx.x * p.slab_stride;
  if constexpr (WKY) {                              // the wave's own taps (kz, ky = wave): straight to the slab
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int fl = J * n + j, kx = fl / CI, ci = fl & (CI - 1);
        const int tap = (kz * KS + wave) * KS + kx, co = 16 * mt + 4 * q;
        if (co < CO && kx < KS) *reinterpret_cast<f32x4 *>(slab0 + (size_t)(tap * CI + ci) * CO + co) = acc[0][j][mt];
      }
    return;
  }
  f32x4 *const part = reinterpret_cast<f32x4 *>(lds);
#pragma unroll
  for (int ky = 0; ky < NKY; ++ky)
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) part[(wave * NACC + (ky * J + j) * MT + mt) * 64 + lane] = acc[ky][j][mt];
  __syncthreads();
  float *const slab = p.slabs + (size_t)blockIdx.x * p.slab_stride;
  constexpr int TPW = NACC / 4;                     // tiles finished by each wave
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int tl = wave * TPW + i;                  // (wave-uniform)
    const int mt = tl % MT, kj = tl / MT, j = kj % J, ky = kj / J;
    f32x4 s = part[tl * 64 + lane];
#pragma unroll
    for (int w2 = 1; w2 < 4; ++w2) {
      const f32x4 v = part[(w2 * NACC + tl) * 64 + lane];
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    // D row = output channel 16 mt + 4 q + r (r = register), column = float J n + j of the run: x-tap, input channel
    const int fl = J * n + j, kx = fl / CI, ci = fl & (CI - 1);
    const int tap = (kz * KS + ky) * KS + kx, co = 16 * mt + 4 * q;
    if (co < CO && kx < KS) *reinterpret_cast<f32x4 *>(slab + (size_t)(tap * CI + ci) * CO + co) = s;
  }
}

// ---- 8 -> 8 (g.d1b, d.d1b): two-block rows.  With 8 output channels half of an m-tile would be zeros; instead the rows are
// (co, s): the gradient voxel ox - s (s = 0, 1 along x) -- it meets the input voxels 2 ox + {0, 1} as the x-taps kx = {0, 1} + 2 s,
// so one k-step over the slots ox = 0 .. OW covers all four x-taps with full tiles and reads every input float ONCE per row
// tap.  The 16 floats (kx 0..1, ci) of a slot are contiguous; with the four ky taps they are the 64 columns of four n-tiles laid
// out so that a lane's 16-byte load (ky = n >> 2, floats 4 (n & 3) .. + 3) is its column of the tiles t = 0..3 (column n of tile
// t = (ky, kx0 = (n & 3) >> 1, ci = 4 (n & 1) + t)).  A wave carries all four kz (16 accumulator tiles): per k-step one 4-byte
// gradient load and four 16-byte input loads feed 16 MFMAs.  One workgroup per row range, wave sum through LDS, one slab.
template <int NBUF>
__global__ __launch_bounds__(256) void bww_s2tb_k(Dev p) {
  constexpr int CI = 8, CO = 8, KS = 4, S = 2, NACC = 16;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4;
  // row ranges that are neighbours along z read two of their four input planes in common: one XCD (one L2) works through
  // a contiguous run of ranges
  const unsigned bid = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int ra = (int)(((long long)bid * p.rows) / p.R), rb = (int)(((long long)(bid + 1) * p.rows) / p.R);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)p.g, 0, p.g_bytes, 0x00020000);

  f32x4 acc[KS][4];
#pragma unroll
  for (int kz = 0; kz < KS; ++kz)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[kz][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  struct Frag { float a; f32x4 b[KS]; };
  const int sa = n >> 3, coa = n & 7;               // A row of this lane: x-shift, output channel
  const int kyb = n >> 2, cg = n & 3;               // B columns of this lane: row tap, 4-float group of the slot's 16 floats
  // A single wave issues one instruction every 4 cycles and an MFMA holds the matrix pipe for 32: whatever is not an MFMA
  // has to fit in the 7 issue slots behind each of them, or it adds to the k-step (measured: 130 bookkeeping instructions per
  // k-step doubled it).  So the row arithmetic (two divisions, the plane / row offsets and their range checks) is kept as
  // state and redone only when the issue position enters a new row; a k-step adds the slot's x terms to it.
  const int qa = q - sa;                            // gradient voxel of slot s: s + qa
  const int Lg = (qa * p.gW + coa) * 4;
  const int Lx = S * q - p.P + (cg >> 1);           // input x of slot s: S s + Lx
  const int LxB = (Lx * p.iW + (cg & 1) * 4) * 4;
  int irow = ra + wave, iox0 = 0;                   // issue position (wave-uniform)
  int rg = 0, rz[KS] = {0, 0, 0, 0};                // row state, uniform: gradient row offset, input plane offsets (bytes)
  bool live = false, zok[KS] = {false, false, false, false};
  int Lrow = 0; bool yok = false;                   // row state, per lane (its ky): input row offset + lane terms, row in range
  auto row_setup = [&]() {
    live = irow < rb;
    const int rowc = live ? irow : ra;
    const int t = (int)fdiv((uint32_t)rowc, (uint32_t)p.OH, p.magicOH), oy = rowc - t * p.OH;
    const int nb = (int)fdiv((uint32_t)t, (uint32_t)p.OD, p.magicOD), oz = t - nb * p.OD;
    rg = (nb * p.gN + oz * p.gD + oy * p.gH) * 4;
    const int iy = S * oy + kyb - p.P;
    yok = live && (unsigned)iy < (unsigned)p.H;
    Lrow = (nb * p.iN + iy * p.iH) * 4 + LxB;
#pragma unroll
    for (int kz = 0; kz < KS; ++kz) {
      const int iz = S * oz + kz - p.P;
      zok[kz] = (unsigned)iz < (unsigned)p.D;
      rz[kz] = iz * p.iD * 4;
    }
  };
  row_setup();
  // One k-step on fragment set f: its 16 MFMAs (MM) with, in the issue slots behind them, the address arithmetic and the
  // loads of the set's next use (the slot NBUF k-steps ahead): a register of the set is reloaded right behind the last MFMA
  // that reads it.  sched_barrier pins the interleave.
  auto step = [&](Frag &f, auto mmtag) {
    constexpr bool MM = decltype(mmtag)::value;
    int goff = 0, offb = 0; bool okb = false;
#pragma unroll
    for (int kz = 0; kz < KS; ++kz) {
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        if constexpr (MM) acc[kz][t4] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a, f.b[kz][t4], acc[kz][t4], 0, 0, 0);
        if (kz == 0 && t4 == 0) {
          const int oxa = iox0 + qa;
          goff = (live && (unsigned)oxa < (unsigned)p.OW) ? rg + Lg + iox0 * p.gW * 4 : OOB;
          asm volatile("" : "+v"(goff));
        }
        if (kz == 0 && t4 == 1) {
          const int x = S * iox0 + Lx;
          okb = yok && iox0 + q <= p.OW && (unsigned)x < (unsigned)p.W;
          offb = Lrow + iox0 * (S * p.iW * 4);
        }
        if (t4 == 3) {
          int off = (okb && zok[kz]) ? offb + rz[kz] : OOB;
          asm volatile("" : "+v"(off));
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);
          f.b[kz] = f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
        }
        if constexpr (MM) __builtin_amdgcn_sched_barrier(0);
      }
    }
    f.a = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(grs, goff, 0, 0));
    iox0 += 4;
    if (iox0 > p.OW) { iox0 = 0; irow += 4; row_setup(); }
    if constexpr (MM) __builtin_amdgcn_sched_barrier(0);
  };
  const int nrow_w = rb > ra + wave ? (rb - ra - wave + 3) >> 2 : 0;
  const int total = nrow_w * ((p.OW + 4) >> 2);     // slots 0 .. OW in k-steps of 4
  Frag f[NBUF];
#pragma unroll
  for (int u = 0; u < NBUF; ++u) step(f[u], std::false_type{});
  int t = 0;
  for (; t + NBUF <= total; t += NBUF) {            // full rounds of the ring: straight-line code
#pragma unroll
    for (int u = 0; u < NBUF; ++u) step(f[u], std::true_type{});
  }
#pragma unroll
  for (int u = 0; u < NBUF - 1; ++u)                // the last total % NBUF sets (already loaded)
    if (t + u < total) {
#pragma unroll
      for (int kz = 0; kz < KS; ++kz)
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
          acc[kz][t4] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[u].a, f[u].b[kz][t4], acc[kz][t4], 0, 0, 0);
    }
  // wave sum through LDS (fixed order); wave w finishes the tiles of kz = w
  f32x4 *const part = reinterpret_cast<f32x4 *>(lds);
#pragma unroll
  for (int kz = 0; kz < KS; ++kz)
#pragma unroll
    for (int t = 0; t < 4; ++t) part[(wave * NACC + kz * 4 + t) * 64 + lane] = acc[kz][t];
  __syncthreads();
  float *const slab = p.slabs + (size_t)bid * p.slab_stride;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int tl = wave * 4 + t;
    f32x4 s = part[tl * 64 + lane];
#pragma unroll
    for (int w2 = 1; w2 < 4; ++w2) {
      const f32x4 v = part[(w2 * NACC + tl) * 64 + lane];
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    // D rows 4 q + r = (co = 4 (q & 1) + r, s = q >> 1); column n = (ky, kx0, channel group)
    const int kx = (cg >> 1) + 2 * (q >> 1), ci = (cg & 1) * 4 + t;
    const int tap = (wave * KS + kyb) * KS + kx;
    *reinterpret_cast<f32x4 *>(slab + (size_t)(tap * CI + ci) * CO + 4 * (q & 1)) = s;
  }
}

// ------------------------------------------------------------------------------------------ host
static uint32_t magic_for(int d) { return (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d); }

static int64_t span_of(const tem_view &v) {
  return (int64_t)(v.N - 1) * v.sN + (int64_t)(v.D - 1) * v.sD + (int64_t)(v.H - 1) * v.sH + (int64_t)(v.W - 1) * v.sW + v.C;
}

static thread_local char *g_name = nullptr;
static thread_local int g_name_len = 0;

template <int CI, int CO, int KS = 4, int S = 2>
static int run(Dev p, int max_slabs, hipStream_t st, bool dry, int *nslab_out) {
  constexpr int MT = (CO + 15) / 16, J = CI / 4, NACC = KS * J * MT;
  const size_t lds_bytes = CI == 32 ? 0 : (size_t)4 * NACC * 64 * 16;
  // row ranges: 128 (two workgroups per CU with the four kz) where the LDS sum leaves room for two, else 64; at least ~4
  // rows per wave
  // (32 input channels: a slab is 110-260 KB; 64 ranges x kz = one workgroup per CU already, and half the slab traffic of 128)
  int R = (lds_bytes <= 80 * 1024 && CI != 32) ? 128 : 64;
  static int rr = -1;
  if (rr < 0) rr = tem_env_int("TEM_BWW_S2_R", 0);
  if (rr > 0) R = rr;
  while (R > 1 && p.rows / R < (CI == 32 ? 2 : 8)) R >>= 1;     // (32 input channels: every wave walks all rows of the range)
  if (R > max_slabs) R = max_slabs;
  if (R < 1) return TEM_EUNSUPPORTED;
  p.R = R;
  if (nslab_out) *nslab_out = R;
  if (g_name) snprintf(g_name, g_name_len, "bww_s2_k<%d, %d, %d, %d>", CI, CO, KS, S);
  if (dry) return TEM_OK;
  static bool attr = false;
  if (!attr && lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)bww_s2_k<CI, CO, KS, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL((bww_s2_k<CI, CO, KS, S>), dim3((unsigned)R, KS), dim3(256), lds_bytes, st, p);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

static int run_tb(Dev p, int max_slabs, hipStream_t st, bool dry, int *nslab_out) {
  const size_t lds_bytes = (size_t)4 * 16 * 64 * 16;            // 64 KB: two workgroups per CU
  int R = 512;
  static int rr = -1;
  if (rr < 0) rr = tem_env_int("TEM_BWW_S2_TBR", 0);
  static int nb = -1;
  if (nb < 0) nb = tem_env_int("TEM_BWW_S2_TBN", 8);
  if (rr > 0) R = rr;
  else while (R > 1 && p.rows / R < 8) R >>= 1;                      // at least two rows per wave
  if (R > max_slabs) R = max_slabs;
  if (R < 1) return TEM_EUNSUPPORTED;
  p.R = R;
  if (nslab_out) *nslab_out = R;
  if (g_name) snprintf(g_name, g_name_len, "bww_s2tb_k<%d>", nb == 6 ? 6 : 8);
  if (dry) return TEM_OK;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void *)bww_s2tb_k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)bww_s2tb_k<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  if (nb == 6) hipLaunchKernelGGL(bww_s2tb_k<6>, dim3((unsigned)R), dim3(256), lds_bytes, st, p);
  else hipLaunchKernelGGL(bww_s2tb_k<8>, dim3((unsigned)R), dim3(256), lds_bytes, st, p);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

static int dispatch(const tem_bww_args *a, hipStream_t st, bool dry, int *nslab_out) {
  const tem_view &i0 = a->in0, &g = a->dout;
  if (a->in1.ptr) return TEM_EUNSUPPORTED;
  const bool k4s2 = a->kd == 4 && a->kh == 4 && a->kw == 4 && a->sd == 2 && a->sh == 2 && a->sw == 2;
  const bool k3s1 = a->kd == 3 && a->kh == 3 && a->kw == 3 && a->sd == 1 && a->sh == 1 && a->sw == 1;
  if (!k4s2 && !k3s1) return TEM_EUNSUPPORTED;
  if (a->pd != a->ph || a->ph != a->pw) return TEM_EUNSUPPORTED;
  static int enabled = -1;
  if (enabled < 0) enabled = tem_env_int("TEM_BWW_S2", 1);
  if (!enabled) return TEM_EUNSUPPORTED;
  if (g.N != i0.N) return TEM_ESHAPE;
  const int64_t ispan = span_of(i0), gspan = span_of(g);
  if (ispan >= ((int64_t)1 << 29) || gspan >= ((int64_t)1 << 29)) return TEM_EUNSUPPORTED;   // byte offsets below 2^31
  if (((uintptr_t)i0.ptr & 15) || i0.sW % 4 || i0.sH % 4 || i0.sD % 4 || i0.sN % 4) return TEM_EUNSUPPORTED;
  const int64_t rows = (int64_t)g.N * g.D * g.H;
  if (rows > (1 << 22) || g.H > 4096 || g.D > 4096) return TEM_EUNSUPPORTED;                 // range of the magic divisions
  const int CI = i0.C, CO = g.C;
  const int64_t stride = a->slab_stride ? a->slab_stride : (int64_t)(k4s2 ? 64 : 27) * CI * CO;
  if (!dry && (((uintptr_t)a->slabs & 15) || stride % 4)) return TEM_EUNSUPPORTED;           // 16-byte slab stores
  Dev p{};
  p.in = i0.ptr; p.iN = (int)i0.sN; p.iD = (int)i0.sD; p.iH = (int)i0.sH; p.iW = (int)i0.sW;
  p.D = i0.D; p.H = i0.H; p.W = i0.W; p.in_bytes = (int)(ispan * 4);
  p.g = g.ptr; p.gN = (int)g.sN; p.gD = (int)g.sD; p.gH = (int)g.sH; p.gW = (int)g.sW;
  p.OD = g.D; p.OH = g.H; p.OW = g.W; p.g_bytes = (int)(gspan * 4);
  p.P = a->pd;
  p.rows = (int)rows;
  p.magicOH = magic_for(g.H); p.magicOD = magic_for(g.D);
  p.slabs = a->slabs; p.slab_stride = stride;
  if (k4s2 && CI == 8 && CO == 8) {                 // g.d1b, d.d1b: two-block rows (with plain rows half of every m-tile is zeros: 73 us)
    static int tb = -1;
    if (tb < 0) tb = tem_env_int("TEM_BWW_S2_TB", 1);
    if (!tb) return TEM_EUNSUPPORTED;
    return run_tb(p, a->nslab, st, dry, nslab_out);
  }
  if (k3s1) {
    // the 3x3x3 layers the Winograd-domain kernel leaves alone (hip_ops.WINO_MIN_VOXELS): g.u2a, d.d3a
    static int k3 = -1;
    if (k3 < 0) k3 = tem_env_int("TEM_BWW_S2_K3", 1);
    if (!k3 || (int64_t)g.D * g.H * g.W > 40000) return TEM_EUNSUPPORTED;
    if (CI == 16 && CO == 32) return run<16, 32, 3, 1>(p, a->nslab, st, dry, nslab_out);
    if (CI == 32 && CO == 32) return run<32, 32, 3, 1>(p, a->nslab, st, dry, nslab_out);
    return TEM_EUNSUPPORTED;
  }
  if (CI == 8 && CO == 16) return run<8, 16>(p, a->nslab, st, dry, nslab_out);     // g.u1b (transposed conv: input and gradient swapped)
  if (CI == 16 && CO == 16) return run<16, 16>(p, a->nslab, st, dry, nslab_out);   // g.d2b
  if (CI == 16 && CO == 32) return run<16, 32>(p, a->nslab, st, dry, nslab_out);   // g.u2b
  if (CI == 32 && CO == 32) return run<32, 32>(p, a->nslab, st, dry, nslab_out);   // d.d2b, d.d3b
  return TEM_EUNSUPPORTED;
}

}  // namespace bwws2

// Called by tem_conv_bwd_weight (conv_bww.hip) ahead of the LDS-ring kernel.
int tem_bww_s2_try(const tem_bww_args *a, hipStream_t st, bool dry, int *nslab_out) { return bwws2::dispatch(a, st, dry, nslab_out); }

int tem_bww_s2_describe(const tem_bww_args *a, char *buf, int len) {
  bwws2::g_name = buf; bwws2::g_name_len = len;
  int n = 0;
  int rc = bwws2::dispatch(a, nullptr, true, &n);
  bwws2::g_name = nullptr;
  return rc == TEM_OK && n == a->nslab ? TEM_OK : TEM_EUNSUPPORTED;
}
