// wino_bww.hip -- kernel gradient of the 3x3x3 stride-1 layers with 16 output channels in the Winograd F(2x2, 3x3)
// domain of the (y, x) axes (the form wino.hip computes the layer in):
//
//   Y[z] = sum_kz A^T [ U[kz] (.) V[z + kz] ] A,   U[kz] = G g[kz] G^T,   V = B^T d B
//   =>  dU[kz]_p (ci, co) = sum_tiles V_p[z + kz][tile][ci] * Z_p[z][tile][co],   Z = A dY A^T   (2x2 -> 4x4)
//       dg[kz] = G^T dU[kz] G                                                                     (16 points -> 3x3 taps)
//
// 48 products per (ci, co) and 2x2 output tile instead of 27 * 4: 2.25x fewer MFMA flops than the direct form.
//
// GEMM per Winograd point p: D[(kz, ci)][co] += A[(kz, ci)][4 tiles] * B[4 tiles][co]   (v_mfma_f32_16x16x4_f32, K = tiles).
//   A operand map: lane = (row = lane & 15, k = lane >> 4) = one ((kz, ci), tile) pair per lane: the lane reads the 4x4 raw
//   voxels of ITS pair from the LDS image of input plane z + kz and transforms them in registers -- the transformed values
//   are the A fragments.  B operand map: lane = (co = lane & 15, tile): the lane loads the 2x2 gradient voxels of its
//   (tile, co) from global memory (64-byte channel runs), expands them to the 4x4 points; Z does not depend on kz, so the
//   z taps ride in the M dimension (C_in = 16: one 16-row tile per tap; C_in = 8: taps 0,1 share a tile).
//   All accumulators (taps x 8 points x 4 registers) stay in registers for the whole run of the workgroup.
//
// Workgroup = 8 waves: four 16-tile subsets of the block x two halves of the 16 points (rows py 0,1 | 2,3 of the
// point grid).  Data movement as wino.hip: BY x BX tiles, z march over a ring of 4 input planes filled by LDS-DMA.
// At the end the four tile subsets are summed through LDS in a fixed order (deterministic) and the workgroup applies
// G^T . G to its partial sums itself: what it writes is an ordinary [27][ci][co] slab, summed over the workgroups by
// tem_reduce_slabs_multi like the slabs of the direct-form kernel.
//
// Reference call sites: Conv3DBackpropFilter of Conv3D(filters, 3) (models/utils.py:73,122; generator.py:96).
#include "tem_common.h"
#include "wino_common.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace wino {

// device-side diagnostics (phase stamps, ablation flags) exist only in -DTEM_DEBUG_KNOBS builds: in the shipped kernels they
// cost registers (18 VGPRs of stamp sums) and scalar branches inside the step loop
#ifdef TEM_DEBUG_KNOBS
#define KDBG(x) (x)
#else
#define KDBG(x) 0
#endif


struct BDev {
  const float *in0, *in1;
  int32_t i0N, i0D, i0H, i0W, i1N, i1D, i1H, i1W;
  int32_t C0;
  int32_t N, D, H, W;
  const float *dy;
  int32_t dN, dD, dH, dW;
  int32_t OD, OH, OW;
  int32_t P;
  int32_t BY, BX, nby, nbx, zsegs, zper, NTZ;
  int32_t E, PLC, subb, slotb, ndma, span0, span1;
  uint32_t magicBX, magicE;
  float *slabs; int64_t slab_stride;   // one kernel-gradient slab [27][ci][co] per workgroup
  int32_t dbg;
};

// PAIR (8 -> 8 channels): 8 rows / columns would leave three quarters of every MFMA tile empty.  Instead the rows are
// (2 input planes x 8 ci) and the columns (2 output planes x 8 co): the four 8x8 blocks of one tile are the products of
// input plane pl = 2 i + s' (type A) or 2 i + 2 + s' (type B) with output plane zo = 2 i + s, i.e. tap kz = pl - zo:
//   A: (s',s) = (0,0) kz 0 | (1,0) kz 1 | (1,1) kz 0 | (0,1) kz -1 (unused);   B: (0,0) kz 2 | (0,1) kz 1 | (1,1) kz 2 | (1,0) unused
// -- every (plane, tap) pair exactly once, three quarters of each tile useful (the blocks are added up at the end).
// 32 channels on either side: the layer is CIH x NB independent (16-channel input half, 16-channel output block)
// problems over the same planes -- a wave pair (the two point halves) owns one such combination and fewer tile subsets
// remain (NS = 4 / (CIH NB)); 32 input channels: 32 tiles per workgroup (the ring of a 32-channel input is twice as big).
// EE: compile-time row pitch of the LDS image (voxels per (row, x parity); 9: tile blocks up to 8 wide, 17: up to 16), as in
// wino.hip: raw reads become lane bases + immediates.
template <int CI, int CO, int NI, bool PAIR, int EE>
__global__ __launch_bounds__(512) void wino_bww_k(BDev p) {
  constexpr int NH = CI / 8, VB = 32;
  constexpr int MT = CI >= 16 ? 3 : 2;                       // accumulator tiles per point: (tap, ci) row tiles, or types A / B
  constexpr int NZO = PAIR ? 1 : 2;                          // output planes handled one after the other per step
  constexpr int CIH = CI >= 16 ? CI / 16 : 1, NB = PAIR ? 1 : CO / 16, NC = CIH * NB, NS = 4 / NC;
  constexpr int TB = CI == 32 ? 32 : 64;                     // tiles per workgroup
  constexpr int JW = TB / NS / 8;                            // tile-pair rounds per wave and plane (8 tiles each)
  constexpr bool PREF = JW <= 2;                             // gradient voxels of the next plane prefetched (register budget)
  static_assert(!PAIR || (CI == 8 && CO == 8), "plane-pair form: 8 -> 8 channels");
  static_assert(JW == 2 || JW == 4, "tile rounds per wave");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char *const ring = reinterpret_cast<char *>(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, q = lane >> 4;
  const int ph = wave & 1;                                   // point rows py = 2 ph, 2 ph + 1
  const int cmb = (wave >> 1) % NC, sub = (wave >> 1) / NC;  // (input half, output block) combination; tile subset of the block
  const int cih = cmb % CIH, nb = cmb / CIH;

  int seg = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int zseg = seg % p.zsegs; seg /= p.zsegs;
  const int bx = seg % p.nbx; seg /= p.nbx;
  const int by = seg % p.nby;
  const int n = seg / p.nby;
  const int tz0 = zseg * p.zper, tz1 = min(p.NTZ, tz0 + p.zper), nsteps = (KDBG(p.dbg & 128)) ? 0 : tz1 - tz0;
  const int oy0 = by * 2 * p.BY, ox0 = bx * 2 * p.BX;
  const int ntile = p.BY * p.BX;

  // ---- LDS-DMA of the input planes (as wino.hip; the sub-images are 64 bytes apart modulo 128 here, see below)
  const bool two_in = p.in1 != p.in0;
  const float *const in0n = p.in0 + (size_t)n * p.i0N, *const in1n = p.in1 + (size_t)n * p.i1N;
  // (chunk offsets inside a plane: computed once and held -- see wino.hip; per step they cost ~40 vector instructions per
  // chunk, and vector instructions are matrix-pipe time on this chip)
  int voff0[NI], voff1[NI];
  {
    const int iy0 = oy0 - p.P, ix0 = ox0 - p.P;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int sp = (wave + 8 * i) * 64 + lane;
      const bool ex = sp < p.PLC;
      const int spc = ex ? sp : 0;
      const int cpos = spc & 1, ve = spc >> 1;
      const int ro = ve / EE, e = ve - ro * EE;
      const int o = ro & 1, yr = ro >> 1;
      const int c = cpos * 4;                                // no chunk swizzle here: see the bank note below
      const int iy = iy0 + yr, ix = ix0 + 2 * e + o;
      const bool ok = ex && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      voff0[i] = ok ? (iy * p.i0H + ix * p.i0W + c) * 4 : (int)0x80000000;
      voff1[i] = ok ? (iy * p.i1H + ix * p.i1W + c) * 4 : (int)0x80000000;
      asm volatile("" : "+v"(voff0[i]), "+v"(voff1[i]));
    }
  }
  auto dma_plane = [&](int iz, int slot) {
    const bool zok = (unsigned)iz < (unsigned)p.D;
    const int izc = zok ? iz : 0;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      char *const dst = ring + slot * p.slotb + h * p.subb;
      if (!two_in || 8 * h < p.C0)
        dma_subimage<NI, (CI <= 16)>(in0n + izc * p.i0D + 8 * h, zok ? p.span0 - 32 * h : 0, voff0, dst, wave, p.ndma);
      else
        dma_subimage<NI, (CI <= 16)>(in1n + izc * p.i1D + (8 * h - p.C0), zok ? p.span1 - 4 * (8 * h - p.C0) : 0, voff1, dst, wave, p.ndma);
    }
  };
  const int izb0 = 2 * tz0 - p.P;
  if (nsteps > 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_plane(izb0 + k, k);
  }

  // ---- per-lane roles.  A: row m = (tap, ci) of M tile mt, tile 4 ks + q of the subset; B: channel m, same tile.
  // LDS image as wino.hip (8-channel sub-images, rows split into even-x / odd-x voxels) but without the chunk XOR, and
  // sub-images / ring slots start 32 bytes past a multiple of 128 (subb, slotb = 32 mod 128): a half-wave's read = 16
  // rows (channels 0..7 | 8..15, or taps 0 | 1 for C_in 8) x the tiles of lane groups q and q + 1 (two tiles = 64 bytes
  // apart) then covers the 32 banks exactly once.
  const int ciA = CI >= 16 ? 16 * cih + m : (m & 7);          // the lane's input channel (A rows)
  constexpr int rowb = EE * VB;
  // tile pair j of the lane: tiles t0 = 8 JW sub + 8 j + 2 q (k-step 2 j, the .x of the packed values) and t0 + 1 (k-step
  // 2 j + 1, .y) -- x neighbours (BX is even): one ds_read2_b32 fetches a raw value of both
  int abase[JW], dybase[JW];                                // byte offset of t0's raw origin inside a plane image; t0's gradient voxel (0,0)
  uint32_t dyok = 0;                                         // bit 8 j + 4 i + o4: gradient voxel o4 of tile t0 + i exists
#pragma unroll
  for (int j = 0; j < JW; ++j) {
    const int t = sub * (8 * JW) + 8 * j + 2 * q;
    const int tc = min(t, ntile - 2);
    const int ty = (int)fdiv((uint32_t)tc, (uint32_t)p.BX, p.magicBX), tx = tc - ty * p.BX;
    abase[j] = (ciA >> 3) * p.subb + (4 * ty * EE + tx) * VB + (ciA & 7) * 4;
    const int oy = oy0 + 2 * ty, ox = ox0 + 2 * tx;
    dybase[j] = oy * p.dH + ox * p.dW + (PAIR ? (m & 7) + (m >> 3) * p.dD : 16 * nb + m);   // PAIR: column block = output plane
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int o4 = 0; o4 < 4; ++o4)
        dyok |= ((t + i < ntile && oy + (o4 >> 1) < p.OH && ox + 2 * i + (o4 & 1) < p.OW) ? 1u : 0u) << (8 * j + 4 * i + o4);
  }
  const float *const dyn = p.dy + (size_t)n * p.dN;

  f32x4 acc[MT][8];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // The gradient voxels come through a buffer descriptor over this sample's dY: a voxel that does not exist (tile over-hang,
  // plane past the tensor) is an out-of-range offset and arrives as zero.  [As plain loads with a select on the VALUE
  // (`ok ? v : 0`) every load was followed by s_waitcnt vmcnt(0) -- the select sits right behind it -- i.e. 16 serialized
  // HBM latencies per plane: 36 % of the kernel's time.]
  const __amdgpu_buffer_rsrc_t dyrs = __builtin_amdgcn_make_buffer_rsrc(
      (void *)dyn, 0, ((p.OD - 1) * p.dD + (p.OH - 1) * p.dH + (p.OW - 1) * p.dW + CO) * 4, 0x00020000);
  auto load_dy = [&](f32x2 (&g)[JW][4], int oz) {            // the lane's 2x2 gradient voxels of its 2 JW tiles, plane oz
    const bool zok = oz + (PAIR ? (m >> 3) : 0) < p.OD;      // (PAIR: plane oz + column block)
#pragma unroll
    for (int j = 0; j < JW; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int o4 = 0; o4 < 4; ++o4) {
          const bool ok = zok && ((dyok >> (8 * j + 4 * i + o4)) & 1u);
          int off = ok ? (oz * p.dD + dybase[j] + (o4 >> 1) * p.dH + (2 * i + (o4 & 1)) * p.dW) * 4 : (int)0x80000000;
          asm volatile("" : "+v"(off));                      // (opaque: one load, not two exec-masked ones)
          const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dyrs, off, 0, 0));
          if (i) g[j][o4].y = v;
          else g[j][o4].x = v;
        }
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the prologue's planes (inline-assembly DMA for C_in <= 16, wino_common.h; at 32 it measured 7-10 % slower)
  __syncthreads();

  // The point half ph is wave-uniform; the loop body is compiled once per value (static row indices: no selects).
  auto load_dy_pair = [&](f32x2 (&g)[4], int oz, int j) {    // ... of tile pair j only
    const bool zok = oz + (PAIR ? (m >> 3) : 0) < p.OD;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int o4 = 0; o4 < 4; ++o4) {
        const bool ok = zok && ((dyok >> (8 * j + 4 * i + o4)) & 1u);
        int off = ok ? (oz * p.dD + dybase[j] + (o4 >> 1) * p.dH + (2 * i + (o4 & 1)) * p.dW) * 4 : (int)0x80000000;
        asm volatile("" : "+v"(off));
        const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dyrs, off, 0, 0));
        if (i) g[o4].y = v;
        else g[o4].x = v;
      }
  };
  auto run = [&](auto phc) {
    constexpr int PH = decltype(phc)::value;
    // 2x2 gradient voxels of the lane's tile pairs (k-steps 2 j, 2 j + 1): two register sets used in turn, the set of the
    // NEXT plane (PREF) / tile pair (!PREF) is in flight under the current one's MFMAs -- every load is issued
    // unconditionally (a plane past the run is out of range and moves nothing) and never copied, so the only waits are the
    // counted ones in front of the consuming blocks
    f32x2 gA[PREF ? JW : 1][4], gB[PREF ? JW : 1][4];
    if constexpr (PREF) load_dy(gA, nsteps > 0 ? 2 * tz0 : p.OD);
    for (int step = 0; step < nsteps; ++step) {
      const int tz = tz0 + step, izb = 2 * tz - p.P;
      const bool more = step + 1 < nsteps;
      const int sA = (step & 1) ? 2 : 0;
      // one (accumulator tile mt, tile pair j) block of output plane zo: 3 raw rows x 4 of both tiles, transforms, 16 MFMAs
      auto block = [&](int zo, int mt, int j, const f32x2 (&gy)[4]) {
        // input plane of this lane's rows: tap kz = mt (C_in >= 16) or 2 mt + (m >> 3) (C_in 8; rows 8..15 of tile 1 repeat tap 2)
        const int kzl = CI >= 16 ? mt : min(2, 2 * mt + (m >> 3));
        const int pl = PAIR ? 2 * mt + (m >> 3) : zo + kzl;    // PAIR: type A rows read planes 0 | 1, type B planes 2 | 3
        const char *plane = ring + ((pl < 2 ? sA : 2 - sA) + (pl & 1)) * p.slotb;
        // raw rows PH .. PH + 2 of the 4x4 tile (the two point rows of this half need three raw rows)
        f32x2 v[3][4];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int dx = 0; dx < 4; ++dx) {
            const int off = (2 * (PH + i) + (dx & 1)) * rowb + (dx >> 1) * VB;     // row (yr, o) of the image, x half
            const float *src = reinterpret_cast<const float *>(plane + abase[j] + off);
            v[i][dx] = f32x2{src[0], src[VB / 4]};                                  // tiles t0, t0 + 1: one ds_read2_b32
          }
        if (PAIR ? (mt == 0 && j == JW - 1) : (zo == 1 && mt == 0 && j == JW - 1)) {
          // every wave is past the step's planes 0 and 1 (zo = 1 starts at plane 1 = its tap 0): they make room for
          // the next step's planes 2, 3
          __syncthreads();
          if (more && !(KDBG(p.dbg & 4))) { dma_plane(izb + 4, sA); dma_plane(izb + 5, sA + 1); }
        }
        // B^T on y for point rows 2 PH, 2 PH + 1, then on x
        f32x2 vp[2][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          vp[0][c] = PH ? v[1][c] - v[0][c] : v[0][c] - v[2][c];
          vp[1][c] = PH ? v[0][c] - v[2][c] : v[1][c] + v[2][c];
        }
        bt4(vp[0][0], vp[0][1], vp[0][2], vp[0][3]);
        bt4(vp[1][0], vp[1][1], vp[1][2], vp[1][3]);
        // Z = A dY A^T for the lane's (co, tile pair), the same two point rows (recomputed per tap: cheaper than
        // keeping it live)
        f32x2 zv[2][4];
        {
          const f32x2 y00 = gy[0], y01 = gy[1], y10 = gy[2], y11 = gy[3];
          const f32x2 a0 = PH ? y00 - y10 : y00, b0 = PH ? y01 - y11 : y01;         // point row 2 PH
          const f32x2 a1 = PH ? -y10 : y00 + y10, b1 = PH ? -y11 : y01 + y11;       // point row 2 PH + 1
          zv[0][0] = a0; zv[0][1] = a0 + b0; zv[0][2] = a0 - b0; zv[0][3] = -b0;
          zv[1][0] = a1; zv[1][1] = a1 + b1; zv[1][2] = a1 - b1; zv[1][3] = -b1;
        }
#pragma unroll
        for (int pt = 0; pt < 8; ++pt)
          acc[mt][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[pt >> 2][pt & 3].x, zv[pt >> 2][pt & 3].x, acc[mt][pt], 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < 8; ++pt)
          acc[mt][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[pt >> 2][pt & 3].y, zv[pt >> 2][pt & 3].y, acc[mt][pt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);               // one tile pair's raw rows in flight at a time (register budget)
      };
      const int oz_next = more ? 2 * tz + 2 : p.OD;          // first plane of the next step (or none)
      if constexpr (PREF && !PAIR) {
        load_dy(gB, 2 * tz + 1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int j = 0; j < JW; ++j) block(0, mt, j, gA[j]);
        load_dy(gA, oz_next);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int j = 0; j < JW; ++j) block(1, mt, j, gB[j]);
      } else if constexpr (PREF) {                           // plane pairs: one set per step, copied (16 registers)
        load_dy(gB, oz_next);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int j = 0; j < JW; ++j) block(0, mt, j, gA[j]);
#pragma unroll
        for (int jj = 0; jj < JW; ++jj)
#pragma unroll
          for (int i = 0; i < 4; ++i) gA[jj][i] = gB[jj][i];
      } else {
        // four tile pairs per wave: pair-major order, the gradient voxels of pair j + 1 fetched under pair j's three taps
        // (fetching the next plane's first pair under the last pair as well costs these register-bound variants 28-78
        // spilled VGPRs: measured, not kept)
#pragma unroll
        for (int zo = 0; zo < NZO; ++zo) {
          load_dy_pair(gA[0], 2 * tz + zo, 0);
#pragma unroll
          for (int j = 0; j < JW; ++j) {
            if (j + 1 < JW) load_dy_pair(gB[0], 2 * tz + zo, j + 1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) block(zo, mt, j, gA[0]);
#pragma unroll
            for (int i = 0; i < 4; ++i) gA[0][i] = gB[0][i];
          }
        }
      }
      // (wino.hip: every wave's plane DMA has landed before the barrier lets anyone read the planes -- explicit, not by the
      // accident of a compiler-placed vmcnt(0); the gradient prefetch in flight is waited for with it: it is consumed next)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  };
  if (ph) run(std::integral_constant<int, 1>{});
  else run(std::integral_constant<int, 0>{});

  // ---- per (input half, output block) combination: sum its NS tile subsets through LDS in a fixed order,
  //   part[((ph * MT + mt) * 8 + pt) * 256 + row * 16 + co],  row = 4 q + r,
  // then the inverse kernel transform dg[kz] = G^T dU[kz] G on the workgroup's partial sums (linear: it commutes with the sum
  // over workgroups) -- the slab that leaves is an ORDINARY kernel-gradient slab [27][ci][co]; tem_reduce_slabs_multi
  // adds the workgroups' slabs in a fixed order like those of the direct-form kernel.
  float *const part = lds;
  constexpr int COB = PAIR ? 8 : 16;                           // columns of the combination's block
  float *const slab = p.slabs + (size_t)blockIdx.x * p.slab_stride;
  for (int c = 0; c < NC; ++c) {
    for (int s4 = 0; s4 < NS; ++s4) {
      if (cmb == c && sub == s4) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int pt = 0; pt < 8; ++pt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float *d = part + ((ph * MT + mt) * 8 + pt) * 256 + (4 * q + r) * 16 + m;
              *d = s4 == 0 ? acc[mt][pt][r] : *d + acc[mt][pt][r];
            }
      }
      __syncthreads();
    }
    const int ci0 = CI >= 16 ? 16 * (c % CIH) : 0, co0 = 16 * (c / CIH);
    constexpr int CIB = CI >= 16 ? 16 : CI;
    if (!(KDBG(p.dbg & 256))) {
      for (int id = tid; id < 3 * CIB * COB; id += 512) {    // (kz, ci, co) of the combination's block
        const int co = id % COB, ci = (id / COB) % CIB, kz = id / (COB * CIB);
        float du[4][4];
#pragma unroll
        for (int py = 0; py < 4; ++py)
#pragma unroll
          for (int px = 0; px < 4; ++px) {
            const int pb = ((py >> 1) * MT) * 8 + ((py & 1) * 4 + px);     // + 8 * tile: block of (point, accumulator tile)
            float v;
            if (!PAIR) {
              const int mt = CI >= 16 ? kz : (kz >> 1), row = CI >= 16 ? ci : ((kz & 1) * 8 + ci);
              v = part[(pb + 8 * mt) * 256 + row * 16 + co];
            } else {                                           // the two 8x8 blocks of tap kz (see the kernel's header)
              v = kz == 0 ? part[pb * 256 + ci * 16 + co] + part[pb * 256 + (8 + ci) * 16 + 8 + co]
                : kz == 1 ? part[pb * 256 + (8 + ci) * 16 + co] + part[(pb + 8) * 256 + ci * 16 + 8 + co]
                          : part[(pb + 8) * 256 + ci * 16 + co] + part[(pb + 8) * 256 + (8 + ci) * 16 + 8 + co];
            }
            du[py][px] = v;
          }
        // G^T (3x4) = [[1, .5, .5, 0], [0, .5, -.5, 0], [0, .5, .5, 1]]
        float t[3][4];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          t[0][px] = du[0][px] + 0.5f * (du[1][px] + du[2][px]);
          t[1][px] = 0.5f * (du[1][px] - du[2][px]);
          t[2][px] = 0.5f * (du[1][px] + du[2][px]) + du[3][px];
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          float *d = slab + (size_t)((kz * 3 + ky) * 3) * CI * CO + (ci0 + ci) * CO + co0 + co;
          d[0] = t[ky][0] + 0.5f * (t[ky][1] + t[ky][2]);
          d[CI * CO] = 0.5f * (t[ky][1] - t[ky][2]);
          d[2 * CI * CO] = 0.5f * (t[ky][1] + t[ky][2]) + t[ky][3];
        }
      }
    }
    if (c + 1 < NC) __syncthreads();                         // the buffer is reused by the next combination
  }
}

// ------------------------------------------------------------------------------------------ host
constexpr int LDS_MAX_B = 160 * 1024;

template <int CI, int CO, int NI>
static int plan_bww(BDev &p, size_t *lds_bytes, int EE, double *cost) {
  constexpr int NH = CI / 8, MT = CI >= 16 ? 3 : 2;
  constexpr int CIH = CI >= 16 ? CI / 16 : 1, NB = CO / 16 > 0 ? CO / 16 : 1, NC = CIH * NB, TB = CI == 32 ? 32 : 64;
  const int TY = (p.OH + 1) / 2, TX = (p.OW + 1) / 2;
  double best = 1e300;
  static const int knob_cus = tem_env_int("TEM_WBWW_CUS", 256), knob_slabw = tem_env_int("TEM_WBWW_SLABW", 100),
                   knob_cuw = tem_env_int("TEM_WBWW_CUW", 300);     // CU-time term as in wino.hip plan()
  for (int by = 1; by <= TY && by <= 64; ++by)
    for (int bx = 1; bx <= TX && bx <= 64; ++bx) {
      const int nt = by * bx;
      if (nt > TB) continue;
      if (bx + 1 > EE) continue;                                       // the kernel's compile-time row pitch
      const int E = EE, plv = (2 * by + 2) * 2 * E;
      const int ndma = (plv * 32 + 1023) / 1024;
      if (bx & 1) continue;                                            // tile pairs (t, t + 1) must not wrap rows
      const int subb = ndma * 1024 + 32, slotb = ((NH * subb + 127) / 128) * 128 + (NH == 1 ? 32 : 0);
      const size_t bytes = std::max((size_t)4 * slotb, (size_t)2 * MT * 8 * 256 * 4);   // ring; reused by the final sums
      if (bytes > (size_t)LDS_MAX_B || ndma > 8 * NI) continue;
      const int nby = (TY + by - 1) / by, nbx = (TX + bx - 1) / bx;
      const int cols = p.N * nby * nbx;
      const double step = 2.0 * 2 * MT * 32 * 45.0 * NC * (TB / 64.0) + 4000.0;   // two waves per SIMD, 2 planes x MT x k-steps x 8 MFMAs each
      const double pro = 8000.0 + 4.0 * slotb / 10.0 + 6000.0;         // prologue + the final LDS reduction and slab write
      for (int zs = 1; zs <= p.NTZ; ++zs) {
        const int zper = (p.NTZ + zs - 1) / zs, zsegs = (p.NTZ + zper - 1) / zper;
        if (zsegs != zs) continue;
        // + the slabs' way to HBM and back (written here, read by reduce_multi_k) at ~2 KB per cycle for the whole chip: g.mid 252 ->
        // 216 slabs of 110 KB per call, 7.39 -> 7.32 ms/step.  [Rounds of 128 instead of 256 workgroups (half the prologues,
        // Winograd-domain finishes and slab bytes; the kernel gradients run beside the dependent chains, so their CU-time counts, not
        // their latency) measured 7.25 ms/step, but every such launch then leaves half the chip idle when it runs alone (g.bww.mid
        // 56 -> 107 us stand-alone): not taken, TEM_WBWW_CUS in knob builds.]
        const double slab = (double)cols * zsegs * (27.0 * CI * CO * 4.0) * 2.0 / 2000.0 * (knob_slabw / 100.0);
        const double t = std::ceil(cols * zsegs / (double)knob_cus) * (pro + zper * step) + slab +
                         knob_cuw / 100.0 * (cols * zsegs / 256.0) * (pro + zper * step);
        if (t < best) {
          best = t; p.BY = by; p.BX = bx; p.nby = nby; p.nbx = nbx; p.zsegs = zsegs; p.zper = zper;
          p.E = E; p.PLC = plv * 2; p.subb = subb; p.slotb = slotb; p.ndma = ndma; *lds_bytes = (bytes + 15) & ~(size_t)15;
        }
      }
    }
  *cost = best;
  return best < 1e300 ? TEM_OK : TEM_EUNSUPPORTED;
}

template <int CI, int CO, int NI>
static int plan_bww_memo(BDev &p, size_t *lds_bytes, int EE, double *cost) {
  struct R { int rc, BY, BX, nby, nbx, zsegs, zper, E, PLC, subb, slotb, ndma; size_t lds; double cost; };
  static tem_plan_cache<5, R> cache;
  const std::array<int, 5> key{p.N, p.OH, p.OW, p.NTZ, EE};
  R r;
  if (!cache.get(key, r)) {
    r.lds = 0; r.cost = 1e300;
    r.rc = plan_bww<CI, CO, NI>(p, &r.lds, EE, &r.cost);
    r.BY = p.BY; r.BX = p.BX; r.nby = p.nby; r.nbx = p.nbx; r.zsegs = p.zsegs; r.zper = p.zper; r.E = p.E; r.PLC = p.PLC;
    r.subb = p.subb; r.slotb = p.slotb; r.ndma = p.ndma;
    cache.put(key, r);
  } else {
    p.BY = r.BY; p.BX = r.BX; p.nby = r.nby; p.nbx = r.nbx; p.zsegs = r.zsegs; p.zper = r.zper; p.E = r.E; p.PLC = r.PLC;
    p.subb = r.subb; p.slotb = r.slotb; p.ndma = r.ndma;
  }
  *cost = r.cost; *lds_bytes = r.lds;
  return r.rc;
}

// mode 0: launch; 1: slab-count query (*nslab); 2: describe
static int run_bww(const tem_bww_args *a, hipStream_t st, int mode, int *nslab, char *name, int name_len) {
  const tem_view &i0 = a->in0, &dy = a->dout;
  const bool cube = a->kd == 3 && a->kh == 3 && a->kw == 3 && a->sd == 1 && a->sh == 1 && a->sw == 1 && a->pd == a->ph &&
                    a->ph == a->pw && a->pd >= 0;
  if (!cube || i0.D < 2) return TEM_EUNSUPPORTED;
  if (!fits32(i0) || !fits32(dy)) return TEM_EUNSUPPORTED;
  BDev p{};
  p.in0 = i0.ptr; p.i0N = (int)i0.sN; p.i0D = (int)i0.sD; p.i0H = (int)i0.sH; p.i0W = (int)i0.sW; p.C0 = i0.C;
  p.in1 = i0.ptr; p.i1N = p.i0N; p.i1D = p.i0D; p.i1H = p.i0H; p.i1W = p.i0W;
  int CI = i0.C;
  auto aligned = [](const tem_view &v) {
    return ((uintptr_t)v.ptr & 15) == 0 && v.sW % 4 == 0 && v.sH % 4 == 0 && v.sD % 4 == 0 && v.sN % 4 == 0 && v.C % 4 == 0;
  };
  if (!aligned(i0)) return TEM_EUNSUPPORTED;
  if (a->in1.ptr) {
    const tem_view &i1 = a->in1;
    if (i1.N != i0.N || i1.D != i0.D || i1.H != i0.H || i1.W != i0.W) return TEM_ESHAPE;
    if (!fits32(i1) || !aligned(i1) || i0.C % 8) return TEM_EUNSUPPORTED;
    p.in1 = i1.ptr; p.i1N = (int)i1.sN; p.i1D = (int)i1.sD; p.i1H = (int)i1.sH; p.i1W = (int)i1.sW;
    CI += i1.C;
  }
  const bool pair = dy.C == 8 && CI == 8;
  const int CO = dy.C;
  // variants: 0: 8 -> 8 (plane pairs), 1: 16 -> 16, 2: 8 -> 16, 3: 16 -> 32, 4: 32 -> 16, 5: 32 -> 32
  const int variant = pair ? 0 : (CI == 16 && CO == 16) ? 1 : (CI == 8 && CO == 16) ? 2 : (CI == 16 && CO == 32) ? 3
                    : (CI == 32 && CO == 16) ? 4 : (CI == 32 && CO == 32) ? 5 : -1;
  if (variant < 0 || dy.N != i0.N) return TEM_EUNSUPPORTED;
  if (dy.D != i0.D + 2 * a->pd - 2 || dy.H != i0.H + 2 * a->ph - 2 || dy.W != i0.W + 2 * a->pw - 2) return TEM_ESHAPE;
  p.N = i0.N; p.D = i0.D; p.H = i0.H; p.W = i0.W;
  p.span0 = (int)(((int64_t)(i0.H - 1) * p.i0H + (int64_t)(i0.W - 1) * p.i0W + i0.C) * 4);
  p.span1 = a->in1.ptr ? (int)(((int64_t)(i0.H - 1) * p.i1H + (int64_t)(i0.W - 1) * p.i1W + a->in1.C) * 4) : p.span0;
  if (((int64_t)(dy.D - 1) * dy.sD + (int64_t)(dy.H - 1) * dy.sH + (int64_t)(dy.W - 1) * dy.sW + dy.C) >= ((int64_t)1 << 29))
    return TEM_EUNSUPPORTED;                                 // byte offsets of the gradient's buffer loads stay below 2^31
  p.dy = dy.ptr; p.dN = (int)dy.sN; p.dD = (int)dy.sD; p.dH = (int)dy.sH; p.dW = (int)dy.sW;
  p.OD = dy.D; p.OH = dy.H; p.OW = dy.W;
  p.NTZ = (p.OD + 1) / 2;
  p.P = a->pd;
  size_t lds_bytes = 0;
  // the two compiled row pitches: the cheaper plan wins (ties: the narrower pitch)
  int rc = TEM_EUNSUPPORTED;
  {
    static int force = -1;
    if (force < 0) force = tem_env_int("TEM_WINO_EE", 0);
    BDev best = p;
    double cbest = 1e300;
    for (int EE : {9, 17}) {
      if (force && EE != force) continue;
      BDev q = p;
      double c = 1e300;
      size_t l = 0;
      const int r = variant == 0 ? plan_bww_memo<8, 8, 2>(q, &l, EE, &c) : variant == 1 ? plan_bww_memo<16, 16, 2>(q, &l, EE, &c)
                  : variant == 2 ? plan_bww_memo<8, 16, 2>(q, &l, EE, &c) : variant == 3 ? plan_bww_memo<16, 32, 2>(q, &l, EE, &c)
                  : variant == 4 ? plan_bww_memo<32, 16, 2>(q, &l, EE, &c) : plan_bww_memo<32, 32, 2>(q, &l, EE, &c);
      if (r == TEM_OK && c < cbest) { best = q; cbest = c; lds_bytes = l; rc = TEM_OK; }
    }
    p = best;
  }
  if (rc != TEM_OK) return rc;
  const int nblocks = p.N * p.nby * p.nbx * p.zsegs;
  if (nblocks > (a->nslab > 0 ? a->nslab : 1024)) return TEM_EUNSUPPORTED;
  if (mode == 1) { *nslab = nblocks; return TEM_OK; }
  if (mode == 2) { if (name) snprintf(name, name_len, "wino_bww_k<%d, %d, 2, %s, %d>", CI, CO, pair ? "true" : "false", p.E); return TEM_OK; }
  if (!a->slabs || a->nslab != nblocks || a->accumulate) return TEM_EINVAL;
  p.magicBX = magic_for(p.BX); p.magicE = magic_for(p.E);
  p.slabs = a->slabs; p.slab_stride = a->slab_stride ? a->slab_stride : (int64_t)27 * CI * dy.C;
  {
    static int dbg = -1;
    if (dbg < 0) dbg = tem_env_int("TEM_DEBUG_FLAGS", 0);
    p.dbg = dbg;
  }
  if (KDBG(p.dbg & 8))
    fprintf(stderr, "wino_bww<%d> O=%dx%dx%d: BY=%d BX=%d nby=%d nbx=%d zsegs=%d zper=%d blocks=%d lds=%zu\n", CI,
            p.OD, p.OH, p.OW, p.BY, p.BX, p.nby, p.nbx, p.zsegs, p.zper, nblocks, lds_bytes);
  static bool attr[12] = {};
  auto go = [&](auto kern) -> int {
    const int ai = variant * 2 + (p.E == 17);
    if (!attr[ai]) {
      hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX_B);
      if (e != hipSuccess) return (int)e;
      attr[ai] = true;
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(512), lds_bytes, st, p);
    TEM_CHECK_LAUNCH();
    return TEM_OK;
  };
  const bool e9 = p.E == 9;
  switch (variant) {
    case 0: return e9 ? go(wino_bww_k<8, 8, 2, true, 9>) : go(wino_bww_k<8, 8, 2, true, 17>);
    case 1: return e9 ? go(wino_bww_k<16, 16, 2, false, 9>) : go(wino_bww_k<16, 16, 2, false, 17>);
    case 2: return e9 ? go(wino_bww_k<8, 16, 2, false, 9>) : go(wino_bww_k<8, 16, 2, false, 17>);
    case 3: return e9 ? go(wino_bww_k<16, 32, 2, false, 9>) : go(wino_bww_k<16, 32, 2, false, 17>);
    case 4: return e9 ? go(wino_bww_k<32, 16, 2, false, 9>) : go(wino_bww_k<32, 16, 2, false, 17>);
    default: return e9 ? go(wino_bww_k<32, 32, 2, false, 9>) : go(wino_bww_k<32, 32, 2, false, 17>);
  }
}

}  // namespace wino

extern "C" int tem_conv_bwd_weight_winograd_nslab(const tem_bww_args *a, char *name, int32_t name_len) {
  if (!a || !tem_view_ok(a->in0) || !tem_view_ok(a->dout)) return TEM_EINVAL;
  int n = 0;
  int rc = wino::run_bww(a, nullptr, 1, &n, nullptr, 0);
  if (rc != TEM_OK) return rc;
  if (name && name_len > 0) wino::run_bww(a, nullptr, 2, nullptr, name, name_len);
  return n;
}

extern "C" int tem_conv_bwd_weight_winograd(const tem_bww_args *a, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!a || !tem_view_ok(a->in0) || !tem_view_ok(a->dout)) return TEM_EINVAL;
  return wino::run_bww(a, (hipStream_t)stream, 0, nullptr, nullptr, 0);
}
