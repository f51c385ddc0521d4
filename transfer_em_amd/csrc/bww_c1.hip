// bww_c1.hip -- kernel gradient of the 3x3x3 stride-1 layers with ONE input channel (g.c0, d.d1a; and g.f2 in the
// swapped form hip_ops.bww_launch gives it: "input" = the 1-channel gradient, "dout" = the 16-channel activation, pad 2):
//
//   dW[tap][co] = sum_v X[v + tap - P] * G[v][co]                       (27 x C_out numbers out of ~70 MB of input)
//
// HBM-bound (AI ~ 7 FLOP/B): the job is to stream G once at full width.  The matrix-core form (bww_lds_k<1,CO>: K =
// voxels, 27 taps in two half-empty row tiles) is bound by its MFMA count and LDS choreography at 1.2 TB/s; here the
// reduction runs on the vector pipe with ALL partial sums in registers:
//   lane = (voxel column (y, x) of a 16-wide patch, channel quad): 27 taps x 4 channels = 108 accumulators;
//   the lane marches along z: per step one 16-byte load of G (two / four lanes of a voxel = one 32 / 64-byte run), the 9
//   new X neighbours of the plane entering its 3x3x3 window from LDS (the 1-channel patch of the whole z run is staged
//   once: a few KB), 108 FMAs.  No barrier inside the march.  At the end: butterfly sum over the lanes of a wave, LDS
//   sum over the waves, one slab of 27 x C_out per workgroup (the ordinary slab reduction finishes, fixed order).
//
// Reference call sites: Conv3DBackpropFilter of models/generator.py:54,110 and discriminator.py:39-40.
#include "tem_common.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace bwwc1 {

struct C1Dev {
  const float *x; int32_t xN, xD, xH, xW;
  int32_t D, H, W;                 // extents of the 1-channel tensor
  const float *g; int32_t gN, gD, gH, gW;
  int32_t OD, OH, OW;              // extents of the CO-channel tensor
  int32_t P;
  int32_t nyb, nxb, zsegs, zper;
  int32_t gspan, xspan;            // bytes one sample of g / x spans
  int32_t dbg;                     // TEM_DEBUG_KNOBS builds: 1 = march without the FMAs, 2 = without the X window reads
  float *slabs; int64_t slab_stride;
};

template <int CO>
__global__ __launch_bounds__(256) void bww_c1_k(C1Dev p) {
  constexpr int NQ = CO / 4;                                  // lanes per voxel (one channel quad each)
  constexpr int PX = 16, PY = 256 / NQ / PX;                  // patch of voxel columns per workgroup: 8 x 16 (CO 8), 4 x 16 (CO 16)
  constexpr int XP = PX + 2, YP = PY + 2;
  constexpr int Q = 6;                                        // planes of G in flight per lane (register queue, 16 bytes per lane and plane)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *const xs = smem;                                     // [zper + 2][YP][XP]: the 1-channel patch of the whole run
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cq = tid % NQ, vox = tid / NQ;
  const int ly = vox / PX, lx = vox % PX;
  int b = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int zseg = b % p.zsegs; b /= p.zsegs;
  const int bx = b % p.nxb; b /= p.nxb;
  const int by = b % p.nyb;
  const int n = b / p.nyb;
  const int z0 = zseg * p.zper, z1 = min(p.OD, z0 + p.zper), nz = z1 - z0;
  const int oy = by * PY + ly, ox = bx * PX + lx;
  const bool vok = oy < p.OH && ox < p.OW;

  // G streams through a per-lane register queue Q planes deep: the lane fetches the 16 bytes of ITS (voxel, channel quad)
  // of plane z + Q right after it has consumed plane z -- plain buffer loads into a statically indexed register array, so
  // the compiler's own vmcnt counting keeps Q - 1 fetches in flight (the march is unrolled Q times and has no branch: past
  // the run the offsets are out of range and the products are zeros).  [Round 2 streamed G through an LDS-DMA ring; the
  // probe tests/tools/stream_probe.hip reads this access pattern at 5.6 TB/s with 4 register loads in flight, the ring's
  // march reached 3.7 TB/s with the FMAs removed.]
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.g + (size_t)n * p.gN), 0, p.gspan, 0x00020000);
  const int goff = vok ? (oy * p.gH + ox * p.gW + 4 * cq) * 4 : (int)0x80000000;
  const float *xl = xs + ly * XP + lx;
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  auto load_g = [&](int z) -> f4 {                            // plane z of the lane's chunk (zeros past the run)
    int off = z < z1 ? goff + z * p.gD * 4 : (int)0x80000000;
    const u4 q = __builtin_amdgcn_raw_buffer_load_b128(grs, off, 0, 0);
    return f4{__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w)};
  };
  // ---- stage X[z0 - P .. z0 - P + nz + 1][by*PY - P ..][bx*PX - P ..] (zeros outside the tensor = out-of-range buffer
  // offsets) by LDS-DMA, 4 bytes per lane: ~33 fetches per wave, ALL in flight at once and no staging registers, then the
  // G queue's first Q planes behind them -- the whole prologue is ONE memory round trip (round 2 staged the patch through
  // registers in five dependent batches of 8 loads: five round trips, ~8 us of a 30 us kernel, before the ring even started).
  {
    const int total = (nz + 2) * YP * XP;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + (size_t)n * p.xN), 0, p.xspan, 0x00020000);
    int cnt = 0;
    for (int i0 = wave * 64; i0 < ((p.dbg & 8) ? 256 : total); i0 += 256) {
      const int i = i0 + lane;
      const int zz = i / (YP * XP), r = i - zz * (YP * XP), yy = r / XP, xx = r - yy * XP;
      const int iz = z0 - p.P + zz, iy = by * PY - p.P + yy, ix = bx * PX - p.P + xx;
      const bool ok = i < total && (unsigned)iz < (unsigned)p.D && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const int off = ok ? (iz * p.xD + iy * p.xH + ix * p.xW) * 4 : (int)0x80000000;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void *)(xs + i0), 4, off, 0, 0, 0);
      if (++cnt == 40) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); cnt = 8; }     // (vmcnt counts to 63)
    }
  }
  f4 gq[Q];
#pragma unroll
  for (int k = 0; k < Q; ++k) gq[k] = load_g(z0 + k);
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(Q) : "memory");     // the patch has landed (the queue's fetches may still fly)
  __builtin_amdgcn_s_barrier();

  typedef float f2_ __attribute__((ext_vector_type(2)));
  f2_ acc[27 * 2];                                            // acc[2 t + h] = channels 2 h, 2 h + 1 of tap t
#pragma unroll
  for (int t = 0; t < 54; ++t) acc[t] = f2_{0.f, 0.f};

  // The 3x3 X neighbours of a plane live as x-PAIRS (one ds_read2_b32 fetches taps dx 0, 1 of a row; tap 2 is the low half
  // of a second pair), and every FMA names the half it broadcasts through op_sel: written as asm because the compiler,
  // given scalar window values, builds the broadcast pairs with a v_mov per odd-register value (62 moves per 162
  // v_pk_fma_f32 of three planes -- the kernel is bound by the vector pipe's issue slots, PMC: 0.49 active x 2 waves).
  typedef float f2 __attribute__((ext_vector_type(2)));
  auto load_w = [&](f2 (&w)[6], int zz) {                     // w[2 dy] = (tap dx 0, tap dx 1), w[2 dy + 1].x = tap dx 2
    const float *pl = xl + ((p.dbg & 2) ? 0 : zz) * (YP * XP);
    if (p.dbg & 2) { if (zz > 1) return; }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      w[2 * dy] = f2{pl[dy * XP], pl[dy * XP + 1]};
      w[2 * dy + 1] = f2{pl[dy * XP + 2], 0.f};
    }
  };
  auto fma_plane = [&](const f2 (&w)[6], int dz, const f4 &g) {
    const f2 g01 = {g.x, g.y}, g23 = {g.z, g.w};
    if (p.dbg & 1) { asm volatile("" :: "v"(g01), "v"(g23), "v"(w[0]), "v"(w[5])); return; }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      f2_ *a = &acc[2 * ((dz * 3 + dy) * 3)];
      asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[0]) : "v"(w[2 * dy]), "v"(g01));
      asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[1]) : "v"(w[2 * dy]), "v"(g23));
      asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(a[2]) : "v"(w[2 * dy]), "v"(g01));
      asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(a[3]) : "v"(w[2 * dy]), "v"(g23));
      asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[4]) : "v"(w[2 * dy + 1]), "v"(g01));
      asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[5]) : "v"(w[2 * dy + 1]), "v"(g23));
    }
  };
  // window of three X planes in registers, rotated by the unrolled march (no register moves): output plane zi multiplies
  // staged planes zi, zi + 1, zi + 2 (taps dz = 0, 1, 2) with G[z0 + zi]
  f2 w0[6], w1[6], w2[6];
  load_w(w0, 0); load_w(w1, 1);
  const int zlast = nz + 1;                                   // staged planes 0 .. nz + 1 exist (steps past the run read the last one: times zero)
  for (int zi = 0; zi < nz; zi += Q) {
#pragma unroll
    for (int u = 0; u < Q; ++u) {
      const f4 g = gq[u];
      gq[u] = load_g(z0 + zi + u + Q);
      if (u % 3 == 0) { load_w(w2, min(zi + u + 2, zlast)); fma_plane(w0, 0, g); fma_plane(w1, 1, g); fma_plane(w2, 2, g); }
      if (u % 3 == 1) { load_w(w0, min(zi + u + 2, zlast)); fma_plane(w1, 0, g); fma_plane(w2, 1, g); fma_plane(w0, 2, g); }
      if (u % 3 == 2) { load_w(w1, min(zi + u + 2, zlast)); fma_plane(w2, 0, g); fma_plane(w0, 1, g); fma_plane(w1, 2, g); }
    }
  }
  asm volatile("" :: "v"(gq[0]), "v"(gq[Q - 1]));             // (the queue's tail fetches are out of range: zeros)

  // ---- sum over the lanes that share a channel quad.  Within a row of 16 lanes: DPP row shifts by multiples of NQ (they keep
  // lane % NQ; vector-pipe speed -- as a __shfl_xor butterfly this was 540 ds_bpermute + 206 waits per wave, about as long as
  // the whole march); the last NQ lanes of each row then hold the row's sums.  Rows and waves: through LDS, fixed order.
  __syncthreads();                                            // the LDS is reused as the cross-row / cross-wave buffer
#pragma unroll
  for (int t = 0; t < 27; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = acc[2 * t + (c >> 1)][c & 1];
      // row_shr:o for o = NQ, 2 NQ, .. 8 (zeros shifted in)
      if (NQ <= 2) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
      acc[2 * t + (c >> 1)][c & 1] = v;
    }
  if (p.dbg & 4) { if (acc[0][0] + acc[53][1] == 12345.f) p.slabs[0] = 1.f; return; }
  float *red = smem;                                          // [4 waves x 4 rows][27][CO]
  if ((lane & 15) >= 16 - NQ) {
    const int part = wave * 4 + (lane >> 4), q4 = (lane & 15) - (16 - NQ);
#pragma unroll
    for (int t = 0; t < 27; ++t)
      *reinterpret_cast<float4 *>(red + (part * 27 + t) * CO + 4 * q4) = make_float4(acc[2 * t][0], acc[2 * t][1], acc[2 * t + 1][0], acc[2 * t + 1][1]);
  }
  __syncthreads();
  float *slab = p.slabs + (size_t)blockIdx.x * p.slab_stride;
  for (int i = tid; i < 27 * CO; i += 256) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += red[k * 27 * CO + i];
    slab[i] = v;
  }
}

// ------------------------------------------------------------------------------------------ matrix-core form (round 3)
// The vector-pipe kernel above is bound by the VECTOR pipe, not by HBM: a v_pk_fma_f32 takes 8 cycles per wave (the fp32
// VALU peak is half the fp32 matrix rate: tests/tools/issue_probe.hip), so its 54 packed FMAs per 16 bytes of G and lane
// allow ~2.5 TB/s at two waves per SIMD -- what it measured.  Here the same reduction runs on v_mfma_f32_16x16x4_f32:
//
//   D[R][n] += sum_k A[R][k] B[k][n],   k = 4 x-consecutive voxels of one row of the patch,
//   C_out = 16:  n = co,              R = tap (dz, dy, dx): 27 of 32 rows (2 MFMAs per k-step),   A[R][k] = X[v_k + tap - P]
//   C_out =  8:  n = (plane h, co),   R = (zi, dy, dx), zi = dz + h in 0..3: 36 of 48 rows (3 MFMAs per k-step and PLANE
//                PAIR; the slab entry of tap (dz, dy, dx) is D[(dz, ..)][(0, co)] + D[(dz + 1, ..)][(1, co)]).
//
// A workgroup owns an 8 x 16 patch of voxel columns and marches along z (1 / 2 planes of G per step); each of its four
// WAVES owns two rows of the patch and is on its own: it fetches its rows of G and its 4 rows of the X halo patch by
// LDS-DMA (16 / 4 bytes per lane, Q - 1 steps ahead) into a private ring and waits for nothing but its own vmcnt -- no
// barrier in the march.  The B fragment of a k-step is 64 consecutive floats of the G rows (one conflict-free
// ds_read_b32 per plane pair), the A fragments are gathers from the 1-channel X rows (lane = (row R, voxel k); an X plane
// pitch of 16 mod 32 floats spreads the planes over the banks).  Every LDS address of the 8 k-steps of a step is one
// per-step base + a compile-time immediate: ~20 vector instructions per step beside the 16 / 24 MFMAs -- which matters,
// because VALU and fp32-MFMA issue serialise on this chip.
template <int CO>
__global__ __launch_bounds__(256) void bww_c1m_k(C1Dev p) {
  constexpr int NZ = CO == 8 ? 2 : 1;                         // planes of G per step
  constexpr int MT = CO == 8 ? 3 : 2;                         // 16-row tiles of D
  constexpr int PY = 8, PX = 16;                              // patch: rows x voxels; a wave owns rows 2 wave, 2 wave + 1
  constexpr int Q = 3;                                        // steps in flight (G fetched Q - 1 steps ahead)
  constexpr int XP = 24, XPL = 112;                           // X rows of a wave: row pitch (6 chunks of 4 floats), plane pitch (floats; 4 rows;
                                                              //   112 = 16 mod 32: the planes a gather touches sit 16 banks apart)
  constexpr int GROW = PX * CO, GPL = 2 * GROW + 16;          // G rows of a wave: floats per row, per plane slot (+16: the planes of a pair 16 banks apart)
  constexpr int RG = NZ * Q;                                  // G planes in the ring
  constexpr int GI = CO / 8;                                  // 16-byte DMA chunks per lane and G plane (2 rows x GROW floats = 64 GI chunks)
  constexpr int RX = NZ * Q + 2;                              // X planes in the ring
  constexpr int ND = NZ * GI + NZ;                            // DMA instructions per step: G chunks + ONE 16-byte fetch per new X plane
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int wlds = RG * GPL + RX * XPL;                   // floats of LDS per wave (~10 KB)
  float *const gs = smem + wave * wlds;                       // [RG][GPL]
  float *const xs = gs + RG * GPL;                            // [RX][XPL]: the wave's 4 rows of the X planes in flight
  const int m = lane & 15, kq = lane >> 4;
  int b = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int zseg = b % p.zsegs; b /= p.zsegs;
  const int bx = b % p.nxb; b /= p.nxb;
  const int by = b % p.nyb;
  const int n = b / p.nyb;
  const int z0 = zseg * p.zper, z1 = min(p.OD, z0 + p.zper), nz = z1 - z0;
  const int nsteps = (nz + NZ - 1) / NZ;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int OOB = (int)0x80000000;

  // ---- DMA roles (constant over the run)
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.g + (size_t)n * p.gN), 0, p.gspan, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + (size_t)n * p.xN), 0, p.xspan, 0x00020000);
  int goff[GI];                                               // byte offset of the lane's chunk(s) inside a plane of G
#pragma unroll
  for (int i = 0; i < GI; ++i) {
    const int c = lane + 64 * i;                              // chunk of the wave's two rows: row c / (GROW / 4), floats (c % (GROW / 4)) * 4 ..
    const int r = c / (GROW / 4), f = (c % (GROW / 4)) * 4;
    const int oy = by * PY + 2 * wave + r, ox = bx * PX + f / CO;
    goff[i] = (oy < p.OH && ox < p.OW) ? (oy * p.gH + ox * p.gW + f % CO) * 4 : OOB;
  }
  auto dma_g = [&](int gp) {                                  // plane z0 + gp of G -> slot gp % RG (zeros past the run)
    const bool ok = gp < nz;
    float *dst = gs + (gp % RG) * GPL;
#pragma unroll
    for (int i = 0; i < GI; ++i) {
      int off = (ok && goff[i] != OOB) ? goff[i] + (z0 + gp) * p.gD * 4 : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(grs, (__attribute__((address_space(3))) void *)(dst + 256 * i), 16, off, 0, 0, 0);
    }
  };
  // The wave's X rows of a plane (zeros outside the tensor): ONE 16-byte LDS-DMA -- 4 rows x 6 chunks on 24 lanes (the
  // other lanes are switched off and write nothing).  The chunk grid is aligned to x = 0 mod 4 (xsh) so that, W being a
  // multiple of 4, a chunk is inside a row or outside it; the global addresses are only 4-byte aligned, which the DMA takes
  // (tests/tools/dma_align_probe.hip).  [As 4-byte fetches -- 2 per plane and wave -- the X traffic was 2/3 of the kernel's
  // memory instructions and the memory pipe's ~16 cycles per wave-instruction, not HBM or the MFMAs, set the pace.]
  const int xsh = p.P <= 0 ? (-p.P / 4) * 4 : -((p.P + 3) / 4) * 4;      // floor(-P / 4) * 4
  const int sh = -p.P - xsh;                                   // 0 .. 3: column of the voxel that output x = 0 reads with tap dx = 0
  int xoffg = OOB;
  if (lane < 24) {
    const int yy = lane / 6, c6 = lane - yy * 6;
    const int iy = by * PY + 2 * wave - p.P + yy, ix = bx * PX + xsh + 4 * c6;
    xoffg = ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? (iy * p.xH + ix) * 4 : OOB;
  }
  auto dma_x = [&](int xp) {                                  // plane z0 - P + xp of X -> slot xp % RX
    const int iz = z0 - p.P + xp;
    int off = ((unsigned)iz < (unsigned)p.D && xoffg != OOB) ? xoffg + iz * p.xD * 4 : OOB;
    if (lane < 24)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void *)(xs + (xp % RX) * XPL), 16, off, 0, 0, 0);
  };
  auto dma_step = [&](int s) {                                // G planes and the NZ new X planes of step s
#pragma unroll
    for (int h = 0; h < NZ; ++h) dma_g(NZ * s + h);
#pragma unroll
    for (int h = 0; h < NZ; ++h) dma_x(NZ * s + 2 + h);
  };
  // ---- prologue: X planes 0, 1 and the steps 0 .. Q - 2
  dma_x(0); dma_x(1);
#pragma unroll
  for (int s = 0; s < Q - 1; ++s) dma_step(s);

  // ---- fragment roles: row R = 16 t + m -> (zi, dy, dx) (C_out 16: zi = dz), voxel k = kq
  int aconst[MT];                                             // float offset of the lane's X element for k-step (row 0, x 0), tile t
  int azi[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    int R = 16 * t + m;
    if (R >= (NZ + 2) * 9) R = 0;                             // padded rows: any valid address (their sums are dropped)
    const int zi = R / 9, dy = (R - zi * 9) / 3, dx = R - zi * 9 - dy * 3;
    azi[t] = zi;
    aconst[t] = dy * XP + dx + kq + sh;
  }
  const int h = CO == 8 ? (m >> 3) : 0;
  const int bconst = (kq * CO + (CO == 8 ? (m & 7) : m)) + h * GPL;          // float offset of the lane's G element (row 0, x kq)

  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int s = 0; s < nsteps; ++s) {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(ND * (Q - 2)) : "memory");      // this wave's fetches for step s have landed
    // step s + Q - 1 goes into the slots step s - 1 used (their reads completed before step s - 1's MFMAs issued); past the
    // run the offsets are out of range: no traffic
    dma_step(s + Q - 1);
    // per-step bases (bytes): X plane (NZ s + zi) % RX, G plane NZ (s % Q)
    const int sx = (NZ * s) % RX;
    const char *xa[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      int sl = sx + azi[t];
      sl = sl >= RX ? sl - RX : sl;
      xa[t] = reinterpret_cast<const char *>(xs) + (sl * XPL + aconst[t]) * 4;
    }
    const char *gb = reinterpret_cast<const char *>(gs) + ((NZ * (s % Q)) * GPL + bconst) * 4;
    float af[8][MT], bf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {                           // k-step: row ks / 4 of the wave, voxels 4 (ks % 4) ..
      const int rr = ks >> 2, j = ks & 3;
      bf[ks] = *reinterpret_cast<const float *>(gb + (rr * GROW + 4 * j * CO) * 4);
#pragma unroll
      for (int t = 0; t < MT; ++t) af[ks][t] = *reinterpret_cast<const float *>(xa[t] + (rr * XP + 4 * j) * 4);
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks][t], bf[ks], acc[t], 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the tail fetches (zeros) before the LDS is reused
  __syncthreads();

  // ---- waves -> one slab: red[wave][R][n], then slab[tap][co] in fixed order
  float *red = smem;                                          // [4][MT * 16][16]
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * MT * 16 + 16 * t + 4 * kq + r) * 16 + m] = acc[t][r];
  __syncthreads();
  float *slab = p.slabs + (size_t)blockIdx.x * p.slab_stride;
  for (int i = tid; i < 27 * CO; i += 256) {
    const int tap = i / CO, co = i - tap * CO;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float *rw = red + w * MT * 16 * 16;
      if (CO == 16) v += rw[tap * 16 + co];
      else v += rw[tap * 16 + co] + rw[(tap + 9) * 16 + 8 + co];          // (dz, dy, dx) x plane 0 + (dz + 1, dy, dx) x plane 1
    }
    slab[i] = v;
  }
}

static int run(const tem_bww_args *a, hipStream_t st, bool dry, int *nslab_out, char *name, int name_len) {
  const tem_view &x = a->in0, &g = a->dout;
  const bool cube = a->kd == 3 && a->kh == 3 && a->kw == 3 && a->sd == 1 && a->sh == 1 && a->sw == 1 && a->pd == a->ph &&
                    a->ph == a->pw && a->pd >= 0;
  if (!cube || x.C != 1 || a->in1.ptr || (g.C != 8 && g.C != 16) || x.D < 2 || x.N != g.N) return TEM_EUNSUPPORTED;
  if (x.sW != 1 || x.W % 4) return TEM_EUNSUPPORTED;          // X rows are fetched as 16-byte chunks of 4 voxels, none across a row end
  // (dout may be any window of the layer's output: voxel o reads x[o + tap - p], zeros outside x -- the region-restricted
  // cycle path passes such windows)
  auto span = [](const tem_view &v) {
    return (int64_t)(v.N - 1) * v.sN + (int64_t)(v.D - 1) * v.sD + (int64_t)(v.H - 1) * v.sH + (int64_t)(v.W - 1) * v.sW + v.C;
  };
  if (span(x) >= ((int64_t)1 << 31) || span(g) >= ((int64_t)1 << 31)) return TEM_EUNSUPPORTED;
  if (((uintptr_t)g.ptr & 15) || g.sW % 4 || g.sH % 4 || g.sD % 4 || g.sN % 4) return TEM_EUNSUPPORTED;
  const int CO = g.C, PY = 8;
  C1Dev p{};
  p.x = x.ptr; p.xN = (int)x.sN; p.xD = (int)x.sD; p.xH = (int)x.sH; p.xW = (int)x.sW;
  p.D = x.D; p.H = x.H; p.W = x.W;
  p.g = g.ptr; p.gN = (int)g.sN; p.gD = (int)g.sD; p.gH = (int)g.sH; p.gW = (int)g.sW;
  p.OD = g.D; p.OH = g.H; p.OW = g.W;
  p.P = a->pd;
  p.nyb = (g.H + PY - 1) / PY; p.nxb = (g.W + 15) / 16;
  // (at most the caller's slab budget; even plane counts for the plane-pair form)
  const int cols = g.N * p.nyb * p.nxb;
  const int cap = a->nslab > 0 ? a->nslab : 1024;
  if (cols > cap) return TEM_EUNSUPPORTED;
  // z runs: ~6 workgroups per CU over the launch (3 are resident: 40 KB of LDS each), >= 8 planes per run
  static int zp = -1;
  if (zp < 0) zp = tem_env_int("TEM_BWWC1_ZPER", 0);
  int zsegs = zp > 0 ? std::max(1, (g.D + zp - 1) / zp) : std::max(1, std::min((1536 + cols / 2) / cols, std::max(1, g.D / 8)));
  int zper = (g.D + zsegs - 1) / zsegs;
  if (CO == 8) zper += zper & 1;
  if ((int64_t)cols * ((g.D + zper - 1) / zper) > cap) return TEM_EUNSUPPORTED;
  zsegs = (g.D + zper - 1) / zper;
  p.zsegs = zsegs; p.zper = zper;
  const int nblocks = cols * zsegs;
  const int NZ = CO == 8 ? 2 : 1, MT = CO == 8 ? 3 : 2;
  const int Q = 3;
  const size_t lds = std::max<size_t>((size_t)4 * ((size_t)NZ * Q * (2 * 16 * CO + 16) + (size_t)(NZ * Q + 2) * 112) * 4, (size_t)4 * MT * 16 * 16 * 4);
  if (lds > 64 * 1024) return TEM_EUNSUPPORTED;
  // bytes one sample of g spans, from the view itself (a batch-1 view may carry any sample stride, 0 included)
  const int64_t gspan = ((int64_t)(g.D - 1) * g.sD + (int64_t)(g.H - 1) * g.sH + (int64_t)(g.W - 1) * g.sW + g.C) * 4;
  if (gspan > (int64_t)0x7fffffff) return TEM_EUNSUPPORTED;
  p.gspan = (int)gspan;
  p.xspan = (int)(((int64_t)(x.D - 1) * x.sD + (int64_t)(x.H - 1) * x.sH + (int64_t)(x.W - 1) * x.sW + 1) * 4);
  { static int dbg = -1; if (dbg < 0) dbg = tem_env_int("TEM_DEBUG_FLAGS", 0); p.dbg = dbg; }
  if (nslab_out) *nslab_out = nblocks;
  if (name) snprintf(name, name_len, "bww_c1m_k<%d>", CO);
  if (dry) return TEM_OK;
  if (!a->slabs || a->nslab != nblocks || a->accumulate) return TEM_EINVAL;
  p.slabs = a->slabs; p.slab_stride = a->slab_stride ? a->slab_stride : (int64_t)27 * CO;
  if (CO == 8) hipLaunchKernelGGL(bww_c1m_k<8>, dim3(nblocks), dim3(256), lds, st, p);
  else hipLaunchKernelGGL(bww_c1m_k<16>, dim3(nblocks), dim3(256), lds, st, p);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}


// ------------------------------------------------------------------------------------------ bf16 inputs (config 5)
// The same march with bf16 X and G (2-byte elements, strides in elements), fp32 MFMAs and fp32 slabs: the shape-generic
// bww_bf16_k spent 59 us on the first layer's kernel gradient (2-byte gathers of the one-channel input per MFMA operand of
// a K = voxels product), 0.6 ms of the bf16 step.  Differences to bww_c1m_k: a G chunk (16 bytes) is 8 elements -- one
// voxel of 8 channels, half a voxel of 16 --, the wave's X rows arrive as ONE 4-byte LDS-DMA on 48 lanes (2 voxels per
// lane: W even, the chunk grid aligned to even x, so no chunk crosses a row end; 16-byte chunks would, at W = 132 / 98),
// and a fragment element is a 2-byte LDS read shifted into the upper half of the register (bf16 -> fp32 is exact).
template <int CO>
__global__ __launch_bounds__(256) void bww_c1m_h_k(C1Dev p) {
  typedef unsigned short u16;
  constexpr int NZ = CO == 8 ? 2 : 1, MT = CO == 8 ? 3 : 2, PY = 8, PX = 16, Q = 3;
  constexpr int XP = 24, XPL = 160;                           // X rows of a wave: row pitch, plane pitch (elements; 320 bytes = 64 mod 256)
  constexpr int GROW = PX * CO, GPL = 2 * GROW + 32;          // G rows of a wave: elements per row, per plane slot (+64 bytes between the planes of a pair)
  constexpr int RG = NZ * Q, RX = NZ * Q + 2;
  constexpr int NCH = 2 * GROW / 8;                           // 16-byte chunks of the wave's two G rows: 32 (C_out 8) or 64
  constexpr int ND = 2 * NZ;                                  // DMA instructions per step: one per G plane, one per new X plane
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int wlds = RG * GPL + RX * XPL;                   // elements of LDS per wave
  u16 *const gs = reinterpret_cast<u16 *>(smem) + wave * wlds;
  u16 *const xs = gs + RG * GPL;
  const int m = lane & 15, kq = lane >> 4;
  int b = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int zseg = b % p.zsegs; b /= p.zsegs;
  const int bx = b % p.nxb; b /= p.nxb;
  const int by = b % p.nyb;
  const int n = b / p.nyb;
  const int z0 = zseg * p.zper, z1 = min(p.OD, z0 + p.zper), nz = z1 - z0;
  const int nsteps = (nz + NZ - 1) / NZ;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int OOB = (int)0x80000000;
  const u16 *const gp_ = reinterpret_cast<const u16 *>(p.g), *const xp_ = reinterpret_cast<const u16 *>(p.x);

  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)(gp_ + (size_t)n * p.gN), 0, p.gspan, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)(xp_ + (size_t)n * p.xN), 0, p.xspan, 0x00020000);
  int goff = OOB;                                             // byte offset of the lane's chunk inside a plane of G
  if (lane < NCH) {
    const int r = lane / (GROW / 8), f = (lane % (GROW / 8)) * 8;
    const int oy = by * PY + 2 * wave + r, ox = bx * PX + f / CO;
    if (oy < p.OH && ox < p.OW) goff = (oy * p.gH + ox * p.gW + f % CO) * 2;
  }
  auto dma_g = [&](int gpl) {                                 // plane z0 + gpl of G -> slot gpl % RG (zeros past the run)
    const int off = (gpl < nz && goff != OOB) ? goff + (z0 + gpl) * p.gD * 2 : OOB;
    if (lane < NCH)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(grs, (__attribute__((address_space(3))) void *)(gs + (gpl % RG) * GPL), 16, off, 0, 0, 0);
  };
  const int xsh = p.P <= 0 ? (-p.P / 2) * 2 : -((p.P + 1) / 2) * 2;      // floor(-P / 2) * 2
  const int sh = -p.P - xsh;                                   // 0 or 1: column of the voxel that output x = 0 reads with tap dx = 0
  int xoffg = OOB;
  if (lane < 48) {
    const int yy = lane / 12, c12 = lane - yy * 12;
    const int iy = by * PY + 2 * wave - p.P + yy, ix = bx * PX + xsh + 2 * c12;
    if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) xoffg = (iy * p.xH + ix) * 2;
  }
  auto dma_x = [&](int xpl) {                                 // plane z0 - P + xpl of X -> slot xpl % RX
    const int iz = z0 - p.P + xpl;
    const int off = ((unsigned)iz < (unsigned)p.D && xoffg != OOB) ? xoffg + iz * p.xD * 2 : OOB;
    if (lane < 48)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void *)(xs + (xpl % RX) * XPL), 4, off, 0, 0, 0);
  };
  auto dma_step = [&](int s) {
#pragma unroll
    for (int h = 0; h < NZ; ++h) dma_g(NZ * s + h);
#pragma unroll
    for (int h = 0; h < NZ; ++h) dma_x(NZ * s + 2 + h);
  };
  dma_x(0); dma_x(1);
#pragma unroll
  for (int s = 0; s < Q - 1; ++s) dma_step(s);

  int aconst[MT], azi[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    int R = 16 * t + m;
    if (R >= (NZ + 2) * 9) R = 0;
    const int zi = R / 9, dy = (R - zi * 9) / 3, dx = R - zi * 9 - dy * 3;
    azi[t] = zi;
    aconst[t] = dy * XP + dx + kq + sh;
  }
  const int h = CO == 8 ? (m >> 3) : 0;
  const int bconst = (kq * CO + (CO == 8 ? (m & 7) : m)) + h * GPL;
  auto up = [](u16 v) { return __uint_as_float((uint32_t)v << 16); };

  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int s = 0; s < nsteps; ++s) {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(ND * (Q - 2)) : "memory");      // this wave's fetches for step s have landed
    dma_step(s + Q - 1);
    const int sx = (NZ * s) % RX;
    const u16 *xa[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      int sl = sx + azi[t];
      sl = sl >= RX ? sl - RX : sl;
      xa[t] = xs + sl * XPL + aconst[t];
    }
    const u16 *gb = gs + (NZ * (s % Q)) * GPL + bconst;
    float af[8][MT], bf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int rr = ks >> 2, j = ks & 3;
      bf[ks] = up(gb[rr * GROW + 4 * j * CO]);
#pragma unroll
      for (int t = 0; t < MT; ++t) af[ks][t] = up(xa[t][rr * XP + 4 * j]);
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks][t], bf[ks], acc[t], 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  float *red = smem;                                          // [4][MT * 16][16]
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * MT * 16 + 16 * t + 4 * kq + r) * 16 + m] = acc[t][r];
  __syncthreads();
  float *slab = p.slabs + (size_t)blockIdx.x * p.slab_stride;
  for (int i = tid; i < 27 * CO; i += 256) {
    const int tap = i / CO, co = i - tap * CO;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float *rw = red + w * MT * 16 * 16;
      if (CO == 16) v += rw[tap * 16 + co];
      else v += rw[tap * 16 + co] + rw[(tap + 9) * 16 + 8 + co];
    }
    slab[i] = v;
  }
}

static int run_h(const tem_bww_args *a, hipStream_t st, bool dry, int *nslab_out, char *name, int name_len) {
  const tem_view &x = a->in0, &g = a->dout;                   // bf16 tensors behind the float* fields, strides in elements
  const bool cube = a->kd == 3 && a->kh == 3 && a->kw == 3 && a->sd == 1 && a->sh == 1 && a->sw == 1 && a->pd == a->ph &&
                    a->ph == a->pw && a->pd >= 0;
  if (!cube || x.C != 1 || a->in1.ptr || (g.C != 8 && g.C != 16) || x.D < 2 || x.N != g.N) return TEM_EUNSUPPORTED;
  // X rows arrive as 4-byte chunks of 2 voxels: even W, even strides and a 4-byte aligned origin
  if (x.sW != 1 || x.W % 2 || x.sH % 2 || x.sD % 2 || x.sN % 2 || ((uintptr_t)x.ptr & 3)) return TEM_EUNSUPPORTED;
  auto span = [](const tem_view &v) {
    return (int64_t)(v.N - 1) * v.sN + (int64_t)(v.D - 1) * v.sD + (int64_t)(v.H - 1) * v.sH + (int64_t)(v.W - 1) * v.sW + v.C;
  };
  if (span(x) >= ((int64_t)1 << 30) || span(g) >= ((int64_t)1 << 30)) return TEM_EUNSUPPORTED;
  if (((uintptr_t)g.ptr & 15) || g.sW % 8 || g.sH % 8 || g.sD % 8 || g.sN % 8) return TEM_EUNSUPPORTED;
  const int CO = g.C, PY = 8;
  C1Dev p{};
  p.x = x.ptr; p.xN = (int)x.sN; p.xD = (int)x.sD; p.xH = (int)x.sH; p.xW = (int)x.sW;
  p.D = x.D; p.H = x.H; p.W = x.W;
  p.g = g.ptr; p.gN = (int)g.sN; p.gD = (int)g.sD; p.gH = (int)g.sH; p.gW = (int)g.sW;
  p.OD = g.D; p.OH = g.H; p.OW = g.W;
  p.P = a->pd;
  p.nyb = (g.H + PY - 1) / PY; p.nxb = (g.W + 15) / 16;
  const int cols = g.N * p.nyb * p.nxb;
  const int cap = a->nslab > 0 ? a->nslab : 1024;
  if (cols > cap) return TEM_EUNSUPPORTED;
  int zsegs = std::max(1, std::min((1536 + cols / 2) / cols, std::max(1, g.D / 8)));
  int zper = (g.D + zsegs - 1) / zsegs;
  if (CO == 8) zper += zper & 1;
  if ((int64_t)cols * ((g.D + zper - 1) / zper) > cap) return TEM_EUNSUPPORTED;
  zsegs = (g.D + zper - 1) / zper;
  p.zsegs = zsegs; p.zper = zper;
  const int nblocks = cols * zsegs;
  const int NZ = CO == 8 ? 2 : 1, MT = CO == 8 ? 3 : 2, Q = 3;
  const size_t lds = std::max<size_t>((size_t)4 * ((size_t)NZ * Q * (2 * 16 * CO + 32) + (size_t)(NZ * Q + 2) * 160) * 2, (size_t)4 * MT * 16 * 16 * 4);
  const int64_t gspan = ((int64_t)(g.D - 1) * g.sD + (int64_t)(g.H - 1) * g.sH + (int64_t)(g.W - 1) * g.sW + g.C) * 2;
  p.gspan = (int)gspan;
  p.xspan = (int)(((int64_t)(x.D - 1) * x.sD + (int64_t)(x.H - 1) * x.sH + (int64_t)(x.W - 1) * x.sW + 1) * 2);
  if (nslab_out) *nslab_out = nblocks;
  if (name) snprintf(name, name_len, "bww_c1m_h_k<%d>", CO);
  if (dry) return TEM_OK;
  if (!a->slabs || a->nslab != nblocks || a->accumulate) return TEM_EINVAL;
  p.slabs = a->slabs; p.slab_stride = a->slab_stride ? a->slab_stride : (int64_t)27 * CO;
  if (CO == 8) hipLaunchKernelGGL(bww_c1m_h_k<8>, dim3(nblocks), dim3(256), lds, st, p);
  else hipLaunchKernelGGL(bww_c1m_h_k<16>, dim3(nblocks), dim3(256), lds, st, p);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

}  // namespace bwwc1

// Same contract as tem_bww_lds_try (conv_bww.hip): dry = only report support and the slab count.
int tem_bww_c1_try(const tem_bww_args *a, hipStream_t st, bool dry, int *nslab_out) {
  return bwwc1::run(a, st, dry, nslab_out, nullptr, 0);
}

int tem_bww_c1_describe(const tem_bww_args *a, char *buf, int len) { return bwwc1::run(a, nullptr, true, nullptr, buf, len); }

// bf16 inputs (bww_bf16.hip dispatches here first for the one-channel layers)
int tem_bww_c1_bf16_try(const tem_bww_args *a, hipStream_t st, bool dry, int *nslab_out, char *name, int name_len) {
  return bwwc1::run_h(a, st, dry, nslab_out, name, name_len);
}
