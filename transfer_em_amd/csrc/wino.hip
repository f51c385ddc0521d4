// wino.hip -- 3x3x3 stride-1 convolution (forward and input-gradient) of the narrow 3-D layers: Winograd
// F(2x2, 3x3) on the (y, x) axes, direct sum over the three z taps, on the fp32 matrix cores.
//
//   Y[z] = sum_kz A^T [ sum_ci (G g[kz] G^T)(ci,co) (.) (B^T d[z + kz] B)(ci) ] A
//
// A 2x2 output tile of one plane needs a 4x4 input tile of three planes and 3 * 16 element-wise products per (ci, co)
// instead of 27 * 4 multiply-adds: 2.25x fewer MFMA flops than the direct form, which is what bounds these layers
// (C_in, C_out <= 16: the direct kernels run at 45 % of the matrix peak, far from HBM).  [The full 3-D form
// F(2x2x2, 3x3x3) saves 3.375x but needs 64 accumulator tiles = all 256 AGPRs of a lane: one wave per SIMD, and with
// nothing to switch to every LDS, VALU and store latency of the transforms is exposed -- measured slower than this.]
//
// GEMM view per Winograd point p (16) and z tap:  M_p[16 tiles][16 co] += V_p[16 tiles][4 ci] * U_p[4 ci][16 co]
//   (v_mfma_f32_16x16x4_f32).  The A operand map of that instruction -- lane = (tile = lane & 15, k = lane >> 4) --
//   is exactly "one (tile, channel) pair per lane": every lane reads the 4x4 raw voxels of ITS pair from the LDS image
//   of the input plane, runs the input transform in its own registers (packed fp32 adds on two channels) and the
//   transformed values ARE the A fragments: no shuffle, no LDS round trip between transform and MFMA.  The C/D map
//   (col = co, row = tile) leaves a lane with all 16 points of (4 tiles, 1 co): the output transform is in-lane too.
//   U (the transformed kernel, tem_winograd_weights) sits in LDS in fragment order (conflict-free ds_read_b64).
//
// Data movement: a workgroup of 8 waves (two per SIMD: one wave's transforms / stores run under the other's MFMAs)
// owns a block of BY x BX tiles (four 16-tile MFMA row blocks, two waves each: one per output plane of the step) and
// marches along z two output planes at a time over a ring of 4 input planes in LDS.  The 2 new planes of a step arrive
// by LDS-DMA (buffer_load ... lds) over the 2 oldest planes once every wave is past them (two thirds into the step).
// Zero padding (input-gradient: pad 2) = out-of-range buffer offsets, which arrive as zeros; concat inputs / split
// outputs as in conv_lds.hip.
//
// Reference call sites: Conv3D(filters, 3) of models/utils.py:73,122 and generator.py:96 and their
// Conv3DBackpropInput.  fp32 throughout; the result differs from the direct form by rounding only
// (|err| ~ 1e-6 relative, tests/test_gpu_wino.py), inside north_star's 1e-3.
#include "tem_common.h"
#include "wino_common.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace wino {

// device-side diagnostics (phase stamps, ablation flags) exist only in -DTEM_DEBUG_KNOBS builds: in the shipped kernels they
// cost registers (18 VGPRs of stamp sums) and scalar branches inside the step loop
#ifdef TEM_DEBUG_KNOBS
#define KDBG(x) (x)
#else
#define KDBG(x) 0
#endif


struct Ep32 {
  float slope;
  const float *gate; int32_t gN, gD, gH, gW; float gate_slope;
  const uint8_t *keep_mask;      // dropout bits drawn by the forward pass (keep_mode 2) or NULL
  int32_t doz, doy, dox, dD, dH, dW;
  int32_t gbytes, mbytes;        // bytes one sample of the gate view spans / bytes of the keep mask: buffer ranges
};

struct Dev {
  const float *in0, *in1;
  int32_t i0N, i0D, i0H, i0W, i1N, i1D, i1H, i1W;
  int32_t C0;
  int32_t N, D, H, W;
  const float *u;
  float *out0, *out1;
  int32_t o0N, o0D, o0H, o0W, o1N, o1D, o1H, o1W;
  int32_t CO0;
  int32_t OD, OH, OW;
  int32_t P;
  int32_t BY, BX, nby, nbx, zsegs, zper, NTZ;
  int32_t E, PLC, subb, slotb;   // even (= odd) x positions per plane row (BX + 1), 16-byte chunks and bytes per sub-image, bytes per ring slot
  int32_t ndma;                  // LDS-DMA wave-instructions per sub-image (subb / 1024)
  int32_t span0, span1;          // bytes one z-plane of in0 / in1 spans (buffer range of the plane's loads)
  uint32_t magicBX, magicE;
  int32_t dbg;
  int32_t obytes;                // bytes one sample of out0 spans (buffer range of the EP 0 / 1 stores)
  unsigned long long *stamps;    // diagnostic: per-phase cycle sums [block][wave][8] (null in normal runs)
  Ep32 ep;
};

// EP: compiled epilogue -- 0: LeakyReLU(slope) (forward layers); 1: LeakyReLU' gate on the saved activation (input-gradients);
// 2: gate, the forward pass's dropout keep bits and a second output tensor (input-gradient of a concat through Dropout)
// C_out = 32: two 16-channel column blocks; the 8 waves are (2 row blocks) x (2 column blocks) x (2 output planes) and a
// workgroup owns 32 tiles.  C_in = 32 (STREAM): the transformed kernel (98 / 196 KB) does not fit beside the ring; it
// streams through two 16 / 32 KB LDS buffers in (z tap, channel-half pair) chunks, one chunk ahead of its use.  32 -> 16:
// the ring of a 32-channel input leaves room for 48 tiles (three row blocks; the fourth pair of waves idles along).
// EE: the LDS image's row pitch in voxels per (row, x parity) = compile time (9 for tile blocks up to 8 wide, 17 up to 16):
// the 16 raw reads of a (z tap, channel half) are then 4 lane bases + immediates, and the DMA's index split is a constant
// division (with a run-time pitch every read had its own address add: ~150 vector instructions per step and wave, a quarter
// of the loop's vector work -- which is matrix-pipe time here).
template <int CI, int CO, int NI, int EP, int EE>
__global__ __launch_bounds__(512) void wino_conv_k(Dev p) {
  constexpr int NH = CI / 8, VB = 32;                        // channel-pair halves = sub-images (8 channels); bytes per sub-image voxel
  constexpr int NB = (CO + 15) / 16;                         // 16-channel column blocks
  constexpr bool STREAM = CI == 32;
  // PAIR (8 -> 8 channels): the 16 columns of a tile are (2 output planes x 8 co) -- input plane pl of the step multiplies
  // with the kernel taps kz = pl - s of both output planes s at once (a zero block where kz leaves 0..2; fragment layout
  // of tem_winograd_weights for 8 -> 8), 4 MFMA passes per k-step for two planes instead of 6 half-empty ones; every
  // wave owns a 16-tile row block and both planes, a workgroup 128 tiles.
  constexpr bool PAIR = CI == 8 && CO == 8;
  constexpr int NKZ = PAIR ? 4 : 3;                          // z passes per step
  constexpr int UCH = 2 * NB * 16 * 128;                     // floats per streamed chunk: 2 channel halves x NB x 16 points x fragment
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char *const ring = reinterpret_cast<char *>(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, q = lane >> 4;
  const int grp = PAIR ? wave : (NB == 1 ? wave >> 1 : wave >> 2);   // 16-tile row block
  const int nb = NB == 1 ? 0 : (wave >> 1) & 1;              // column block
  // C/D roles.  The kernel fragments are the MFMA's A side and the transformed input its B side: D rows = output channels,
  // columns = tiles, so a lane ends up with FOUR CONSECUTIVE CHANNELS (rows 4 q .. 4 q + 3) of ONE tile (column m) -- its 2x2
  // output voxels leave as four 16-byte stores and need four gate / keep-byte fetches and four offset computations (round 2
  // had tiles as rows: 16 4-byte stores and 16 fetches per lane and plane, ~200 vector instructions per step more -- which
  // on this chip is matrix-pipe time: VALU and fp32 MFMA issue serialise).
  const int zb = PAIR ? (q >> 1) : (wave & 1);               // output plane of the step: per wave, or (PAIR) per row half
  const int co = PAIR ? 4 * (q & 1) : nb * 16 + 4 * q;       // this lane's first output channel (it owns co .. co + 3)
  float *const uld = reinterpret_cast<float *>(ring + 4 * p.slotb);

  int seg = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int zseg = seg % p.zsegs; seg /= p.zsegs;
  const int bx = seg % p.nbx; seg /= p.nbx;
  const int by = seg % p.nby;
  const int n = seg / p.nby;
  const int tz0 = zseg * p.zper, tz1 = min(p.NTZ, tz0 + p.zper), nsteps = tz1 - tz0;
  const int oy0 = by * 2 * p.BY, ox0 = bx * 2 * p.BX;
  const int ntile = p.BY * p.BX;

  // ---- LDS-DMA: byte offset of the lane's chunk inside a z-plane of the source view, relative to the sub-image's first
  // channel (the same for every sub-image, plane and step): computed ONCE and pinned in 2 NI VGPRs.  [Round 2 recomputed
  // them in every step -- ~40 vector instructions per chunk (index split, swizzle, bounds, two 64-bit multiply-adds),
  // 160 per step and wave -- on the belief that a value held across the step would be spilled; the kernels use ~160 of the
  // 256 VGPRs two waves per SIMD allow, and on this chip a vector instruction costs matrix-pipe time: v_mfma_f32 and VALU
  // issue serialise (tests/tools/issue_probe.hip: 4 MFMA + 16 v_add_u32 = 211 cycles, 128 + 76 apart).]
  // Out of range = zero padding / slot padding.
  const bool two_in = p.in1 != p.in0;                      // kernel-uniform
  const float *const in0n = p.in0 + (size_t)n * p.i0N, *const in1n = p.in1 + (size_t)n * p.i1N;
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
      const int c = (cpos ^ swz(e, yr)) * 4;
      const int iy = iy0 + yr, ix = ix0 + 2 * e + o;
      const bool ok = ex && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      voff0[i] = ok ? (iy * p.i0H + ix * p.i0W + c) * 4 : (int)0x80000000;
      voff1[i] = ok ? (iy * p.i1H + ix * p.i1W + c) * 4 : (int)0x80000000;
      asm volatile("" : "+v"(voff0[i]), "+v"(voff1[i]));     // opaque: held, not rematerialised inside the step loop
    }
  }
  auto dma_plane = [&](int iz, int slot) {                 // input plane iz -> ring slot (zeros outside the input)
    const bool zok = (unsigned)iz < (unsigned)p.D;
    const int izc = zok ? iz : 0;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      char *const dst = ring + slot * p.slotb + h * p.subb;
      if (!two_in || 8 * h < p.C0)
        dma_subimage<NI, !STREAM>(in0n + izc * p.i0D + 8 * h, zok ? p.span0 - 32 * h : 0, voff0, dst, wave, p.ndma);
      else
        dma_subimage<NI, !STREAM>(in1n + izc * p.i1D + (8 * h - p.C0), zok ? p.span1 - 4 * (8 * h - p.C0) : 0, voff1, dst, wave, p.ndma);
    }
  };

  // ---- prologue: the four planes of the first step, U into LDS
  const int izb0 = 2 * tz0 - p.P;
  if (nsteps > 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_plane(izb0 + k, k);
  }
  auto dma_u = [&](int chunk, int buf) {                   // streamed kernel chunk (z tap chunk / 2, half pair chunk % 2) -> buffer
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)(p.u + (size_t)chunk * UCH), 0, UCH * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < UCH * 4 / 1024 / 8; ++i) {
      const int j = wave + 8 * i;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)((char *)uld + buf * (UCH * 4) + j * 1024), 16,
                                               (j * 64 + lane) * 16, 0, 0, 0);
    }
  };
  if (STREAM) {
    dma_u(0, 0);
  } else {
    // (LDS-DMA like the planes: as a copy loop through registers hipcc emitted load -> s_waitcnt vmcnt(0) -> ds_write per
    // iteration -- 3 to 12 serialized memory round trips, each also waiting for the planes' DMA, in EVERY workgroup's prologue)
    constexpr int UF4 = NKZ * NH * NB * 16 * 64 * 2 / 4;
    static_assert(UF4 % 512 == 0, "whole 1 KB wave transfers");
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void *)p.u, 0, UF4 * 16, 0x00020000);
#pragma unroll
    for (int i = 0; i < UF4 / 512; ++i) {
      const int j = wave + 8 * i;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ur, (__attribute__((address_space(3))) void *)((char *)uld + j * 1024), 16,
                                               (j * 64 + lane) * 16, 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the prologue's planes (inline-assembly DMA: no wait of the compiler's)
  __syncthreads();

  constexpr int rowb = EE * VB;                              // bytes per (yr, o) row of a sub-image
  const Ep32 &ep = p.ep;
  float *const out0n = p.out0 + (size_t)n * p.o0N, *const out1n = EP == 2 ? p.out1 + (size_t)n * p.o1N : nullptr;
  const float *const gaten = EP >= 1 ? ep.gate + (size_t)n * ep.gN : nullptr;
  const bool in0c = EP != 2 || co < p.CO0;                   // this lane's channels go to out0 (with the full epilogue)
  // The epilogue runs on the vector pipe the transforms need: everything that does not depend on the output voxel is a lane
  // constant or a scalar.  The lane's destination tensor (EP 2: out0 or out1 by its channel quad) is chosen once; gate
  // values (16 bytes = the lane's 4 channels) and keep bytes come through buffer descriptors -- per fetch ONE select (offset
  // or out-of-range = zero), the 2x2 voxel's displacement in the scalar offset; the keep bits of the lane are bits
  // (co & 4) .. (co & 4) + 3 of its voxel's byte.
  float *const obase = EP == 2 ? (in0c ? out0n : out1n) : out0n;      // (EP 0, 1: wave-uniform -- scalar base + 32-bit offset)
  const int oco = in0c ? co : co - p.CO0;
  const int oD = in0c ? p.o0D : p.o1D, oH = in0c ? p.o0H : p.o1H, oW = in0c ? p.o0W : p.o1W;
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)gaten, 0, EP >= 1 ? ep.gbytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void *)ep.keep_mask, 0, EP == 2 ? ep.mbytes : 0, 0x00020000);
  const int mpv = EP == 2 ? p.CO0 >> 3 : 0;                  // mask bytes per voxel
  const int kshift = co & 4;
  int gso[4], mso[4], oso[4];                                // scalar offsets of the 2x2 voxels (bytes)
#pragma unroll
  for (int o4 = 0; o4 < 4; ++o4) {
    gso[o4] = EP >= 1 ? ((o4 >> 1) * ep.gH + (o4 & 1) * ep.gW) * 4 : 0;
    mso[o4] = EP == 2 ? ((o4 >> 1) * ep.dW + (o4 & 1)) * mpv : 0;
    oso[o4] = ((o4 >> 1) * p.o0H + (o4 & 1) * p.o0W) * 4;
  }
  // EP 0 / 1 (one output tensor; C_in < 32): stores through a buffer descriptor too -- an output outside the tensor is an out-of-range
  // offset (dropped) instead of an exec-mask region with its branch around every store
  const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)out0n, 0, EP != 2 ? p.obytes : 0, 0x00020000);

  // A role: this lane's (tile, channel pair) of the wave's row block
  const int tA = min(grp * 16 + m, ntile - 1);
  const int tyA = (int)fdiv((uint32_t)tA, (uint32_t)p.BX, p.magicBX), txA = tA - tyA * p.BX;
  const int a0 = (4 * tyA * EE + txA) * VB + (q & 1) * 8;
  int cs[2][2];                                              // chunk byte offset for (e = tx + a, tile row ty + b)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) cs[a][b] = ((q >> 1) ^ swz(txA + a, 2 * (tyA + b))) * 16;

#ifdef TEM_DEBUG_KNOBS
  unsigned long long t_last = p.stamps ? clock64() : 0, t_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i) do { if (p.stamps) { unsigned long long t_now = clock64(); t_sum[i] += t_now - t_last; t_last = t_now; } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
  f32x4 acc[16];
  f32x4 gv[4];                                             // gate values / keep bytes of the lane's 4 output voxels (2x2 of its tile, its 4
  uint32_t kb[4];                                          //   channels), fetched a step ahead of use
  // the lane's tile (column m of the row block): its output voxel (0, 0) and the validity of its 2x2 voxels -- computed once
  // (the tiles do not move along z); offsets are a few multiply-adds from these
  uint32_t tyx, tok4 = 0;
  {
    const int t = grp * 16 + m;
    const int ty = (int)fdiv((uint32_t)t, (uint32_t)p.BX, p.magicBX), tx = t - ty * p.BX;
    const int oy = oy0 + 2 * ty, ox = ox0 + 2 * tx;
    const bool tok = co < CO && t < ntile;
#pragma unroll
    for (int o4 = 0; o4 < 4; ++o4) tok4 |= ((tok && oy + (o4 >> 1) < p.OH && ox + (o4 & 1) < p.OW) ? 1u : 0u) << o4;
    tyx = ((uint32_t)oy << 16) | (uint32_t)ox;
  }
  auto tile_geom = [&](int oz, int &o0, int &go, int &mb, uint32_t &okm) {
    uint32_t yx = tyx;
    asm volatile("" : "+v"(yx));                             // per-use recompute: no per-output offsets hoisted out of the step loop
    const int oy = (int)(yx >> 16), ox = (int)(yx & 0xffffu);
    okm = oz < p.OD ? tok4 : 0u;
    o0 = oz * oD + oy * oH + ox * oW + oco;
    go = EP >= 1 ? (oz * ep.gD + oy * ep.gH + ox * ep.gW + co) * 4 : 0;
    mb = EP == 2 ? (int)(((((uint32_t)n * ep.dD + (oz + ep.doz)) * ep.dH + (oy + ep.doy)) * ep.dW + (ox + ep.dox)) * (uint32_t)mpv) + (co >> 3) : 0;
  };
  auto fetch_ep = [&](int oz) {
    int o0, go, mb;
    uint32_t okm;
    tile_geom(oz, o0, go, mb, okm);
    if (!in0c) okm = 0;
#pragma unroll
    for (int o4 = 0; o4 < 4; ++o4) {
      const bool okf = (okm >> o4) & 1u;
      int goff = okf ? go : (int)0x80000000;
      asm volatile("" : "+v"(goff));
      const u32x4 g4 = __builtin_amdgcn_raw_buffer_load_b128(grs, goff, gso[o4], 0);
      gv[o4] = f32x4{__uint_as_float(g4.x), __uint_as_float(g4.y), __uint_as_float(g4.z), __uint_as_float(g4.w)};
      if (EP == 2) {
        int moff = okf ? mb : (int)0x80000000;
        asm volatile("" : "+v"(moff));
        kb[o4] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(mrs, moff, mso[o4], 0) >> kshift;
      }
    }
  };
  // output transform A^T M A on (y, x) and epilogue of plane oz; lane = (tile m of the row block, channels co .. co + 3: the
  // f32x4 components of the accumulators)
  auto finish = [&](int oz) {
    f32x4 yx[4][2], yy[2][2];
#pragma unroll
    for (int a = 0; a < 4; ++a) at4(acc[a * 4 + 0], acc[a * 4 + 1], acc[a * 4 + 2], acc[a * 4 + 3], yx[a][0], yx[a][1]);
#pragma unroll
    for (int b = 0; b < 2; ++b) at4(yx[0][b], yx[1][b], yx[2][b], yx[3][b], yy[0][b], yy[1][b]);
    int o0, go, mb;
    uint32_t okm;
    tile_geom(oz, o0, go, mb, okm);
#pragma unroll
    for (int o4 = 0; o4 < 4; ++o4) {
      const bool ok = (okm >> o4) & 1u;
      f32x4 val = yy[o4 >> 1][o4 & 1];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (EP == 0) val[r] = fmaxf(val[r], ep.slope * val[r]);         // LeakyReLU for 0 <= slope <= 1 (host): max instead of compare + select
        if (EP >= 1 && in0c) {
          val[r] = gv[o4][r] > 0.f ? val[r] : ep.gate_slope * val[r];
          if (EP == 2) val[r] = ((kb[o4] >> r) & 1u) ? 2.f * val[r] : 0.f;
        }
      }
      if (EP == 2 || STREAM) {                               // (32 input channels: measured 5 % slower with buffer stores)
        if (ok) *reinterpret_cast<f32x4 *>(obase + o0 + (o4 >> 1) * oH + (o4 & 1) * oW) = val;
      } else {
        // (the voxel's displacement goes into the VECTOR offset, the scalar offset stays 0: a 16-byte buffer store with an
        // SGPR offset whose data registers the next vector instruction overwrites lost data on gfx950 -- the gated epilogue's
        // v_pk_mul right behind the store corrupted its second component; hipcc pads that hazard only for immediate offsets)
        int so = ok ? o0 * 4 + oso[o4] : (int)0x80000000;
        asm volatile("" : "+v"(so));
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(val[0]), __float_as_uint(val[1]), __float_as_uint(val[2]), __float_as_uint(val[3])},
                                               ors, so, 0, 0);
      }
    }
  };

  for (int step = 0; step < nsteps; ++step) {
    const int tz = tz0 + step, oz = 2 * tz + zb, izb = 2 * tz - p.P;
    const bool more = step + 1 < nsteps;
    const int sA = (step & 1) ? 2 : 0;                     // ring slots of the step's input planes 0,1 (2,3 are in the other pair)
    STAMP(0);                                              // barrier B / loop overhead
    if (EP >= 1) fetch_ep(oz);

#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kz = 0; kz < NKZ; ++kz) {
      // input plane zb + kz of the step (PAIR: plane kz): planes 0,1 live in slots sA, sA+1, planes 2,3 in the other pair
      const int pl = PAIR ? kz : zb + kz;
      const char *plane = ring + ((pl < 2 ? sA : 2 - sA) + (pl & 1)) * p.slotb + a0;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        f32x2 v[4][4];
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
          for (int dx = 0; dx < 4; ++dx)
            v[dy][dx] = *reinterpret_cast<const f32x2 *>(plane + h * p.subb + (2 * dy + (dx & 1)) * rowb + (dx >> 1) * VB + cs[dx >> 1][dy >> 1]);
        if (kz == 1 && h == NH - 1) {
          // every wave is past the step's planes 0 and 1 (wave zb = 0 reads 0,1,2; zb = 1 reads 1,2,3 in this order):
          // they make room for the next step's.  (hipcc puts vmcnt(0) in front of the next LDS read -- it cannot tell the
          // ring slots apart -- so the fetch is waited for inside the step; handing the slots over after the step's LAST
          // reads, with the explicit wait at the step's end, measured 2-4 % slower: less work left to cover the latency.)
          __syncthreads();
          if (more && !(KDBG(p.dbg & 4))) { dma_plane(izb + 4, sA); dma_plane(izb + 5, sA + 1); }
        }
        // input transform B^T d B on (y, x), both channels of the pair at once
#pragma unroll
        for (int a = 0; a < 4; ++a) bt4(v[a][0], v[a][1], v[a][2], v[a][3]);
#pragma unroll
        for (int b = 0; b < 4; ++b) bt4(v[0][b], v[1][b], v[2][b], v[3][b]);
        if (STREAM && !(h & 1)) {
          // chunk boundary: every wave is done with the previous chunk's buffer and this chunk has landed (the barrier of
          // the step's end serves chunk 0); the next chunk -- or the next step's first -- starts flying into the other buffer
          const int c = kz * 2 + (h >> 1);
          if (c > 0) __syncthreads();
          if (c < 5) dma_u(c + 1, (c + 1) & 1);
          else if (more) dma_u(0, 0);
        }
        const float *uh = STREAM ? uld + ((kz * 2 + (h >> 1)) & 1) * UCH + (((h & 1) * NB + nb) * 16) * 128 + lane * 2
                                 : uld + (((kz * NH + h) * NB + nb) * 16) * 128 + lane * 2;
        f32x2 uf[16];
#pragma unroll
        for (int pt = 0; pt < 16; ++pt) uf[pt] = *reinterpret_cast<const f32x2 *>(uh + pt * 128);
#pragma unroll
        for (int pt = 0; pt < 16; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[pt].x, v[pt >> 2][pt & 3].x, acc[pt], 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < 16; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[pt].y, v[pt >> 2][pt & 3].y, acc[pt], 0, 0, 0);
      }
    }
    STAMP(1);                                              // reads + transforms + MFMAs (+ ring hand-over)

    finish(oz);
    STAMP(3);                                              // epilogue
    // every wave's DMA of the next step's planes (and, streamed kernels, of the next kernel chunk) has LANDED before the
    // barrier lets anyone read them.  Explicit: until round 3 this held only because hipcc puts vmcnt(0) in front of the first
    // LDS read that follows an LDS-DMA in program order -- which drained the fetch inside the same step by accident of the
    // schedule; the 8 -> 8 gated variant, rescheduled by the new epilogue, read planes another wave's DMA had not delivered.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  static_assert(!PAIR || EP <= 1, "8 -> 8: forward and gated input-gradient only");

#ifdef TEM_DEBUG_KNOBS
  if (p.stamps && lane == 0) {
    for (int i = 0; i < 8; ++i) p.stamps[((size_t)blockIdx.x * 8 + wave) * 8 + i] = t_sum[i];
  }
#endif
#undef STAMP
}

// ------------------------------------------------------------------------------------------ weights
// U[kz]_p(ci,co) = sum_{ky,kx} G[py][ky] G[px][kx] w((kz,ky,kx), ci, co), stored in B-fragment order:
//   u[((((kz*NH + h)*NB + nb)*16 + p)*64 + (q*16 + co % 16))*2 + j]  with  ci = 8h + 2q + j, nb = co / 16,
//   NH = ci/8, NB = ceil(co/16)   (co >= C_out: 0)
__global__ __launch_bounds__(256) void wino_weights_k(const float *theta, float *u, const tem_wino_layer *layers) {
  const tem_wino_layer L = layers[blockIdx.x];
  const int nb = blockIdx.z, NB = (L.co + 15) / 16;
  const int co = nb * 16 + (threadIdx.x & 15), ci = (threadIdx.x >> 4) + 16 * blockIdx.y;
  if (ci >= L.ci || nb >= NB) return;
  const int NH = L.ci / 8;
  const bool pair = L.ci == 8 && L.co == 8;                  // 8 -> 8: columns = (output plane s, co), one fragment set per input plane
  const int h = ci >> 3, q = (ci & 7) >> 1, j = ci & 1;
  if (pair) {
    if (nb) return;
    const int c16 = threadIdx.x & 15, s_ = c16 >> 3, c8 = c16 & 7;
    auto g4p = [](float a, float b, float c, float (&o)[4]) {
      o[0] = a; o[1] = 0.5f * (a + b + c); o[2] = 0.5f * (a - b + c); o[3] = c;
    };
    for (int pl = 0; pl < 4; ++pl) {
      const int kz = pl - s_;
      float gy[4][4];
      if (kz >= 0 && kz <= 2) {
        float g[3][3], gx[3][4];
        for (int t9 = 0; t9 < 9; ++t9) {
          const int t = kz * 9 + t9;
          g[t9 / 3][t9 % 3] = L.flip ? theta[L.src_off + ((int64_t)(26 - t) * 8 + c8) * 8 + ci] : theta[L.src_off + ((int64_t)t * 8 + ci) * 8 + c8];
        }
        for (int a = 0; a < 3; ++a) g4p(g[a][0], g[a][1], g[a][2], gx[a]);
        for (int x = 0; x < 4; ++x) {
          float o[4];
          g4p(gx[0][x], gx[1][x], gx[2][x], o);
          for (int y = 0; y < 4; ++y) gy[y][x] = o[y];
        }
      } else {
        for (int y = 0; y < 4; ++y) for (int x = 0; x < 4; ++x) gy[y][x] = 0.f;
      }
      float *d = u + L.dst_off + ((int64_t)(pl * 16) * 64 + (q * 16 + c16)) * 2 + j;
      for (int pt = 0; pt < 16; ++pt) d[pt * 128] = gy[pt >> 2][pt & 3];
    }
    return;
  }
  auto g4 = [](float a, float b, float c, float (&o)[4]) {
    o[0] = a; o[1] = 0.5f * (a + b + c); o[2] = 0.5f * (a - b + c); o[3] = c;
  };
#pragma unroll
  for (int kz = 0; kz < 3; ++kz) {
    float g[3][3];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      const int t = kz * 9 + t9;
      float w = 0.f;
      if (co < L.co) w = L.flip ? theta[L.src_off + ((int64_t)(26 - t) * L.co + co) * L.ci + ci]
                                : theta[L.src_off + ((int64_t)t * L.ci + ci) * L.co + co];
      g[t9 / 3][t9 % 3] = w;
    }
    float gx[3][4], gy[4][4];
#pragma unroll
    for (int a = 0; a < 3; ++a) g4(g[a][0], g[a][1], g[a][2], gx[a]);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      float o[4];
      g4(gx[0][x], gx[1][x], gx[2][x], o);
#pragma unroll
      for (int y = 0; y < 4; ++y) gy[y][x] = o[y];
    }
    float *d = u + L.dst_off + ((int64_t)(((kz * NH + h) * NB + nb) * 16) * 64 + (q * 16 + (co & 15))) * 2 + j;
#pragma unroll
    for (int pt = 0; pt < 16; ++pt) d[pt * 128] = gy[pt >> 2][pt & 3];
  }
}

// ------------------------------------------------------------------------------------------ host
constexpr int LDS_MAX = 160 * 1024;
static thread_local char *g_name = nullptr;
static thread_local int g_name_len = 0;

template <int CI, int CO, int NI>
int plan(Dev &p, double *cost, size_t *lds_bytes, int EE) {
  constexpr int NH = CI / 8, NB = (CO + 15) / 16;
  constexpr bool PAIR = CI == 8 && CO == 8;
  const size_t ubytes = CI == 32 ? (size_t)2 * (2 * NB * 16 * 128) * 4                       // streamed: two chunk buffers
                                 : (size_t)(PAIR ? 4 : 3) * NH * NB * 16 * 64 * 2 * 4;
  const int TY = (p.OH + 1) / 2, TX = (p.OW + 1) / 2;
  double best = 1e300;
  static const int knob_cus = tem_env_int("TEM_WINO_CUS", 256), knob_cuw = tem_env_int("TEM_WINO_CUW", 400);
  for (int by = 1; by <= TY && by <= 64; ++by)
    for (int bx = 1; bx <= TX && bx <= 64; ++bx) {
      const int nt = by * bx;
      if (nt > (PAIR ? 128 : 64 / NB)) continue;
      if (bx + 1 > EE) continue;                             // the kernel's compile-time row pitch
      const int E = EE, plv = (2 * by + 2) * 2 * E;
      const int subb = (plv * 32 + 1023) & ~1023, slotb = NH * subb;
      const size_t bytes = (size_t)4 * slotb + ubytes;
      const int ndma = subb / 1024;
      if (bytes > (size_t)LDS_MAX || ndma > 8 * NI) continue;
      const int nby = (TY + by - 1) / by, nbx = (TX + bx - 1) / bx;
      const int cols = p.N * nby * nbx;
      // cycles per step and SIMD: the MFMA stream of its two waves (3 z taps x NH x 32 each, 45 reference cycles per
      // MFMA at the throttled matrix clock) with the transforms / stores of one wave under the other's -- a SIMD with
      // ONE busy wave (<= 2 row blocks) takes as long, its latencies exposed (measured); a row block that wraps tile
      // rows pays a few LDS bank conflicts; prologue = four planes + U at the CU's HBM share
      const int wraps = (16 % bx) ? 1 : 0;
      const double step = 2.0 * (PAIR ? 4 : 3) * NH * 32 * 45.0 * (1.0 + 0.03 * wraps) + 3500.0 + (CI == 32 ? 3000.0 : 0.0);
      const double pro = 6000.0 + (4.0 * slotb + ubytes) / 10.0;
      for (int zs = 1; zs <= p.NTZ; ++zs) {
        const int zper = (p.NTZ + zs - 1) / zs, zsegs = (p.NTZ + zper - 1) / zper;
        if (zsegs != zs) continue;
        // latency of the launch alone (rounds of 256 workgroups) + 4 x its CU-time share: the step runs four streams that end
        // together (cgan.py), so a launch's CU-time is what it costs -- a plan with fewer, longer z-runs pays fewer prologues
        // (four planes + U: ~1.2 steps) even where it leaves CUs idle when the launch is timed alone.  fp32 step by this weight
        // (with 3 x in wino_bww_k): 0: 7.24 ms, 0.5: 7.11, 1.5: 7.06, 4: 7.02, 10: 7.05; the dominant launch alone 79.9 -> 84 us.
        const double t = std::ceil(cols * zsegs / (double)knob_cus) * (pro + zper * step) +
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
int plan_memo(Dev &p, double *cost, size_t *lds_bytes, int EE) {
  struct R { int rc, BY, BX, nby, nbx, zsegs, zper, E, PLC, subb, slotb, ndma; size_t lds; double cost; };
  static tem_plan_cache<5, R> cache;
  const std::array<int, 5> key{p.N, p.OH, p.OW, p.NTZ, EE};
  R r;
  if (!cache.get(key, r)) {
    r.lds = 0; r.cost = 1e300;
    r.rc = plan<CI, CO, NI>(p, &r.cost, &r.lds, EE);
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

template <int CI, int CO, int NI, int EP, int EE>
static int launch(const Dev &p, size_t lds_bytes, hipStream_t st) {
  const int nblocks = p.N * p.nby * p.nbx * p.zsegs;
  auto kern = wino_conv_k<CI, CO, NI, EP, EE>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(512), lds_bytes, st, p);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

template <int CI, int CO, int NI, int EP>
int run_best(Dev p, hipStream_t st, bool dry) {
  // the compiled row pitches: the cheapest plan wins (ties: the narrower pitch)
  static int force = -1;
  if (force < 0) force = tem_env_int("TEM_WINO_EE", 0);
  const int pitches[2] = {9, 17};      // (11 measured too: never better than 9 on the step's shapes)
  Dev best = p;
  double cbest = 1e300;
  size_t lds_bytes = 0;
  for (int i = 0; i < 2; ++i) {
    if (force && pitches[i] != force) continue;
    Dev q = p;
    double c = 1e300;
    size_t l = 0;
    if (plan_memo<CI, CO, NI>(q, &c, &l, pitches[i]) == TEM_OK && c < cbest) { best = q; cbest = c; lds_bytes = l; }
  }
  if (cbest >= 1e300) return TEM_EUNSUPPORTED;
  p = best;
  p.magicBX = magic_for(p.BX); p.magicE = magic_for(p.E);
  if (dry) {
    if (g_name) snprintf(g_name, g_name_len, "wino_conv_k<%d, %d, %d, %d, %d>", CI, CO, NI, EP, p.E);
    return TEM_OK;
  }
  const int nblocks = p.N * p.nby * p.nbx * p.zsegs;
  if (KDBG(p.dbg & 8))
    fprintf(stderr, "wino<%d,%d,%d> O=%dx%dx%d: BY=%d BX=%d nby=%d nbx=%d zsegs=%d zper=%d blocks=%d lds=%zu\n", CI, CO, EP,
            p.OD, p.OH, p.OW, p.BY, p.BX, p.nby, p.nbx, p.zsegs, p.zper, nblocks, lds_bytes);
  return p.E == 9 ? launch<CI, CO, NI, EP, 9>(p, lds_bytes, st) : launch<CI, CO, NI, EP, 17>(p, lds_bytes, st);
}

int dispatch(const tem_conv_args *a, hipStream_t st, bool dry) {
  const tem_view &i0 = a->in0, &o0 = a->out0;
  const bool cube = a->kd == 3 && a->kh == 3 && a->kw == 3 && a->sd == 1 && a->sh == 1 && a->sw == 1 && a->pd == a->ph &&
                    a->ph == a->pw;
  if (!cube || a->w_layout != TEM_W_WINOGRAD || a->pd < 0) return TEM_EUNSUPPORTED;
  if (!fits32(i0) || !fits32(o0)) return TEM_EUNSUPPORTED;
  Dev p{};
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
    if (!fits32(i1) || !aligned(i1)) return TEM_EUNSUPPORTED;
    p.in1 = i1.ptr; p.i1N = (int)i1.sN; p.i1D = (int)i1.sD; p.i1H = (int)i1.sH; p.i1W = (int)i1.sW;
    CI += i1.C;
    if (i0.C % 8) return TEM_EUNSUPPORTED;                 // a sub-image (8 channels) has one source
  }
  p.N = i0.N; p.D = i0.D; p.H = i0.H; p.W = i0.W;
  p.span0 = (int)(((int64_t)(i0.H - 1) * p.i0H + (int64_t)(i0.W - 1) * p.i0W + i0.C) * 4);
  p.span1 = a->in1.ptr ? (int)(((int64_t)(i0.H - 1) * p.i1H + (int64_t)(i0.W - 1) * p.i1W + a->in1.C) * 4) : p.span0;
  p.u = a->w;
  p.out0 = o0.ptr; p.o0N = (int)o0.sN; p.o0D = (int)o0.sD; p.o0H = (int)o0.sH; p.o0W = (int)o0.sW; p.CO0 = o0.C;
  {
    const int64_t ospan = (int64_t)(o0.D - 1) * o0.sD + (int64_t)(o0.H - 1) * o0.sH + (int64_t)(o0.W - 1) * o0.sW + o0.C;
    if (ospan >= ((int64_t)1 << 29)) return TEM_EUNSUPPORTED;            // byte offsets below 2^31
    p.obytes = (int)(ospan * 4);
  }
  int CO = o0.C;
  if (a->out1.ptr) {
    const tem_view &o1 = a->out1;
    if (o1.N != o0.N || o1.D != o0.D || o1.H != o0.H || o1.W != o0.W) return TEM_ESHAPE;
    if (!fits32(o1)) return TEM_EUNSUPPORTED;
    p.out1 = o1.ptr; p.o1N = (int)o1.sN; p.o1D = (int)o1.sD; p.o1H = (int)o1.sH; p.o1W = (int)o1.sW;
    CO += o1.C;
  }
  if (o0.N != i0.N) return TEM_ESHAPE;
  if (o0.D != i0.D + 2 * a->pd - 2 || o0.H != i0.H + 2 * a->ph - 2 || o0.W != i0.W + 2 * a->pw - 2) return TEM_ESHAPE;
  p.OD = o0.D; p.OH = o0.H; p.OW = o0.W;
  p.NTZ = (p.OD + 1) / 2;
  p.P = a->pd;
  {
    static int dbg = -1;
    if (dbg < 0) dbg = tem_env_int("TEM_DEBUG_FLAGS", 0);
    p.dbg = dbg;
    static unsigned long long stamp_ptr = ~0ull;
    if (stamp_ptr == ~0ull) stamp_ptr = tem_env_hex("TEM_WINO_STAMP_BUF");
    p.stamps = (unsigned long long *)stamp_ptr;
  }
  const tem_epilogue &e = a->ep;
  Ep32 &q = p.ep;
  if (e.bias || e.add.ptr) return TEM_EUNSUPPORTED;        // no k3 s1 layer of the step has them
  q.slope = e.slope; q.gate_slope = e.gate_slope;
  if (!(e.slope >= 0.f && e.slope <= 1.f)) return TEM_EUNSUPPORTED;      // the epilogue takes max(v, slope v)
  int EP = 0;
  if (e.gate.ptr) {
    const tem_view &g = e.gate;
    if (g.N != o0.N || g.D != o0.D || g.H != o0.H || g.W != o0.W || g.C < o0.C) return TEM_ESHAPE;
    if (!fits32(g) || e.slope != 1.f) return TEM_EUNSUPPORTED;
    q.gate = g.ptr; q.gN = (int)g.sN; q.gD = (int)g.sD; q.gH = (int)g.sH; q.gW = (int)g.sW;
    const int64_t gspan = (int64_t)(g.D - 1) * g.sD + (int64_t)(g.H - 1) * g.sH + (int64_t)(g.W - 1) * g.sW + g.C;
    if (gspan >= ((int64_t)1 << 29)) return TEM_EUNSUPPORTED;            // byte offsets below 2^31
    q.gbytes = (int)(gspan * 4);
    EP = 1;
  }
  if (e.dropout || a->out1.ptr) {
    // only the pair the step uses: the forward pass's keep bits (the Philox form would cost one block per element here)
    // together with the split output, behind a gate
    if (!(e.dropout && e.keep_mask && e.keep_mode == 2 && a->out1.ptr && EP == 1)) return TEM_EUNSUPPORTED;
    q.keep_mask = e.keep_mask;
    q.doz = e.drop_org[0]; q.doy = e.drop_org[1]; q.dox = e.drop_org[2];
    q.dD = e.drop_dims[0] ? e.drop_dims[0] : o0.D; q.dH = e.drop_dims[0] ? e.drop_dims[1] : o0.H;
    q.dW = e.drop_dims[0] ? e.drop_dims[2] : o0.W;
    if ((int64_t)o0.N * q.dD * q.dH * q.dW * o0.C >= ((int64_t)1 << 32)) return TEM_EUNSUPPORTED;
    if (o0.C % 8) return TEM_EUNSUPPORTED;                               // a lane's keep bit = bit (co & 7) of byte voxel * C/8 + co/8
    q.mbytes = (int)(((int64_t)o0.N * q.dD * q.dH * q.dW * o0.C + 7) / 8);
    EP = 2;
  }
#define WINO_CASE(ci, co, ni, epi) if (CI == ci && CO == co && EP == epi) return run_best<ci, co, ni, epi>(p, st, dry);
  WINO_CASE(16, 16, 2, 0) WINO_CASE(16, 16, 2, 1) WINO_CASE(16, 16, 2, 2)
  WINO_CASE(8, 8, 3, 0) WINO_CASE(8, 8, 3, 1)
  WINO_CASE(8, 16, 2, 0) WINO_CASE(8, 16, 2, 1)
  WINO_CASE(16, 8, 2, 0) WINO_CASE(16, 8, 2, 1)
  WINO_CASE(16, 32, 2, 0) WINO_CASE(16, 32, 2, 1)
  WINO_CASE(32, 32, 2, 0) WINO_CASE(32, 32, 2, 1) WINO_CASE(32, 32, 2, 2)
  WINO_CASE(32, 16, 2, 0) WINO_CASE(32, 16, 2, 1)
#undef WINO_CASE
  return TEM_EUNSUPPORTED;
}

}  // namespace wino

int tem_conv_wino_try(const tem_conv_args *a, hipStream_t st, bool dry) { return wino::dispatch(a, st, dry); }

int tem_conv_wino_describe(const tem_conv_args *a, char *buf, int len) {
  wino::g_name = buf; wino::g_name_len = len;
  int rc = wino::dispatch(a, nullptr, true);
  wino::g_name = nullptr;
  return rc;
}

extern "C" int tem_winograd_weights(const float *theta, float *u, const tem_wino_layer *layers_dev, int32_t nlayers,
                                    tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!theta || !u || !layers_dev || nlayers <= 0) return TEM_EINVAL;
  hipLaunchKernelGGL(wino::wino_weights_k, dim3(nlayers, 2, 2), dim3(256), 0, (hipStream_t)stream, theta, u, layers_dev);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}
