// conv3_bf16.hip -- 3x3x3 stride-1 and 4x4x4 stride-2 convolution forward / input-gradient of the bf16 mode (BASELINE
// config 5), z-marching.
//
// The shape-generic conv_bf16_k spends ~500 vector instructions per 16-voxel tile (index arithmetic, the LDS transpose
// of its epilogue, a kernel copy staged per output plane) around 14..54 matrix instructions -- vector and matrix
// instructions do not overlap on gfx950 (tests/tools/issue_probe.hip), so it runs at ~1/5 of what the bytes allow.
// This kernel serves the 3x3x3 layers with 8 / 16 / 32 channels, which hold most of the bf16 step:
//   * a workgroup owns a TX x TY column of the output and marches along z; the input planes live in a 4-slot LDS ring
//     filled by 16-byte LDS-DMA (no registers, no vector instructions; zero padding, cropped views and the volume
//     border are out-of-range buffer offsets = zeros), every input plane is fetched once per column;
//   * the ring plane is the flat [rows][TX + 2] voxel array; a tile is 16 CONSECUTIVE voxels of that flat index (the
//     two halo columns per row are computed and dropped: 2 / (TX + 2) waste instead of ragged row ends), so the LDS
//     address of (tile, tap) is one per-lane register per k-step plus the tile's byte displacement;
//   * one ds_read_b128 per matrix instruction is exactly the LDS array's rate (256 B/clk/CU against 16 cycles per
//     v_mfma_f32_16x16x32_bf16), so the reads must be conflict-free: measured (tests/tools/lds_b128_probe*.hip), the plain
//     channels-last image is -- 16 and 32 channels for any row length, 8 channels when the row length is 2 mod 16
//     voxels (the host pads the ring row to that); an XOR permutation of the chunks made it 2x WORSE;
//   * the kernel taps are the A operand (rows = output channels), held in registers for the whole march; the
//     activations are the B operand (columns = voxels): a lane ends up with 4 consecutive output channels of one
//     voxel, which is the layout of the bf16 store, the gate and the skip-gradient -- no transpose;
//   * epilogue as conv_bf16_k: bias (folded into the accumulator's initial value), skip-gradient add, LeakyReLU' gate,
//     dropout by the keep mask the forward pass wrote (a launch that has to DRAW the mask stays in conv_bf16_k),
//     LeakyReLU, split outputs.
// MFMA k index = (tap, ci) as in conv_bf16_k, so the packed kernel [tap][co][ci] is read as it is.
#include "tem_common.h"
#include <cstdio>
#include <cstdlib>

namespace conv3_bf16 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int OOB = (int)0x80000000;
constexpr int LDS_MAX = 160 * 1024;

struct Dev {
  const u16 *in0, *in1;            // (a descriptor each: which kernel runs must not depend on where the allocator put the two
  uint32_t in0_bytes, in1_bytes;   //  inputs of a concat -- a single descriptor over both needs them within 2 GB)
  int32_t ndma_cnt;                // vmcnt events per plane: ndma, twice that for a concat (a wave issues for in0 and for in1)
  int32_t i0N, i0D, i0H, i0W, i1N, i1D, i1H, i1W, C0;
  int32_t D, H, W, P;
  const u16 *w;                    // packed bf16 kernel [tap][co][ci]
  int32_t flip;
  u16 *out0, *out1;                // split outputs: equal strides (one scalar offset per tile), a descriptor each
  uint32_t out0_bytes, out1_bytes;
  int32_t oN, oD, oH, oW, CO0;
  int32_t OD, OH, OW;
  int32_t TX, TY, nbx, nby, zsegs, zper;
  int32_t RW, PV, ndma, slot_bytes;
  int32_t RD;                      // ring slots: 3 planes under the taps + RD - 3 planes in flight
  uint32_t magicRW;                // ceil(2^22 / RW)
  uint32_t magicT;                 // ceil(2^32 / (RW / 16)): tile index -> ring row
  const float *bias;
  float slope;
  const u16 *gate; int32_t gN, gD, gH, gW, gbytes; float gate_slope;
  const u16 *add;  int32_t aN, aD, aH, aW, aoz, aoy, aox, aDd, aHh, aWw, abytes;
  const uint8_t *keep;             // dropout keep mask to READ (1 bit per element of the dropout tensor), or null
  int32_t doz, doy, dox, dD, dH, dW, mbytes;
  int32_t dbg;                     // TEM_DEBUG_KNOBS builds: 1 no stores, 2 no matrix chain, 4 no DMA (timing experiments)
};

__device__ __forceinline__ float bflo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfhi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));     // round to nearest even
}

template <int CI, int CO, int K, int S, int NW, bool SPLIT, int EPI>
__global__ __launch_bounds__((NW + 1) * 64) void conv3_bf16_k(Dev p) {
  // SPLIT: the n-tiles of a 32-channel output go to different waves (kernel fragments of both would not fit the registers)
  constexpr int NTAP = K * K * K, NSTEP = (NTAP * CI + 31) / 32, NTA = (CO + 15) / 16, NT = SPLIT ? 1 : NTA, NPG = SPLIT ? NW / NTA : NW;
  constexpr int CPV = CI / 8, TILEB = S * 32 * CI;           // bytes between the B fragments of x-adjacent tiles
  static_assert((K == 3 && S == 1) || (K == 4 && S == 2), "3x3x3 stride 1 or 4x4x4 stride 2");
  constexpr int MAXJ = 40;                                  // LDS-DMA instructions per plane (slot <= 40 KB)
  static_assert(CI == 8 || CI == 16 || CI == 32, "8 / 16 / 32 input channels");
  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const int SB = p.slot_bytes, RW = p.RW;
  const int ntb = SPLIT ? wave % NTA : 0, pg = SPLIT ? wave / NTA : wave;      // first n-tile, tile-pair group of the wave

  int b = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int zs = b % p.zsegs; b /= p.zsegs;
  const int bx = b % p.nbx; b /= p.nbx;
  const int by = b % p.nby;
  const int n = b / p.nby;
  const int oz0 = zs * p.zper, nz = min(p.zper, p.OD - oz0);
  if (nz <= 0) return;
  const int ox0 = bx * p.TX, oy0 = by * p.TY;
  const int TXo = min(p.TX, p.OW - ox0), TYo = min(p.TY, p.OH - oy0);

  const uint32_t ring0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)ring;

  // ---- wave NW is the producer: it alone issues the LDS-DMA of the ring and waits for it, so the compute waves' vmcnt
  // only ever counts their own gate / skip-gradient loads and stores (in-order returns: a wave that issued the DMA
  // itself would wait for the plane's fetch at the first gate it reads).  Instruction j of a plane fills the 64 chunks
  // [64 j, 64 j + 64) of the slot.  (Inline assembly: for the builtin, hipcc cannot tell the ring slots apart and puts
  // s_waitcnt vmcnt(0) in front of the first LDS read after it.)
  if (wave == NW) {
    const uint32_t r0_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)p.in0), r0_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)p.in0 >> 32) & 0xffffu);
    const uint32_t r1_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)p.in1), r1_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)p.in1 >> 32) & 0xffffu);
    // (a lane's chunk index inside a voxel is the same for every instruction -- 64 j is a multiple of the chunks per voxel -- so
    // which input of a concat it reads and that input's plane stride are lane constants)
    const bool s0 = (lane & (CPV - 1)) * 8 < p.C0;
    const int zsl = (s0 ? p.i0D : p.i1D) * 2;
    const bool concat = p.in1 != p.in0;                        // kernel-uniform
    // one instruction per chunk row, two for a concat (the lanes of in0, then the lanes of in1: the descriptor is scalar).  The
    // in1 lanes are switched on by their EXEC mask inside the assembly -- as a divergent branch per instruction the unrolled
    // producer grew to 15,000 lines and its instruction fetch, not the DMA, paced the ring (g.d1b 17 -> 29 us)
    const uint64_t m1 = __builtin_amdgcn_ballot_w64(!s0), m0 = ~m1;
    int voff[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      voff[j] = OOB;
      if (j < p.ndma) {                                         // (wave-uniform)
        const int g = 64 * j + lane;
        const int v = g / CPV, h = g & (CPV - 1);
        const int r = (int)(((uint32_t)v * p.magicRW) >> 22), cx = v - r * RW;
        const int iy = S * oy0 - p.P + r, ix = S * ox0 - p.P + cx;
        const bool ok = v < p.PV && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const int c = 8 * h;
        if (ok) voff[j] = s0 ? (n * p.i0N + iy * p.i0H + ix * p.i0W + c) * 2 : (n * p.i1N + iy * p.i1H + ix * p.i1W + (c - p.C0)) * 2;
      }
    }
    int slot = 0;
    auto dma = [&](int k) {                                    // input plane S oz0 - P + k -> the next slot of the ring
      const int iz = S * oz0 - p.P + k;
      const bool zok = (unsigned)iz < (unsigned)p.D;
      const uint32_t dst = ring0 + slot * SB;
      slot = slot + 1 == p.RD ? 0 : slot + 1;
      const int zo = iz * zsl;
      // (a chunk outside the input keeps its out-of-range offset: 0x80000000 + zo, zo >= 0 for a plane inside the input -- planes
      // outside it get a zero-length descriptor -- is still beyond every range; a compare per chunk would live in 40 scalar
      // register pairs and spill.  Descriptors and the concat decision once per plane: the producer's ~6 instructions per DMA
      // are what keeps it ahead of the compute waves -- at ~25 it paced the ring: g.f1 31 -> 42 us)
      const u32x4 ra = u32x4{r0_lo, r0_hi, zok ? p.in0_bytes : 0u, 0x00020000u};
      if (p.dbg & 4) return;
      if (!concat) {
#pragma unroll
        for (int j = 0; j < MAXJ; ++j)
          if (j < p.ndma)
            asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst + j * 1024), "v"(voff[j] + zo), "s"(ra) : "memory");
      } else {
        const u32x4 rb = u32x4{r1_lo, r1_hi, zok ? p.in1_bytes : 0u, 0x00020000u};
#pragma unroll
        for (int j = 0; j < MAXJ; ++j)
          if (j < p.ndma)
            asm volatile("s_mov_b32 m0, %0\n\ts_mov_b64 exec, %4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
                         "s_mov_b64 exec, %5\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b64 exec, -1"
                         ::"s"(dst + j * 1024), "v"(voff[j] + zo), "s"(ra), "s"(rb), "s"(m0), "s"(m1) : "memory");
      }
    };
    // Step j (output plane oz0 + j) reads the planes S j .. S j + K - 1; the ring has RD >= K + S slots, so the planes up to
    // S j + RD - 1 may be in it or in flight.  vmcnt returns in order, so "planes <= S j + K - 1 have landed" is
    // vmcnt <= (planes issued after them) x ndma -- an immediate, hence the switch (the host keeps it below 64).
    const int last = S * (nz - 1) + K - 1;
    int issued = -1;                                           // last plane issued
    for (int j = -1; j < nz; ++j) {                            // (j = -1: the prologue -- ONE inlined copy of the unrolled dma())
      const int fly = j < 0 ? -1 : (issued - (S * j + K - 1)) * p.ndma_cnt;
      if (j >= 0)
      switch (fly > 0 ? fly : 0) {
#define TEM_W1(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
#define TEM_W8(a) TEM_W1(a##0) TEM_W1(a##1) TEM_W1(a##2) TEM_W1(a##3) TEM_W1(a##4) TEM_W1(a##5) TEM_W1(a##6) TEM_W1(a##7)
        TEM_W1(0) TEM_W1(1) TEM_W1(2) TEM_W1(3) TEM_W1(4) TEM_W1(5) TEM_W1(6) TEM_W1(7) TEM_W1(8) TEM_W1(9)
        TEM_W1(10) TEM_W1(11) TEM_W1(12) TEM_W1(13) TEM_W1(14) TEM_W1(15) TEM_W1(16) TEM_W1(17) TEM_W1(18) TEM_W1(19)
        TEM_W1(20) TEM_W1(21) TEM_W1(22) TEM_W1(23) TEM_W1(24) TEM_W1(25) TEM_W1(26) TEM_W1(27) TEM_W1(28) TEM_W1(29)
        TEM_W1(30) TEM_W1(31) TEM_W1(32) TEM_W1(33) TEM_W1(34) TEM_W1(35) TEM_W1(36) TEM_W1(37) TEM_W1(38) TEM_W1(39)
        TEM_W1(40) TEM_W1(41) TEM_W1(42) TEM_W1(43) TEM_W1(44) TEM_W1(45) TEM_W1(46) TEM_W1(47) TEM_W1(48) TEM_W1(49)
        TEM_W1(50) TEM_W1(51) TEM_W1(52) TEM_W1(53) TEM_W1(54) TEM_W1(55) TEM_W1(56) TEM_W1(57) TEM_W1(58) TEM_W1(59)
        TEM_W1(60) TEM_W1(61) TEM_W1(62)
#undef TEM_W1
#undef TEM_W8
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
      // the step's planes have landed; past the barrier nobody reads the planes below S j any more
      if (j >= 0) asm volatile("s_barrier" ::: "memory");
      const int upto = (j < 0 ? 0 : S * j) + p.RD - 1;
#pragma unroll 1
      while (issued < last && issued < upto) dma(++issued);
    }
    return;
  }

  // ---- kernel taps: A fragments (row = output channel nt*16 + m, k = 32 s + 8 kq ..+7) in registers for the march
  bf16x8 wf[NSTEP][NT];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    const int e0 = 32 * s + 8 * kq, tap = e0 / CI, c0 = e0 - tap * CI;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = (ntb + nt) * 16 + m;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (tap < NTAP && co < CO) v = *reinterpret_cast<const u32x4 *>(p.w + ((p.flip ? NTAP - 1 - tap : tap) * CO + co) * CI + c0);
      wf[s][nt] = __builtin_bit_cast(bf16x8, v);
    }
  }

  // ---- LDS address of this lane's B fragment (voxel m of the wave's first tile, k-range of kq) per k-step
  uint32_t cur[NSTEP];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    const int e0 = 32 * s + 8 * kq;
    const int tap = min(e0 / CI, NTAP - 1), h = (e0 % CI) >> 3;   // (a padded k-range multiplies zero kernel rows: any address)
    const int dz = tap / (K * K), dy = (tap - K * K * dz) / K, dx = tap - K * K * dz - K * dy;
    const int vrel = S * m + dy * RW + dx;
    cur[s] = ring0 + dz * SB + (vrel * CPV + h) * 16;
  }

  // ---- epilogue roles: a lane holds 4 consecutive channels eco(nt) of the voxel in column m of the tile.  The ring row is a
  // multiple of 16 voxels, so a tile lies in ONE row at x0 = 16 c: its coordinates are wave-uniform, every tensor offset
  // splits into a per-lane constant (m, channel; out of range for a lane without a role) and a SCALAR part that rides in
  // the buffer instructions' soffset -- no vector instruction per tile for addresses.
  constexpr bool GATE = EPI & 1, ADD = EPI & 2, KEEP = EPI & 4, LEAKY = EPI & 8;      // LEAKY: 0 < slope < 1, max(v, slope v)
  const __amdgpu_buffer_rsrc_t ors0 = __builtin_amdgcn_make_buffer_rsrc((void *)p.out0, 0, p.out0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ors1 = __builtin_amdgcn_make_buffer_rsrc((void *)p.out1, 0, p.out1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)p.gate, 0, GATE ? p.gbytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.keep, 0, KEEP ? p.mbytes : 0, 0x00020000);
  int vo[NT], vg[NT], va[NT], vk[NT];
  f32x4 bias4[NT];
  bool first[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int eco = (ntb + nt) * 16 + 4 * kq;
    const bool live = eco < CO;
    first[nt] = eco < p.CO0;
    vo[nt] = !live ? OOB : (first[nt] ? eco * 2 : (eco - p.CO0) * 2) + m * p.oW * 2;
    vg[nt] = (live && first[nt]) ? (eco + m * p.gW) * 2 : OOB;
    va[nt] = (eco + m * p.aW) * 2;
    vk[nt] = (live && first[nt]) ? (m * p.CO0 + eco) >> 3 : OOB;
    bias4[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias && first[nt] && live) bias4[nt] = f32x4{p.bias[eco], p.bias[eco + 1], p.bias[eco + 2], p.bias[eco + 3]};
  }
  const bool split = p.out1 != p.out0;
  const int axl = ox0 - p.aox + m;                             // this lane's column of the skip-gradient window, tile at x0 = 0
  const float gs = p.gate_slope;
  const f32x2 slope2 = f32x2{p.slope, p.slope};
  // stride 1: the ring row is a multiple of 16 voxels and the tiles run flat over the rows (a pair may span two rows);
  // stride 2: tiles are 16 output voxels = 32 input voxels of a ring row, pairs stay inside an output row
  const int TPR = S == 1 ? RW >> 4 : (p.TX + 31) >> 5;        // tiles (stride 2: tile pairs; an edge column masks the surplus) per row
  const int npair = S == 1 ? (TYo * TPR + 1) >> 1 : TYo * TPR;

  // (the kernel / bias loads are consumed here, in front of the loops: left pending into the loop, their first use
  // inside it gets s_waitcnt vmcnt(0) in every iteration -- the waitcnt pass cannot tell that they landed long ago)
#pragma unroll
  for (int s = 0; s < NSTEP; ++s)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(wf[s][nt]));
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(bias4[nt]));
  const uint32_t ring_len = (uint32_t)p.RD * (uint32_t)SB, ring_end = ring0 + ring_len;
  for (int j = 0; j < nz; ++j) {
    asm volatile("s_barrier" ::: "memory");                // the producer's: the step's K input planes are in the ring
    const int oz = oz0 + j;
    const int az = oz - p.aoz;
    const __amdgpu_buffer_rsrc_t ars =
        __builtin_amdgcn_make_buffer_rsrc((void *)p.add, 0, (ADD && (unsigned)az < (unsigned)p.aDd) ? p.abytes : 0, 0x00020000);
    const int so_z = (n * p.oN + oz * p.oD + oy0 * p.oH + ox0 * p.oW) * 2;
    const int sg_z = (n * p.gN + oz * p.gD + oy0 * p.gH + ox0 * p.gW) * 2;
    const int sa_z = (n * p.aN + az * p.aD + (oy0 - p.aoy) * p.aH + (ox0 - p.aox) * p.aW) * 2;
    const int sk_z = ((n * p.dD + oz + p.doz) * p.dH + p.doy + oy0) * p.dW + ox0 + p.dox;       // voxel index of the dropout tensor
    for (int pi = pg; pi < npair; pi += NPG) {                  // wave-uniform
      int rr[2], x0[2], rem[2];
      uint32_t toff;                                            // byte offset of the pair's first B fragment in a ring plane
      if constexpr (S == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int t = 2 * pi + i;
          rr[i] = TPR == 1 ? t : (int)__umulhi((uint32_t)t, p.magicT);    // (the magic of 1 does not fit 32 bits)
          x0[i] = (t - rr[i] * TPR) << 4;
          rem[i] = rr[i] < TYo ? TXo - x0[i] : 0;               // columns of the tile that are output voxels
        }
        toff = (uint32_t)pi * (2 * TILEB);
      } else {
        rr[0] = TPR == 1 ? pi : (int)__umulhi((uint32_t)pi, p.magicT);
        rr[1] = rr[0];
        x0[0] = (pi - rr[0] * TPR) << 5; x0[1] = x0[0] + 16;
        rem[0] = TXo - x0[0]; rem[1] = TXo - x0[1];
        toff = (uint32_t)((S * rr[0] * RW + S * x0[0]) * (CI * 2));
      }
      if (rem[0] > 0 || rem[1] > 0) {
        u32x2 gv[2][NT], av[2][NT];
        uint32_t kb[2][NT];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (m < rem[i]) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              if constexpr (GATE) gv[i][nt] = __builtin_amdgcn_raw_buffer_load_b64(grs, vg[nt], sg_z + (rr[i] * p.gH + x0[i] * p.gW) * 2, 0);
              if constexpr (KEEP)                               // element e = voxel * C_out0 + channel: bit e & 7 of byte e >> 3
                kb[i][nt] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(mrs, vk[nt], (sk_z + rr[i] * p.dW + x0[i]) * (p.CO0 >> 3), 0);
              if constexpr (ADD) {
                const bool ain = first[nt] && (unsigned)(oy0 - p.aoy + rr[i]) < (unsigned)p.aHh && (unsigned)(axl + x0[i]) < (unsigned)p.aWw;
                const int aoff = ain ? va[nt] + sa_z + (rr[i] * p.aH + x0[i] * p.aW) * 2 : OOB;
                av[i][nt] = __builtin_amdgcn_raw_buffer_load_b64(ars, aoff, 0, 0);
              }
            }
          }
        }
        f32x4 acc[2][NT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[i][nt] = bias4[nt];
        // B fragments PD k-steps ahead of their matrix instructions (sched_barrier: hipcc otherwise sinks each read to
        // its use and waits lgkmcnt(0) in front of every instruction of the chain)
        constexpr int PD = NSTEP < 4 ? NSTEP : 4;
        if (!(p.dbg & 2)) {
          typedef const __attribute__((address_space(3))) u32x4 *lptr;
          u32x4 bq[NSTEP][2];
#pragma unroll
          for (int s = 0; s < PD; ++s) {
            bq[s][0] = *reinterpret_cast<lptr>(cur[s] + toff);
            bq[s][1] = *reinterpret_cast<lptr>(cur[s] + toff + TILEB);
          }
#pragma unroll
          for (int s = 0; s < NSTEP; ++s) {
            __builtin_amdgcn_sched_barrier(0);
            if (s + PD < NSTEP) {
              bq[s + PD][0] = *reinterpret_cast<lptr>(cur[s + PD] + toff);
              bq[s + PD][1] = *reinterpret_cast<lptr>(cur[s + PD] + toff + TILEB);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][nt], __builtin_bit_cast(bf16x8, bq[s][0]), acc[0][nt], 0, 0, 0);
              acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][nt], __builtin_bit_cast(bf16x8, bq[s][1]), acc[1][nt], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (m < rem[i] && !(p.dbg & 1)) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              f32x4 vv = acc[i][nt];
              if (first[nt]) {
                if constexpr (ADD) {
                  vv[0] += bflo(av[i][nt].x); vv[1] += bfhi(av[i][nt].x); vv[2] += bflo(av[i][nt].y); vv[3] += bfhi(av[i][nt].y);
                }
                if constexpr (GATE) {
                  vv[0] = bflo(gv[i][nt].x) > 0.f ? vv[0] : gs * vv[0];
                  vv[1] = bfhi(gv[i][nt].x) > 0.f ? vv[1] : gs * vv[1];
                  vv[2] = bflo(gv[i][nt].y) > 0.f ? vv[2] : gs * vv[2];
                  vv[3] = bfhi(gv[i][nt].y) > 0.f ? vv[3] : gs * vv[3];
                }
                if constexpr (KEEP) {                           // (C_out0 a multiple of 8: e & 4 == 4 kq & 4)
                  const uint32_t bits = kb[i][nt] >> ((kq & 1) * 4);
#pragma unroll
                  for (int c = 0; c < 4; ++c) vv[c] = ((bits >> c) & 1u) ? 2.f * vv[c] : 0.f;
                }
                if constexpr (LEAKY) {                          // (plain v_max_f32: fmaxf() canonicalises both operands first)
                  const f32x2 lo = slope2 * f32x2{vv[0], vv[1]}, hi = slope2 * f32x2{vv[2], vv[3]};
                  asm("v_max_f32 %0, %1, %2" : "=v"(vv[0]) : "v"(vv[0]), "v"(lo[0]));
                  asm("v_max_f32 %0, %1, %2" : "=v"(vv[1]) : "v"(vv[1]), "v"(lo[1]));
                  asm("v_max_f32 %0, %1, %2" : "=v"(vv[2]) : "v"(vv[2]), "v"(hi[0]));
                  asm("v_max_f32 %0, %1, %2" : "=v"(vv[3]) : "v"(vv[3]), "v"(hi[1]));
                }
              }
              const u32x2 ov = u32x2{pack2(vv[0], vv[1]), pack2(vv[2], vv[3])};
              const int so = so_z + (rr[i] * p.oH + x0[i] * p.oW) * 2;
              if (!split || first[nt]) __builtin_amdgcn_raw_buffer_store_b64(ov, ors0, vo[nt], so, 0);     // (split: kernel-uniform)
              else __builtin_amdgcn_raw_buffer_store_b64(ov, ors1, vo[nt], so, 0);
            }
          }
        }
      }
    }
    // next step: the planes move one slot on
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      cur[s] += (uint32_t)(S * SB);
      if (cur[s] >= ring_end) cur[s] -= ring_len;
    }
  }
}

// ------------------------------------------------------------------------------------------ host
static bool fits31(int64_t v) { return v >= 0 && v < ((int64_t)1 << 30); }
static int64_t span(const tem_view &v) {
  return (int64_t)(v.N - 1) * v.sN + (int64_t)(v.D - 1) * v.sD + (int64_t)(v.H - 1) * v.sH + (int64_t)(v.W - 1) * v.sW + v.C;
}

template <int CI, int CO, int K, int S, int NW, bool SPLIT, int EPI>
static int run(Dev p, int N, hipStream_t st, bool dry, char *name, int name_len) {
  constexpr int NSTEP = (K * K * K * CI + 31) / 32, CPV = CI / 8, MAXJ = 40, NPG = SPLIT ? NW / ((CO + 15) / 16) : NW;
  // output column (TX x TY) and z segments: fewest rounds of (march length + ring prologue) x tile pairs per wave
  double best = 1e30;
  struct Plan { double best; int TX, TY, nbx, nby, zsegs, zper, RW, PV, ndma, slot_bytes, RD, ndma_cnt; };
  static tem_plan_cache<5, Plan> cache;                     // (the search below costs 5-30 us of host time per launch)
  const std::array<int, 5> key{N, p.OD, p.OH, p.OW, p.in1 != p.in0 ? 1 : 0};
  Plan memo;
  if (cache.get(key, memo)) {
    best = memo.best;
    p.TX = memo.TX; p.TY = memo.TY; p.nbx = memo.nbx; p.nby = memo.nby; p.zsegs = memo.zsegs; p.zper = memo.zper; p.RW = memo.RW;
    p.PV = memo.PV; p.ndma = memo.ndma; p.slot_bytes = memo.slot_bytes; p.RD = memo.RD; p.ndma_cnt = memo.ndma_cnt;
  } else {
  const int knob_ty = tem_env_int("TEM_C3B_TY", 0), knob_nbx = tem_env_int("TEM_C3B_NBX", 0), knob_zs = tem_env_int("TEM_C3B_ZSEGS", 0);
  const int knob_rd = tem_env_int("TEM_C3B_RD", 0), knob_cuw = tem_env_int("TEM_C3B_CUW", 300);   // CU-time term as in wino.hip plan(): bf16 step 3.89 -> 3.75 ms (1 x: 3.79, 10 x: 3.79)
  for (int nbx = 1; nbx <= 4; ++nbx) {
    if (knob_nbx && nbx != knob_nbx) continue;
    const int TX = (p.OW + nbx - 1) / nbx;
    if (nbx > 1 && TX < 16) break;
    // stride 1: a tile (16 voxels) never straddles ring rows; stride 2: the row holds the 2 TX + 2 input columns
    const int RW = S == 1 ? (TX + 2 + 15) & ~15 : (S * (TX - 1) + K + 1) & ~1;
    for (int TY = 2; TY <= 24 && TY <= p.OH + 1; ++TY) {
      if (knob_ty && TY != knob_ty) continue;
      const int PV = (S * (TY - 1) + K) * RW;
      const int slot = (int)((((int64_t)(PV + S * 32 + K + 1) * CI * 2) + 1023) & ~(int64_t)1023);     // + the last pair's over-read
      const int ndma = (PV * CPV + 63) / 64;
      if (ndma > MAXJ) continue;
      const int nby = (p.OH + TY - 1) / TY;
      const int cols = N * nbx * nby;
      const int pairs = ((S == 1 ? (TY * (RW / 16) + 1) / 2 : TY * ((TX + 31) / 32)) + NPG - 1) / NPG;
      // measured (tests/tools/c3b_sweep.py): ~3 us of launch + column start-up, a step costs its barrier plus, per tile pair,
      // the matrix chain and ~500 cycles of scalar / epilogue work; deeper rings (more planes in flight) bought nothing
      const double step = 1000.0 + pairs * (2.0 * NSTEP * 16.0 * ((CO + 15) / 16) + 500.0);
      for (int RD = K + S; RD <= 8; ++RD) {
        if (knob_rd ? RD != knob_rd : RD != K + S) continue;
        if (RD < K + S || RD * slot > LDS_MAX || (RD - K) * ndma * (p.in1 != p.in0 ? 2 : 1) > 62) continue;
        for (int zsegs = 1; zsegs <= p.OD; ++zsegs) {
          if (knob_zs && zsegs != knob_zs) continue;
          const int zper = (p.OD + zsegs - 1) / zsegs, zs = (p.OD + zper - 1) / zper;
          if (zs != zsegs) continue;
          const int64_t wgs = (int64_t)cols * zs;
          const double rounds = (double)((wgs + 255) / 256);
          const double cost = (rounds + knob_cuw / 100.0 * (double)wgs / 256.0) * ((zper + (K - S) / (double)S) * step + 7000.0);
          if (cost < best) {
            best = cost;
            p.TX = TX; p.TY = TY; p.nbx = nbx; p.nby = nby; p.zsegs = zs; p.zper = zper;
            p.RW = RW; p.PV = PV; p.ndma = ndma; p.slot_bytes = slot; p.RD = RD;
            p.ndma_cnt = p.in1 != p.in0 ? 2 * ndma : ndma;
          }
        }
      }
    }
  }
  memo = Plan{best, p.TX, p.TY, p.nbx, p.nby, p.zsegs, p.zper, p.RW, p.PV, p.ndma, p.slot_bytes, p.RD, p.ndma_cnt};
  cache.put(key, memo);
  }
  if (best >= 1e30) return TEM_EUNSUPPORTED;
  p.dbg = tem_env_int("TEM_C3B_DBG", 0);
  p.magicRW = (uint32_t)(((1u << 22) + p.RW - 1) / p.RW);
  {
    const uint32_t tpr = S == 1 ? p.RW / 16 : (p.TX + 31) / 32;      // (the kernel's TPR for a full column; edge columns have fewer pairs)
    p.magicT = (uint32_t)((((uint64_t)1 << 32) + tpr - 1) / tpr);
  }
  {                                                                                     // (v * magic) >> 22 exact, in 32 bits
    const int64_t vmax = (int64_t)(S * (p.TY - 1) + K) * p.RW + 64;
    if (vmax >= (1 << 22) / p.RW || vmax * p.magicRW >= ((int64_t)1 << 32) || p.magicRW >= (1u << 24)) return TEM_EUNSUPPORTED;
  }
  if (dry) {
    if (name) snprintf(name, name_len, "conv3_bf16_k<%d, %d, %d, %d, %d, %s, %d>", CI, CO, K, S, NW, SPLIT ? "true" : "false", EPI);
    return TEM_OK;
  }
  static int dbg = -1;
  if (dbg < 0) dbg = tem_env_int("TEM_DEBUG_FLAGS", 0);
  const int nblocks = N * p.nby * p.nbx * p.zsegs;
  const size_t lds = (size_t)p.RD * p.slot_bytes;
  if (dbg & 8)
    fprintf(stderr, "conv3_bf16<%d,%d> O=%dx%dx%d P=%d: TX=%d TY=%d RW=%d zsegs=%d zper=%d blocks=%d lds=%zu ndma=%d RD=%d\n", CI, CO, p.OD,
            p.OH, p.OW, p.P, p.TX, p.TY, p.RW, p.zsegs, p.zper, nblocks, lds, p.ndma, p.RD);
  auto kern = conv3_bf16_k<CI, CO, K, S, NW, SPLIT, EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3((NW + 1) * 64), lds, st, p);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

// TEM_EUNSUPPORTED: the caller falls back to conv_bf16_k
int dispatch(const tem_conv_args *a, hipStream_t st, bool dry, char *name, int name_len) {
  const tem_view &i0 = a->in0, &o0 = a->out0;
  const bool k3s1 = a->kd == 3 && a->kh == 3 && a->kw == 3 && a->sd == 1 && a->sh == 1 && a->sw == 1;
  const bool k4s2 = a->kd == 4 && a->kh == 4 && a->kw == 4 && a->sd == 2 && a->sh == 2 && a->sw == 2;
  if (!((k3s1 || k4s2) && a->pd == a->ph && a->ph == a->pw)) return TEM_EUNSUPPORTED;
  if (a->ep.dropout && !(a->ep.keep_mask && a->ep.keep_mode == 2)) return TEM_EUNSUPPORTED;     // reads a keep mask, draws none
  if (o0.N != i0.N) return TEM_ESHAPE;
  auto U = [](const float *q) { return reinterpret_cast<const u16 *>(q); };
  auto al16 = [](const tem_view &v) {
    return ((uintptr_t)v.ptr & 15) == 0 && v.C % 8 == 0 && v.sW % 8 == 0 && v.sH % 8 == 0 && v.sD % 8 == 0 && v.sN % 8 == 0;
  };
  auto al8 = [](const tem_view &v) {
    return ((uintptr_t)v.ptr & 7) == 0 && v.C % 4 == 0 && v.sW % 4 == 0 && v.sH % 4 == 0 && v.sD % 4 == 0 && v.sN % 4 == 0;
  };
  Dev p{};
  if (!al16(i0) || !fits31(span(i0))) return TEM_EUNSUPPORTED;
  int CI = i0.C;
  p.in0 = U(i0.ptr); p.in1 = p.in0;
  p.in0_bytes = (uint32_t)(span(i0) * 2); p.in1_bytes = p.in0_bytes;
  p.i0N = (int)i0.sN; p.i0D = (int)i0.sD; p.i0H = (int)i0.sH; p.i0W = (int)i0.sW; p.C0 = i0.C;
  p.i1N = p.i0N; p.i1D = p.i0D; p.i1H = p.i0H; p.i1W = p.i0W;
  if (a->in1.ptr) {
    const tem_view &i1 = a->in1;
    if (i1.N != i0.N || i1.D != i0.D || i1.H != i0.H || i1.W != i0.W) return TEM_ESHAPE;
    if (!al16(i1) || !fits31(span(i1))) return TEM_EUNSUPPORTED;
    p.in1 = U(i1.ptr); p.in1_bytes = (uint32_t)(span(i1) * 2);
    p.i1N = (int)i1.sN; p.i1D = (int)i1.sD; p.i1H = (int)i1.sH; p.i1W = (int)i1.sW;
    CI += i1.C;
  }
  p.D = i0.D; p.H = i0.H; p.W = i0.W; p.P = a->pd;
  p.w = U(a->w); p.flip = a->w_layout == TEM_W_FLIP_CO_CI;
  if (!al8(o0) || !fits31(span(o0))) return TEM_EUNSUPPORTED;
  p.oN = (int)o0.sN; p.oD = (int)o0.sD; p.oH = (int)o0.sH; p.oW = (int)o0.sW;
  p.CO0 = o0.C;
  int CO = o0.C;
  p.out0 = const_cast<u16 *>(U(o0.ptr)); p.out1 = p.out0;
  p.out0_bytes = (uint32_t)(span(o0) * 2); p.out1_bytes = p.out0_bytes;
  if (a->out1.ptr) {
    const tem_view &o1 = a->out1;
    if (o1.N != o0.N || o1.D != o0.D || o1.H != o0.H || o1.W != o0.W) return TEM_ESHAPE;
    if (!al8(o1) || !fits31(span(o1))) return TEM_EUNSUPPORTED;
    if (o1.sN != o0.sN || o1.sD != o0.sD || o1.sH != o0.sH || o1.sW != o0.sW) return TEM_EUNSUPPORTED;    // one scalar offset for both
    p.out1 = const_cast<u16 *>(U(o1.ptr)); p.out1_bytes = (uint32_t)(span(o1) * 2);
    CO += o1.C;
  }
  p.OD = o0.D; p.OH = o0.H; p.OW = o0.W;
  {
    const int K = k3s1 ? 3 : 4, S = k3s1 ? 1 : 2;
    if (p.OD != (p.D + 2 * p.P - K) / S + 1 || p.OH != (p.H + 2 * p.P - K) / S + 1 || p.OW != (p.W + 2 * p.P - K) / S + 1) return TEM_ESHAPE;
  }
  const tem_epilogue &e = a->ep;
  p.bias = e.bias; p.slope = e.slope; p.gate_slope = e.gate_slope;
  if (e.bias && o0.C % 4) return TEM_EUNSUPPORTED;
  if (e.gate.ptr) {
    const tem_view &g = e.gate;
    if (g.N != o0.N || g.D != o0.D || g.H != o0.H || g.W != o0.W || g.C < o0.C) return TEM_ESHAPE;
    if (!al8(g) || !fits31(span(g))) return TEM_EUNSUPPORTED;
    p.gate = U(g.ptr); p.gN = (int)g.sN; p.gD = (int)g.sD; p.gH = (int)g.sH; p.gW = (int)g.sW; p.gbytes = (int)(span(g) * 2);
  }
  if (e.add.ptr) {
    const tem_view &ad = e.add;
    if (ad.C < o0.C || ad.N != o0.N) return TEM_ESHAPE;
    if (!al8(ad) || !fits31(span(ad))) return TEM_EUNSUPPORTED;
    p.add = U(ad.ptr); p.aN = (int)ad.sN; p.aD = (int)ad.sD; p.aH = (int)ad.sH; p.aW = (int)ad.sW;
    p.aoz = e.add_off[0]; p.aoy = e.add_off[1]; p.aox = e.add_off[2];
    p.aDd = ad.D; p.aHh = ad.H; p.aWw = ad.W; p.abytes = (int)(span(ad) * 2);
  }
  if (e.dropout) {
    if (o0.C % 8) return TEM_EUNSUPPORTED;
    p.keep = e.keep_mask;
    p.doz = e.drop_org[0]; p.doy = e.drop_org[1]; p.dox = e.drop_org[2];
    p.dD = e.drop_dims[0] ? e.drop_dims[0] : o0.D; p.dH = e.drop_dims[0] ? e.drop_dims[1] : o0.H;
    p.dW = e.drop_dims[0] ? e.drop_dims[2] : o0.W;
    const int64_t vox = (int64_t)o0.N * p.dD * p.dH * p.dW;
    if (vox >= (1 << 24) || vox * o0.C >= ((int64_t)1 << 31)) return TEM_EUNSUPPORTED;     // 24-bit multiplies, 31-bit element index
    p.mbytes = (int)((vox * o0.C + 7) / 8);
  }
  const int N = i0.N;
  const bool lmax = p.slope > 0.f && p.slope < 1.f;
  if (p.slope != 1.f && !(lmax && !p.gate && !p.add && !p.keep)) return TEM_EUNSUPPORTED;    // LeakyReLU only without gate / add / mask
  const int epi = (p.gate ? 1 : 0) | (p.add ? 2 : 0) | (p.keep ? 4 : 0) | (lmax ? 8 : 0);
#define C3E(ci, co, k, s, nw, sp, ep) \
  if (CI == ci && CO == co && (k == 3) == k3s1 && epi == ep) return run<ci, co, k, s, nw, sp, ep>(p, N, st, dry, name, name_len);
#define C3(ci, co, k, s, nw, sp) C3E(ci, co, k, s, nw, sp, 8) C3E(ci, co, k, s, nw, sp, 0) C3E(ci, co, k, s, nw, sp, 1) \
                                 C3E(ci, co, k, s, nw, sp, 3) C3E(ci, co, k, s, nw, sp, 5) C3E(ci, co, k, s, nw, sp, 7)
  C3(8, 8, 3, 1, 8, false) C3(8, 16, 3, 1, 8, false) C3(16, 8, 3, 1, 8, false) C3(16, 16, 3, 1, 8, false) C3(16, 32, 3, 1, 4, false)
  C3(32, 16, 3, 1, 4, false) C3(32, 32, 3, 1, 4, true)
  // 4x4x4 stride 2 (Downsample blocks; input-gradients of the transposed convolutions): 16 / 32 k-steps of kernel fragments in
  // registers; 32 input channels (64 k-steps) stay on conv_bf16_k
  C3(8, 8, 4, 2, 8, false) C3(8, 16, 4, 2, 8, false) C3(16, 16, 4, 2, 4, false) C3(16, 32, 4, 2, 4, true)
#undef C3E
#undef C3
  return TEM_EUNSUPPORTED;
}

}  // namespace conv3_bf16
