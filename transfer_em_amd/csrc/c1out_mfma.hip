// c1out_mfma.hip -- the 3x3x3 stride-1 convolutions with ONE output channel (generator.py:110 last conv; the
// input-gradients of the first convs, generator.py:54 / discriminator.py:39-40) with the channel sum on the matrix cores.
//
//   out[o] = sum_{t = (dz,dy,dx)} P[o + t][t],      P[v][t] = sum_ci X[v][ci] * w[t][ci]
//
// The layer is HBM-bound (12 FLOP per byte of input), but as a VALU stencil every output voxel costs 27 C_in FMAs AND
// 27 C_in / 4 LDS reads of 16 bytes -- the LDS pipe, not HBM, sets the pace (c1_stencil_k: 1.4-1.9 TB/s).  Here the sum
// over the input channels is a [voxels x C_in] x [C_in x 27] product on v_mfma_f32_16x16x4_f32 (the 27 taps are the
// columns: two 16-wide n-tiles, 27 of 32 used), whose A fragments come STRAIGHT from HBM/L2: lane (voxel m, k group kq)
// loads the channels 4 kq .. 4 kq + 3 of its voxel, 16 voxels = 1 KB contiguous per wave
// instruction, no LDS image of the input at all.  What remains per output voxel is the 27-term shifted sum over P,
// which goes through LDS as single floats (27 b32 reads + 27 adds per output instead of 27 C_in FMAs).
//
// A workgroup owns a 16 x 16 patch of output columns and marches along z: per input plane it computes P for the
// 18 x 18 halo patch (21 tiles over its 4 waves), and each thread adds the plane's three dz-slices into the three
// rotating accumulators of its output column (as c1_stencil_k does); the plane two steps back is complete and leaves
// through the fused epilogue (bias, LeakyReLU-gradient gate, LeakyReLU).  The planes are software-pipelined: P is
// double-buffered, the matrix cores work on plane j + 1 while the wave gathers plane j, and plane j + 2's fragments are
// in flight (one barrier per plane).
#include "tem_common.h"
#include <cstdio>
#include <cstdlib>

namespace c1out {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2f __attribute__((ext_vector_type(2)));

struct Dev {
  const float *in;
  int32_t iN, iD, iH, iW, D, H, W, in_bytes;
  const float *w;
  float *out;
  int32_t oN, oD, oH, oW, OD, OH, OW, out_bytes;
  int32_t P;
  int32_t ntx, nty, zsegs, zper;
  float slope;
  const float *gate; int32_t gN, gD, gH, gW, gate_bytes; float gate_slope;
  const float *bias;
};

constexpr int TX = 16, TY = 16, COLS = TX + 2, ROWS = TY + 2, PV = ROWS * COLS;      // output patch, halo patch
constexpr int NTILE = (PV + 15) / 16, NTW = (NTILE + 3) / 4;                        // 16-voxel tiles per plane, per wave
// P in LDS is TAP-major: P[tap][voxel of the halo patch].  A lane of the MFMA result holds 4 consecutive voxels of one tap = ONE
// 16-byte store per n-tile (voxel-major rows of 29 floats took 8 four-byte stores per tile: 48 of the ~75 LDS instructions per plane
// and wave, and the LDS pipe -- not HBM, and in bf16 not the matrix cores -- sets this kernel's pace); the gather reads P[tap][v + shift]
// with consecutive lanes on consecutive voxels.  Row pitch = 4 (mod 32) floats: the 16 tap-lanes of a store hit 16 distinct bank quads.
constexpr int PVP = 356;                                                             // floats per tap row (>= NTILE * 16)
constexpr int PBUF = 28 * PVP;                                                       // floats of one P buffer (27 taps + a spare row)
static_assert(PVP >= NTILE * 16 && PVP % 32 == 4, "tap row pitch");
constexpr int OOB = (int)0x80000000;

template <int CI, bool FLIP, bool GATE>
__global__ __launch_bounds__(256, 2) void c1out_mfma_k(Dev p) {
  static_assert(CI == 16 || CI == 8, "one 16- or 8-byte load per lane and voxel");
  constexpr int KS = CI / 4;                          // k-steps; lane group kq owns channels KS kq .. KS kq + KS - 1
  extern __shared__ __attribute__((aligned(16))) float P_[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;

  int b = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);     // x-neighbours (shared halos) meet in one L2
  const int zseg = b % p.zsegs; b /= p.zsegs;
  const int txi = b % p.ntx; b /= p.ntx;
  const int tyi = b % p.nty;
  const int n = b / p.nty;
  const int ox0 = txi * TX, oy0 = tyi * TY;
  const int oz0 = zseg * p.zper, oz1 = min(p.OD, oz0 + p.zper);
  const int nplanes = oz1 - oz0 + 2;
  const int iz0 = oz0 - p.P;

  // ---- B fragments: column t = 16 nt + m (tap), k-step s multiplies channel KS kq + s
  float B[KS][2];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int t = 16 * nt + m;
      B[s][nt] = t < 27 ? p.w[(FLIP ? 26 - t : t) * CI + KS * kq + s] : 0.f;
    }

  // ---- this lane's A voxels: tile wave + 4 i, voxel 16 tile + m of the halo patch (row r, column c).  Nothing per plane is decided
  // per lane (round 3: with two workgroups per CU the step is bound by instruction issue beside the matrix pipe): a lane without a
  // voxel / an output carries an out-of-range offset for good -- the plane term added to it stays out of range, and a plane outside
  // the volume is a plane term of 2^30 >= any tensor here --, gate and output go through buffer descriptors, the taps 27..31 of the
  // second n-tile land in one spare row of P.
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)p.gate, 0, GATE ? p.gate_bytes : 0, 0x00020000);
  int goff[NTW];                                     // byte offset inside plane iz = 0 (or OOB: outside the image / the patch)
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int v = (wave + 4 * i) * 16 + m;
    const int r = v / COLS, c = v - r * COLS;
    const int iy = oy0 - p.P + r, ix = ox0 - p.P + c;
    const bool ok = v < PV && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    goff[i] = ok ? (n * p.iN + iy * p.iH + ix * p.iW + KS * kq) * 4 : OOB;
  }
  auto load_plane = [&](float (&a)[NTW][KS], int iz) {
    const int zo = (unsigned)iz < (unsigned)p.D ? iz * p.iD * 4 : 0x40000000;      // (block-uniform)
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      if constexpr (KS == 4) {
        const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(xrs, goff[i] + zo, 0, 0);
        a[i][0] = __uint_as_float(q.x); a[i][1] = __uint_as_float(q.y); a[i][2] = __uint_as_float(q.z); a[i][3] = __uint_as_float(q.w);
      } else {
        const u32x2f q = __builtin_amdgcn_raw_buffer_load_b64(xrs, goff[i] + zo, 0, 0);
        a[i][0] = __uint_as_float(q.x); a[i][1] = __uint_as_float(q.y);
      }
    }
  };

  // ---- this thread's output column
  const int ty = tid >> 4, tx = tid & 15;
  const int ox = ox0 + tx, oy = oy0 + ty;
  const bool owner = ox < p.OW && oy < p.OH;
  const float *pbase = P_ + (ty * COLS + tx);
  const int ooff = owner ? (n * p.oN + oy * p.oH + ox * p.oW) * 4 : OOB;
  const int gbase = owner ? (n * p.gN + oy * p.gH + ox * p.gW) * 4 : OOB;
  const float bias = p.bias ? p.bias[0] : 0.f;
  const int row1 = (m < 11 ? 16 + m : 27) * PVP;             // second n-tile: taps 16..26, the rest into the spare row

  float acc[3] = {0.f, 0.f, 0.f};
  // P of a plane from loaded fragments into buffer `buf`: per tile KS k-steps x 2 n-tiles
  auto p_phase = [&](const float (&a)[NTW][KS], int buf) {
    float *const Pb = P_ + buf * PBUF + 4 * kq;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      if (i < NTW - 1 || wave + 4 * i < NTILE) {             // (only the last round has waves without a tile)
        f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], B[s][0], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], B[s][1], c1, 0, 0, 0);
        }
        float *d = Pb + (wave + 4 * i) * 16;                 // voxels 4 kq .. 4 kq + 3 of the tile (stored under the next tile's MFMAs)
        *reinterpret_cast<f32x4 *>(d + m * PVP) = c0;
        *reinterpret_cast<f32x4 *>(d + row1) = c1;
      }
    }
  };
  // Software pipeline over the planes (P double-buffered, ONE barrier per plane): while the matrix cores work on plane
  // j + 1, the same wave gathers plane j's shifted sums out of the other buffer, and planes j + 2, j + 3 are in flight
  // (three fragment sets).
  float a0[NTW][KS], a1[NTW][KS], a2[NTW][KS];
  load_plane(a0, iz0);
  load_plane(a1, nplanes > 1 ? iz0 + 1 : -1);
  load_plane(a2, nplanes > 2 ? iz0 + 2 : -1);
  p_phase(a0, 0);
  auto step = [&](float (&anow)[NTW][KS], float (&aload)[NTW][KS], int j, int r3) {
    // anow = fragments of plane j + 1 (loaded), aload = set to refill with plane j + 3
    __syncthreads();                                         // P(j) is complete; nobody still reads the buffer P(j + 1) goes to
    const int ozf = oz0 + j - 2;                             // the output plane this step completes
    float gv = 1.f;
    if (GATE && j >= 2) gv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(grs, gbase, ozf * p.gD * 4, 0));
    if (j + 1 < nplanes) p_phase(anow, (j + 1) & 1);         // block-uniform
    load_plane(aload, j + 3 < nplanes ? iz0 + j + 3 : -1);   // (past the run: out of range, moves no data)
    // ---- shifted sum: plane j feeds output planes j (dz 0), j - 1 (dz 1), j - 2 (dz 2): slot (j - dz) mod 3
    const float *pb = pbase + (j & 1) * PBUF;
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      float s = 0.f;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) s += pb[((dz * 3 + dy) * 3 + dx) * PVP + dy * COLS + dx];
      acc[(r3 + 3 - dz) % 3] += s;
    }
    if (j >= 2) {                                            // output plane j - 2 is complete (block-uniform)
      float v = acc[(r3 + 1) % 3] + bias;
      if (GATE) v = gv > 0.f ? v : p.gate_slope * v;
      v = v > 0.f ? v : p.slope * v;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ors, ooff, ozf * p.oD * 4, 0);
    }
    acc[(r3 + 1) % 3] = 0.f;
  };
  // (unrolled by 6: accumulator slot and fragment set = plane mod 3, P buffer = plane mod 2 are compile-time)
  for (int j0 = 0; j0 < nplanes; j0 += 6) {
#pragma unroll
    for (int r6 = 0; r6 < 6; ++r6) {
      const int j = j0 + r6;
      if (j < nplanes) {                                     // block-uniform
        if (r6 % 3 == 0) step(a1, a0, j, 0);
        else if (r6 % 3 == 1) step(a2, a1, j, 1);
        else step(a0, a2, j, 2);
      }
    }
  }
}

// ---- bf16 mode (conv_bf16.hip tries this for the one-output-channel 3x3x3 layers): the same march with bf16 fragments.
// One v_mfma_f32_16x16x16_bf16 spans all 16 input channels (8: the lane groups 2 and 3 load nothing and multiply zeros), so a
// 16-voxel tile costs one 8-byte load per lane (512 contiguous bytes per wave instruction) and two MFMAs instead of eight:
// in fp32 the kernel is bound by 48 MFMAs of 32 cycles per plane and wave, here by the shifted sum and HBM.  P, the three
// accumulators and the epilogue stay fp32; input, gate and output are bf16, the kernel is the packed bf16 copy [tap][ci].
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

struct DevH {
  const u16 *in;
  int32_t iN, iD, iH, iW, D, H, W, in_bytes;
  const u16 *w;
  u16 *out;
  int32_t oN, oD, oH, oW, OD, OH, OW, out_bytes;
  int32_t P;
  int32_t ntx, nty, zsegs, zper;
  float slope;
  const u16 *gate; int32_t gN, gD, gH, gW, gate_bytes; float gate_slope;
  const float *bias;
};

template <int CI, bool FLIP, bool GATE>
__global__ __launch_bounds__(256, 2) void c1out_h_k(DevH p) {
  static_assert(CI == 16 || CI == 8, "one 8-byte load per lane and voxel");
  extern __shared__ __attribute__((aligned(16))) float P_[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;

  int b = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int zseg = b % p.zsegs; b /= p.zsegs;
  const int txi = b % p.ntx; b /= p.ntx;
  const int tyi = b % p.nty;
  const int n = b / p.nty;
  const int ox0 = txi * TX, oy0 = tyi * TY;
  const int oz0 = zseg * p.zper, oz1 = min(p.OD, oz0 + p.zper);
  const int nplanes = oz1 - oz0 + 2;
  const int iz0 = oz0 - p.P;

  // ---- B fragments: column t = 16 nt + m (tap), the lane's k = channels 4 kq .. 4 kq + 3
  s16x4 B[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int t = 16 * nt + m;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      B[nt][i] = (t < 27 && 4 * kq + i < CI) ? (short)p.w[(FLIP ? 26 - t : t) * CI + 4 * kq + i] : (short)0;
  }

  // The step is bound by instruction issue (two workgroups per CU: ~300 instructions per plane and wave were 1.1 us per plane),
  // so nothing per plane is decided per lane: a lane without a voxel / output carries an out-of-range offset for good (the plane
  // term added to it stays out of range; a plane outside the volume is a plane term of 2^30 >= any tensor here), gate and output go
  // through buffer descriptors, the taps 27..31 of the second n-tile land in one spare row of P.
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)p.gate, 0, GATE ? p.gate_bytes : 0, 0x00020000);
  int goff[NTW];                                     // byte offset inside plane iz = 0 (or OOB: outside the image / the patch)
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int v = (wave + 4 * i) * 16 + m;
    const int r = v / COLS, c = v - r * COLS;
    const int iy = oy0 - p.P + r, ix = ox0 - p.P + c;
    const bool ok = v < PV && 4 * kq < CI && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    goff[i] = ok ? (n * p.iN + iy * p.iH + ix * p.iW + 4 * kq) * 2 : OOB;
  }
  auto load_plane = [&](s16x4 (&a)[NTW], int iz) {
    const int zo = (unsigned)iz < (unsigned)p.D ? iz * p.iD * 2 : 0x40000000;      // (block-uniform)
#pragma unroll
    for (int i = 0; i < NTW; ++i)
      a[i] = __builtin_bit_cast(s16x4, __builtin_amdgcn_raw_buffer_load_b64(xrs, goff[i] + zo, 0, 0));
  };

  const int ty = tid >> 4, tx = tid & 15;
  const int ox = ox0 + tx, oy = oy0 + ty;
  const bool owner = ox < p.OW && oy < p.OH;
  const float *pbase = P_ + (ty * COLS + tx);
  const int ooff = owner ? (n * p.oN + oy * p.oH + ox * p.oW) * 2 : OOB;
  const int gbase = owner ? (n * p.gN + oy * p.gH + ox * p.gW) * 2 : OOB;
  const float bias = p.bias ? p.bias[0] : 0.f;
  const int row1 = (m < 11 ? 16 + m : 27) * PVP;             // second n-tile: taps 16..26, the rest into the spare row

  float acc[3] = {0.f, 0.f, 0.f};
  auto p_phase = [&](const s16x4 (&a)[NTW], int buf) {
    float *const Pb = P_ + buf * PBUF + 4 * kq;
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 c0[NTW], c1[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {                          // (all products first: the stores then find them finished)
      c0[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[i], B[0], z, 0, 0, 0);
      c1[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[i], B[1], z, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      if (i < NTW - 1 || wave + 4 * i < NTILE) {             // (only the last round has waves without a tile)
        float *d = Pb + (wave + 4 * i) * 16;                 // voxels 4 kq .. 4 kq + 3 of the tile
        *reinterpret_cast<f32x4 *>(d + m * PVP) = c0[i];
        *reinterpret_cast<f32x4 *>(d + row1) = c1[i];
      }
    }
  };
  // (three fragment sets: plane j + 3 is requested while plane j + 1 goes through the matrix cores)
  s16x4 a0[NTW], a1[NTW], a2[NTW];
  load_plane(a0, iz0);
  load_plane(a1, nplanes > 1 ? iz0 + 1 : -1);
  load_plane(a2, nplanes > 2 ? iz0 + 2 : -1);
  p_phase(a0, 0);
  auto step = [&](s16x4 (&anow)[NTW], s16x4 (&aload)[NTW], int j, int r3) {
    __syncthreads();                                         // P(j) is complete; nobody still reads the buffer P(j + 1) goes to
    const int ozf = oz0 + j - 2;                             // the output plane this step completes
    short gq = 0;
    if (GATE && j >= 2) gq = __builtin_amdgcn_raw_buffer_load_b16(grs, gbase, ozf * p.gD * 2, 0);
    if (j + 1 < nplanes) p_phase(anow, (j + 1) & 1);         // block-uniform
    load_plane(aload, j + 3 < nplanes ? iz0 + j + 3 : -1);   // (past the run: out of range, moves no data)
    const float *pb = pbase + (j & 1) * PBUF;
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      float s = 0.f;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) s += pb[((dz * 3 + dy) * 3 + dx) * PVP + dy * COLS + dx];
      acc[(r3 + 3 - dz) % 3] += s;
    }
    if (j >= 2) {                                            // output plane j - 2 is complete (block-uniform)
      float v = acc[(r3 + 1) % 3] + bias;
      if (GATE) v = __uint_as_float((uint32_t)(unsigned short)gq << 16) > 0.f ? v : p.gate_slope * v;
      v = v > 0.f ? v : p.slope * v;
      __builtin_amdgcn_raw_buffer_store_b16((short)__builtin_bit_cast(u16, (__bf16)v), ors, ooff, ozf * p.oD * 2, 0);   // RNE
    }
    acc[(r3 + 1) % 3] = 0.f;
  };
  // (unrolled by 6: accumulator slot and fragment set = plane mod 3, P buffer = plane mod 2 are compile-time; step j multiplies
  // the set of plane j + 1 and refills the set of plane j, consumed one step earlier)
  for (int j0 = 0; j0 < nplanes; j0 += 6) {
#pragma unroll
    for (int r6 = 0; r6 < 6; ++r6) {
      const int j = j0 + r6;
      if (j < nplanes) {                                     // block-uniform
        if (r6 % 3 == 0) step(a1, a0, j, 0);
        else if (r6 % 3 == 1) step(a2, a1, j, 1);
        else step(a0, a2, j, 2);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ host
static bool fits32(const tem_view &v) {
  int64_t span = (int64_t)(v.N - 1) * v.sN + (int64_t)(v.D - 1) * v.sD + (int64_t)(v.H - 1) * v.sH +
                 (int64_t)(v.W - 1) * v.sW + v.C;
  return span < ((int64_t)1 << 29);                  // (byte offsets of the buffer loads stay below 2^31)
}

static thread_local char *g_name = nullptr;
static thread_local int g_name_len = 0;

template <int CI, bool FLIP, bool GATE>
static int run1(Dev p, int N, hipStream_t st, bool dry) {
  if (dry) {
    if (g_name) snprintf(g_name, g_name_len, "c1out_mfma_k<%d, %s, %s>", CI, FLIP ? "true" : "false", GATE ? "true" : "false");
    return TEM_OK;
  }
  p.ntx = (p.OW + TX - 1) / TX; p.nty = (p.OH + TY - 1) / TY;
  // z-run: ~2 workgroups per CU, but >= 8 output planes per run where the volume allows (an input plane is processed
  // (zper + 2) / zper times)
  const int tiles = p.ntx * p.nty * N;
  int zsegs = (512 + tiles - 1) / tiles;
  if (zsegs < 1) zsegs = 1;
  int zper = (p.OD + zsegs - 1) / zsegs;
  if (zper < 8) zper = p.OD < 8 ? p.OD : 8;
  static int zp = -1;
  if (zp < 0) zp = tem_env_int("TEM_C1OUT_ZPER", 0);
  if (zp > 0) zper = zp < p.OD ? zp : p.OD;
  p.zper = zper;
  p.zsegs = (p.OD + zper - 1) / zper;
  const size_t lds_bytes = (size_t)2 * PBUF * 4;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void *)c1out_mfma_k<CI, FLIP, GATE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  const int nblocks = p.zsegs * p.ntx * p.nty * N;
  hipLaunchKernelGGL((c1out_mfma_k<CI, FLIP, GATE>), dim3((unsigned)nblocks), dim3(256), lds_bytes, st, p);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

template <int CI, bool FLIP>
static int run(Dev p, int N, hipStream_t st, bool dry) {
  return p.gate ? run1<CI, FLIP, true>(p, N, st, dry) : run1<CI, FLIP, false>(p, N, st, dry);
}

static int dispatch(const tem_conv_args *a, hipStream_t st, bool dry) {
  const tem_view &i0 = a->in0, &o0 = a->out0;
  if (a->in1.ptr || a->out1.ptr || o0.C != 1) return TEM_EUNSUPPORTED;
  if (a->kd != 3 || a->kh != 3 || a->kw != 3 || a->sd != 1 || a->sh != 1 || a->sw != 1) return TEM_EUNSUPPORTED;
  if (a->pd != a->ph || a->ph != a->pw) return TEM_EUNSUPPORTED;
  if (a->ep.dropout || a->ep.add.ptr) return TEM_EUNSUPPORTED;
  // (8 -> 1, the input-gradients of the first convolutions: slower than c1_stencil_k until the kernel's instruction diet -- 51 vs 38 us
  // in round 2, 22.5 vs 20.2 before it, 18.7 vs 20.2-22 after; TEM_C1OUT_8=0 in knob builds selects the stencil)
  if (i0.C != 16 && !(i0.C == 8 && tem_env_int("TEM_C1OUT_8", 1))) return TEM_EUNSUPPORTED;
  static int enabled = -1;
  if (enabled < 0) enabled = tem_env_int("TEM_C1OUT_MFMA", 1);
  if (!enabled) return TEM_EUNSUPPORTED;
  if (o0.N != i0.N) return TEM_ESHAPE;
  if (!fits32(i0) || !fits32(o0)) return TEM_EUNSUPPORTED;
  if (((uintptr_t)i0.ptr & 15) || i0.sW % 4 || i0.sH % 4 || i0.sD % 4 || i0.sN % 4) return TEM_EUNSUPPORTED;
  Dev p{};
  p.in = i0.ptr; p.iN = (int)i0.sN; p.iD = (int)i0.sD; p.iH = (int)i0.sH; p.iW = (int)i0.sW;
  p.D = i0.D; p.H = i0.H; p.W = i0.W;
  p.in_bytes = (int)(((int64_t)(i0.N - 1) * i0.sN + (int64_t)(i0.D - 1) * i0.sD + (int64_t)(i0.H - 1) * i0.sH +
                      (int64_t)(i0.W - 1) * i0.sW + i0.C) * 4);
  p.w = a->w;
  p.out = o0.ptr; p.oN = (int)o0.sN; p.oD = (int)o0.sD; p.oH = (int)o0.sH; p.oW = (int)o0.sW;
  p.OD = o0.D; p.OH = o0.H; p.OW = o0.W;
  auto bytes_of = [](const tem_view &v) {
    return (int)(((int64_t)(v.N - 1) * v.sN + (int64_t)(v.D - 1) * v.sD + (int64_t)(v.H - 1) * v.sH + (int64_t)(v.W - 1) * v.sW + v.C) * 4);
  };
  p.out_bytes = bytes_of(o0);
  p.P = a->pd;
  p.slope = a->ep.slope; p.gate_slope = a->ep.gate_slope; p.bias = a->ep.bias;
  if (a->ep.gate.ptr) {
    const tem_view &g = a->ep.gate;
    if (g.N != o0.N || g.D != o0.D || g.H != o0.H || g.W != o0.W || g.C < o0.C) return TEM_ESHAPE;
    if (!fits32(g)) return TEM_EUNSUPPORTED;
    p.gate = g.ptr; p.gN = (int)g.sN; p.gD = (int)g.sD; p.gH = (int)g.sH; p.gW = (int)g.sW;
    p.gate_bytes = bytes_of(g);
  }
  const bool flip = a->w_layout == TEM_W_FLIP_CO_CI;
  if (a->w_layout != TEM_W_TAP_CI_CO && !flip) return TEM_EUNSUPPORTED;
  const int N = i0.N;
  if (i0.C == 8) return flip ? run<8, true>(p, N, st, dry) : run<8, false>(p, N, st, dry);   // input-gradients of the first convolutions
  return flip ? run<16, true>(p, N, st, dry) : run<16, false>(p, N, st, dry);     // g.f2 forward
}


template <int CI, bool FLIP, bool GATE>
static int run_h1(DevH p, int N, hipStream_t st, bool dry, char *name, int name_len) {
  if (name) snprintf(name, name_len, "c1out_h_k<%d, %s, %s>", CI, FLIP ? "true" : "false", GATE ? "true" : "false");
  if (dry) return TEM_OK;
  p.ntx = (p.OW + TX - 1) / TX; p.nty = (p.OH + TY - 1) / TY;
  const int tiles = p.ntx * p.nty * N;
  int zsegs = (512 + tiles - 1) / tiles;
  if (zsegs < 1) zsegs = 1;
  int zper = (p.OD + zsegs - 1) / zsegs;
  if (zper < 8) zper = p.OD < 8 ? p.OD : 8;
  p.zper = zper;
  p.zsegs = (p.OD + zper - 1) / zper;
  const size_t lds_bytes = (size_t)2 * PBUF * 4;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void *)c1out_h_k<CI, FLIP, GATE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  const int nblocks = p.zsegs * p.ntx * p.nty * N;
  hipLaunchKernelGGL((c1out_h_k<CI, FLIP, GATE>), dim3((unsigned)nblocks), dim3(256), lds_bytes, st, p);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

template <int CI, bool FLIP>
static int run_h(DevH p, int N, hipStream_t st, bool dry, char *name, int name_len) {
  return p.gate ? run_h1<CI, FLIP, true>(p, N, st, dry, name, name_len) : run_h1<CI, FLIP, false>(p, N, st, dry, name, name_len);
}

// bf16 tensors behind the float* fields of tem_conv_args (strides in elements), `w` = the packed bf16 kernel [tap][ci]
static int dispatch_h(const tem_conv_args *a, hipStream_t st, bool dry, char *name, int name_len) {
  const tem_view &i0 = a->in0, &o0 = a->out0;
  if (a->in1.ptr || a->out1.ptr || o0.C != 1 || (i0.C != 16 && i0.C != 8)) return TEM_EUNSUPPORTED;
  if (a->kd != 3 || a->kh != 3 || a->kw != 3 || a->sd != 1 || a->sh != 1 || a->sw != 1) return TEM_EUNSUPPORTED;
  if (a->pd != a->ph || a->ph != a->pw) return TEM_EUNSUPPORTED;
  if (a->ep.dropout || a->ep.add.ptr) return TEM_EUNSUPPORTED;
  if (o0.N != i0.N) return TEM_ESHAPE;
  if (!fits32(i0) || !fits32(o0)) return TEM_EUNSUPPORTED;
  if (((uintptr_t)i0.ptr & 7) || i0.sW % 4 || i0.sH % 4 || i0.sD % 4 || i0.sN % 4) return TEM_EUNSUPPORTED;
  auto U = [](const float *q) { return reinterpret_cast<const u16 *>(q); };
  DevH p{};
  p.in = U(i0.ptr); p.iN = (int)i0.sN; p.iD = (int)i0.sD; p.iH = (int)i0.sH; p.iW = (int)i0.sW;
  p.D = i0.D; p.H = i0.H; p.W = i0.W;
  p.in_bytes = (int)(((int64_t)(i0.N - 1) * i0.sN + (int64_t)(i0.D - 1) * i0.sD + (int64_t)(i0.H - 1) * i0.sH +
                      (int64_t)(i0.W - 1) * i0.sW + i0.C) * 2);
  p.w = U(a->w);
  p.out = const_cast<u16 *>(U(o0.ptr)); p.oN = (int)o0.sN; p.oD = (int)o0.sD; p.oH = (int)o0.sH; p.oW = (int)o0.sW;
  p.OD = o0.D; p.OH = o0.H; p.OW = o0.W;
  auto bytes_of = [](const tem_view &v) {
    return (int)(((int64_t)(v.N - 1) * v.sN + (int64_t)(v.D - 1) * v.sD + (int64_t)(v.H - 1) * v.sH + (int64_t)(v.W - 1) * v.sW + v.C) * 2);
  };
  p.out_bytes = bytes_of(o0);
  p.P = a->pd;
  p.slope = a->ep.slope; p.gate_slope = a->ep.gate_slope; p.bias = a->ep.bias;
  if (a->ep.gate.ptr) {
    const tem_view &g = a->ep.gate;
    if (g.N != o0.N || g.D != o0.D || g.H != o0.H || g.W != o0.W || g.C < o0.C) return TEM_ESHAPE;
    if (!fits32(g)) return TEM_EUNSUPPORTED;
    p.gate = U(g.ptr); p.gN = (int)g.sN; p.gD = (int)g.sD; p.gH = (int)g.sH; p.gW = (int)g.sW;
    p.gate_bytes = bytes_of(g);
  }
  const bool flip = a->w_layout == TEM_W_FLIP_CO_CI;
  if (a->w_layout != TEM_W_TAP_CI_CO && !flip) return TEM_EUNSUPPORTED;
  const int N = i0.N;
  if (i0.C == 16) return flip ? run_h<16, true>(p, N, st, dry, name, name_len) : run_h<16, false>(p, N, st, dry, name, name_len);
  return flip ? run_h<8, true>(p, N, st, dry, name, name_len) : run_h<8, false>(p, N, st, dry, name, name_len);
}

}  // namespace c1out

// Called by tem_conv (dispatch.hip) ahead of the VALU stencil.
int tem_conv_c1out_try(const tem_conv_args *a, hipStream_t st, bool dry) { return c1out::dispatch(a, st, dry); }

int tem_conv_c1out_describe(const tem_conv_args *a, char *buf, int len) {
  c1out::g_name = buf; c1out::g_name_len = len;
  int rc = c1out::dispatch(a, nullptr, true);
  c1out::g_name = nullptr;
  return rc;
}

// bf16 mode (conv_bf16.hip tries this for the one-output-channel 3x3x3 layers)
int tem_conv_c1out_bf16_try(const tem_conv_args *a, hipStream_t st, bool dry, char *name, int name_len) {
  return c1out::dispatch_h(a, st, dry, name, name_len);
}
