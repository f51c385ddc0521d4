// stencil_c1.hip -- the HBM-bound edge layers of both networks: 3x3x3 stride-1 convolutions with ONE input
// channel (generator.py:54 first conv, discriminator.py:39-40; input-gradient of generator.py:110) or ONE output
// channel (generator.py:110 last conv; input-gradients of the first convs).  Arithmetic intensity ~12 FLOP/B:
// these are the only fp32 layers bound by HBM, not by the matrix/vector rate, so the kernel is built
// around bytes:
//
//   * a workgroup owns a TX x TY patch of output columns and marches along z over `zper` output planes;
//   * every input plane of its (TX+2) x (TY+2) halo patch is fetched ONCE (16-byte coalesced loads, issued one
//     plane ahead into registers, written to a single LDS image after the barrier) and contributes to the three
//     output planes it touches through three rotating accumulator sets (out[z] += P_dz[z + dz]): an input byte
//     crosses HBM/L2 -> LDS (zper+2)/zper times and is never re-read per tap;
//   * the 27 x C kernel taps come through the scalar path (wave-uniform), the x/y taps from LDS
//     (voxel pitch C+4 floats: conflict-free ds_read_b128);
//   * outputs leave as whole channel runs (16-byte stores), with the LeakyReLU / LeakyReLU-gradient gate fused.
//
// Patch shape (TX, TY) is picked per layer on the host so that the patches tile the plane with little waste
// (edges 130, 98, 96, 94 are not multiples of a power of two).
#include "tem_common.h"
#include <cstdio>
#include <cstdlib>

namespace stencil_c1 {

struct Dev {
  const float *in;
  int32_t iN, iD, iH, iW;          // input strides (elements), extents
  int32_t D, H, W;
  const float *w;
  float *out;
  int32_t oN, oD, oH, oW;
  int32_t OD, OH, OW;
  int32_t P;
  int32_t TX, TY, ntx, nty, zsegs, zper;
  uint32_t magicTX, magicCols;      // ceil(2^32 / TX), ceil(2^32 / (TX+2)) for the index splits
  float slope;
  const float *gate; int32_t gN, gD, gH, gW; float gate_slope;
  const float *bias;
  int32_t dbg;                     // ablation switches (TEM_DEBUG_FLAGS, perf triage only): 1 no stores, 2 one tap only, 4 no plane loads
};

// acc[co] += sum_ci xv[ci] * wt[ci*CO + co]   (wt wave-uniform -> scalar loads)
template <int CI, int CO>
__device__ __forceinline__ void fma_tap(float (&acc)[CO], const float (&xv)[CI], const float *__restrict__ wt) {
#pragma unroll
  for (int ci = 0; ci < CI; ++ci)
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = fmaf(xv[ci], wt[ci * CO + co], acc[co]);
}

// PF: loader chunks per thread per plane (16-byte chunks for CI >= 4, dwords for CI == 1)
template <int CI, int CO, bool FLIP, int PF>
__global__ __launch_bounds__(256, CI == 16 ? 3 : 4) void c1_stencil_k(Dev p, const float *__restrict__ wgt) {
  static_assert((CI == 1) != (CO == 1), "exactly one side has a single channel");
  constexpr int CIP = CI == 1 ? 1 : CI + 4;               // LDS voxel pitch (floats): 16-byte aligned, bank-spread
  constexpr int CH = CI == 1 ? 1 : 4;                     // floats per loader chunk
  constexpr int CPV = CI / CH;                            // chunks per voxel
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int cols = p.TX + 2, rows = p.TY + 2;

  int b = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);     // x-neighbours (shared halos) meet in one L2
  const int zseg = b % p.zsegs; b /= p.zsegs;
  const int txi = b % p.ntx; b /= p.ntx;
  const int tyi = b % p.nty;
  const int n = b / p.nty;
  const int ox0 = txi * p.TX, oy0 = tyi * p.TY;
  const int oz0 = zseg * p.zper, oz1 = min(p.OD, oz0 + p.zper);
  const int nplanes = oz1 - oz0 + 2;

  // this thread's output column
  const int ty = (int)__umulhi((uint32_t)tid, p.magicTX), tx = tid - ty * p.TX;
  const bool active = ty < p.TY;
  const int ox = ox0 + tx, oy = oy0 + ty;
  const bool owner = active && ox < p.OW && oy < p.OH;
  const float *lbase = lds + ((active ? ty : 0) * cols + (active ? tx : 0)) * CIP;

  // loader chunk descriptors (constant over the planes): global offset inside a plane, LDS offset, validity
  int goff[PF], loff[PF];
  uint32_t okmask = 0;
  const int total = rows * cols * CPV;
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const int id = tid + i * 256;
    const bool ex = id < total;
    const int vox = ex ? id / CPV : 0, c = ex ? (id - vox * CPV) * CH : 0;
    const int r = (int)__umulhi((uint32_t)vox, p.magicCols), cx = vox - r * cols;
    const int iy = oy0 - p.P + r, ix = ox0 - p.P + cx;
    const bool inb = ex && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    goff[i] = n * p.iN + iy * p.iH + ix * p.iW + c;
    loff[i] = ex ? vox * CIP + c : -1;
    okmask |= (inb ? 1u : 0u) << i;
  }
  float pf[PF][CH];
  auto load_plane = [&](int iz) {
    const bool zin = (unsigned)iz < (unsigned)p.D;
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const bool ok = zin && ((okmask >> i) & 1u);
      const float *src = p.in + (goff[i] + iz * p.iD);
      if constexpr (CH == 4) {
        const float4 t = ok ? *reinterpret_cast<const float4 *>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
        pf[i][0] = t.x; pf[i][1] = t.y; pf[i][2] = t.z; pf[i][3] = t.w;
      } else {
        pf[i][0] = ok ? *src : 0.f;
      }
    }
  };
  auto store_plane = [&]() {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      if (loff[i] >= 0) {
        if constexpr (CH == 4) *reinterpret_cast<float4 *>(lds + loff[i]) = make_float4(pf[i][0], pf[i][1], pf[i][2], pf[i][3]);
        else lds[loff[i]] = pf[i][0];
      }
    }
  };

  float acc[3][CO];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[s][c] = 0.f;

  auto finish = [&](float (&v)[CO], int oz) {
    if (!owner) return;
    if (p.bias) {
#pragma unroll
      for (int c = 0; c < CO; ++c) v[c] += p.bias[c];
    }
    if (p.gate) {
      const float *g = p.gate + (n * p.gN + oz * p.gD + oy * p.gH + ox * p.gW);
      if constexpr (CO % 4 == 0) {
#pragma unroll
        for (int c = 0; c < CO; c += 4) {
          const float4 g4 = *reinterpret_cast<const float4 *>(g + c);
          v[c] = g4.x > 0.f ? v[c] : p.gate_slope * v[c];
          v[c + 1] = g4.y > 0.f ? v[c + 1] : p.gate_slope * v[c + 1];
          v[c + 2] = g4.z > 0.f ? v[c + 2] : p.gate_slope * v[c + 2];
          v[c + 3] = g4.w > 0.f ? v[c + 3] : p.gate_slope * v[c + 3];
        }
      } else {
#pragma unroll
        for (int c = 0; c < CO; ++c) v[c] = g[c] > 0.f ? v[c] : p.gate_slope * v[c];
      }
    }
    if (p.slope != 1.f) {
#pragma unroll
      for (int c = 0; c < CO; ++c) v[c] = v[c] > 0.f ? v[c] : p.slope * v[c];
    }
    float *o = p.out + (n * p.oN + oz * p.oD + oy * p.oH + ox * p.oW);
    if (p.dbg & 1) return;
    if constexpr (CO % 4 == 0) {
#pragma unroll
      for (int c = 0; c < CO; c += 4) *reinterpret_cast<float4 *>(o + c) = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
    } else {
#pragma unroll
      for (int c = 0; c < CO; ++c) o[c] = v[c];
    }
  };

  const int iz0 = oz0 - p.P;
  load_plane(iz0);
  store_plane();
  __syncthreads();

  for (int j0 = 0; j0 < nplanes; j0 += 3) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int j = j0 + r;
      if (j < nplanes) {                                     // block-uniform
        const bool more = j + 1 < nplanes;
        if (more && !(p.dbg & 4)) load_plane(iz0 + j + 1);   // next plane's HBM/L2 reads fly under this plane's FMAs
        // plane j feeds output planes j (tap dz 0), j-1 (dz 1), j-2 (dz 2): accumulator slot (j - dz) mod 3
        const int ndy = (p.dbg & 2) ? 1 : 3;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
          if (dy < ndy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            float xv[CI];
            const float *src = lbase + (dy * cols + dx) * CIP;
            if constexpr (CI == 1) {
              xv[0] = src[0];
            } else {
#pragma unroll
              for (int c = 0; c < CI; c += 4) {
                const float4 t = *reinterpret_cast<const float4 *>(src + c);
                xv[c] = t.x; xv[c + 1] = t.y; xv[c + 2] = t.z; xv[c + 3] = t.w;
              }
            }
#pragma unroll
            for (int dz = 0; dz < 3; ++dz) {
              const int tap = (dz * 3 + dy) * 3 + dx;
              fma_tap<CI, CO>(acc[(r + 3 - dz) % 3], xv, wgt + (FLIP ? 26 - tap : tap) * (CI * CO));
            }
          }
        // output plane j-2 is complete
        if (j >= 2) finish(acc[(r + 1) % 3], oz0 + j - 2);
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[(r + 1) % 3][c] = 0.f;
        __syncthreads();                                     // every wave is done reading plane j
        if (more) store_plane();
        __syncthreads();
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ C_in == 1 on the matrix cores
// out[v][co] = sum_tap x[v + tap] * w[tap][co] is a [27 x C_out]^T x [27 x voxels] product.  v_mfma_f32_16x16x4_f32 with the
// 27 taps as the K dimension keeps every kernel tap in REGISTERS (one VGPR per k-step, loaded once per wave) -- the VALU form
// has to stream all 27 x C_out taps through the scalar path for every output voxel.  The kernel is the M side (rows =
// output channels), 16 consecutive voxels of a row the N side: the C/D map (column = lane & 15, rows 4 (lane >> 4) + r) then
// leaves a lane with FOUR CONSECUTIVE CHANNELS OF ONE VOXEL -- a 16-byte store straight from the accumulator, no transpose
// (round 2 had the voxels as rows and sent every tile through an LDS patch: 5 LDS operations and a wait per tile).
// C_out == 8 would fill half of the 16 rows: the other half takes the NEXT output plane (rows = (plane, co)), K then spans
// the 4 input planes the plane pair touches (36 = 9 k-steps, 27 of the 36 products per row are real).  C_out == 16: K = 27
// padded to 28 (7 k-steps).  A workgroup loads its input patch (1 channel: a few KB) into LDS once -- 16-byte loads, all of a
// thread's loads in flight before the first LDS write -- and every wave gathers its voxel-side fragments from it (ds_read_b32,
// lane -> (voxel, tap)), two tiles at a time; the gate values of a tile pair are requested before its MFMA chain.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct DevM {
  const float *in;
  int32_t iN, iD, iH, D, H, W, in_bytes;
  float *out;
  int32_t oN, oD, oH, oW, OD, OH, OW;
  int32_t P;
  int32_t TXT, TY, nty, nzg;        // 16-voxel tiles per row, rows per patch, patches in y, plane groups in z
  int32_t colsP, cpr;               // LDS row pitch (floats, multiple of 4) and 16-byte chunks per row
  int32_t xs, sh;                   // input x of LDS column 0 (a multiple of 4: chunks never straddle x = 0 or x = W) and
                                    // the column of the first voxel output 0 reads: -P - xs in 0..3
  uint32_t magicCpr, magicTXT;
  float slope;
  const float *gate; int32_t gN, gD, gH, gW; float gate_slope;
  int32_t out_bytes, gate_bytes;    // bytes one sample of out / gate spans (buffer ranges)
  int32_t dbg;
};

// K order: the taps are enumerated as NP * 3 ROWS (zi, dy) of 3 x-taps; lane group kq owns the rows kq, kq + 4, kq + 8 and
// k-step s = 3 j + dx multiplies x-tap dx of the group's row 4 j + kq.  A lane's three taps of a row are adjacent floats of the
// patch: ONE ds_read2_b32 + one ds_read_b32 per row instead of three gathers, one address add per row (3 per tile, not 9).
// With the row pitch = 16 mod 32 floats and an odd TY the rows of the two lane groups of an LDS cycle are 16 banks apart:
// conflict-free.  EPI: 0 = LeakyReLU(slope) (slope 1: linear), 1 = LeakyReLU' gate on a saved activation -- compiled in, and
// invalid lanes (tile over-hang) are out-of-range buffer offsets instead of exec-mask regions: the loop body is branch-free
// and its loads / stores need no vmcnt(0) at a join (round 2's form waited for every tile's STORE to complete).
template <int CO, bool FLIP, int EPI>
__global__ __launch_bounds__(256) void c1_mfma_k(DevM p, const float *__restrict__ wgt) {
  static_assert(CO == 8 || CO == 16, "rows = (plane pair, 8 channels) or 16 channels");
  constexpr int NZ = CO == 8 ? 2 : 1;                     // output planes per tile
  constexpr int NP = NZ + 2;                              // input planes per patch
  constexpr int NR = (NP * 3 + 3) / 4;                    // tap rows (zi, dy) per lane group: 3
  constexpr int KS = NR * 3;                              // k-steps of 4 taps: 9
  constexpr int NCH = 2;                                  // 16-byte patch chunks per thread and plane (rows * cpr <= 512)
  constexpr int OOB = (int)0x80000000;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const int rows = p.TY + 2;
  const int plane = rows * p.colsP;

  int b = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);     // z-neighbours share input planes: one XCD's L2
  const int zg = b % p.nzg; b /= p.nzg;
  const int typ = b % p.nty;
  const int n = b / p.nty;
  const int oy0 = typ * p.TY, oz0 = zg * NZ;

  // ---- kernel fragments (the MFMA's A side): lane (row = m, k-step s = 3 j + dx -> tap row 4 j + kq = (zi, dy), x-tap dx)
  float B[KS];
  int rowoff[NR];                                         // byte offset of the lane's tap row j in the patch (+ voxel m)
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int r = 4 * j + kq;
    const bool real = r < NP * 3;
    const int zi = real ? r / 3 : 0, dy = real ? r - zi * 3 : 0;
    rowoff[j] = (zi * plane + dy * p.colsP + m + p.sh) * 4;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      float v = 0.f;
      if (real) {
        const int zo = CO == 8 ? (m >> 3) : 0, dz = zi - zo;
        if (dz >= 0 && dz <= 2) {
          const int tap = dz * 9 + dy * 3 + dx;
          v = wgt[(FLIP ? 26 - tap : tap) * CO + (CO == 8 ? (m & 7) : m)];
        }
      }
      B[3 * j + dx] = v;
    }
  }

  // ---- load the patch: NP planes x rows x colsP, zeros outside the input (padding / volume border).  One 16-byte load per
  // chunk (4 x-consecutive voxels of the 1-channel input; any 4-byte alignment), every load of the thread issued before the
  // first LDS write: one memory round trip per workgroup.
  {
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, p.in_bytes, 0x00020000);
    const int nchunk = rows * p.cpr;                       // host: <= NCH * 256
    u32x4 pf[NP][NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + i * 256;
      const int r = (int)__umulhi((uint32_t)id, p.magicCpr), c4 = id - r * p.cpr;
      const int iy = oy0 - p.P + r;
      const int ix0 = p.xs + 4 * c4;                       // W and xs are multiples of 4: a chunk is inside the row or outside
      const bool okxy = id < nchunk && (unsigned)iy < (unsigned)p.H && (unsigned)ix0 < (unsigned)p.W;
      const int base = (n * p.iN + iy * p.iH + ix0) * 4;
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
        const int iz = oz0 - p.P + pl;
        const int off = (okxy && (unsigned)iz < (unsigned)p.D && !(p.dbg & 4)) ? base + iz * p.iD * 4 : OOB;
        pf[pl][i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + i * 256;
      if (id < nchunk) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<u32x4 *>(lds + pl * plane + id * 4) = pf[pl][i];
      }
    }
  }
  __syncthreads();

  // lane's output role: voxel m of the tile, channels co0 .. co0 + 3 of output plane oz0 + zo
  const int zo = CO == 8 ? (kq >> 1) : 0, co0 = CO == 8 ? 4 * (kq & 1) : 4 * kq;
  const int oz = oz0 + zo;
  const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)(p.out + (size_t)n * p.oN), 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.gate + (size_t)n * p.gN), 0, EPI == 1 ? p.gate_bytes : 0, 0x00020000);
  const bool zok = oz < p.OD && !(p.dbg & 1);
  const int obase = (oz * p.oD + oy0 * p.oH + m * p.oW + co0) * 4;          // + trow * oH * 4 + tcol * 16 * oW * 4
  const int gbase = EPI == 1 ? (oz * p.gD + oy0 * p.gH + m * p.gW + co0) * 4 : 0;
  const int ntiles = (p.dbg & 2) ? 0 : p.TXT * p.TY;
  const char *const ldsb = reinterpret_cast<const char *>(lds);
  // two tiles per iteration (tiles t and t + 4; a wave walks t = wave, wave + 8, ...): their MFMA chains -- each a dependent
  // chain on its own accumulator -- interleave.  (trow, tcol) advance as scalars.
  int tr0 = 0, tc0 = wave;
  while (tc0 >= p.TXT) { tc0 -= p.TXT; ++tr0; }
  for (int t = wave; t < ntiles; t += 8) {                // wave-uniform
    int tr1 = tr0, tc1 = tc0 + 4;
    while (tc1 >= p.TXT) { tc1 -= p.TXT; ++tr1; }
    const bool two = t + 4 < ntiles;
    if (!two) { tr1 = tr0; tc1 = tc0; }
    const bool ok0 = zok && tc0 * 16 + m < p.OW && oy0 + tr0 < p.OH;
    const bool ok1 = zok && two && tc1 * 16 + m < p.OW && oy0 + tr1 < p.OH;
    int oo0 = ok0 ? obase + (tr0 * p.oH + tc0 * 16 * p.oW) * 4 : OOB, oo1 = ok1 ? obase + (tr1 * p.oH + tc1 * 16 * p.oW) * 4 : OOB;
    u32x4 g0 = {0u, 0u, 0u, 0u}, g1 = g0;
    if (EPI == 1) {                                       // requested ahead of the MFMA chain
      const int go0 = ok0 ? gbase + (tr0 * p.gH + tc0 * 16 * p.gW) * 4 : OOB, go1 = ok1 ? gbase + (tr1 * p.gH + tc1 * 16 * p.gW) * 4 : OOB;
      g0 = __builtin_amdgcn_raw_buffer_load_b128(grs, go0, 0, 0);
      g1 = __builtin_amdgcn_raw_buffer_load_b128(grs, go1, 0, 0);
    }
    const int tb0 = (tr0 * p.colsP + tc0 * 16) * 4, tb1 = (tr1 * p.colsP + tc1 * 16) * 4;
    float a0[KS], a1[KS];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const float *s0 = reinterpret_cast<const float *>(ldsb + rowoff[j] + tb0), *s1 = reinterpret_cast<const float *>(ldsb + rowoff[j] + tb1);
      a0[3 * j] = s0[0]; a0[3 * j + 1] = s0[1]; a0[3 * j + 2] = s0[2];
      a1[3 * j] = s1[0]; a1[3 * j + 1] = s1[1]; a1[3 * j + 2] = s1[2];
    }
    __builtin_amdgcn_sched_barrier(0);                    // (every read in flight before the first MFMA: left alone, the
                                                          // scheduler sinks each read next to its MFMA -- ds_read, lgkmcnt(0), MFMA)
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(B[s], a0[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(B[s], a1[s], acc1, 0, 0, 0);
    }
    u32x4 o0, o1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v0 = acc0[c], v1 = acc1[c];
      if (EPI == 1) {
        v0 = __uint_as_float(g0[c]) > 0.f ? v0 : p.gate_slope * v0;
        v1 = __uint_as_float(g1[c]) > 0.f ? v1 : p.gate_slope * v1;
      } else {
        v0 = v0 > 0.f ? v0 : p.slope * v0;                // (slope 1: the same value either way)
        v1 = v1 > 0.f ? v1 : p.slope * v1;
      }
      o0[c] = __float_as_uint(v0); o1[c] = __float_as_uint(v1);
    }
    asm volatile("" : "+v"(oo0), "+v"(oo1));
    __builtin_amdgcn_raw_buffer_store_b128(o0, ors, oo0, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(o1, ors, oo1, 0, 0);
    tc0 += 8;
    while (tc0 >= p.TXT) { tc0 -= p.TXT; ++tr0; }
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

template <int CI, int CO, bool FLIP, int PF>
int run(Dev p, int N, hipStream_t st, bool dry) {
  constexpr int CIP = CI == 1 ? 1 : CI + 4, CPV = CI == 1 ? 1 : CI / 4;
  // patch shape: maximise useful lanes x (1 / halo over-fetch), subject to the loader's register budget
  double best = -1.0;
  for (int TX = 8; TX <= 128; ++TX) {
    const int TY = 256 / TX;
    if (TY < 2) break;
    if ((TX + 2) * (TY + 2) * CPV > PF * 256) continue;
    const int ntx = (p.OW + TX - 1) / TX, nty = (p.OH + TY - 1) / TY;
    const double lanes = (double)p.OW * p.OH / ((double)ntx * nty * 256.0);
    const double halo = (double)(TX * TY) / ((TX + 2) * (TY + 2));
    const double score = lanes * (CI == 1 ? 1.0 : halo);         // the one-channel input is cheap to over-fetch
    if (score > best) { best = score; p.TX = TX; p.TY = TY; p.ntx = ntx; p.nty = nty; }
  }
  if (best < 0) return TEM_EUNSUPPORTED;
  // z-run: enough workgroups to keep every CU's queue full, but >= 4 planes per run (an input plane is fetched
  // (zper+2)/zper times) where the volume allows
  const int tiles = p.ntx * p.nty * N;
  int zsegs = (3072 + tiles - 1) / tiles;
  if (zsegs < 1) zsegs = 1;
  int zper = (p.OD + zsegs - 1) / zsegs;
  if (zper < 4) zper = p.OD < 4 ? p.OD : 4;
  p.zper = zper;
  p.zsegs = (p.OD + zper - 1) / zper;
  p.magicTX = magic_for(p.TX);
  p.magicCols = magic_for(p.TX + 2);
  if (dry) {
    if (g_name) snprintf(g_name, g_name_len, "c1_stencil_k<%d, %d, %s, %d>", CI, CO, FLIP ? "true" : "false", PF);
    return TEM_OK;
  }
  static int dbg = -1;
  if (dbg < 0) dbg = tem_env_int("TEM_DEBUG_FLAGS", 0);
  p.dbg = dbg;
  if (dbg & 8)
    fprintf(stderr, "c1_stencil<%d,%d> OW=%d OH=%d OD=%d: TX=%d TY=%d zper=%d zsegs=%d blocks=%d\n", CI, CO, p.OW, p.OH,
            p.OD, p.TX, p.TY, p.zper, p.zsegs, p.zsegs * p.ntx * p.nty * N);
  const size_t lds_bytes = (((size_t)(p.TX + 2) * (p.TY + 2) * CIP * 4) + 15) & ~(size_t)15;
  const int nblocks = p.zsegs * p.ntx * p.nty * N;
  hipLaunchKernelGGL((c1_stencil_k<CI, CO, FLIP, PF>), dim3((unsigned)nblocks), dim3(256), lds_bytes, st, p, p.w);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

template <int CO, bool FLIP>
int run_mfma(const Dev &q, int N, hipStream_t st, bool dry) {
  constexpr int NZ = CO == 8 ? 2 : 1, NP = NZ + 2;
  DevM p{};
  if (q.iW != 1 || q.W % 4) return TEM_EUNSUPPORTED;      // 16-byte loads of 4 x-consecutive voxels, never across a row end
  p.in = q.in; p.iN = q.iN; p.iD = q.iD; p.iH = q.iH; p.D = q.D; p.H = q.H; p.W = q.W;
  {
    const int64_t span = ((int64_t)(N - 1) * q.iN + (int64_t)(q.D - 1) * q.iD + (int64_t)(q.H - 1) * q.iH + q.W) * 4;
    if (span >= ((int64_t)1 << 31)) return TEM_EUNSUPPORTED;
    p.in_bytes = (int)span;
  }
  p.out = q.out; p.oN = q.oN; p.oD = q.oD; p.oH = q.oH; p.oW = q.oW; p.OD = q.OD; p.OH = q.OH; p.OW = q.OW;
  p.P = q.P; p.slope = q.slope; p.gate = q.gate; p.gN = q.gN; p.gD = q.gD; p.gH = q.gH; p.gW = q.gW;
  p.gate_slope = q.gate_slope; p.dbg = q.dbg;
  if (q.bias) return TEM_EUNSUPPORTED;                    // (no one-channel layer of the path has a bias: the VALU form takes it)
  if (q.gate && q.slope != 1.f) return TEM_EUNSUPPORTED;  // compiled epilogues: LeakyReLU, or the gate
  {
    const int64_t ospan = ((int64_t)(q.OD - 1) * q.oD + (int64_t)(q.OH - 1) * q.oH + (int64_t)(q.OW - 1) * q.oW + CO) * 4;
    const int64_t gspan = q.gate ? ((int64_t)(q.OD - 1) * q.gD + (int64_t)(q.OH - 1) * q.gH + (int64_t)(q.OW - 1) * q.gW + CO) * 4 : 0;
    if (ospan >= ((int64_t)1 << 31) || gspan >= ((int64_t)1 << 31)) return TEM_EUNSUPPORTED;
    p.out_bytes = (int)ospan; p.gate_bytes = (int)gspan;
  }
  p.TXT = (p.OW + 15) / 16;
  p.xs = p.P <= 0 ? (-p.P / 4) * 4 : -((p.P + 3) / 4) * 4;  // floor(-P / 4) * 4
  p.sh = -p.P - p.xs;
  p.colsP = p.TXT * 16 + 16;                              // columns sh .. sh + 16 TXT + 1 are read (sh <= 3); pitch = 16 mod 32
  p.cpr = p.colsP / 4;
  // rows per patch: ~64 tiles per workgroup (16 per wave; the patch's two halo rows are (TY + 2) / TY of the input reads),
  // bounded by the loader's two chunks per thread and plane
  p.TY = (64 + p.TXT - 1) / p.TXT;
  if (p.TY > p.OH) p.TY = p.OH;
  while (p.TY > 1 && (p.TY + 2) * p.cpr > 2 * 256) --p.TY;
  if ((p.TY + 2) * p.cpr > 2 * 256) return TEM_EUNSUPPORTED;
  p.nty = (p.OH + p.TY - 1) / p.TY;
  p.TY = (p.OH + p.nty - 1) / p.nty;                       // even bands
  if (!(p.TY & 1) && (p.TY + 3) * p.cpr <= 2 * 256) ++p.TY;   // odd: tap rows of neighbouring planes 16 banks apart
  p.nty = (p.OH + p.TY - 1) / p.TY;
  p.nzg = (p.OD + NZ - 1) / NZ;
  p.magicCpr = magic_for(p.cpr);
  p.magicTXT = magic_for(p.TXT);
  const size_t lds_bytes = (size_t)NP * (p.TY + 2) * p.colsP * 4;
  if (lds_bytes > 64 * 1024) return TEM_EUNSUPPORTED;
  if (dry) {
    if (g_name) snprintf(g_name, g_name_len, "c1_mfma_k<%d, %s, %d>", CO, FLIP ? "true" : "false", q.gate ? 1 : 0);
    return TEM_OK;
  }
  const int nblocks = N * p.nty * p.nzg;
  if (q.gate) hipLaunchKernelGGL((c1_mfma_k<CO, FLIP, 1>), dim3((unsigned)nblocks), dim3(256), lds_bytes, st, p, q.w);
  else hipLaunchKernelGGL((c1_mfma_k<CO, FLIP, 0>), dim3((unsigned)nblocks), dim3(256), lds_bytes, st, p, q.w);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

int dispatch(const tem_conv_args *a, hipStream_t st, bool dry) {
  const tem_view &i0 = a->in0, &o0 = a->out0;
  if (a->in1.ptr || a->out1.ptr) return TEM_EUNSUPPORTED;
  if (a->kd != 3 || a->kh != 3 || a->kw != 3 || a->sd != 1 || a->sh != 1 || a->sw != 1) return TEM_EUNSUPPORTED;
  if (a->pd != a->ph || a->ph != a->pw) return TEM_EUNSUPPORTED;
  if (a->ep.dropout || a->ep.add.ptr) return TEM_EUNSUPPORTED;
  if (o0.N != i0.N) return TEM_ESHAPE;
  if (!fits32(i0) || !fits32(o0)) return TEM_EUNSUPPORTED;
  static int enabled = -1;
  if (enabled < 0) enabled = tem_env_int("TEM_STENCIL_C1", 1);
  if (!enabled) return TEM_EUNSUPPORTED;
  const int CI = i0.C, CO = o0.C;
  auto aligned = [](const tem_view &v) {
    return v.C % 4 != 0 || (((uintptr_t)v.ptr & 15) == 0 && v.sW % 4 == 0 && v.sH % 4 == 0 && v.sD % 4 == 0 && v.sN % 4 == 0);
  };
  if (!aligned(i0) || !aligned(o0)) return TEM_EUNSUPPORTED;
  Dev p{};
  p.in = i0.ptr; p.iN = (int)i0.sN; p.iD = (int)i0.sD; p.iH = (int)i0.sH; p.iW = (int)i0.sW;
  p.D = i0.D; p.H = i0.H; p.W = i0.W;
  p.w = a->w;
  p.out = o0.ptr; p.oN = (int)o0.sN; p.oD = (int)o0.sD; p.oH = (int)o0.sH; p.oW = (int)o0.sW;
  p.OD = o0.D; p.OH = o0.H; p.OW = o0.W;
  p.P = a->pd;
  p.slope = a->ep.slope; p.gate_slope = a->ep.gate_slope; p.bias = a->ep.bias;
  if (a->ep.gate.ptr) {
    const tem_view &g = a->ep.gate;
    if (g.N != o0.N || g.D != o0.D || g.H != o0.H || g.W != o0.W || g.C < o0.C) return TEM_ESHAPE;
    if (!fits32(g) || !aligned(g)) return TEM_EUNSUPPORTED;
    p.gate = g.ptr; p.gN = (int)g.sN; p.gD = (int)g.sD; p.gH = (int)g.sH; p.gW = (int)g.sW;
  }
  const bool flip = a->w_layout == TEM_W_FLIP_CO_CI;
  const int N = i0.N;
  { static int dbg = -1; if (dbg < 0) dbg = tem_env_int("TEM_DEBUG_FLAGS", 0); p.dbg = dbg; }
#define C1_CASE(ci, co, pf) \
  if (CI == ci && CO == co) return flip ? run<ci, co, true, pf>(p, N, st, dry) : run<ci, co, false, pf>(p, N, st, dry);
  static int use_mfma = -1;
  if (use_mfma < 0) use_mfma = tem_env_int("TEM_C1_MFMA", 1);
  if (use_mfma && CI == 1 && CO == 8) {    // g.c0, d.d1a forward
    const int rc = flip ? run_mfma<8, true>(p, N, st, dry) : run_mfma<8, false>(p, N, st, dry);
    if (rc != TEM_EUNSUPPORTED) return rc;  // (odd row lengths / strided x: the VALU form below)
  }
  if (use_mfma && CI == 1 && CO == 16) {   // input-gradient of g.f2 (gated by f1)
    const int rc = flip ? run_mfma<16, true>(p, N, st, dry) : run_mfma<16, false>(p, N, st, dry);
    if (rc != TEM_EUNSUPPORTED) return rc;
  }
  C1_CASE(1, 8, 3)
  C1_CASE(1, 16, 3)
  C1_CASE(16, 1, 6)      // g.f2 forward
  C1_CASE(8, 1, 3)       // input-gradients of g.c0 / d.d1a (the cycle and adversarial paths need dx)
#undef C1_CASE
  return TEM_EUNSUPPORTED;
}


// ------------------------------------------------------------------------------------------ bf16 in / out (config 5)
// c1_mfma_k with bf16 tensors (the shape-generic conv_bf16_k took 30 us for g.c0 and 23 us for the input-gradient of g.f2:
// the 27 taps of a one-channel input are ONE k-step of the bf16 MFMA, so its per-tile index work is all there is).  The LDS
// patch stays fp32 -- the loader widens its 8-byte chunks of 4 voxels once (bf16 -> fp32 is a shift) -- so the fragment reads
// and the fp32 MFMA chain are c1_mfma_k's; the kernel copy (packed bf16 [tap][co]) is widened into the B registers, the gate
// arrives as 8 bytes per lane and the lane's 4 channels leave as an 8-byte bf16 store.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
typedef float f32x2_ __attribute__((ext_vector_type(2)));

template <int CO, bool FLIP, int EPI>
__global__ __launch_bounds__(256) void c1_mfma_h_k(DevM p, const unsigned short *__restrict__ wgt) {
  typedef unsigned short u16;
  constexpr int NZ = CO == 8 ? 2 : 1, NP = NZ + 2, NR = (NP * 3 + 3) / 4, KS = NR * 3, NCH = 2;
  constexpr int OOB = (int)0x80000000;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const int rows = p.TY + 2;
  const int plane = rows * p.colsP;
  int b = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int zg = b % p.nzg; b /= p.nzg;
  const int typ = b % p.nty;
  const int n = b / p.nty;
  const int oy0 = typ * p.TY, oz0 = zg * NZ;
  auto up = [](uint32_t h) { return __uint_as_float(h << 16); };

  float B[KS];
  int rowoff[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int r = 4 * j + kq;
    const bool real = r < NP * 3;
    const int zi = real ? r / 3 : 0, dy = real ? r - zi * 3 : 0;
    rowoff[j] = (zi * plane + dy * p.colsP + m + p.sh) * 4;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      float v = 0.f;
      if (real) {
        const int zo = CO == 8 ? (m >> 3) : 0, dz = zi - zo;
        if (dz >= 0 && dz <= 2) {
          const int tap = dz * 9 + dy * 3 + dx;
          v = up(wgt[(FLIP ? 26 - tap : tap) * CO + (CO == 8 ? (m & 7) : m)]);
        }
      }
      B[3 * j + dx] = v;
    }
  }
  {
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, p.in_bytes, 0x00020000);
    const int nchunk = rows * p.cpr;
    u32x2 pf[NP][NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + i * 256;
      const int r = (int)__umulhi((uint32_t)id, p.magicCpr), c4 = id - r * p.cpr;
      const int iy = oy0 - p.P + r;
      const int ix0 = p.xs + 4 * c4;
      const bool okxy = id < nchunk && (unsigned)iy < (unsigned)p.H && (unsigned)ix0 < (unsigned)p.W;
      const int base = (n * p.iN + iy * p.iH + ix0) * 2;
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
        const int iz = oz0 - p.P + pl;
        const int off = (okxy && (unsigned)iz < (unsigned)p.D) ? base + iz * p.iD * 2 : OOB;
        pf[pl][i] = __builtin_amdgcn_raw_buffer_load_b64(xrs, off, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + i * 256;
      if (id < nchunk) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
          const u32x2 v = pf[pl][i];
          *reinterpret_cast<u32x4 *>(lds + pl * plane + id * 4) = u32x4{v.x << 16, v.x & 0xffff0000u, v.y << 16, v.y & 0xffff0000u};
        }
      }
    }
  }
  __syncthreads();

  const int zo = CO == 8 ? (kq >> 1) : 0, co0 = CO == 8 ? 4 * (kq & 1) : 4 * kq;
  const int oz = oz0 + zo;
  const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)((u16 *)p.out + (size_t)n * p.oN), 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)((const u16 *)p.gate + (size_t)n * p.gN), 0, EPI == 1 ? p.gate_bytes : 0, 0x00020000);
  const bool zok = oz < p.OD;
  const int obase = (oz * p.oD + oy0 * p.oH + m * p.oW + co0) * 2;
  const int gbase = EPI == 1 ? (oz * p.gD + oy0 * p.gH + m * p.gW + co0) * 2 : 0;
  const int ntiles = p.TXT * p.TY;
  const char *const ldsb = reinterpret_cast<const char *>(lds);
  int tr0 = 0, tc0 = wave;
  while (tc0 >= p.TXT) { tc0 -= p.TXT; ++tr0; }
  for (int t = wave; t < ntiles; t += 8) {
    int tr1 = tr0, tc1 = tc0 + 4;
    while (tc1 >= p.TXT) { tc1 -= p.TXT; ++tr1; }
    const bool two = t + 4 < ntiles;
    if (!two) { tr1 = tr0; tc1 = tc0; }
    const bool ok0 = zok && tc0 * 16 + m < p.OW && oy0 + tr0 < p.OH;
    const bool ok1 = zok && two && tc1 * 16 + m < p.OW && oy0 + tr1 < p.OH;
    int oo0 = ok0 ? obase + (tr0 * p.oH + tc0 * 16 * p.oW) * 2 : OOB, oo1 = ok1 ? obase + (tr1 * p.oH + tc1 * 16 * p.oW) * 2 : OOB;
    u32x2 g0 = {0u, 0u}, g1 = g0;
    if (EPI == 1) {
      const int go0 = ok0 ? gbase + (tr0 * p.gH + tc0 * 16 * p.gW) * 2 : OOB, go1 = ok1 ? gbase + (tr1 * p.gH + tc1 * 16 * p.gW) * 2 : OOB;
      g0 = __builtin_amdgcn_raw_buffer_load_b64(grs, go0, 0, 0);
      g1 = __builtin_amdgcn_raw_buffer_load_b64(grs, go1, 0, 0);
    }
    const int tb0 = (tr0 * p.colsP + tc0 * 16) * 4, tb1 = (tr1 * p.colsP + tc1 * 16) * 4;
    float a0[KS], a1[KS];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const float *s0 = reinterpret_cast<const float *>(ldsb + rowoff[j] + tb0), *s1 = reinterpret_cast<const float *>(ldsb + rowoff[j] + tb1);
      a0[3 * j] = s0[0]; a0[3 * j + 1] = s0[1]; a0[3 * j + 2] = s0[2];
      a1[3 * j] = s1[0]; a1[3 * j + 1] = s1[1]; a1[3 * j + 2] = s1[2];
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(B[s], a0[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(B[s], a1[s], acc1, 0, 0, 0);
    }
    float v0[4], v1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      v0[c] = acc0[c]; v1[c] = acc1[c];
      if (EPI == 1) {
        const uint32_t h0 = (c & 1) ? (g0[c >> 1] & 0xffff0000u) : (g0[c >> 1] << 16), h1 = (c & 1) ? (g1[c >> 1] & 0xffff0000u) : (g1[c >> 1] << 16);
        v0[c] = __uint_as_float(h0) > 0.f ? v0[c] : p.gate_slope * v0[c];
        v1[c] = __uint_as_float(h1) > 0.f ? v1[c] : p.gate_slope * v1[c];
      } else {
        v0[c] = v0[c] > 0.f ? v0[c] : p.slope * v0[c];
        v1[c] = v1[c] > 0.f ? v1[c] : p.slope * v1[c];
      }
    }
    auto pk = [](float a, float b_) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_{a, b_}, bf16x2_)); };
    asm volatile("" : "+v"(oo0), "+v"(oo1));
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{pk(v0[0], v0[1]), pk(v0[2], v0[3])}, ors, oo0, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{pk(v1[0], v1[1]), pk(v1[2], v1[3])}, ors, oo1, 0, 0);
    tc0 += 8;
    while (tc0 >= p.TXT) { tc0 -= p.TXT; ++tr0; }
  }
}

template <int CO, bool FLIP>
static int run_mfma_h(const Dev &q, int N, hipStream_t st, bool dry, char *name, int name_len) {
  constexpr int NZ = CO == 8 ? 2 : 1, NP = NZ + 2;
  DevM p{};
  if (q.iW != 1 || q.W % 4 || q.iH % 2 || q.iD % 2 || q.iN % 2 || ((uintptr_t)q.in & 3)) return TEM_EUNSUPPORTED;   // 8-byte loads of 4 voxels
  p.in = q.in; p.iN = q.iN; p.iD = q.iD; p.iH = q.iH; p.D = q.D; p.H = q.H; p.W = q.W;
  {
    const int64_t span = ((int64_t)(N - 1) * q.iN + (int64_t)(q.D - 1) * q.iD + (int64_t)(q.H - 1) * q.iH + q.W) * 2;
    if (span >= ((int64_t)1 << 31)) return TEM_EUNSUPPORTED;
    p.in_bytes = (int)span;
  }
  p.out = q.out; p.oN = q.oN; p.oD = q.oD; p.oH = q.oH; p.oW = q.oW; p.OD = q.OD; p.OH = q.OH; p.OW = q.OW;
  p.P = q.P; p.slope = q.slope; p.gate = q.gate; p.gN = q.gN; p.gD = q.gD; p.gH = q.gH; p.gW = q.gW;
  p.gate_slope = q.gate_slope;
  if (q.bias) return TEM_EUNSUPPORTED;
  if (q.gate && q.slope != 1.f) return TEM_EUNSUPPORTED;
  {
    const int64_t ospan = ((int64_t)(q.OD - 1) * q.oD + (int64_t)(q.OH - 1) * q.oH + (int64_t)(q.OW - 1) * q.oW + CO) * 2;
    const int64_t gspan = q.gate ? ((int64_t)(q.OD - 1) * q.gD + (int64_t)(q.OH - 1) * q.gH + (int64_t)(q.OW - 1) * q.gW + CO) * 2 : 0;
    if (ospan >= ((int64_t)1 << 31) || gspan >= ((int64_t)1 << 31)) return TEM_EUNSUPPORTED;
    p.out_bytes = (int)ospan; p.gate_bytes = (int)gspan;
  }
  p.TXT = (p.OW + 15) / 16;
  p.xs = p.P <= 0 ? (-p.P / 4) * 4 : -((p.P + 3) / 4) * 4;
  p.sh = -p.P - p.xs;
  p.colsP = p.TXT * 16 + 16;
  p.cpr = p.colsP / 4;
  p.TY = (64 + p.TXT - 1) / p.TXT;
  if (p.TY > p.OH) p.TY = p.OH;
  while (p.TY > 1 && (p.TY + 2) * p.cpr > 2 * 256) --p.TY;
  if ((p.TY + 2) * p.cpr > 2 * 256) return TEM_EUNSUPPORTED;
  p.nty = (p.OH + p.TY - 1) / p.TY;
  p.TY = (p.OH + p.nty - 1) / p.nty;
  if (!(p.TY & 1) && (p.TY + 3) * p.cpr <= 2 * 256) ++p.TY;
  p.nty = (p.OH + p.TY - 1) / p.TY;
  p.nzg = (p.OD + NZ - 1) / NZ;
  p.magicCpr = magic_for(p.cpr);
  p.magicTXT = magic_for(p.TXT);
  const size_t lds_bytes = (size_t)NP * (p.TY + 2) * p.colsP * 4;
  if (lds_bytes > 64 * 1024) return TEM_EUNSUPPORTED;
  if (name) snprintf(name, name_len, "c1_mfma_h_k<%d, %s, %d>", CO, FLIP ? "true" : "false", q.gate ? 1 : 0);
  if (dry) return TEM_OK;
  const int nblocks = N * p.nty * p.nzg;
  const unsigned short *w = reinterpret_cast<const unsigned short *>(q.w);
  if (q.gate) hipLaunchKernelGGL((c1_mfma_h_k<CO, FLIP, 1>), dim3((unsigned)nblocks), dim3(256), lds_bytes, st, p, w);
  else hipLaunchKernelGGL((c1_mfma_h_k<CO, FLIP, 0>), dim3((unsigned)nblocks), dim3(256), lds_bytes, st, p, w);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

// bf16 tensors behind the float* fields of tem_conv_args (strides in elements), `w` = the packed bf16 kernel [tap][co]
int dispatch_h(const tem_conv_args *a, hipStream_t st, bool dry, char *name, int name_len) {
  const tem_view &i0 = a->in0, &o0 = a->out0;
  if (a->in1.ptr || a->out1.ptr || i0.C != 1 || (o0.C != 8 && o0.C != 16)) return TEM_EUNSUPPORTED;
  if (a->kd != 3 || a->kh != 3 || a->kw != 3 || a->sd != 1 || a->sh != 1 || a->sw != 1) return TEM_EUNSUPPORTED;
  if (a->pd != a->ph || a->ph != a->pw) return TEM_EUNSUPPORTED;
  if (a->ep.dropout || a->ep.add.ptr) return TEM_EUNSUPPORTED;
  if (o0.N != i0.N) return TEM_ESHAPE;
  if (!fits32(i0) || !fits32(o0)) return TEM_EUNSUPPORTED;
  auto al8 = [](const tem_view &v) {      // 8-byte accesses of 4 bf16 channels
    return ((uintptr_t)v.ptr & 7) == 0 && v.sW % 4 == 0 && v.sH % 4 == 0 && v.sD % 4 == 0 && v.sN % 4 == 0;
  };
  if (!al8(o0)) return TEM_EUNSUPPORTED;
  Dev p{};
  p.in = i0.ptr; p.iN = (int)i0.sN; p.iD = (int)i0.sD; p.iH = (int)i0.sH; p.iW = (int)i0.sW;
  p.D = i0.D; p.H = i0.H; p.W = i0.W;
  p.w = a->w;
  p.out = o0.ptr; p.oN = (int)o0.sN; p.oD = (int)o0.sD; p.oH = (int)o0.sH; p.oW = (int)o0.sW;
  p.OD = o0.D; p.OH = o0.H; p.OW = o0.W;
  p.P = a->pd;
  p.slope = a->ep.slope; p.gate_slope = a->ep.gate_slope; p.bias = a->ep.bias;
  if (a->ep.gate.ptr) {
    const tem_view &g = a->ep.gate;
    if (g.N != o0.N || g.D != o0.D || g.H != o0.H || g.W != o0.W || g.C < o0.C) return TEM_ESHAPE;
    if (!fits32(g) || !al8(g)) return TEM_EUNSUPPORTED;
    p.gate = g.ptr; p.gN = (int)g.sN; p.gD = (int)g.sD; p.gH = (int)g.sH; p.gW = (int)g.sW;
  }
  const bool flip = a->w_layout == TEM_W_FLIP_CO_CI;
  const int N = i0.N;
  if (o0.C == 8) return flip ? run_mfma_h<8, true>(p, N, st, dry, name, name_len) : run_mfma_h<8, false>(p, N, st, dry, name, name_len);
  return flip ? run_mfma_h<16, true>(p, N, st, dry, name, name_len) : run_mfma_h<16, false>(p, N, st, dry, name, name_len);
}

}  // namespace stencil_c1

// Called by tem_conv (dispatch.hip) after the LDS/MFMA-tiled kernel declined.
int tem_conv_c1_try(const tem_conv_args *a, hipStream_t st, bool dry) { return stencil_c1::dispatch(a, st, dry); }

int tem_conv_c1_describe(const tem_conv_args *a, char *buf, int len) {
  stencil_c1::g_name = buf; stencil_c1::g_name_len = len;
  int rc = stencil_c1::dispatch(a, nullptr, true);
  stencil_c1::g_name = nullptr;
  return rc;
}

// bf16 mode (conv_bf16.hip tries this first for the one-input-channel layers)
int tem_conv_c1_bf16_try(const tem_conv_args *a, hipStream_t st, bool dry, char *name, int name_len) {
  return stencil_c1::dispatch_h(a, st, dry, name, name_len);
}
