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
// out[v][co] = sum_tap x[v + tap] * w[tap][co] is a [voxels x 27] x [27 x C_out] product.  v_mfma_f32_16x16x4_f32 with
// M = 16 consecutive voxels of a row and the 27 taps as the K dimension keeps every kernel tap in REGISTERS (the B
// fragments: one VGPR per k-step, loaded once per wave) -- the VALU form has to stream all 27 x C_out taps through the
// scalar path for every output voxel.  C_out == 8 would fill half of the 16 columns: the other half takes the NEXT
// output plane (columns = (plane, co)), K then spans the 4 input planes the plane pair touches (36 = 9 k-steps, 27 of
// the 36 products per column are real).  C_out == 16: K = 27 padded to 28 (7 k-steps).
// A workgroup loads its input patch (1 channel: a few KB) into LDS once, every wave gathers its A fragments from it
// (ds_read_b32, lane -> (voxel, tap)), and the 16x16 result tile is transposed through a private LDS patch so that
// each lane stores 16 contiguous bytes (1 KB per wave instruction), with bias / LeakyReLU / gate fused.
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct DevM {
  const float *in;
  int32_t iN, iD, iH, iW, D, H, W;
  float *out;
  int32_t oN, oD, oH, oW, OD, OH, OW;
  int32_t P;
  int32_t TXT, TY, nty, nzg;        // 16-voxel tiles per row, rows per patch, patches in y, plane groups in z
  uint32_t magicCols, magicTXT;
  float slope;
  const float *gate; int32_t gN, gD, gH, gW; float gate_slope;
  const float *bias;
  int32_t dbg;
};

template <int CO, bool FLIP>
__global__ __launch_bounds__(256) void c1_mfma_k(DevM p, const float *__restrict__ wgt) {
  static_assert(CO == 8 || CO == 16, "columns = (plane pair, 8 channels) or 16 channels");
  constexpr int NZ = CO == 8 ? 2 : 1;                     // output planes per tile
  constexpr int NP = NZ + 2;                              // input planes per patch
  constexpr int KS = CO == 8 ? 9 : 7;                     // k-steps of 4 taps
  constexpr int TPITCH = 20;                              // floats per row of the transpose patch (conflict-free)
  constexpr int PFM = 4;                                  // patch voxels per thread and plane (rows*cols <= 1024)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const int cols = p.TXT * 16 + 2, rows = p.TY + 2;
  const int plane = rows * cols;

  int b = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);     // z-neighbours share input planes: one XCD's L2
  const int zg = b % p.nzg; b /= p.nzg;
  const int typ = b % p.nty;
  const int n = b / p.nty;
  const int oy0 = typ * p.TY, oz0 = zg * NZ;

  // ---- B fragments: lane (n_col = m, k = 4s + kq)
  float B[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = 4 * s + kq;
    float v = 0.f;
    if (CO == 8) {
      const int zi = k / 9, rem = k - zi * 9;             // k = zi*9 + dy*3 + dx, zi in 0..3
      const int zo = m >> 3, dz = zi - zo;
      if (dz >= 0 && dz <= 2) {
        const int tap = dz * 9 + rem;
        v = wgt[(FLIP ? 26 - tap : tap) * 8 + (m & 7)];
      }
    } else {
      if (k < 27) v = wgt[(FLIP ? 26 - k : k) * 16 + m];
    }
    B[s] = v;
  }
  // ---- A gather offsets inside the patch: tap k -> (zi, dy, dx)
  int offk[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    int k = 4 * s + kq;
    if (k >= NP * 9) k = 0;                               // padded k-step (its B is zero): any finite value
    const int zi = k / 9, rem = k - zi * 9, dy = rem / 3, dx = rem - dy * 3;
    offk[s] = zi * plane + dy * cols + dx + m;
  }

  // ---- load the patch: NP planes x rows x cols, zeros outside the input (padding / volume border).  All of a
  // thread's loads are issued before the first LDS write: one memory round trip per workgroup, not one per element.
  {
    const int rc = rows * cols;                            // host: rc <= PFM * 256
    float pf[NP][PFM];
#pragma unroll
    for (int i = 0; i < PFM; ++i) {
      const int r2 = tid + i * 256;
      const int r = (int)__umulhi((uint32_t)r2, p.magicCols), cx = r2 - r * cols;
      const int iy = oy0 - p.P + r, ix = cx - p.P;
      const bool okxy = r2 < rc && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const float *src = p.in + (n * p.iN + iy * p.iH + ix * p.iW);
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
        const int iz = oz0 - p.P + pl;
        pf[pl][i] = (okxy && (unsigned)iz < (unsigned)p.D && !(p.dbg & 4)) ? src[iz * p.iD] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < PFM; ++i) {
      const int r2 = tid + i * 256;
      if (r2 < rc) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) lds[pl * plane + r2] = pf[pl][i];
      }
    }
  }
  __syncthreads();

  float *tp = lds + ((NP * plane + 3) & ~3) + wave * (16 * TPITCH);
  const int ti = lane >> 2, tcq = lane & 3;               // transposed role: voxel of the tile, channel quad
  const int zo = CO == 8 ? (tcq >> 1) : 0, co0 = CO == 8 ? 4 * (tcq & 1) : 4 * tcq;
  const int ntiles = (p.dbg & 2) ? 0 : p.TXT * p.TY;
  // two tiles per iteration: their MFMA chains (each a dependent chain on its own accumulator) and LDS round
  // trips interleave, so one wave keeps the matrix pipe fed while the other tile's operands are in flight
  auto tile_src = [&](int t, int &trow, int &tcol) -> const float * {
    trow = p.TXT == 1 ? t : (int)__umulhi((uint32_t)t, p.magicTXT);   // magic(1) overflows
    tcol = t - trow * p.TXT;
    return lds + trow * cols + tcol * 16;
  };
  auto epilogue = [&](const f32x4 &acc, int trow, int tcol) {
    // C/D map: col = lane&15, row = 4*(lane>>4)+reg  ->  lane (voxel ti, channels 4*tcq..)
#pragma unroll
    for (int q = 0; q < 4; ++q) tp[(kq * 4 + q) * TPITCH + m] = acc[q];
    __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): this wave's own LDS writes have landed
    const float4 v4 = *reinterpret_cast<const float4 *>(tp + ti * TPITCH + tcq * 4);
    const int ox = tcol * 16 + ti, oy = oy0 + trow, oz = oz0 + zo;
    if (ox < p.OW && oy < p.OH && oz < p.OD && !(p.dbg & 1)) {
      float v[4] = {v4.x, v4.y, v4.z, v4.w};
      if (p.bias) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] += p.bias[co0 + c];
      }
      if (p.gate) {
        const float4 g4 = *reinterpret_cast<const float4 *>(p.gate + (n * p.gN + oz * p.gD + oy * p.gH + ox * p.gW + co0));
        v[0] = g4.x > 0.f ? v[0] : p.gate_slope * v[0];
        v[1] = g4.y > 0.f ? v[1] : p.gate_slope * v[1];
        v[2] = g4.z > 0.f ? v[2] : p.gate_slope * v[2];
        v[3] = g4.w > 0.f ? v[3] : p.gate_slope * v[3];
      }
      if (p.slope != 1.f) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = v[c] > 0.f ? v[c] : p.slope * v[c];
      }
      *reinterpret_cast<float4 *>(p.out + (n * p.oN + oz * p.oD + oy * p.oH + ox * p.oW + co0)) =
          make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  for (int t = wave; t < ntiles; t += 8) {                // wave-uniform
    const int t2 = t + 4;
    const bool two = t2 < ntiles;
    int r0, c0, r1, c1;
    const float *s0 = tile_src(t, r0, c0);
    const float *s1 = tile_src(two ? t2 : t, r1, c1);
    float a0[KS], a1[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) { a0[s] = s0[offk[s]]; a1[s] = s1[offk[s]]; }
    __builtin_amdgcn_sched_barrier(0);                    // (all 2 KS reads in flight before the first MFMA: left alone, the
                                                          // scheduler sinks each read next to its MFMA -- ds_read, lgkmcnt(0), MFMA)
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], B[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], B[s], acc1, 0, 0, 0);
    }
    epilogue(acc0, r0, c0);
    if (two) epilogue(acc1, r1, c1);
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
  p.in = q.in; p.iN = q.iN; p.iD = q.iD; p.iH = q.iH; p.iW = q.iW; p.D = q.D; p.H = q.H; p.W = q.W;
  p.out = q.out; p.oN = q.oN; p.oD = q.oD; p.oH = q.oH; p.oW = q.oW; p.OD = q.OD; p.OH = q.OH; p.OW = q.OW;
  p.P = q.P; p.slope = q.slope; p.gate = q.gate; p.gN = q.gN; p.gD = q.gD; p.gH = q.gH; p.gW = q.gW;
  p.gate_slope = q.gate_slope; p.bias = q.bias; p.dbg = q.dbg;
  p.TXT = (p.OW + 15) / 16;
  // rows per patch: ~32 tiles per workgroup (8 per wave) keeps the patch load + barrier a small share
  p.TY = 32 / p.TXT;
  if (p.TY < 1) p.TY = 1;
  if (p.TY > p.OH) p.TY = p.OH;
  while (p.TY > 1 && (p.TY + 2) * (p.TXT * 16 + 2) > 4 * 256) --p.TY;      // the loader holds 4 voxels per thread and plane
  if ((p.TY + 2) * (p.TXT * 16 + 2) > 4 * 256) return TEM_EUNSUPPORTED;
  p.nty = (p.OH + p.TY - 1) / p.TY;
  p.nzg = (p.OD + NZ - 1) / NZ;
  p.magicCols = magic_for(p.TXT * 16 + 2);
  p.magicTXT = magic_for(p.TXT);
  const size_t patch = (size_t)NP * (p.TY + 2) * (p.TXT * 16 + 2);
  const size_t lds_bytes = (((patch + 3) & ~(size_t)3) + 4 * 16 * 20) * 4;
  if (lds_bytes > 64 * 1024) return TEM_EUNSUPPORTED;
  if (dry) {
    if (g_name) snprintf(g_name, g_name_len, "c1_mfma_k<%d, %s>", CO, FLIP ? "true" : "false");
    return TEM_OK;
  }
  const int nblocks = N * p.nty * p.nzg;
  hipLaunchKernelGGL((c1_mfma_k<CO, FLIP>), dim3((unsigned)nblocks), dim3(256), lds_bytes, st, p, q.w);
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
  if (use_mfma && CI == 1 && CO == 8)      // g.c0, d.d1a forward
    return flip ? run_mfma<8, true>(p, N, st, dry) : run_mfma<8, false>(p, N, st, dry);
  if (use_mfma && CI == 1 && CO == 16)     // input-gradient of g.f2 (gated by f1)
    return flip ? run_mfma<16, true>(p, N, st, dry) : run_mfma<16, false>(p, N, st, dry);
  C1_CASE(1, 8, 3)
  C1_CASE(1, 16, 3)
  C1_CASE(16, 1, 6)      // g.f2 forward
  C1_CASE(8, 1, 3)       // input-gradients of g.c0 / d.d1a (the cycle and adversarial paths need dx)
#undef C1_CASE
  return TEM_EUNSUPPORTED;
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
