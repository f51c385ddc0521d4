// bww_lds.hip -- kernel gradient of the 3-D convolutions, LDS-tiled on the fp32 matrix cores.
//
//   dW[(tap,ci)][co] = sum over output voxels o of  X[o*S + tap - P][ci] * G[o][co]
//
// One workgroup owns a (n, R output rows in y, a run of output planes in z) column of the
// volume and ALL rows of dW: the K^3*CI x CO accumulator tiles (16x16, v_mfma_f32_16x16x4_f32)
// are dealt round-robin to its waves and live in registers for the whole run, so there is no
// cross-wave reduction and each workgroup writes exactly one partial slab.
//
// Data movement per z-step: the workgroup marches along z holding a ring of K input planes
// (each (R-1)*S+K rows of the full x extent, channels-last with a CI+2 voxel pitch that
// spreads the 16-voxel and 16-channel fragment reads over the LDS banks) plus the R gradient
// rows of the current output plane.  Only the S newest planes and the next gradient rows are
// fetched per step, as coalesced 16-byte global loads issued BEFORE the step's MFMA work and
// written to LDS after it (register staging), so HBM/L2 latency hides under the matrix work and
// every input byte is read ~(R+K-S)/R times per workgroup column instead of K^3 times.
// Both MFMA fragments are plain ds_read_b32: A = [4 voxels][16 (tap,ci) rows], B = [4 voxels]
// [16 co] -- the channels-last layout as it is.
//
// C_out == 8 (XSH): an 8-wide gradient would leave half of every 16-column MFMA tile empty.  Columns
// 8..15 are fed the SAME gradient shifted by one output voxel in x,
//   D[(tz,ty,tx,ci)][8 + co] = sum_v X[v*S + tx][ci] * G[v - 1][co] = dW[(tz,ty,tx + S,ci)][co],
// so a tile row delivers two x-taps at once and only the x-taps tx < K - S need rows of their own:
// 9 instead of 14 tiles for k3 s1 (8 -> 8), 16 instead of 32 for k4 s2.
#include "tem_common.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace bwwlds {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Dev {
  const float *in0, *in1;
  int64_t i0N, i0D, i0H, i0W, i1N, i1D, i1H, i1W;
  int32_t C0;                    // channels taken from in0 (rest from in1)
  int32_t N, D, H, W;            // input extents
  const float *g;
  int64_t gN, gD, gH, gW;
  int32_t OD, OH, OW;
  int32_t P;                     // zero padding (all three axes)
  float *slabs;
  int64_t slab_stride;
  int32_t R, nych, zsegs, zper;  // rows per y-chunk, #y-chunks, #z-segments, z-steps per segment
  int32_t YR, WX, WXp, OWp;      // rows per ring slot, voxels loaded per X row, allocated voxels, padded OW
  uint32_t magicX, magicG;       // ceil(2^32 / chunks-per-row) for the loaders' index split
  int32_t chunksX, chunksG;      // chunks per X row / per G row
};


template <int CI, int CO, int K, int S, int NW, bool XSH = false>
struct Cfg {
  static_assert(!XSH || CO == 8, "the x-shift form pairs two 8-channel gradients in one n-tile");
  static constexpr int CIP = CI == 1 ? 1 : CI + 2;
  static constexpr int COP = CO + 2;
  static constexpr int KXR = XSH ? K - S : K;           // x-taps that own tile rows
  static constexpr int ROWS = K * K * KXR * CI;
  static constexpr int MTILES = (ROWS + 15) / 16;
  static constexpr int NT = (CO + 15) / 16;
  static constexpr int T = MTILES * NT;
  // When there are fewer tiles than waves (C_in == 1: 2 tiles) the waves split the voxel rows instead:
  // KSPL waves own the same tiles for interleaved rows and are summed through LDS at the end.
  static constexpr int KSPL = (T * 2 <= NW) ? NW / ((T + NT - 1) / NT * NT) : 1;
  static constexpr int WG = NW / KSPL;                    // wave groups = distinct tile owners
  static constexpr int TPW = (T + WG - 1) / WG;
  static constexpr int CHX = CI % 4 == 0 ? 4 : 1;     // floats per loader chunk
  static constexpr int CHG = CO % 4 == 0 ? 4 : 1;
  // one tile more than waves (k3 s1 x-shift form: 9 tiles, 8 waves): every wave owns one tile, the last tile
  // is summed by wave r for output row r and reduced across the waves at the end
  static constexpr bool SHR = XSH && KSPL == 1 && T == WG + 1;
  static_assert(NW % NT == 0 && WG % NT == 0 && NW % KSPL == 0, "a wave keeps one n-tile");
};

template <int CH>
struct Chunk { float v[CH]; };

template <int CH>
__device__ __forceinline__ Chunk<CH> gload(const float *p, bool ok) {
  Chunk<CH> c;
  if constexpr (CH == 4) {
    float4 t = ok ? *reinterpret_cast<const float4 *>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
    c.v[0] = t.x; c.v[1] = t.y; c.v[2] = t.z; c.v[3] = t.w;
  } else {
    c.v[0] = ok ? *p : 0.f;
  }
  return c;
}

template <int CH>
__device__ __forceinline__ void lstore(float *p, const Chunk<CH> &c) {
  if constexpr (CH == 4) {   // voxel pitch is even => 8-byte aligned
    *reinterpret_cast<float2 *>(p) = make_float2(c.v[0], c.v[1]);
    *reinterpret_cast<float2 *>(p + 2) = make_float2(c.v[2], c.v[3]);
  } else {
    *p = c.v[0];
  }
}

// PFX / PFG: register-staged loader chunks per thread per z-step (X planes / gradient rows)
template <int CI, int CO, int K, int S, int NW, int MAXPFX, int MAXPFG, bool XSH = false>
__global__ __launch_bounds__(NW * 64) void bww_lds_k(Dev p) {
  using C = Cfg<CI, CO, K, S, NW, XSH>;
  constexpr int CIP = C::CIP, COP = C::COP, NT = C::NT, TPW = C::TPW, CHX = C::CHX, CHG = C::CHG;
  constexpr int NTHR = NW * 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int rowpitch = p.WXp * CIP;
  const int slotpitch = p.YR * rowpitch;
  float *Xs = lds;
  float *Gs = lds + K * slotpitch + (XSH ? COP : 0);     // XSH: one zero voxel in front of row 0 (G[-1])
  const int growpitch = p.OWp * COP;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps tile bookkeeping on the scalar unit
  const int m = lane & 15, kq = lane >> 4;

  // ---- which column of the volume is ours
  int seg = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);   // neighbouring columns share an XCD's L2
  const int zseg = seg % p.zsegs; seg /= p.zsegs;
  const int ych = seg % p.nych;
  const int n = seg / p.nych;
  const int oy0 = ych * p.R;
  const int oz0 = zseg * p.zper;
  const int oz1 = min(p.OD, oz0 + p.zper);
  const int nsteps = oz1 - oz0;

  // ---- zero the whole LDS image once (slack voxels and never-written rows must stay finite)
  {
    const int total4 = (K * slotpitch + p.R * growpitch + (XSH ? COP : 0) + 3) / 4;      // allocation is rounded up to 16 B
    for (int i = tid; i < total4; i += NTHR) reinterpret_cast<float4 *>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();

  // ---- loaders.  X chunk id -> (plane-in-step, row, chunk-in-row); G chunk id -> (row, chunk)
  auto load_x = [&](Chunk<CHX> (&pf)[MAXPFX], int iz_first, int nplanes) {
    const int per_plane = p.YR * p.chunksX;
    const int total = nplanes * per_plane;
#pragma unroll
    for (int i = 0; i < MAXPFX; ++i) {
      int id = tid + i * NTHR;
      asm volatile("" : "+v"(id));                         // recompute per step: keeps index math out of live registers
      bool ok = id < total;
      int rowid = __umulhi((uint32_t)id, p.magicX);       // id / chunksX
      int pos = id - rowid * p.chunksX;
      int pl = rowid >= p.YR ? 1 : 0;                      // a batch is at most S <= 2 planes
      int yr = rowid - pl * p.YR;
      int vox = CHX == 4 ? pos / (CI / 4 > 0 ? CI / 4 : 1) : pos;
      int c = CHX == 4 ? (pos - vox * (CI / 4 > 0 ? CI / 4 : 1)) * 4 : 0;
      int iz = iz_first + pl, iy = oy0 * S - p.P + yr, ix = vox - p.P;
      ok = ok && iz >= 0 && iz < p.D && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const float *src = c < p.C0 ? p.in0 + n * p.i0N + iz * p.i0D + iy * p.i0H + ix * p.i0W + c
                                  : p.in1 + n * p.i1N + iz * p.i1D + iy * p.i1H + ix * p.i1W + (c - p.C0);
      pf[i] = gload<CHX>(src, ok);
    }
  };
  auto store_x = [&](const Chunk<CHX> (&pf)[MAXPFX], int iz_first, int nplanes) {
    const int per_plane = p.YR * p.chunksX;
    const int total = nplanes * per_plane;
#pragma unroll
    for (int i = 0; i < MAXPFX; ++i) {
      int id = tid + i * NTHR;
      asm volatile("" : "+v"(id));
      if (id < total) {
        int rowid = __umulhi((uint32_t)id, p.magicX);
        int pos = id - rowid * p.chunksX;
        int pl = rowid >= p.YR ? 1 : 0;
        int yr = rowid - pl * p.YR;
        int vox = CHX == 4 ? pos / (CI / 4 > 0 ? CI / 4 : 1) : pos;
        int c = CHX == 4 ? (pos - vox * (CI / 4 > 0 ? CI / 4 : 1)) * 4 : 0;
        int slot = ((iz_first + pl) % K + K) % K;
        lstore<CHX>(Xs + slot * slotpitch + yr * rowpitch + vox * CIP + c, pf[i]);
      }
    }
  };
  auto load_g = [&](Chunk<CHG> (&pf)[MAXPFG], int oz) {
    const int total = p.R * p.chunksG;
#pragma unroll
    for (int i = 0; i < MAXPFG; ++i) {
      int id = tid + i * NTHR;
      asm volatile("" : "+v"(id));
      bool ok = id < total;
      int r = __umulhi((uint32_t)id, p.magicG);
      int pos = id - r * p.chunksG;
      int vox = CHG == 4 ? pos / (CO / 4 > 0 ? CO / 4 : 1) : pos;
      int c = CHG == 4 ? (pos - vox * (CO / 4 > 0 ? CO / 4 : 1)) * 4 : 0;
      int oy = oy0 + r;
      ok = ok && oy < p.OH && vox < p.OW;
      pf[i] = gload<CHG>(p.g + n * p.gN + oz * p.gD + oy * p.gH + vox * p.gW + c, ok);
    }
  };
  auto store_g = [&](const Chunk<CHG> (&pf)[MAXPFG]) {
    const int total = p.R * p.chunksG;
#pragma unroll
    for (int i = 0; i < MAXPFG; ++i) {
      int id = tid + i * NTHR;
      asm volatile("" : "+v"(id));
      if (id < total) {
        int r = __umulhi((uint32_t)id, p.magicG);
        int pos = id - r * p.chunksG;
        int vox = CHG == 4 ? pos / (CO / 4 > 0 ? CO / 4 : 1) : pos;
        int c = CHG == 4 ? (pos - vox * (CO / 4 > 0 ? CO / 4 : 1)) * 4 : 0;
        lstore<CHG>(Gs + r * growpitch + vox * COP + c, pf[i]);
      }
    }
  };

  // ---- this wave's accumulator tiles: t = wgid + j*WG, n-tile fixed per wave; kshare = which rows it sums
  constexpr int WG = C::WG, KSPL = C::KSPL;
  const int wgid = wave % WG, kshare = wave / WG;
  const int nt = wgid % NT;
  int aconst[TPW], adz[TPW];
  f32x4 acc[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    int t = min(wgid + j * WG, C::T - 1);                  // tiles past T: recomputed copy, never stored
    int mt = t / NT;
    int row = min(mt * 16 + m, C::ROWS - 1);              // rows >= ROWS are never stored
    int tap = row / CI, ci = row - tap * CI;
    int dx = tap % C::KXR, dy = (tap / C::KXR) % K;
    adz[j] = tap / (C::KXR * K);
    aconst[j] = dy * rowpitch + dx * CIP + ci + kq * S * CIP;
    acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool bvalid = XSH || (nt * 16 + m) < CO;
  // XSH: columns 8..15 read the gradient one voxel to the left (zero in front of x = 0)
  const int boff = XSH ? (kq - (m >> 3)) * COP + (m & 7) : kq * COP + nt * 16 + (bvalid ? m : 0);

  Chunk<CHX> pfx[MAXPFX];
  Chunk<CHG> pfg[MAXPFG];

  // ---- prologue: all K planes of the first step, and its gradient rows
  if (nsteps > 0) {
    const int iz0 = oz0 * S - p.P;
    if constexpr (K > S) {
      // K planes may not fit one register batch: load them S at a time
      for (int pl = 0; pl < K; pl += S) {
        int np = min(S, K - pl);
        load_x(pfx, iz0 + pl, np);
        store_x(pfx, iz0 + pl, np);
      }
    } else {
      load_x(pfx, iz0, K);
      store_x(pfx, iz0, K);
    }
    load_g(pfg, oz0);
    store_g(pfg);
  }
  __syncthreads();

  for (int step = 0; step < nsteps; ++step) {
    const int oz = oz0 + step;
    const int izb = oz * S - p.P;
    const bool more = step + 1 < nsteps;
    if (more) {                                            // issue next step's HBM/L2 reads now
      load_x(pfx, izb + K, S);
      load_g(pfg, oz + 1);
    }
    int abase[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      int slot = ((izb + adz[j]) % K + K) % K;
      abase[j] = slot * slotpitch + aconst[j];
    }
    // k-steps of 4 voxels, two per iteration with ping-pong fragment registers: the ds_reads of
    // the next k-step are issued BEFORE the current k-step's MFMAs (sched_barrier pins that order),
    // so their latency hides under TPW back-to-back MFMAs.  OWp is a multiple of 8: nk is even.
    const int nk = p.OWp >> 2;
    auto row_pass = [&](auto ntl_tag, int r) {             // NTL = accumulator tiles this wave feeds on row r
      constexpr int NTL = decltype(ntl_tag)::value;
      const float *xr = Xs + r * S * rowpitch;
      const float *gr = Gs + r * growpitch + boff;
      float a0[NTL], a1[NTL], b0, b1;
      b0 = gr[0];
#pragma unroll
      for (int j = 0; j < NTL; ++j) a0[j] = xr[abase[j]];
      for (int k = 0; k < nk; k += 2) {
        b1 = gr[(k + 1) * 4 * COP];
#pragma unroll
        for (int j = 0; j < NTL; ++j) a1[j] = xr[abase[j] + (k + 1) * 4 * S * CIP];
        __builtin_amdgcn_sched_barrier(0);
        {
          const float b = bvalid ? b0 : 0.f;
#pragma unroll
          for (int j = 0; j < NTL; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b, acc[j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int k2 = k + 2 < nk ? k + 2 : k;             // past the end: re-read, never used
        b0 = gr[k2 * 4 * COP];
#pragma unroll
        for (int j = 0; j < NTL; ++j) a0[j] = xr[abase[j] + k2 * 4 * S * CIP];
        __builtin_amdgcn_sched_barrier(0);
        {
          const float b = bvalid ? b1 : 0.f;
#pragma unroll
          for (int j = 0; j < NTL; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b, acc[j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    for (int r = kshare; r < p.R; r += KSPL) {
      if constexpr (C::SHR) {
        if ((r % NW) == wave) row_pass(std::integral_constant<int, 2>{}, r);    // wave-uniform
        else row_pass(std::integral_constant<int, 1>{}, r);
      } else {
        row_pass(std::integral_constant<int, TPW>{}, r);
      }
    }
    __syncthreads();                                       // everyone is done with the oldest planes / G rows
    if (more) {
      store_x(pfx, izb + K, S);
      store_g(pfg);
    }
    __syncthreads();
  }

  // ---- one partial slab per workgroup, each tile written by the wave group that owns it
  if constexpr (KSPL > 1) {                                // sum the KSPL row shares through LDS (the ring is dead now)
    float *red = lds;
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) red[((wave * TPW + j) * 4 + q) * 64 + lane] = acc[j][q];
    __syncthreads();
    if (kshare == 0) {
#pragma unroll
      for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v = acc[j][q];
          for (int ks = 1; ks < KSPL; ++ks) v += red[(((ks * WG + wgid) * TPW + j) * 4 + q) * 64 + lane];
          acc[j][q] = v;
        }
    }
  }
  if constexpr (C::SHR) {                                  // the shared last tile: sum the waves' row shares
    float *red = lds;
#pragma unroll
    for (int q = 0; q < 4; ++q) red[(wave * 4 + q) * 64 + lane] = acc[1][q];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v = 0.f;
        for (int w = 0; w < NW; ++w) v += red[(w * 4 + q) * 64 + lane];
        acc[1][q] = v;
      }
    }
  }
  float *slab = p.slabs + (int64_t)blockIdx.x * p.slab_stride;
  if (kshare == 0) {
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      int t = wgid + j * WG;
      if (t < C::T) {
        int mt = t / NT;
        int co = nt * 16 + m;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int row = mt * 16 + kq * 4 + q;                  // C/D map: row = 4*(lane>>4)+reg, col = lane&15
          if constexpr (XSH) {
            // row = ((tz*K + ty)*KXR + txr)*CI + ci; column m = shift*8 + co  ->  tap tx = txr + shift*S
            const int tap = row / CI, ci = row - tap * CI;
            const int txr = tap % C::KXR, tzy = tap / C::KXR, sh = m >> 3;
            const int tx = txr + sh * S;
            const bool dup = sh == 0 && txr >= S;           // also delivered as (txr - S, shifted): keep that one
            if (row < C::ROWS && !dup) slab[((int64_t)(tzy * K + tx) * CI + ci) * CO + (m & 7)] = acc[j][q];
          } else {
            if (row < C::ROWS && co < CO) slab[(int64_t)row * CO + co] = acc[j][q];
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ host
constexpr int LDS_BUDGET = 150 * 1024;
static thread_local char *g_name = nullptr;
static thread_local int g_name_len = 0;
constexpr int TARGET_BLOCKS = 256;   // one workgroup per CU: swept 128..1024 on the 132^3 step (TEM_BWW_BLOCKS), 256 is fastest

static uint32_t magic_for(int d) { return (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d); }

// Fills the plan fields of `p`; returns false when the geometry does not fit this kernel.
template <int CI, int CO, int K, int S, int NW, int MAXPFX, int MAXPFG, bool XSH>
bool plan(Dev &p, int max_slabs, size_t &lds_bytes, int &nblocks) {
  using C = Cfg<CI, CO, K, S, NW, XSH>;
  const int NTHR = NW * 64;
  p.WX = (p.OW - 1) * S + K;
  p.OWp = (p.OW + (XSH ? 1 : 0) + 7) & ~7; // even number of 4-voxel k-steps (XSH: one more voxel for the shifted columns)
  p.WXp = p.WX + (p.OWp - p.OW) * S + 1;   // slack voxels the padded k-steps read (kept zero)
  p.chunksX = C::CHX == 4 ? p.WX * (CI / 4) : p.WX;
  p.chunksG = C::CHG == 4 ? p.OW * (CO / 4) : p.OW;
  p.magicX = magic_for(p.chunksX);
  p.magicG = magic_for(p.chunksG);
  int R = p.OH < 8 ? p.OH : 8;
  for (; R >= 1; --R) {
    int YR = (R - 1) * S + K;
    size_t bytes = ((size_t)K * YR * p.WXp * C::CIP + (size_t)R * p.OWp * C::COP + (XSH ? C::COP : 0)) * 4;
    bool fits = bytes <= (size_t)LDS_BUDGET && (size_t)S * YR * p.chunksX <= (size_t)MAXPFX * NTHR &&
                (size_t)R * p.chunksG <= (size_t)MAXPFG * NTHR;
    if (fits) {
      p.R = R; p.YR = YR;
      size_t red = (C::KSPL > 1 || C::SHR) ? (size_t)NW * C::TPW * 4 * 64 * 4 : 0;   // the row-share reduction reuses the ring
      lds_bytes = ((bytes > red ? bytes : red) + 15) & ~(size_t)15;
      break;
    }
  }
  if (R < 1) return false;
  p.nych = (p.OH + p.R - 1) / p.R;
  int cols = p.N * p.nych;
  static int target = -1;
  if (target < 0) target = tem_env_int("TEM_BWW_BLOCKS", TARGET_BLOCKS);
  int want = max_slabs < target ? max_slabs : target;
  int zsegs = want / cols;
  if (zsegs < 1) zsegs = 1;
  if (zsegs > p.OD) zsegs = p.OD;
  p.zper = (p.OD + zsegs - 1) / zsegs;
  p.zsegs = (p.OD + p.zper - 1) / p.zper;
  nblocks = cols * p.zsegs;
  return nblocks <= max_slabs;
}

template <int CI, int CO, int K, int S, int NW, int MAXPFX, int MAXPFG, bool XSH = false>
int run(Dev &p, int max_slabs, hipStream_t st, bool dry, int *nslab_out) {
  size_t lds_bytes = 0;
  int nblocks = 0;
  if (!plan<CI, CO, K, S, NW, MAXPFX, MAXPFG, XSH>(p, max_slabs, lds_bytes, nblocks)) return TEM_EUNSUPPORTED;
  if (nslab_out) *nslab_out = nblocks;
  if (dry) {
    if (g_name)
      snprintf(g_name, g_name_len, "bww_lds_k<%d, %d, %d, %d, %d, %d, %d, %s>", CI, CO, K, S, NW, MAXPFX, MAXPFG,
               XSH ? "true" : "false");
    return TEM_OK;
  }
  auto kern = bww_lds_k<CI, CO, K, S, NW, MAXPFX, MAXPFG, XSH>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(NW * 64), lds_bytes, st, p);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

#define BWW_CASE(ci, co, k, s, nw, pfx, pfg) \
  if (CI == ci && CO == co && K == k && S == s) return run<ci, co, k, s, nw, pfx, pfg>(p, max_slabs, st, dry, nslab_out);
#define BWW_CASE_XSH(ci, co, k, s, nw, pfx, pfg) \
  if (CI == ci && CO == co && K == k && S == s && xsh) return run<ci, co, k, s, nw, pfx, pfg, true>(p, max_slabs, st, dry, nslab_out);

// Dispatch; `dry` only computes the number of slabs the launch would write.
int dispatch(const tem_bww_args *a, hipStream_t st, bool dry, int *nslab_out) {
  const tem_view &i0 = a->in0, &g = a->dout;
  const bool cube_k = a->kd == a->kh && a->kh == a->kw, cube_s = a->sd == a->sh && a->sh == a->sw;
  const bool cube_p = a->pd == a->ph && a->ph == a->pw;
  if (!cube_k || !cube_s || !cube_p || a->kd < 3) return TEM_EUNSUPPORTED;
  Dev p{};
  p.in0 = i0.ptr; p.i0N = i0.sN; p.i0D = i0.sD; p.i0H = i0.sH; p.i0W = i0.sW;
  p.C0 = i0.C;
  p.in1 = i0.ptr; p.i1N = i0.sN; p.i1D = i0.sD; p.i1H = i0.sH; p.i1W = i0.sW;
  int CI = i0.C;
  if (a->in1.ptr) {
    const tem_view &i1 = a->in1;
    p.in1 = i1.ptr; p.i1N = i1.sN; p.i1D = i1.sD; p.i1H = i1.sH; p.i1W = i1.sW;
    CI += i1.C;
    if (i0.C % 4 || i1.C % 4) return TEM_EUNSUPPORTED;
  }
  p.N = i0.N; p.D = i0.D; p.H = i0.H; p.W = i0.W;
  p.g = g.ptr; p.gN = g.sN; p.gD = g.sD; p.gH = g.sH; p.gW = g.sW;
  p.OD = g.D; p.OH = g.H; p.OW = g.W;
  p.P = a->pd;
  p.slabs = a->slabs;
  const int CO = g.C, K = a->kd, S = a->sd;
  p.slab_stride = a->slab_stride ? a->slab_stride : (int64_t)K * K * K * CI * CO;
  const int max_slabs = a->nslab;
  // 16-byte loader chunks need 16-byte aligned channel vectors
  auto aligned = [](const tem_view &v) {
    return v.C % 4 != 0 || (((uintptr_t)v.ptr & 15) == 0 && v.sW % 4 == 0 && v.sH % 4 == 0 && v.sD % 4 == 0 && v.sN % 4 == 0);
  };
  if (!aligned(i0) || (a->in1.ptr && !aligned(a->in1)) || !aligned(g)) return TEM_EUNSUPPORTED;
  //        CI  CO  K  S  waves  X-chunks  G-chunks     (tiles per wave = ceil(K^3*CI/16 * ceil(CO/16) / waves))
  static int xsh = -1;
  if (xsh < 0) xsh = tem_env_int("TEM_BWW_XSH", 1);
  BWW_CASE_XSH(8, 8, 3, 1, 8, 4, 3)  // g.d1a: 9 tiles in the x-shift form
  BWW_CASE_XSH(8, 8, 4, 2, 8, 6, 2)  // g.d1b / d.d1b: 16 tiles
  BWW_CASE(1, 8, 3, 1, 8, 3, 4)      // g.c0 / d.d1a: 2 tiles x 4 row shares, HBM-bound on the gradient stream
  BWW_CASE(8, 8, 3, 1, 8, 4, 3)      // g.d1a: 14 tiles
  BWW_CASE(8, 16, 3, 1, 8, 4, 3)     // g.d2a / d.hack
  BWW_CASE(16, 16, 3, 1, 8, 4, 3)    // g.f1: 27 tiles, 4 per wave (two waves per SIMD hide LDS latency)
  BWW_CASE(16, 32, 3, 1, 8, 4, 3)    // g.u2a / d.d2a: 54 tiles
  BWW_CASE(32, 32, 3, 1, 8, 5, 3)    // g.mid / d.d3a: 108 tiles, 14 per wave
  BWW_CASE(32, 16, 3, 1, 8, 5, 3)    // g.u1a: 54 tiles
  BWW_CASE(16, 1, 3, 1, 4, 8, 5)     // (kept for callers that do not use the swapped C_out = 1 form)
  BWW_CASE(1, 16, 3, 1, 8, 3, 4)     // g.f2 swapped: X := dy (1 ch), G := f1 (16 ch), pad 2 -> dW with flipped taps
  BWW_CASE(8, 8, 4, 2, 8, 6, 2)      // g.d1b / d.d1b: 32 tiles
  BWW_CASE(16, 16, 4, 2, 8, 6, 3)    // g.d2b: 64 tiles
  BWW_CASE(8, 16, 4, 2, 8, 6, 2)     // g.u1b (transposed conv: roles of input and gradient swapped)
  BWW_CASE(16, 32, 4, 2, 4, 12, 4)   // g.u2b: 128 tiles, 32 per wave (1 wave per SIMD: 512 registers)
  // (32,32,k4,s2) -- the discriminator's two deep stride-2 layers, 1 GFLOP together -- would need 256
  // accumulator tiles per workgroup; they stay on the global-load kernel in conv_bww.hip.
  return TEM_EUNSUPPORTED;
}

}  // namespace bwwlds

// Called by tem_conv_bwd_weight (conv_bww.hip) before it falls back to the global-load kernel.
int tem_bww_lds_try(const tem_bww_args *a, hipStream_t st, bool dry, int *nslab_out) {
  return bwwlds::dispatch(a, st, dry, nslab_out);
}

int tem_bww_lds_describe(const tem_bww_args *a, char *buf, int len) {
  bwwlds::g_name = buf; bwwlds::g_name_len = len;
  int n = 0;
  int rc = bwwlds::dispatch(a, nullptr, true, &n);
  bwwlds::g_name = nullptr;
  return rc == TEM_OK && n == a->nslab ? TEM_OK : TEM_EUNSUPPORTED;
}
