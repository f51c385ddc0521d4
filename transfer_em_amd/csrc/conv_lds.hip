// conv_lds.hip -- forward / input-gradient convolution of the 3-D layers, LDS-tiled on the
// fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 fmaf chains, no reduced precision).
//
//   out[o][co] = epilogue( sum_{tap,ci} X[o*S + tap - P][ci] * W(tap,ci,co) )
//
// GEMM view per tap: D[16 voxels][16 co] += A[16 voxels][4 ci] * B[4 ci][16 co].
//   A comes from an LDS image of the input: channels-last rows with a CI+2 voxel pitch, which
//     makes the 16-voxel x 2-channel footprint of a ds_read_b32 half-wave hit 32 distinct banks;
//   B (the kernel taps) is read straight from HBM/L2 into registers, one tap ahead of its use
//     ([4 ci][16 co] is one contiguous 256-byte run of the Keras kernel layout);
//   D accumulates in registers and leaves through the fused epilogue (bias, skip-gradient add,
//     LeakyReLU / LeakyReLU-gradient gate, Philox dropout) as 64-byte channel runs.
//
// Data movement: a workgroup owns (n, R output rows, a run of output planes) and marches along
// z with a ring of K input planes in LDS; per step only the S new planes are fetched (coalesced
// 16-byte global loads issued before the step's MFMA work, written to LDS after it), so an input
// voxel is read from HBM/L2 ~(R+K-S)/R times instead of K^3 times, and never re-read for the
// other output channels.  For stride 1 the R x W output patch is linearised (v = r*WP + x) so
// that 16-voxel MFMA tiles run across row ends: only the K-1 halo columns per row are wasted.
// Concat inputs (generator.py:74-86) are gathered by the loader; split outputs are routed by the
// epilogue; zero padding / cropping is the loader's bounds check.
//
// Two waves per SIMD (512-thread workgroups): one wave's loader / epilogue VALU work overlaps the
// other's MFMA stream.  All index arithmetic is 32-bit (the host checks that every view spans
// fewer than 2^31 elements).
#include "tem_common.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace convlds {

// Steady-state loader: true = every wave fetches its share of the next planes (the chunk count per thread and
// the descriptor registers halve); false = only the upper half of the waves loads.
constexpr bool LOADER_ALL = true;
constexpr int pfx_of(int half_loader_chunks) { return LOADER_ALL ? (half_loader_chunks + 1) / 2 : half_loader_chunks; }

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Ep32 {                    // epilogue with 32-bit strides
  const float *bias;
  float slope;
  const float *gate; int32_t gN, gD, gH, gW; float gate_slope;
  const float *add;  int32_t aN, aD, aH, aW, aoz, aoy, aox, aDd, aHh, aWw;
  int32_t dropout;
  DropoutStream ds;
  const uint32_t *step_dev;
  int32_t doz, doy, dox, dD, dH, dW;
  const uint8_t *keep_mask;      // keep_mode 2: dropout bits drawn by the forward pass (else NULL)
};

struct Dev {
  const float *in0, *in1;
  int32_t i0N, i0D, i0H, i0W, i1N, i1D, i1H, i1W;
  int32_t C0;
  int32_t N, D, H, W;
  const float *w;
  int32_t flip;
  float *out0, *out1;
  int32_t o0N, o0D, o0H, o0W, o1N, o1D, o1H, o1W;
  int32_t CO0;                   // channels routed to out0 (rest to out1)
  int32_t OD, OH, OW;
  int32_t P;
  int32_t R, nych, zsegs, zper;
  int32_t YR, WX, WP;            // rows per ring slot, voxels loaded per row, LDS row pitch (voxels)
  int32_t ntiles, nseg;          // m-tiles per z-step; x-segments per row (S == 2 only)
  uint32_t magicX, magicWP, magicSeg;
  int32_t chunksX;
  unsigned long long *stamps;    // diagnostic: per-phase cycle sums [block][wave][8] (null in normal runs)
  int32_t patch;                 // epilogue transpose: row pitch (floats) of the 16-row LDS patch per wave (20: conflict-free,
                                 // 16: 4-way conflicts on its four writes, 2 KB less); 0 = ds_bpermute (no LDS)
  int32_t dbg;                   // ablation switches (TEM_DEBUG_FLAGS env, perf triage only): 1 no stores, 2 no MFMA, 4 no prefetch
  Ep32 ep;
};

template <int CI, int CO, int K, int S, int NW, int MAXPFX, int MTW, bool DROP>
__global__ __launch_bounds__(NW * 64) void conv_lds_k(Dev p) {
  constexpr int CIP = CI + 2, NT = (CO + 15) / 16, KS = CI / 4, NTHR = NW * 64, NTAP = K * K * K;
  constexpr int TAIL = S == 1 ? 18 : 40;                  // voxels the last tiles over-read past the ring (S == 1: <= 17)
  static_assert(CI % 16 == 0 && NW % NT == 0, "C_in 16 or 32: an even number of k-step pairs per tap");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int rowpitch = p.WP * CIP;
  const int slotpitch = p.YR * rowpitch;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps tile bookkeeping on the scalar unit
  const int m = lane & 15, kq = lane >> 4;

  int seg = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);   // neighbouring columns share an XCD's L2
  const int zseg = seg % p.zsegs; seg /= p.zsegs;
  const int ych = seg % p.nych;
  const int n = seg / p.nych;
  const int oy0 = ych * p.R;
  const int oz0 = zseg * p.zper;
  const int oz1 = min(p.OD, oz0 + p.zper);
  const int nsteps = oz1 - oz0;

  unsigned long long t_last = p.stamps ? clock64() : 0, t_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i) do { if (p.stamps) { unsigned long long t_now = clock64(); t_sum[i] += t_now - t_last; t_last = t_now; } } while (0)
  {  // the loader rewrites every ring voxel (out-of-range ones as zeros); only the tail the last tiles
     // over-read must be cleared once so that it stays finite
    float *tail = lds + K * slotpitch;
    for (int i = tid; i < TAIL * CIP; i += NTHR) tail[i] = 0.f;
  }

  // ---- loader of S new input planes per step (register staged)
  const int in0_n = n * p.i0N, in1_n = n * p.i1N;
  const int iy_base = oy0 * S - p.P;
  auto load_x = [&](float4 (&pf)[MAXPFX], int iz_first, int nplanes, int ltid, int lthreads) {
    const int total = nplanes * p.YR * p.chunksX;
#pragma unroll
    for (int i = 0; i < MAXPFX; ++i) {
      int id = ltid + i * lthreads;
      asm volatile("" : "+v"(id));                         // recompute per step: keeps index math out of live registers
      bool ok = id < total;
      int rowid = __umulhi((uint32_t)id, p.magicX);
      int pos = id - rowid * p.chunksX;
      int pl = rowid >= p.YR ? 1 : 0;
      int yr = rowid - pl * p.YR;
      int vox = pos / (CI / 4);
      int c = (pos - vox * (CI / 4)) * 4;
      int iz = iz_first + pl, iy = iy_base + yr, ix = vox - p.P;
      ok = ok && (unsigned)iz < (unsigned)p.D && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const float *src = c < p.C0 ? p.in0 + (in0_n + iz * p.i0D + iy * p.i0H + ix * p.i0W + c)
                                  : p.in1 + (in1_n + iz * p.i1D + iy * p.i1H + ix * p.i1W + (c - p.C0));
      pf[i] = ok ? *reinterpret_cast<const float4 *>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_x = [&](const float4 (&pf)[MAXPFX], int iz_first, int nplanes, int ltid, int lthreads) {
    const int total = nplanes * p.YR * p.chunksX;
    const int slot0 = ((iz_first % K) + K) % K, slot1 = (slot0 + 1) % K;
#pragma unroll
    for (int i = 0; i < MAXPFX; ++i) {
      int id = ltid + i * lthreads;
      asm volatile("" : "+v"(id));
      if (id < total) {
        int rowid = __umulhi((uint32_t)id, p.magicX);
        int pos = id - rowid * p.chunksX;
        int pl = rowid >= p.YR ? 1 : 0;
        int yr = rowid - pl * p.YR;
        int vox = pos / (CI / 4);
        int c = (pos - vox * (CI / 4)) * 4;
        float *d = lds + (pl ? slot1 : slot0) * slotpitch + yr * rowpitch + vox * CIP + c;   // 8-byte aligned
        *reinterpret_cast<float2 *>(d) = make_float2(pf[i].x, pf[i].y);
        *reinterpret_cast<float2 *>(d + 2) = make_float2(pf[i].z, pf[i].w);
      }
    }
  };

  // ---- this wave's output tiles: pair index = mtile*NT + nt, dealt round-robin (nt fixed per wave)
  const int nt = wave % NT;
  const int npairs = p.ntiles * NT;
  int abase[MTW];                                        // LDS float index of this lane's A voxel, tap (0,0,0)
#pragma unroll
  for (int j = 0; j < MTW; ++j) {
    int pr = min(wave + j * NW, npairs - 1);             // surplus tiles recompute the last one, never stored
    int t = pr / NT;
    int vox;
    if (S == 1) {
      vox = t * 16 + m;                                  // linearised: v = r*WP + x
    } else {
      int r = p.nseg == 1 ? t : (int)__umulhi((uint32_t)t, p.magicSeg), x0 = (t - r * p.nseg) * 16;
      vox = r * S * p.WP + (x0 + m) * S;
    }
    abase[j] = vox * CIP + 2 * kq;                         // this lane's channel pair of k-step pair 0
  }
  // ---- kernel-tap fragments.  MFMA k-step u = 2p + w of a tap multiplies channels ci = 8p + 2*kq + w
  // (kq = lane>>4): a lane's two channels of a k-step PAIR are adjacent, so its A fragment for both
  // k-steps is ONE ds_read_b64 -- half the LDS instructions and two k-steps of MFMA work (8 MFMAs at
  // MTW 4) per LDS round trip.  The sum over ci is order-free, B just follows the same map.
  const int co = nt * 16 + m;
  const bool bvalid = co < CO;
  const int cob = bvalid ? co : 0;
  auto load_b = [&](float (&b)[KS], int tap) {
    const float *wt = p.w + (p.flip ? NTAP - 1 - tap : tap) * (CI * CO);
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const int ci = 8 * (u / 2) + 2 * kq + (u & 1);
      b[u] = bvalid ? wt[p.flip ? cob * CI + ci : ci * CO + cob] : 0.f;
    }
  };
  auto tap_origin = [&](int tap, int izb) -> const float * {
    const int dz = tap / (K * K), rem = tap - dz * (K * K), dy = rem / K, dx = rem - dy * K;
    const int slot = (((izb + dz) % K) + K) % K;
    return lds + (slot * slotpitch + (dy * p.WP + dx) * CIP);
  };

  // ---- epilogue.  The MFMA leaves each lane with ONE channel of FOUR voxels (C/D map: col = lane&15,
  // row = 4*(lane>>4)+reg).  Every wave transposes its 16x16 tile so that a lane owns FOUR consecutive
  // channels of ONE voxel: the gate / add loads and the stores become single 16-byte accesses (a full
  // 64-byte channel run per voxel, 1 KB per wave-instruction), and the dropout bits of the four channels
  // come from one Philox call.  The transpose goes through a private 1.25 KB LDS patch per wave, or -- when
  // those 10 KB are what keeps one more output row per workgroup out of the 160 KB (p.patch == 0, host's
  // choice) -- through ds_bpermute (the LDS crossbar, no storage, ~900 cycles more per tile).
  const int TPITCH = p.patch;                             // floats per row of the transpose patch
  float *tp = lds + ((K * slotpitch + TAIL * CIP + 3) & ~3) + wave * (16 * TPITCH);   // 16-byte aligned
  const int ti = lane >> 2, tcq = lane & 3;               // transposed role: voxel row, channel quad
  const int tco = nt * 16 + tcq * 4;
  const int tsrc = (16 * (ti >> 2) + 4 * tcq) * 4;        // byte address of the lane holding (row ti, channel 4*tcq)
  const int tq = ti & 3;                                  // ... in accumulator register tq
  auto epilogue = [&](const f32x4 (&acc)[MTW], int oz) {
    const bool first = tco < p.CO0;
    const bool cvalid = tco < CO;
    float *optr = first ? p.out0 + (n * p.o0N + oz * p.o0D + tco) : p.out1 + (n * p.o1N + oz * p.o1D + (tco - p.CO0));
    const int oH = first ? p.o0H : p.o1H, oW = first ? p.o0W : p.o1W;
    const Ep32 &ep = p.ep;
    const float *gptr = ep.gate ? ep.gate + (n * ep.gN + oz * ep.gD + tco) : nullptr;
    const int az = oz - ep.aoz;
    const float *aptr = (ep.add && (unsigned)az < (unsigned)ep.aDd) ? ep.add + (n * ep.aN + az * ep.aD + tco) : nullptr;
    DropoutStream ds = ep.ds;
    if (DROP && ep.dropout && ep.step_dev) ds.step = *ep.step_dev;
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
      const int pr = wave + j * NW;
      if (pr < npairs) {                                   // wave-uniform
        int t = pr / NT;
        asm volatile("" : "+s"(t));                        // per-step recompute: no hoisted per-tile registers
        float4 v4;
        if (p.patch) {                                     // kernel-uniform
#pragma unroll
          for (int q = 0; q < 4; ++q) tp[(kq * 4 + q) * TPITCH + m] = acc[j][q];
          __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): this wave's own LDS writes have landed
          v4 = *reinterpret_cast<const float4 *>(tp + ti * TPITCH + tcq * 4);
        } else {
          float t4[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
              r[q] = __int_as_float(__builtin_amdgcn_ds_bpermute(tsrc + 4 * c, __float_as_int(acc[j][q])));
            t4[c] = tq == 0 ? r[0] : (tq == 1 ? r[1] : (tq == 2 ? r[2] : r[3]));
          }
          v4 = make_float4(t4[0], t4[1], t4[2], t4[3]);
        }
        int r, ox;
        if (S == 1) {
          const int v = t * 16 + ti;
          r = __umulhi((uint32_t)v, p.magicWP);
          ox = v - r * p.WP;
        } else {
          r = p.nseg == 1 ? t : (int)__umulhi((uint32_t)t, p.magicSeg);   // magic for d == 1 overflows 32 bits
          ox = (t - r * p.nseg) * 16 + ti;
        }
        const int oy = oy0 + r;
        if (cvalid && r < p.R && oy < p.OH && ox < p.OW) {
          float v[4] = {v4.x, v4.y, v4.z, v4.w};
          if (first) {
            if (ep.bias) {
#pragma unroll
              for (int c = 0; c < 4; ++c) v[c] += ep.bias[tco + c];
            }
            if (aptr) {
              const int ay = oy - ep.aoy, ax = ox - ep.aox;
              if ((unsigned)ay < (unsigned)ep.aHh && (unsigned)ax < (unsigned)ep.aWw) {
                const float4 a4 = *reinterpret_cast<const float4 *>(aptr + ay * ep.aH + ax * ep.aW);
                v[0] += a4.x; v[1] += a4.y; v[2] += a4.z; v[3] += a4.w;
              }
            }
            if (gptr) {
              const float4 g4 = *reinterpret_cast<const float4 *>(gptr + oy * ep.gH + ox * ep.gW);
              v[0] = g4.x > 0.f ? v[0] : ep.gate_slope * v[0];
              v[1] = g4.y > 0.f ? v[1] : ep.gate_slope * v[1];
              v[2] = g4.z > 0.f ? v[2] : ep.gate_slope * v[2];
              v[3] = g4.w > 0.f ? v[3] : ep.gate_slope * v[3];
            }
            if (DROP && ep.dropout) {
              const uint64_t e = ((((uint64_t)n * ep.dD + (oz + ep.doz)) * ep.dH + (oy + ep.doy)) * ep.dW + (ox + ep.dox)) *
                                     (uint64_t)p.CO0 + tco;
              if (ep.keep_mask) {                            // kernel-uniform: four bits of the forward pass's mask
                const uint32_t bits = (uint32_t)ep.keep_mask[e >> 3] >> (uint32_t)(e & 4u);
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = ((bits >> c) & 1u) ? 2.f * v[c] : 0.f;
              } else {
                const Philox128 ph = ds.block(e >> 7);
                const uint32_t eb = (uint32_t)(e & 127);
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = DropoutStream::bit(ph, eb + c) ? 2.f * v[c] : 0.f;
              }
            }
            if (ep.slope != 1.f) {
#pragma unroll
              for (int c = 0; c < 4; ++c) v[c] = v[c] > 0.f ? v[c] : ep.slope * v[c];
            }
          }
          if (!(p.dbg & 1)) *reinterpret_cast<float4 *>(optr + oy * oH + ox * oW) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
  };

  float4 pfx[MAXPFX];
  if (nsteps > 0) {                                        // all K planes of the first step in one flight
    const int iz0 = oz0 * S - p.P;
    constexpr int NB = (K + S - 1) / S;
    float4 pro[NB][MAXPFX];
#pragma unroll
    for (int b = 0; b < NB; ++b) load_x(pro[b], iz0 + b * S, min(S, K - b * S), tid, NTHR);
#pragma unroll
    for (int b = 0; b < NB; ++b) store_x(pro[b], iz0 + b * S, min(S, K - b * S), tid, NTHR);
  }
  __syncthreads();

  STAMP(0);                                                // zero-fill + prologue
  const bool late = wave >= NW / 2;
  // Steady-state loader (late half only): per-chunk descriptors are computed ONCE -- global offset inside a
  // plane, LDS offset inside a slot, and three bit masks (chunk valid in y/x, second plane of a stride-2
  // step, taken from in1) -- so that a step's loader is ~4 VALU per 16-byte chunk.
  int goff[MAXPFX], loff[MAXPFX];
  uint32_t okmask = 0, plmask = 0, srcmask = 0;
  constexpr int LTHR = LOADER_ALL ? NTHR : NTHR / 2;      // threads that run the steady-state loader
  const bool loads = LOADER_ALL || late;
  {
    const int ltid = LOADER_ALL ? tid : tid - NTHR / 2, total = S * p.YR * p.chunksX;
#pragma unroll
    for (int i = 0; i < MAXPFX; ++i) {
      int id = ltid + i * LTHR;
      bool ok = loads && id < total;
      int rowid = __umulhi((uint32_t)(ok ? id : 0), p.magicX);
      int pos = (ok ? id : 0) - rowid * p.chunksX;
      int pl = rowid >= p.YR ? 1 : 0;
      int yr = rowid - pl * p.YR;
      int vox = pos / (CI / 4);
      int c = (pos - vox * (CI / 4)) * 4;
      int iy = iy_base + yr, ix = vox - p.P;
      bool inb = ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      bool s1 = c >= p.C0;
      goff[i] = s1 ? in1_n + iy * p.i1H + ix * p.i1W + (c - p.C0) : in0_n + iy * p.i0H + ix * p.i0W + c;
      loff[i] = yr * rowpitch + vox * CIP + c;
      okmask |= (inb ? 1u : 0u) << i;
      plmask |= ((ok && pl) ? 1u : 0u) << i;
      srcmask |= (s1 ? 1u : 0u) << i;
      if (!ok) loff[i] = -1;                               // chunk does not exist for this thread
    }
  }
  auto load_fast = [&](float4 (&pf)[MAXPFX], int iz_first) {
#pragma unroll
    for (int i = 0; i < MAXPFX; ++i) {
      const int pl = (plmask >> i) & 1u;
      const int iz = iz_first + pl;
      const bool s1 = (srcmask >> i) & 1u;
      const bool ok = ((okmask >> i) & 1u) && (unsigned)iz < (unsigned)p.D;
      const float *src = (s1 ? p.in1 : p.in0) + (goff[i] + iz * (s1 ? p.i1D : p.i0D));
      pf[i] = ok ? *reinterpret_cast<const float4 *>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_fast = [&](const float4 (&pf)[MAXPFX], int iz_first) {
    const int slot0 = ((iz_first % K) + K) % K, slot1 = (slot0 + 1) % K;
#pragma unroll
    for (int i = 0; i < MAXPFX; ++i) {
      if (loff[i] >= 0) {
        float *d = lds + ((((plmask >> i) & 1u) ? slot1 : slot0) * slotpitch + loff[i]);
        *reinterpret_cast<float2 *>(d) = make_float2(pf[i].x, pf[i].y);
        *reinterpret_cast<float2 *>(d + 2) = make_float2(pf[i].z, pf[i].w);
      }
    }
  };
  f32x4 acc_prev[MTW];
#pragma unroll
  for (int j = 0; j < MTW; ++j) acc_prev[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int step = 0; step < nsteps; ++step) {
    const int oz = oz0 + step;
    const int izb = oz * S - p.P;
    const bool more = step + 1 < nsteps;
    // Only the late half fetches the next planes: the early half goes straight to its MFMA stream, so the
    // matrix pipe is busy while the loader's address arithmetic and load issue run on the partner waves.
    if (loads && more && !(p.dbg & 4)) load_fast(pfx, izb + K);

    STAMP(1);                                              // load_x issue
    if (late && step > 0) epilogue(acc_prev, oz - 1);
    STAMP(2);                                              // early epilogue (late half)

    f32x4 acc[MTW];
#pragma unroll
    for (int j = 0; j < MTW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Flat pipeline over (tap, k-step): the A fragments of the NEXT group are read from LDS before
    // the current group's MFMAs issue; B fragments of the next tap are fetched one tap ahead.
    // Kernel taps live in a ring of three register sets: tap t multiplies with set t%3 while the loads
    // of tap t+2 fill set (t+2)%3 -- two whole taps (>= 1000 cycles) of L2 latency cover and no
    // register-to-register rotation (a copy would force the wait one tap early).  The tap loop is
    // unrolled by 3 so that the set indices are static.
    constexpr int PP = KS / 2;                            // k-step pairs per tap (even: 2 or 4)
    const int ntap_run = ((p.dbg & 2) || ((p.dbg & 64) && late) || ((p.dbg & 128) && !late)) ? 1 : NTAP;   // ablations
    float B[3][KS];
    float2 A[2][MTW];
    load_b(B[0], 0);
    load_b(B[1], NTAP > 1 ? 1 : 0);
    const float *xt = tap_origin(0, izb);
#pragma unroll
    for (int j = 0; j < MTW; ++j) A[0][j] = *reinterpret_cast<const float2 *>(xt + abase[j]);
    for (int tap0 = 0; tap0 < ntap_run; tap0 += 3) {
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
        const int tap = tap0 + tb;
        if (tap < ntap_run) {                              // wave-uniform
          const int tn = tap + 1 < NTAP ? tap + 1 : tap;
          const float *xn = tap_origin(tn, izb);
          load_b(B[(tb + 2) % 3], tap + 2 < NTAP ? tap + 2 : tap);
#pragma unroll
          for (int pp = 0; pp < PP; ++pp) {
            const int cur = pp & 1, nxt = cur ^ 1;
            // fragments of the NEXT pair (same tap, or pair 0 of the next tap) fly while this pair's MFMAs issue
#pragma unroll
            for (int j = 0; j < MTW; ++j)
              A[nxt][j] = *reinterpret_cast<const float2 *>((pp + 1 < PP ? xt + 8 * (pp + 1) : xn) + abase[j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < MTW; ++j)
              acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[cur][j].x, B[tb][2 * pp], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < MTW; ++j)
              acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[cur][j].y, B[tb][2 * pp + 1], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
          xt = xn;
        }
      }
    }

    STAMP(3);                                              // MFMA phase
    // Waves NW/2.. share their SIMDs with waves 0..NW/2-1.  The upper half runs one step's epilogue
    // (pure VALU + stores) at the START of the next step, i.e. while its SIMD partner streams MFMAs,
    // and streams its own MFMAs while the partner runs its epilogue: matrix and vector pipes overlap.
    if (!late) {
      epilogue(acc, oz);
    } else {
#pragma unroll
      for (int j = 0; j < MTW; ++j) acc_prev[j] = acc[j];
    }
    STAMP(4);                                              // epilogue (early half) / acc copy
    __syncthreads();                                       // all waves are done reading the oldest planes
    STAMP(5);                                              // barrier 1 wait
    if (loads && more && !(p.dbg & 4)) store_fast(pfx, izb + K);
    STAMP(6);                                              // store_x
    __syncthreads();
    STAMP(7);                                              // barrier 2 wait
  }
  if (late && nsteps > 0) epilogue(acc_prev, oz1 - 1);
  if (p.stamps && lane == 0) {
    for (int i = 0; i < 8; ++i) p.stamps[((size_t)blockIdx.x * NW + wave) * 8 + i] = t_sum[i];
  }
#undef STAMP
}

// ------------------------------------------------------------------------------------------ host
constexpr int LDS_MAX = 160 * 1024;                     // gfx950: 160 KiB per workgroup
static thread_local char *g_name = nullptr;   // set by tem_conv_describe around a dry run
static thread_local int g_name_len = 0;
constexpr int TARGET_BLOCKS = 512;

static uint32_t magic_for(int d) { return (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d); }

static bool fits32(const tem_view &v) {
  int64_t span = (int64_t)(v.N - 1) * v.sN + (int64_t)(v.D - 1) * v.sD + (int64_t)(v.H - 1) * v.sH +
                 (int64_t)(v.W - 1) * v.sW + v.C;
  return span < (int64_t)1 << 31 && v.sN < ((int64_t)1 << 31);
}

// mode 0: launch; 1: dry run; 2: only predict (*cost receives the model's cycles)
template <int CI, int CO, int K, int S, int NW, int MAXPFX, int MTW, bool DROP>
int run(Dev p, hipStream_t st, int mode, double *cost) {
  constexpr int CIP = CI + 2, NT = (CO + 15) / 16;
  const int NTHR = NW * 64;
  p.WX = (p.OW - 1) * S + K;
  p.WP = p.WX;                                             // LDS row pitch in voxels
  p.chunksX = p.WX * (CI / 4);
  p.magicX = magic_for(p.chunksX);
  p.magicWP = magic_for(p.WP);
  p.nseg = (p.OW + 15) / 16;
  p.magicSeg = magic_for(p.nseg);
  size_t lds_bytes = 0;
  // Pick rows-per-workgroup R and the z-run length by a small cost model (cycles per SIMD):
  //   step     = MFMA slots of the two waves sharing a SIMD + loader/epilogue work that does not overlap
  //   prologue = first K planes + LDS clear, paid once per workgroup
  //   total    = ceil(workgroups / 256 CUs) * (prologue + steps * step)
  int R = 0;
  double best = 1e300;
  const int NTAPS = K * K * K, KSTEPS = CI / 4;
  for (int r = 1; r <= (p.OH < 16 ? p.OH : 16); ++r) {
    int YR = (r - 1) * S + K;
    int ntiles = S == 1 ? (r * p.WP + 15) / 16 : r * p.nseg;
   for (int patch = 20; patch >= 0; patch = patch == 20 ? 16 : (patch == 16 ? 0 : -1)) {
    // ring + tail the last tiles over-read (+ one 16-row transpose patch per wave)
    size_t bytes = ((((size_t)K * YR * p.WP * CIP + (S == 1 ? 18 : 40) * CIP + 3) & ~(size_t)3) + NW * 16 * patch) * 4;
    bool fits = bytes <= (size_t)LDS_MAX && (size_t)S * YR * p.chunksX <= (size_t)MAXPFX * (LOADER_ALL ? NTHR : NTHR / 2) &&
                ntiles * NT <= MTW * NW;
    if (!fits) continue;
    int rounds = (ntiles * NT + NW - 1) / NW;                      // tile slots each wave executes per step
    int nych = (p.OH + r - 1) / r;
    int cols = p.N * nych;
    if (rounds > MTW) continue;
    double step = 2.0 * MTW * NTAPS * KSTEPS * 32.0 * 1.35 + 2500.0 + 600.0 * MTW;   // every wave runs MTW slots
    if (patch == 16) step += 100.0 * MTW;             // bank conflicts on the patch writes
    if (patch == 0) step += 2500.0 * MTW;             // measured: the ds_bpermute transpose costs 2-3k cycles per tile
    double pro = 12000.0 + bytes / 12.0;              // first K planes arrive at the CU's HBM share (~12 B/clk)
    for (int zs = 1; zs <= p.OD; ++zs) {
      int zper = (p.OD + zs - 1) / zs, zsegs = (p.OD + zper - 1) / zper;
      if (zsegs != zs) continue;
      double t = std::ceil(cols * zsegs / 256.0) * (pro + zper * step);
      if (t < best) {
        best = t; R = r; p.R = r; p.YR = YR; p.ntiles = ntiles; lds_bytes = (bytes + 15) & ~(size_t)15;
        p.nych = nych; p.zper = zper; p.zsegs = zsegs; p.patch = patch;
      }
    }
   }
  }
  if (R < 1) return TEM_EUNSUPPORTED;
  if (cost) *cost = best;
  if (mode == 2) return TEM_OK;
  const bool dry = mode == 1;
  int nblocks = p.N * p.nych * p.zsegs;
  if (dry) {
    if (g_name)
      snprintf(g_name, g_name_len, "conv_lds_k<%d, %d, %d, %d, %d, %d, %d, %s>", CI, CO, K, S, NW, MAXPFX, MTW,
               DROP ? "true" : "false");
    return TEM_OK;
  }
  if (p.dbg & 8)
    fprintf(stderr, "conv_lds<%d,%d,%d,%d> OW=%d OH=%d OD=%d: R=%d YR=%d ntiles=%d nych=%d zsegs=%d zper=%d blocks=%d lds=%zu patch=%d\n", CI, CO, K, S,
            p.OW, p.OH, p.OD, p.R, p.YR, p.ntiles, p.nych, p.zsegs, p.zper, nblocks, lds_bytes, p.patch);
  auto kern = conv_lds_k<CI, CO, K, S, NW, MAXPFX, MTW, DROP>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(NTHR), lds_bytes, st, p);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

// each geometry is built with 4 and with 2 (and 1) accumulator tiles per wave; the cost model picks
template <int CI, int CO, int K, int S, int NW, int MAXPFX, bool DROP>
int run_best(const Dev &p, hipStream_t st, bool dry) {
  double c4 = 1e300, c2 = 1e300, c1 = 1e300;
  run<CI, CO, K, S, NW, MAXPFX, 4, DROP>(p, st, 2, &c4);
  run<CI, CO, K, S, NW, MAXPFX, 2, DROP>(p, st, 2, &c2);
  run<CI, CO, K, S, NW, MAXPFX, 1, DROP>(p, st, 2, &c1);
  if (c4 >= 1e300 && c2 >= 1e300 && c1 >= 1e300) return TEM_EUNSUPPORTED;
  if (c4 <= c2 && c4 <= c1) return run<CI, CO, K, S, NW, MAXPFX, 4, DROP>(p, st, dry ? 1 : 0, nullptr);
  if (c2 <= c1) return run<CI, CO, K, S, NW, MAXPFX, 2, DROP>(p, st, dry ? 1 : 0, nullptr);
  return run<CI, CO, K, S, NW, MAXPFX, 1, DROP>(p, st, dry ? 1 : 0, nullptr);
}

#define CONV_CASE(ci, co, k, s, nw, pfx, mtw) \
  if (CI == ci && CO == co && K == k && S == s && !a->ep.dropout) return run_best<ci, co, k, s, nw, pfx_of(pfx), false>(p, st, dry);
#define CONV_CASE_DROP(ci, co, k, s, nw, pfx, mtw) \
  if (CI == ci && CO == co && K == k && S == s && a->ep.dropout) return run_best<ci, co, k, s, nw, pfx_of(pfx), true>(p, st, dry);

int dispatch(const tem_conv_args *a, hipStream_t st, bool dry) {
  const tem_view &i0 = a->in0, &o0 = a->out0;
  const bool cube = a->kd == a->kh && a->kh == a->kw && a->sd == a->sh && a->sh == a->sw && a->pd == a->ph &&
                    a->ph == a->pw;
  if (!cube || a->kd < 3) return TEM_EUNSUPPORTED;
  Dev p{};
  if (!fits32(i0) || !fits32(o0)) return TEM_EUNSUPPORTED;
  p.in0 = i0.ptr; p.i0N = (int)i0.sN; p.i0D = (int)i0.sD; p.i0H = (int)i0.sH; p.i0W = (int)i0.sW; p.C0 = i0.C;
  p.in1 = i0.ptr; p.i1N = p.i0N; p.i1D = p.i0D; p.i1H = p.i0H; p.i1W = p.i0W;
  int CI = i0.C;
  if (a->in1.ptr) {
    const tem_view &i1 = a->in1;
    if (i1.N != i0.N || i1.D != i0.D || i1.H != i0.H || i1.W != i0.W) return TEM_ESHAPE;
    if (!fits32(i1)) return TEM_EUNSUPPORTED;
    p.in1 = i1.ptr; p.i1N = (int)i1.sN; p.i1D = (int)i1.sD; p.i1H = (int)i1.sH; p.i1W = (int)i1.sW;
    CI += i1.C;
    if (i0.C % 4 || i1.C % 4) return TEM_EUNSUPPORTED;
  }
  p.N = i0.N; p.D = i0.D; p.H = i0.H; p.W = i0.W;
  p.w = a->w; p.flip = a->w_layout == TEM_W_FLIP_CO_CI;
  p.out0 = o0.ptr; p.o0N = (int)o0.sN; p.o0D = (int)o0.sD; p.o0H = (int)o0.sH; p.o0W = (int)o0.sW; p.CO0 = o0.C;
  int CO = o0.C;
  if (a->out1.ptr) {
    const tem_view &o1 = a->out1;
    if (o1.N != o0.N || o1.D != o0.D || o1.H != o0.H || o1.W != o0.W) return TEM_ESHAPE;
    if (!fits32(o1)) return TEM_EUNSUPPORTED;
    p.out1 = o1.ptr; p.o1N = (int)o1.sN; p.o1D = (int)o1.sD; p.o1H = (int)o1.sH; p.o1W = (int)o1.sW;
    CO += o1.C;
  }
  if (o0.N != i0.N) return TEM_ESHAPE;
  p.OD = o0.D; p.OH = o0.H; p.OW = o0.W;
  p.P = a->pd;
  {
    static int dbg = -1;
    if (dbg < 0) dbg = tem_env_int("TEM_DEBUG_FLAGS", 0);
    p.dbg = dbg;
    static unsigned long long stamp_ptr = ~0ull;
    if (stamp_ptr == ~0ull) stamp_ptr = tem_env_hex("TEM_STAMP_BUF");
    p.stamps = (unsigned long long *)stamp_ptr;
  }
  const tem_epilogue &e = a->ep;
  Ep32 &q = p.ep;
  q.bias = e.bias; q.slope = e.slope; q.gate_slope = e.gate_slope;
  if (e.gate.ptr) {
    const tem_view &g = e.gate;
    if (g.N != o0.N || g.D != o0.D || g.H != o0.H || g.W != o0.W || g.C < o0.C) return TEM_ESHAPE;
    if (!fits32(g)) return TEM_EUNSUPPORTED;
    q.gate = g.ptr; q.gN = (int)g.sN; q.gD = (int)g.sD; q.gH = (int)g.sH; q.gW = (int)g.sW;
  }
  if (e.add.ptr) {
    const tem_view &ad = e.add;
    if (ad.C < o0.C || ad.N != o0.N) return TEM_ESHAPE;
    if (!fits32(ad)) return TEM_EUNSUPPORTED;
    q.add = ad.ptr; q.aN = (int)ad.sN; q.aD = (int)ad.sD; q.aH = (int)ad.sH; q.aW = (int)ad.sW;
    q.aoz = e.add_off[0]; q.aoy = e.add_off[1]; q.aox = e.add_off[2];
    q.aDd = ad.D; q.aHh = ad.H; q.aWw = ad.W;
  }
  q.dropout = e.dropout;
  q.ds.k0 = (uint32_t)e.seed; q.ds.k1 = (uint32_t)(e.seed >> 32); q.ds.site = e.site; q.ds.step = e.step;
  q.step_dev = e.step_dev;
  if (e.dropout && e.keep_mask && e.keep_mode == 1) return TEM_EUNSUPPORTED;      // the direct kernels write the mask
  q.keep_mask = (e.dropout && e.keep_mask && e.keep_mode == 2 && o0.C % 8 == 0) ? e.keep_mask : nullptr;
  q.doz = e.drop_org[0]; q.doy = e.drop_org[1]; q.dox = e.drop_org[2];
  q.dD = e.drop_dims[0] ? e.drop_dims[0] : o0.D; q.dH = e.drop_dims[0] ? e.drop_dims[1] : o0.H;
  q.dW = e.drop_dims[0] ? e.drop_dims[2] : o0.W;
  auto aligned = [](const tem_view &v) {
    return ((uintptr_t)v.ptr & 15) == 0 && v.sW % 4 == 0 && v.sH % 4 == 0 && v.sD % 4 == 0 && v.sN % 4 == 0;
  };
  if (!aligned(i0) || (a->in1.ptr && !aligned(a->in1))) return TEM_EUNSUPPORTED;
  // the epilogue moves 4 channels per lane as one 16-byte access
  if (!aligned(o0) || o0.C % 4 || (a->out1.ptr && (!aligned(a->out1) || a->out1.C % 4))) return TEM_EUNSUPPORTED;
  if (e.gate.ptr && !aligned(e.gate)) return TEM_EUNSUPPORTED;
  if (e.add.ptr && !aligned(e.add)) return TEM_EUNSUPPORTED;
  const int K = a->kd, S = a->sd;
  //         CI  CO  K  S  waves  X-chunks  tiles/wave
  // (8,8,k3) and (16,8,k3): C_out = 8 fills half of a 16-wide MFMA tile; the direct VALU kernel is
  // faster there (46 vs 27 TFLOP/s measured on g.d1a) -- they are left to conv_direct.hip.
  // (8,16,k3): left to the direct kernel (same speed there, 30 TFLOP/s, and C_in = 8 has a single k-step pair)
  CONV_CASE(16, 16, 3, 1, 8, 12, 4)     // g.f1 fwd (concat 8+8) and its input-gradient (split 8|8)
  CONV_CASE(16, 32, 3, 1, 8, 12, 4)     // g.u2a, d.d2a fwd; input-gradient of g.u1a
  CONV_CASE(32, 16, 3, 1, 8, 12, 4)     // g.u1a fwd; input-gradients of the 16->32 layers
  CONV_CASE(32, 32, 3, 1, 8, 12, 4)     // g.mid fwd (concat 16+16) and input-gradient (split 16|16); d.d3a
  // (8,8,k4,s2) and (8,16,k4,s2): 64 taps x 2 k-steps with half-empty N tiles -- the direct kernel is
  // faster (46 vs 12 TFLOP/s measured on g.d1b); left to conv_direct.hip.
  CONV_CASE(16, 16, 4, 2, 8, 14, 4)     // g.d2b fwd
  CONV_CASE(32, 32, 4, 2, 8, 14, 4)     // d.d2b, d.d3b fwd
  CONV_CASE(16, 32, 4, 2, 8, 14, 4)     // input-gradient of g.u2b
  CONV_CASE_DROP(16, 16, 3, 1, 8, 12, 4)  // input-gradient of g.f1 through Dropout (split 8|8)
  CONV_CASE_DROP(32, 32, 3, 1, 8, 12, 4)  // input-gradient of g.mid through Dropout (split 16|16)
  return TEM_EUNSUPPORTED;
}

}  // namespace convlds

// Called by tem_conv (dispatch.hip) before it falls back to the direct kernel.
int tem_conv_lds_try(const tem_conv_args *a, hipStream_t st, bool dry) { return convlds::dispatch(a, st, dry); }

int tem_conv_lds_describe(const tem_conv_args *a, char *buf, int len) {
  convlds::g_name = buf; convlds::g_name_len = len;
  int rc = convlds::dispatch(a, nullptr, true);
  convlds::g_name = nullptr;
  return rc;
}
