// conv_s2.hip -- the k4 s2 convolutions (Downsample blocks, models/utils.py:80, and the input-gradient of the
// Conv3DTranspose layers, models/utils.py:129-130) as a split-K GEMM on the fp32 matrix cores, without an LDS image.
//
//   out[o][co] = epilogue( sum_{(kz,ky,kx), ci} X[2 o + k - P][ci] * W[(kz,ky,kx)][ci][co] ),   K = 64 C_in
//
// The reduction index of v_mfma_f32_16x16x4_f32 may be permuted freely as long as A and B agree, so lane group
// kq = lane >> 4 (the instruction's k index) is given the x-tap kx = kq: for a row tap (kz, ky) the lane of output
// voxel m reads the C_in CONTIGUOUS floats of input voxel (2 oz + kz - P, 2 oy + ky - P, 2 ox + kq - P) as 16-byte
// loads straight from HBM/L2 (the four lane groups of a voxel cover one 4 C_in run; neighbouring voxels overlap by
// half, which the L1 absorbs) and k-step j multiplies channel ci = j.  No LDS staging, no bank conflicts; the loads of
// the next tiles are issued one by one between the current tiles' MFMAs (ping-pong fragment sets, no copies).
//
// The reduction is SPLIT OVER THE WAVES of a workgroup: C_in / 2 waves, each owning 32 k-steps = 32 / C_in row taps
// (one kz plane: 4, 2 or 1 values of ky), so a wave keeps its whole B fragment in 32 VGPRs per n-tile for the life
// of the workgroup and even the deep layers with a few hundred output voxels use every SIMD of a CU.  The waves'
// partial 16 x 16 tiles meet in LDS (double-buffered, one barrier per iteration) and are summed in wave order --
// deterministic -- by lanes of every wave, which run the fused epilogue (bias, skip-gradient add, LeakyReLU-gradient
// gate, LeakyReLU; their gate / add values were requested before the MFMA chain) and store 16-byte channel runs.
// Tiles are 16 consecutive output voxels of one plane, linearised over (oy, ox).  One workgroup per resident slot; the
// workgroups of an XCD walk that XCD's contiguous eighth of the tiles interleaved (see the kernel).
// Measured stand-alone (132^3 step shapes, warm): 462 us for the eleven k4 s2 launches on the previous kernels
// (conv_direct_k / conv_lds_k), 216 us here; e.g. input-gradient of g.u1b 60 -> 29 us (71 TFLOP/s), d.d2b 64 -> 25 us,
// g.d1b (8 -> 8, two-block form below) 56 -> 34 us.
#include "tem_common.h"
#include <cstdio>
#include <cstdlib>

namespace convs2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Ep {                      // epilogue with 32-bit strides (no dropout on these layers: dispatch leaves those to the other kernels)
  const float *bias;
  float slope;
  const float *gate; int32_t gN, gD, gH, gW, gbytes; float gate_slope;
  const float *add;  int32_t aN, aD, aH, aW, aoz, aoy, aox, aDd, aHh, aWw, abytes;
};

struct Dev {
  const float *in;
  int32_t iN, iD, iH, iW, D, H, W;
  const float *w;
  float *out;
  int32_t oN, oD, oH, oW, OD, OH, OW, CO, oN_count;
  int32_t P;
  int32_t RL;                                       // slots per output row: OW (TB: OW + 1 input voxel pairs)
  int32_t dbg;                                      // TEM_DEBUG_FLAGS (perf triage): 2 no epilogue
  int32_t in_bytes;                                 // extent of the input view (bytes), the buffer descriptor's range
  int32_t tiles_pp, plane_vox, total, iters;        // tiles per output plane, voxels per plane, tiles, iterations (T tiles each)
  uint32_t magicOW, magicTpp, magicOD;
  Ep ep;
};

constexpr int PITCH = 20;                           // row pitch (floats) of a partial 16 x 16 tile in LDS
constexpr int OOB = (int)0x80000000;                // buffer offset past every descriptor range: the load returns zeros

__device__ __forceinline__ uint32_t fdiv(uint32_t v, uint32_t d, uint32_t magic) { return d == 1 ? v : __umulhi(v, magic); }

// TB (8 -> 8 channels, "two blocks"): with C_out = 8 half of an n-tile would be zeros.  Here a tile row is an INPUT voxel
// pair j (x = 2 j - P, 2 j + 1 - P; 16 contiguous floats: one 16-byte load per lane and row tap, no voxel read twice) and
// the two halves of the columns are its two uses: columns 0..7 = taps k_x 0, 1 of output voxel j, columns 8..15 = taps
// k_x 2, 3 of output voxel j - 1.  Half the k-steps, every MFMA column useful; the epilogue adds row j's first block and
// row j + 1's second block.  A row of the output has OW + 1 pair slots; tiles are 16 consecutive slots, 15 apart (the
// sixteenth row only lends its second block).
template <int CI, int NT, int T, bool TB>
__global__ __launch_bounds__(CI / 2 * 64) void conv_s2_k(Dev p) {
  constexpr int NWAVE = CI / 2, KQ = 32 / CI;
  constexpr int RUNF = TB ? 4 : CI;                 // floats per lane and row tap: its x-tap's channels (TB: a quarter of the pair)
  constexpr int CPL = RUNF / 4;                     // 16-byte loads per lane and row tap
  constexpr int NA = KQ * CPL;                      // ... per lane and tile
  constexpr int VPT = TB ? 15 : 16;                 // output voxels (TB: new pair slots) per tile
  static_assert(!TB || (CI == 8 && NT == 1), "two blocks: 8 -> 8 channels");
  constexpr bool PREF = CI < 32;                    // 16 waves (four per SIMD, 128 VGPRs each) hide the latency by themselves
  constexpr int TILEF = 16 * PITCH;                 // floats of one partial tile
  constexpr int IPW = T * 64 * NT / NWAVE;          // epilogue items (row of a tile, column quad) per wave
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const int rt0 = wave * KQ, kz = rt0 >> 2, ky0 = rt0 & 3;   // this wave's row taps: (kz, ky0 .. ky0 + KQ - 1)

  const int cob = blockIdx.y * (16 * NT);           // first output channel of this workgroup (32 -> 32: two halves)
  // Workgroups are dealt round-robin over the 8 XCDs (tem_common.h): XCD k owns the contiguous eighth [x0, it1) of the
  // iterations and its workgroups walk it INTERLEAVED (workgroup j takes x0 + j, x0 + j + step, ...), so at any moment one
  // XCD works on ~step consecutive iterations -- a couple of output planes whose input planes (each used by two output
  // planes and two output rows) stay in that XCD's 4 MB L2 instead of being fetched again over the fabric.
  const bool one = gridDim.x < 8;                   // (fewer workgroups than XCDs: one interleaved walk over everything)
  const int xcd = one ? 0 : (int)(blockIdx.x & 7), step = one ? (int)gridDim.x : ((int)gridDim.x - xcd + 7) >> 3;
  const int x0 = one ? 0 : (int)(((long long)xcd * p.iters) >> 3), it1 = one ? p.iters : (int)(((long long)(xcd + 1) * p.iters) >> 3);
  const int it0 = x0 + (one ? (int)blockIdx.x : (int)(blockIdx.x >> 3));

  // Every global read goes through a buffer descriptor: a lane whose voxel / row tap / column lies outside (zero padding,
  // the over-hang of the last tile, an absent gate or skip-gradient view) sends an offset past the descriptor's range and
  // receives zeros -- no branches and no selects on the data, and ONE load schedule for the compiler's vmcnt bookkeeping
  // (a load behind a branch makes it wait for the newest loads on the path that issued them).
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, 64 * CI * p.CO * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)p.ep.gate, 0, p.ep.gate ? p.ep.gbytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void *)p.ep.add, 0, p.ep.add ? p.ep.abytes : 0, 0x00020000);

  // tile g -> (n, oz, t); wave-uniform
  auto decode = [&](int g, int &n, int &oz, int &t) {
    const int pl = (int)fdiv((uint32_t)g, (uint32_t)p.tiles_pp, p.magicTpp);      // n * OD + oz
    t = g - pl * p.tiles_pp;
    n = (int)fdiv((uint32_t)pl, (uint32_t)p.OD, p.magicOD);
    oz = pl - n * p.OD;
  };
  // A fragments of tile g: float jj of row tap rr of the lane = float kq RUNF + jj of the run that starts at input voxel
  // (2 oz + kz - P, 2 oy + ky - P, 2 ox - P), ox = the row's (first) output voxel.  `offsets` does the index arithmetic of
  // an iteration (byte offsets, OOB where nothing is to be read), `fetch` issues one 16-byte load.
  auto offsets = [&](int (&off)[T][NA], int it) {
    const bool live = it < it1;
#pragma unroll
    for (int tt = 0; tt < T; ++tt) {
      const int g = it * T + tt;
      int n, oz, t;
      decode(min(g, p.total - 1), n, oz, t);
      const int v = t * VPT + m;
      const int oy = (int)fdiv((uint32_t)v, (uint32_t)p.RL, p.magicOW), ox = v - oy * p.RL;      // (TB: ox = pair slot j)
      const int iz = 2 * oz + kz - p.P, x0 = 2 * ox - p.P;
      const bool okv = live && g < p.total && v < p.plane_vox && (unsigned)iz < (unsigned)p.D;
      const int base = (n * p.iN + iz * p.iD) * 4;
      int xoff[CPL];                                 // byte offset of chunk c inside the input row, or OOB
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int fo = kq * RUNF + 4 * c, xin = x0 + fo / CI;            // (CI is a power of two: shifts)
        xoff[c] = (okv && (unsigned)xin < (unsigned)p.W) ? base + (xin * p.iW + (fo & (CI - 1))) * 4 : OOB;
      }
#pragma unroll
      for (int rr = 0; rr < KQ; ++rr) {
        const int iy = 2 * oy + ky0 + rr - p.P;
        const bool oky = (unsigned)iy < (unsigned)p.H;
#pragma unroll
        for (int c = 0; c < CPL; ++c) off[tt][rr * CPL + c] = (oky && xoff[c] != OOB) ? xoff[c] + iy * p.iH * 4 : OOB;
      }
    }
  };
  auto fetch = [&](int off) -> float4 {
    const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);
    return make_float4(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w));
  };
  auto load_it = [&](float4 (&a)[T][NA], int it) {
    int off[T][NA];
    offsets(off, it);
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int tt = 0; tt < T; ++tt) a[tt][i] = fetch(off[tt][i]);
  };

  // ---- B fragment: k-step (rr, jj) multiplies W[(kz, ky0 + rr, kx)][ci][co] with (x-tap, ci) = float kq RUNF + jj of the run;
  // column n of the lane: co = cob + 16 nt + n, kx = x-tap
  float B[KQ * RUNF][NT];
#pragma unroll
  for (int rr = 0; rr < KQ; ++rr)
#pragma unroll
    for (int jj = 0; jj < RUNF; ++jj)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int f = kq * RUNF + jj, xt = f / CI, ci = f & (CI - 1);
        const int co = TB ? (m & 7) : cob + nt * 16 + m, kx = TB ? xt + 2 * (m >> 3) : xt;
        const bool ok = co < p.CO && (unsigned)kx < 4u;
        B[rr * RUNF + jj][nt] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
            wrs, ok ? (((((rt0 + rr) * 4 + kx) * CI + ci) * p.CO) + co) * 4 : OOB, 0, 0));
      }

  // epilogue role of the lane: item = wave IPW + lane = (tile tt of the iteration, row mi, column quad cq)
  const int item = wave * IPW + lane;
  const int ett = __builtin_amdgcn_readfirstlane(item / (64 * NT));       // (a wave's items lie in one tile)
  const int erem = item - ett * (64 * NT);
  const int emi = erem / (4 * NT), ecq = erem - emi * (4 * NT);
  const int ent = ecq >> 2, ecl = 4 * (ecq & 3);
  const int ecol = TB ? 4 * (ecq & 1) : cob + 4 * ecq;                   // first of the lane's 4 output channels
  const bool elane = lane < IPW && (TB ? (ecq < 2 && emi < 15) : ecol < p.CO) && !(p.dbg & 2);
  const Ep &ep = p.ep;

  // one iteration on loaded fragments: (gate / skip-gradient loads issued,) MFMA chains, partial tiles to LDS, barrier,
  // wave sum + epilogue
  auto compute = [&](const float4 (&a)[T][NA], float4 (&an)[T][NA], int it, int itn, int par) {
    // ---- the lane's output voxel; its gate and skip-gradient values travel under the MFMA chain
    const int g = it * T + ett;
    int en, eoz, et;
    decode(min(g, p.total - 1), en, eoz, et);
    const int ev = et * VPT + emi;
    const int eoy = (int)fdiv((uint32_t)ev, (uint32_t)p.RL, p.magicOW), eox = ev - eoy * p.RL;
    const bool evalid = elane && g < p.total && ev < p.plane_vox && eox < p.OW;
    const u32x4 gq = __builtin_amdgcn_raw_buffer_load_b128(
        grs, evalid ? (en * ep.gN + eoz * ep.gD + eoy * ep.gH + eox * ep.gW + ecol) * 4 : OOB, 0, 0);
    const int az = eoz - ep.aoz, ay = eoy - ep.aoy, ax = eox - ep.aox;
    const bool ain = evalid && (unsigned)az < (unsigned)ep.aDd && (unsigned)ay < (unsigned)ep.aHh && (unsigned)ax < (unsigned)ep.aWw;
    const u32x4 aq = __builtin_amdgcn_raw_buffer_load_b128(
        ars, ain ? (en * ep.aN + az * ep.aD + ay * ep.aH + ax * ep.aW + ecol) * 4 : OOB, 0, 0);

    int offn[PREF ? T : 1][NA];                       // the next iteration's loads are issued one by one BETWEEN the MFMAs
    if constexpr (PREF) offsets(offn, itn);           // (a burst would fill the memory pipeline's queue and stall the wave at issue)
    f32x4 acc[T][NT][2];                              // two chains per tile (even / odd k-steps), added at the end
#pragma unroll
    for (int tt = 0; tt < T; ++tt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[tt][nt][0] = acc[tt][nt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int rr = i / CPL, c = i - rr * CPL;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = rr * RUNF + 4 * c + e;
#pragma unroll
        for (int tt = 0; tt < T; ++tt) {
          const float av = e == 0 ? a[tt][i].x : e == 1 ? a[tt][i].y : e == 2 ? a[tt][i].z : a[tt][i].w;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[tt][nt][T * NT == 1 ? (e & 1) : 0] =
                __builtin_amdgcn_mfma_f32_16x16x4f32(av, B[k][nt], acc[tt][nt][T * NT == 1 ? (e & 1) : 0], 0, 0, 0);
        }
        if constexpr (PREF) {
          if (e < T) {                                // tile e's chunk i of the next iteration behind k-step e
            __builtin_amdgcn_sched_barrier(0);        // (keeps the load HERE: the scheduler would gather them into one burst)
            an[e < T ? e : 0][i] = fetch(offn[e < T ? e : 0][i]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    // ---- partial tiles to LDS: part[buf][wave][tt][nt][row 4 kq + r][col m]
    float *const buf = lds + par * (NWAVE * T * NT * TILEF);
#pragma unroll
    for (int tt = 0; tt < T; ++tt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float *d = buf + ((wave * T + tt) * NT + nt) * TILEF;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          d[(4 * kq + r) * PITCH + m] = T * NT == 1 ? acc[tt][nt][0][r] + acc[tt][nt][1][r] : acc[tt][nt][0][r];
      }
    __syncthreads();
    // ---- sum over the waves (fixed order) + epilogue
    if (evalid) {
      const float *s = buf + (ett * NT + ent) * TILEF + emi * PITCH + (TB ? ecol : ecl);
      float4 sum = *reinterpret_cast<const float4 *>(s);
#pragma unroll
      for (int w2 = 1; w2 < NWAVE; ++w2) {
        const float4 q = *reinterpret_cast<const float4 *>(s + w2 * (T * NT * TILEF));
        sum.x += q.x; sum.y += q.y; sum.z += q.z; sum.w += q.w;
      }
      if constexpr (TB) {                               // + the next row's second block (taps k_x 2, 3 of this output voxel)
#pragma unroll
        for (int w2 = 0; w2 < NWAVE; ++w2) {
          const float4 q = *reinterpret_cast<const float4 *>(s + PITCH + 8 + w2 * (T * NT * TILEF));
          sum.x += q.x; sum.y += q.y; sum.z += q.z; sum.w += q.w;
        }
      }
      float vv[4] = {sum.x + __uint_as_float(aq.x), sum.y + __uint_as_float(aq.y), sum.z + __uint_as_float(aq.z),
                     sum.w + __uint_as_float(aq.w)};
      if (ep.bias) {                                  // kernel-uniform (before the skip-gradient add in tem_epilogue's order; sums commute)
#pragma unroll
        for (int c = 0; c < 4; ++c) vv[c] += ep.bias[ecol + c];
      }
      if (ep.gate) {                                  // kernel-uniform
        const float gg[4] = {__uint_as_float(gq.x), __uint_as_float(gq.y), __uint_as_float(gq.z), __uint_as_float(gq.w)};
#pragma unroll
        for (int c = 0; c < 4; ++c) vv[c] = gg[c] > 0.f ? vv[c] : ep.gate_slope * vv[c];
      }
      if (ep.slope != 1.f) {
#pragma unroll
        for (int c = 0; c < 4; ++c) vv[c] = vv[c] > 0.f ? vv[c] : ep.slope * vv[c];
      }
      *reinterpret_cast<float4 *>(p.out + (en * p.oN + eoz * p.oD + eoy * p.oH + eox * p.oW + ecol)) =
          make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
  };

  if constexpr (PREF) {
    // two fragment sets in ping-pong (no register copies: the loads of iteration i + 1 are in flight under iteration i;
    // a dead iteration's loads are all out of range and move no data)
    float4 a0[T][NA], a1[T][NA];
    load_it(a0, it0);
    for (int it = it0; it < it1; it += 2 * step) {
      compute(a0, a1, it, it + step, 0);
      if (it + step < it1) compute(a1, a0, it + step, it + 2 * step, 1);       // workgroup-uniform
    }
  } else {
    float4 a0[T][NA];
    int par = 0;
    for (int it = it0; it < it1; it += step, par ^= 1) {
      load_it(a0, it);
      compute(a0, a0, it, it, par);
    }
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

template <int CI, int NT, int T, bool TB = false>
static int run(Dev p, hipStream_t st, bool dry) {
  constexpr int NWAVE = CI / 2, VPT = TB ? 15 : 16;
  // every geometry check comes BEFORE the dry-run answer: tem_conv_is_tiled / Launch.meta must name the kernel the launch
  // really runs (a 260^3 model's g.d1b has tiles_pp = 1084 and falls through to the generic kernel)
  p.RL = TB ? p.OW + 1 : p.OW;
  p.plane_vox = p.OH * p.RL;
  p.tiles_pp = (p.plane_vox + VPT - 1) / VPT;
  const int64_t total = (int64_t)p.oN_count * p.OD * p.tiles_pp;
  if (total > (1 << 22) || p.tiles_pp > 1024 || p.OW > 1024) return TEM_EUNSUPPORTED;    // range of the magic divisions
  if (dry) {
    if (g_name) snprintf(g_name, g_name_len, "conv_s2_k<%d, %d, %d, %s>", CI, NT, T, TB ? "true" : "false");
    return TEM_OK;
  }
  p.total = (int)total;
  p.magicOW = magic_for(p.RL); p.magicTpp = magic_for(p.tiles_pp); p.magicOD = magic_for(p.OD);
  p.iters = (p.total + T - 1) / T;
  // as many workgroups as are resident at once (each keeps its B fragment for its whole contiguous range of tiles: the
  // ~3 us prologue -- B loads, first A loads -- is paid once per CU slot; more, shorter workgroups measured 15-25 % slower)
  static int mult = -1;
  if (mult < 0) mult = tem_env_int("TEM_S2_WGS", 1);
  const int ny = (p.CO + 16 * NT - 1) / (16 * NT);
  const int resident = 256 * (CI == 8 ? 2 : 1) * mult / ny;
  const int nblocks = p.iters < resident ? p.iters : resident;
  const size_t lds_bytes = (size_t)2 * NWAVE * T * NT * 16 * PITCH * 4;
  static bool attr = false;
  if (!attr && lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)conv_s2_k<CI, NT, T, TB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL((conv_s2_k<CI, NT, T, TB>), dim3((unsigned)nblocks, (unsigned)ny), dim3(NWAVE * 64), lds_bytes, st, p);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

static int64_t span_of(const tem_view &v) {
  return (int64_t)(v.N - 1) * v.sN + (int64_t)(v.D - 1) * v.sD + (int64_t)(v.H - 1) * v.sH + (int64_t)(v.W - 1) * v.sW + v.C;
}

static int dispatch(const tem_conv_args *a, hipStream_t st, bool dry) {
  const tem_view &i0 = a->in0, &o0 = a->out0;
  if (a->in1.ptr || a->out1.ptr || a->w_layout != TEM_W_TAP_CI_CO) return TEM_EUNSUPPORTED;
  if (a->kd != 4 || a->kh != 4 || a->kw != 4 || a->sd != 2 || a->sh != 2 || a->sw != 2) return TEM_EUNSUPPORTED;
  if (a->pd != a->ph || a->ph != a->pw) return TEM_EUNSUPPORTED;
  static int enabled = -1;
  if (enabled < 0) enabled = tem_env_int("TEM_CONV_S2", 1);
  if (!enabled) return TEM_EUNSUPPORTED;
  if (o0.N != i0.N) return TEM_ESHAPE;
  if (!fits32(i0) || !fits32(o0)) return TEM_EUNSUPPORTED;
  const int64_t in_span = span_of(i0);
  if (in_span >= ((int64_t)1 << 29)) return TEM_EUNSUPPORTED;            // byte offsets of the buffer loads stay below 2^31
  auto aligned = [](const tem_view &v) {
    return ((uintptr_t)v.ptr & 15) == 0 && v.sW % 4 == 0 && v.sH % 4 == 0 && v.sD % 4 == 0 && v.sN % 4 == 0;
  };
  if (!aligned(i0) || !aligned(o0) || o0.C % 4) return TEM_EUNSUPPORTED;
  const tem_epilogue &e = a->ep;
  if (e.dropout) return TEM_EUNSUPPORTED;                                // (no k4 s2 layer of the model sits under a Dropout)
  Dev p{};
  Ep &q = p.ep;
  q.bias = e.bias; q.slope = e.slope; q.gate_slope = e.gate_slope;
  if (e.gate.ptr) {
    const tem_view &g = e.gate;
    if (g.N != o0.N || g.D != o0.D || g.H != o0.H || g.W != o0.W || g.C < o0.C) return TEM_ESHAPE;
    if (!aligned(g) || span_of(g) >= ((int64_t)1 << 29)) return TEM_EUNSUPPORTED;
    q.gate = g.ptr; q.gN = (int)g.sN; q.gD = (int)g.sD; q.gH = (int)g.sH; q.gW = (int)g.sW; q.gbytes = (int)(span_of(g) * 4);
  }
  if (e.add.ptr) {
    const tem_view &ad = e.add;
    if (ad.C < o0.C || ad.N != o0.N) return TEM_ESHAPE;
    if (!aligned(ad) || span_of(ad) >= ((int64_t)1 << 29)) return TEM_EUNSUPPORTED;
    q.add = ad.ptr; q.aN = (int)ad.sN; q.aD = (int)ad.sD; q.aH = (int)ad.sH; q.aW = (int)ad.sW; q.abytes = (int)(span_of(ad) * 4);
    q.aoz = e.add_off[0]; q.aoy = e.add_off[1]; q.aox = e.add_off[2];
    q.aDd = ad.D; q.aHh = ad.H; q.aWw = ad.W;
  }
  p.in = i0.ptr; p.iN = (int)i0.sN; p.iD = (int)i0.sD; p.iH = (int)i0.sH; p.iW = (int)i0.sW;
  p.D = i0.D; p.H = i0.H; p.W = i0.W;
  p.w = a->w; p.in_bytes = (int)(in_span * 4);
  p.out = o0.ptr; p.oN = (int)o0.sN; p.oD = (int)o0.sD; p.oH = (int)o0.sH; p.oW = (int)o0.sW;
  p.OD = o0.D; p.OH = o0.H; p.OW = o0.W; p.CO = o0.C; p.oN_count = o0.N;
  p.P = a->pd;
  { static int dbg = -1; if (dbg < 0) dbg = tem_env_int("TEM_DEBUG_FLAGS", 0); p.dbg = dbg; }
  p.plane_vox = o0.H * o0.W;
  const int CI = i0.C, CO = o0.C;
  static int tb = -1;
  if (tb < 0) tb = tem_env_int("TEM_S2_TB", 1);
  if (tb && CI == 8 && CO == 8) return run<8, 1, 4, true>(p, st, dry);           // g.d1b, d.d1b: two-block form
  if (CI == 8 && (CO == 8 || CO == 16)) return run<8, 1, 2>(p, st, dry);     // 8 -> 8 without the two-block form (fallback); input-gradient of g.u1b
  if (CI == 16 && CO == 16) return run<16, 1, 2>(p, st, dry);                // g.d2b
  if (CI == 16 && CO == 32) return run<16, 2, 2>(p, st, dry);                // input-gradient of g.u2b
  if (CI == 32 && CO == 32) return run<32, 1, 1>(p, st, dry);                // d.d2b, d.d3b (two workgroups per tile: one per 16 output channels)
  return TEM_EUNSUPPORTED;
}

}  // namespace convs2

// Called by tem_conv (dispatch.hip) ahead of the LDS-ring kernel.
int tem_conv_s2_try(const tem_conv_args *a, hipStream_t st, bool dry) { return convs2::dispatch(a, st, dry); }

int tem_conv_s2_describe(const tem_conv_args *a, char *buf, int len) {
  convs2::g_name = buf; convs2::g_name_len = len;
  int rc = convs2::dispatch(a, nullptr, true);
  convs2::g_name = nullptr;
  return rc;
}
