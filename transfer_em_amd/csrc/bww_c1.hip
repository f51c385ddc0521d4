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
  int32_t gspan;                   // bytes one sample of g spans
  float *slabs; int64_t slab_stride;
};

template <int CO>
__global__ __launch_bounds__(256) void bww_c1_k(C1Dev p) {
  constexpr int NQ = CO / 4;                                  // lanes per voxel (one channel quad each)
  constexpr int PX = 16, PY = 256 / NQ / PX;                  // patch of voxel columns per workgroup: 8 x 16 (CO 8), 4 x 16 (CO 16)
  constexpr int XP = PX + 2, YP = PY + 2;
  constexpr int RING = 8;                                     // planes of G in flight per lane (LDS-DMA ring, 16 bytes per lane and plane)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *const gring = smem;                                  // [RING][256 lanes][4]
  float *const xs = smem + RING * 256 * 4;                    // [zper + 2][YP][XP]: the 1-channel patch of the whole run
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

  // ---- stage X[z0 - P .. z0 - P + nz + 1][by*PY - P ..][bx*PX - P ..] (zeros outside the tensor): clamped addresses and
  // batches of 8 loads in flight (a load under a divergent branch would be waited for one by one)
  {
    const int total = (nz + 2) * YP * XP;
    const float *xn = p.x + (size_t)n * p.xN;
    for (int i0 = tid; i0 < total; i0 += 8 * 256) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = min(i0 + k * 256, total - 1);
        const int zz = i / (YP * XP), r = i - zz * (YP * XP), yy = r / XP, xx = r - yy * XP;
        const int iz = z0 - p.P + zz, iy = by * PY - p.P + yy, ix = bx * PX - p.P + xx;
        const bool ok = (unsigned)iz < (unsigned)p.D && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const float t = xn[ok ? iz * p.xD + iy * p.xH + ix * p.xW : 0];
        v[k] = ok ? t : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (i0 + k * 256 < total) xs[i0 + k * 256] = v[k];
    }
  }
  __syncthreads();

  float acc[27][4];
#pragma unroll
  for (int t = 0; t < 27; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = 0.f;

  // G streams through a per-lane LDS-DMA ring RING planes deep: every lane fetches the 16 bytes of ITS (voxel, channel
  // quad) of plane z + RING while it works on plane z -- HBM latency (~2 us) is covered without a register per plane in
  // flight, and a lane only ever reads what it fetched itself: no barrier in the march, just a counted vmcnt wait.
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.g + (size_t)n * p.gN), 0, p.gspan, 0x00020000);
  const int goff = vok ? (oy * p.gH + ox * p.gW + 4 * cq) * 4 : (int)0x80000000;
  const float *xl = xs + ly * XP + lx;
  auto dma_g = [&](int z) {                                  // plane z of the lane's chunk -> ring slot z % RING (zeros past the run)
    const int off = z < z1 ? goff + z * p.gD * 4 : (int)0x80000000;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(grs, (__attribute__((address_space(3))) void *)(gring + ((z - z0) % RING) * 1024 + wave * 256),
                                             16, off, 0, 0, 0);
  };
  typedef float f4 __attribute__((ext_vector_type(4)));
  const uint32_t gaddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)(gring + tid * 4);
  // The lane's chunk of a plane is read one step AHEAD of its use (read_issue at the head of step z for plane z + 1,
  // read_done at its tail): the LDS round trip runs under the step's 108 FMAs instead of in front of them.  Counted
  // wait + read in ONE asm statement: as a plain LDS load hipcc would put vmcnt(0) in front of it (it cannot count the
  // DMA) and drain the ring every plane.  vmcnt(N): all but the newest N fetches have landed.
  auto read_first = [&](f4 &v, int z) {                     // plane z0: all RING fetches issued, the oldest must have landed
    asm volatile("s_waitcnt vmcnt(%2)\n\tds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v) : "v"(gaddr + (uint32_t)(((z - z0) % RING) * 4096)), "n"(RING - 1) : "memory");
  };
  auto read_issue = [&](f4 &v, int z) {                     // plane z = current + 1: fetches current + 2 .. current + RING - 1 may be in flight
    asm volatile("s_waitcnt vmcnt(%2)\n\tds_read_b128 %0, %1"
                 : "=&v"(v) : "v"(gaddr + (uint32_t)(((z - z0) % RING) * 4096)), "n"(RING - 2) : "memory");
  };
  auto read_done = [&](f4 &v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) : : "memory"); };
#pragma unroll
  for (int k = 0; k < RING; ++k) dma_g(z0 + k);
  auto load_w = [&](float (&w)[9], int zz) {                  // the 3x3 X neighbours of the lane's column in staged plane zz
    const float *pl = xl + zz * (YP * XP);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) w[dy * 3 + dx] = pl[dy * XP + dx];
  };
  auto fma_plane = [&](const float (&w)[9], int dz, const float4 &g) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      acc[dz * 9 + t][0] = fmaf(w[t], g.x, acc[dz * 9 + t][0]);
      acc[dz * 9 + t][1] = fmaf(w[t], g.y, acc[dz * 9 + t][1]);
      acc[dz * 9 + t][2] = fmaf(w[t], g.z, acc[dz * 9 + t][2]);
      acc[dz * 9 + t][3] = fmaf(w[t], g.w, acc[dz * 9 + t][3]);
    }
  };
  // window of three X planes in registers, rotated by a 3x unrolled march (no register moves): output plane zi multiplies
  // staged planes zi, zi + 1, zi + 2 (taps dz = 0, 1, 2) with G[z0 + zi]; G is fetched one plane ahead
  float w0[9], w1[9], w2[9];
  load_w(w0, 0); load_w(w1, 1);
  f4 ga, gb, gc;
  read_first(ga, z0);
  auto G4 = [](const f4 &v) { return make_float4(v.x, v.y, v.z, v.w); };
  for (int zi = 0; zi < nz; zi += 3) {
    {
      read_issue(gb, z0 + zi + 1); dma_g(z0 + zi + RING);
      load_w(w2, zi + 2);
      const float4 g = G4(ga);
      fma_plane(w0, 0, g); fma_plane(w1, 1, g); fma_plane(w2, 2, g);
      read_done(gb);
    }
    if (zi + 1 < nz) {
      read_issue(gc, z0 + zi + 2); dma_g(z0 + zi + 1 + RING);
      load_w(w0, zi + 3);
      const float4 g = G4(gb);
      fma_plane(w1, 0, g); fma_plane(w2, 1, g); fma_plane(w0, 2, g);
      read_done(gc);
    }
    if (zi + 2 < nz) {
      read_issue(ga, z0 + zi + 3); dma_g(z0 + zi + 2 + RING);
      load_w(w1, zi + 4);
      const float4 g = G4(gc);
      fma_plane(w2, 0, g); fma_plane(w0, 1, g); fma_plane(w1, 2, g);
      read_done(ga);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the ring's tail fetches (zeros) before the LDS is reused

  // ---- sum over the lanes that share a channel quad.  Within a row of 16 lanes: DPP row shifts by multiples of NQ (they keep
  // lane % NQ; vector-pipe speed -- as a __shfl_xor butterfly this was 540 ds_bpermute + 206 waits per wave, about as long as
  // the whole march); the last NQ lanes of each row then hold the row's sums.  Rows and waves: through LDS, fixed order.
  __syncthreads();                                            // the LDS is reused as the cross-row / cross-wave buffer
#pragma unroll
  for (int t = 0; t < 27; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = acc[t][c];
      // row_shr:o for o = NQ, 2 NQ, .. 8 (zeros shifted in)
      if (NQ <= 2) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
      acc[t][c] = v;
    }
  float *red = smem;                                          // [4 waves x 4 rows][27][CO]
  if ((lane & 15) >= 16 - NQ) {
    const int part = wave * 4 + (lane >> 4), q4 = (lane & 15) - (16 - NQ);
#pragma unroll
    for (int t = 0; t < 27; ++t)
      *reinterpret_cast<float4 *>(red + (part * 27 + t) * CO + 4 * q4) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
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

static int run(const tem_bww_args *a, hipStream_t st, bool dry, int *nslab_out, char *name, int name_len) {
  const tem_view &x = a->in0, &g = a->dout;
  const bool cube = a->kd == 3 && a->kh == 3 && a->kw == 3 && a->sd == 1 && a->sh == 1 && a->sw == 1 && a->pd == a->ph &&
                    a->ph == a->pw && a->pd >= 0;
  if (!cube || x.C != 1 || a->in1.ptr || (g.C != 8 && g.C != 16) || x.D < 2 || x.N != g.N) return TEM_EUNSUPPORTED;
  // (dout may be any window of the layer's output: voxel o reads x[o + tap - p], zeros outside x -- the region-restricted
  // cycle path passes such windows)
  auto span = [](const tem_view &v) {
    return (int64_t)(v.N - 1) * v.sN + (int64_t)(v.D - 1) * v.sD + (int64_t)(v.H - 1) * v.sH + (int64_t)(v.W - 1) * v.sW + v.C;
  };
  if (span(x) >= ((int64_t)1 << 31) || span(g) >= ((int64_t)1 << 31)) return TEM_EUNSUPPORTED;
  if (((uintptr_t)g.ptr & 15) || g.sW % 4 || g.sH % 4 || g.sD % 4 || g.sN % 4) return TEM_EUNSUPPORTED;
  const int CO = g.C, NQ = CO / 4, PY = 256 / NQ / 16;
  C1Dev p{};
  p.x = x.ptr; p.xN = (int)x.sN; p.xD = (int)x.sD; p.xH = (int)x.sH; p.xW = (int)x.sW;
  p.D = x.D; p.H = x.H; p.W = x.W;
  p.g = g.ptr; p.gN = (int)g.sN; p.gD = (int)g.sD; p.gH = (int)g.sH; p.gW = (int)g.sW;
  p.OD = g.D; p.OH = g.H; p.OW = g.W;
  p.P = a->pd;
  p.nyb = (g.H + PY - 1) / PY; p.nxb = (g.W + 15) / 16;
  // z runs: one round of workgroups (two per CU at 219 VGPRs: 512), long runs -- the per-workgroup costs (staging, ring
  // fill at HBM latency, the 108-value butterfly) are ~8 us; at most the caller's slab budget
  const int cols = g.N * p.nyb * p.nxb;
  const int cap = a->nslab > 0 ? a->nslab : 1024;
  int zsegs = std::max(1, std::min(std::min(cap / cols, 512 / cols), std::max(1, g.D / 8)));
  if (cols > cap) return TEM_EUNSUPPORTED;
  int zper = (g.D + zsegs - 1) / zsegs;
  zsegs = (g.D + zper - 1) / zper;
  p.zsegs = zsegs; p.zper = zper;
  const int nblocks = cols * zsegs;
  const size_t lds = 8 * 256 * 16 + (size_t)(zper + 2) * (PY + 2) * 18 * 4;       // G ring + X patch (>= the 4 x 27 x CO reduction buffer)
  if (lds > 96 * 1024) return TEM_EUNSUPPORTED;
  // bytes one sample of g spans, from the view itself (a batch-1 view may carry any sample stride, 0 included)
  const int64_t gspan = ((int64_t)(g.D - 1) * g.sD + (int64_t)(g.H - 1) * g.sH + (int64_t)(g.W - 1) * g.sW + g.C) * 4;
  if (gspan > (int64_t)0x7fffffff) return TEM_EUNSUPPORTED;
  p.gspan = (int)gspan;
  if (nslab_out) *nslab_out = nblocks;
  if (name) snprintf(name, name_len, "bww_c1_k<%d>", CO);
  if (dry) return TEM_OK;
  if (!a->slabs || a->nslab != nblocks || a->accumulate) return TEM_EINVAL;
  p.slabs = a->slabs; p.slab_stride = a->slab_stride ? a->slab_stride : (int64_t)27 * CO;
  static bool attr[2] = {false, false};
  if (!attr[CO == 16]) {
    hipError_t e = CO == 8 ? hipFuncSetAttribute((const void *)bww_c1_k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)
                           : hipFuncSetAttribute((const void *)bww_c1_k<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return (int)e;
    attr[CO == 16] = true;
  }
  if (CO == 8) hipLaunchKernelGGL(bww_c1_k<8>, dim3(nblocks), dim3(256), lds, st, p);
  else hipLaunchKernelGGL(bww_c1_k<16>, dim3(nblocks), dim3(256), lds, st, p);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

}  // namespace bwwc1

// Same contract as tem_bww_lds_try (conv_bww.hip): dry = only report support and the slab count.
int tem_bww_c1_try(const tem_bww_args *a, hipStream_t st, bool dry, int *nslab_out) {
  return bwwc1::run(a, st, dry, nslab_out, nullptr, 0);
}

int tem_bww_c1_describe(const tem_bww_args *a, char *buf, int len) { return bwwc1::run(a, nullptr, true, nullptr, buf, len); }
