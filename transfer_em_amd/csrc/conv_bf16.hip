// conv_bf16.hip -- convolution forward / input-gradient for the bf16 mixed-precision mode (BASELINE config 5):
// bf16 activations and kernel copies, fp32 accumulation on v_mfma_f32_16x16x32_bf16, fp32 epilogue, bf16 stores.
//
//   out[o][co] = epilogue( sum_{tap,ci} X[o*S + tap - P][ci] * W(tap,ci,co) )
//
// With bf16 operands the matrix pipe runs 16x the fp32 rate, so EVERY layer is bound by bytes: one
// shape-generic kernel serves all of them, built around data movement rather than MFMA packing:
//   * a workgroup owns (n, one output plane, a TX x TY patch of it); it loads the K-plane input patch ONCE into LDS
//     (16-byte coalesced loads, all issued before the first LDS write; concat inputs gathered by channel, zero
//     padding / cropping by the bounds check) -- z/y/x neighbours re-read their halos from L2;
//   * the kernel taps of the launch are staged in LDS as ready-made B fragments ([k-step][n-tile][lane] x 8 bf16),
//     read back with conflict-free ds_read_b128 and shared by the two tiles a wave keeps in flight;
//   * A fragments are gathered from the channels-last patch (voxel pitch 2*C_in + 16 bytes: conflict-free
//     ds_read_b128 of 8 channels): MFMA k index = (tap, ci), a lane's 8 k-values are 8 channels of one tap;
//     C_in == 1 uses the taps themselves as K (27 -> one k-step);
//   * narrow outputs (C_out 1 or 8) simply leave MFMA columns empty -- irrelevant at this arithmetic intensity;
//   * fused epilogue as in the fp32 kernels (bias, skip-gradient add, LeakyReLU' gate, Philox dropout with keep-mask
//     write / read, LeakyReLU, split outputs), 8-byte bf16 stores of 4 channels per lane.
#include "tem_common.h"
#include <cstdio>
#include <cstdlib>

namespace conv_bf16 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }   // round to nearest even

// x / d for 0 <= x < 2^31 with magic = ceil(2^32 / d) (d == 1: the magic does not fit 32 bits)
__device__ __forceinline__ int fdiv(int x, int d, uint32_t magic) { return d == 1 ? x : (int)__umulhi((uint32_t)x, magic); }

struct Ep {
  const float *bias;
  float slope;
  const u16 *gate; int32_t gN, gD, gH, gW; float gate_slope;
  const u16 *add;  int32_t aN, aD, aH, aW, aoz, aoy, aox, aDd, aHh, aWw;
  int32_t dropout;
  DropoutStream ds;
  const uint32_t *step_dev;
  int32_t doz, doy, dox, dD, dH, dW;
  uint8_t *keep_mask;
  int32_t keep_mode;
  int32_t gbytes, abytes, mbytes;  // extents (bytes) of the gate / add views and of the keep mask: buffer ranges
};

struct Dev {
  const u16 *in0, *in1;
  int32_t i0N, i0D, i0H, i0W, i1N, i1D, i1H, i1W, C0;
  int32_t D, H, W;
  const u16 *w;                    // packed bf16 kernel [tap][co][ci]
  int32_t flip;                    // taps reversed (input-gradient of a stride-1 convolution)
  u16 *out0, *out1;
  int32_t o0N, o0D, o0H, o0W, o1N, o1D, o1H, o1W, CO0;
  int32_t OD, OH, OW, P;
  int32_t TX, TY, nbx, nby;        // output patch, patches per plane
  int32_t cols, rows;              // input patch extents (voxels)
  uint32_t magicCols, magicPlane, magicTX;
  Ep ep;
};

// BLDS: kernel taps staged in LDS as B fragments (large layers); false: read from the packed kernel in L2 per k-step
// (the two deep k4 s2 32 -> 32 layers of the discriminators: 128 KB of fragments, a few thousand voxels)
template <int CI, int CO, int K, int S, int PF, bool BLDS>
__global__ __launch_bounds__(256) void conv_bf16_k(Dev p) {
  constexpr int NTAP = K * K * K, KTOT = NTAP * CI, NSTEP = (KTOT + 31) / 32, NT = (CO + 15) / 16;
  constexpr int WPN = NT >= 4 ? 1 : 4 / NT;                // waves per n-tile
  constexpr int NTW = NT >= 4 ? NT / 4 : 1;               // n-tiles per wave (C_out 64 would need 1; kept general)
  constexpr int PITCH = CI >= 8 ? CI : 1;                 // LDS voxel pitch in bf16 elements (un-padded: lds_b128_probe3.hip)
  constexpr int CPV = CI >= 8 ? CI / 8 : 1;               // 16-byte chunks per voxel
  constexpr int TPITCH = 20;
  static_assert(CI == 1 || CI % 8 == 0, "C_in 1 or a multiple of 8");
  static_assert(NTW == 1, "up to 64 output channels");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const int plane = p.rows * p.cols;

  // LDS carve: patch | B fragments | gather table | per-wave transpose patches
  u16 *patch = reinterpret_cast<u16 *>(smem);
  const int patch_bytes = ((K * plane * PITCH * 2) + 15) & ~15;
  u16 *Bl = reinterpret_cast<u16 *>(smem + patch_bytes);
  constexpr int B_BYTES = BLDS ? NSTEP * NT * 64 * 16 : 0;
  int *tab = reinterpret_cast<int *>(smem + patch_bytes + B_BYTES);
  constexpr int TAB_INTS = CI == 1 ? 32 : (BLDS ? 1 : 2) * ((NSTEP * 4 + 3) & ~3);      // !BLDS: + kernel offsets
  float *tp = reinterpret_cast<float *>(smem + patch_bytes + B_BYTES + TAB_INTS * 4) + wave * (16 * TPITCH);

  int b = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);     // z-neighbours of a patch share K-S input planes: one L2
  const int oz = b % p.OD; b /= p.OD;
  const int bx = b % p.nbx; b /= p.nbx;
  const int by = b % p.nby;
  const int n = b / p.nby;
  const int ox0 = bx * p.TX, oy0 = by * p.TY;
  const int TXo = min(p.TX, p.OW - ox0), TYo = min(p.TY, p.OH - oy0);

  // ---- B fragments of the whole launch -> LDS: slot (s, nt, l) holds B[k = 32 s + 8 (l>>4) + j][col = nt*16 + (l&15)]
  if constexpr (BLDS) {
    auto bfrag = [&](int idx) -> uint4 {
      const int l = idx & 63, snt = idx >> 6, nt = snt % NT, s = snt / NT;
      const int co = nt * 16 + (l & 15), e0 = 32 * s + 8 * (l >> 4);
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (idx < NSTEP * NT * 64 && co < CO) {
        if constexpr (CI >= 8) {
          const int tap = e0 / CI, c0 = e0 - tap * CI;
          if (tap < NTAP) v = *reinterpret_cast<const uint4 *>(p.w + ((p.flip ? NTAP - 1 - tap : tap) * CO + co) * CI + c0);
        } else {
          u16 h[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int tap = e0 + j;
            h[j] = tap < NTAP ? p.w[(p.flip ? NTAP - 1 - tap : tap) * CO + co] : (u16)0;
          }
          v = make_uint4(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16), h[4] | ((uint32_t)h[5] << 16),
                         h[6] | ((uint32_t)h[7] << 16));
        }
      }
      return v;
    };
    for (int base = tid; base < NSTEP * NT * 64; base += 4 * 256) {      // 4 loads in flight per thread
      uint4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = bfrag(base + i * 256);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (base + i * 256 < NSTEP * NT * 64) *reinterpret_cast<uint4 *>(Bl + (size_t)(base + i * 256) * 8) = v[i];
    }
  }
  // ---- gather table: patch offset (bf16 elements) of k-range e0 = 32 s + 8 kq  ->  (tap, first channel)
  if constexpr (CI >= 8) {
    for (int i = tid; i < NSTEP * 4; i += 256) {
      const int e0 = 32 * (i >> 2) + 8 * (i & 3);
      int tap = e0 / CI;
      const int c0 = e0 - tap * CI;
      if (tap >= NTAP) tap = 0;                            // padded k-range (its B is zero): any finite operand
      const int dz = tap / (K * K), rem = tap - dz * (K * K), dy = rem / K, dx = rem - dy * K;
      tab[i] = ((dz * p.rows + dy) * p.cols + dx) * PITCH + c0;
      if constexpr (!BLDS) {                               // element offset of (tap, co = 0, c0) in the packed kernel, -1: padded
        const int t0 = e0 / CI;
        tab[((NSTEP * 4 + 3) & ~3) + i] = t0 < NTAP ? ((p.flip ? NTAP - 1 - t0 : t0) * CO) * CI + c0 : -1;
      }
    }
  } else {
    if (tid < 32) {
      const int tap = tid < NTAP ? tid : 0;
      const int dz = tap / (K * K), rem = tap - dz * (K * K), dy = rem / K, dx = rem - dy * K;
      tab[tid] = (dz * p.rows + dy) * p.cols + dx;
    }
  }

  // ---- input patch: K planes x rows x cols voxels; zeros outside the input (padding, cropped views, volume border)
  {
    const int iz0 = oz * S - p.P, iy0 = oy0 * S - p.P, ix0 = ox0 * S - p.P;
    if constexpr (CI >= 8) {
      const int total = K * plane * CPV;
      uint4 pf[PF];
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int id = tid + i * 256;
        const int vox = id / CPV, c = (id - vox * CPV) * 8;
        const int pl = fdiv(vox, plane, p.magicPlane), r2 = vox - pl * plane;
        const int r = fdiv(r2, p.cols, p.magicCols), cx = r2 - r * p.cols;
        const int iz = iz0 + pl, iy = iy0 + r, ix = ix0 + cx;
        const bool ok = id < total && (unsigned)iz < (unsigned)p.D && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const u16 *src = c < p.C0 ? p.in0 + (n * p.i0N + iz * p.i0D + iy * p.i0H + ix * p.i0W + c)
                                  : p.in1 + (n * p.i1N + iz * p.i1D + iy * p.i1H + ix * p.i1W + (c - p.C0));
        pf[i] = ok ? *reinterpret_cast<const uint4 *>(src) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int id = tid + i * 256;
        if (id < total) {
          const int vox = id / CPV, c = (id - vox * CPV) * 8;
          *reinterpret_cast<uint4 *>(patch + vox * PITCH + c) = pf[i];
        }
      }
    } else {
      const int total = K * plane;
      u16 pf[PF];
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int id = tid + i * 256;
        const int pl = fdiv(id, plane, p.magicPlane), r2 = id - pl * plane;
        const int r = fdiv(r2, p.cols, p.magicCols), cx = r2 - r * p.cols;
        const int iz = iz0 + pl, iy = iy0 + r, ix = ix0 + cx;
        const bool ok = id < total && (unsigned)iz < (unsigned)p.D && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        pf[i] = ok ? p.in0[n * p.i0N + iz * p.i0D + iy * p.i0H + ix * p.i0W] : (u16)0;
      }
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int id = tid + i * 256;
        if (id < total) patch[id] = pf[i];
      }
    }
  }
  __syncthreads();

  const int nt = wave % NT;
  const int ti = lane >> 2, tcq = lane & 3;               // transposed role: voxel of the tile, channel quad
  const int eco = nt * 16 + tcq * 4;                      // first of this lane's 4 output channels
  const int L = TYo * p.TX, ntiles = (L + 15) >> 4;      // tiles run across row ends of the (full-width) patch
  const Ep &ep = p.ep;
  DropoutStream ds = ep.ds;
  if (ep.dropout && ep.step_dev) ds.step = *ep.step_dev;
  const bool first = eco < p.CO0;                         // routed to out0 (epilogue applies) or out1 (raw)

  auto a_base = [&](int t) -> int {
    const int v = min(t * 16 + m, L - 1);                 // lanes past the patch recompute a voxel inside it, never stored
    const int r = fdiv(v, p.TX, p.magicTX), x = min(v - r * p.TX, TXo - 1);
    return (r * S * p.cols + x * S) * PITCH;
  };
  auto gather = [&](int base, int s) -> bf16x8 {
    if constexpr (CI >= 8) {
      return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(patch + base + tab[s * 4 + kq]));
    } else {
      u16 h[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = patch[base + tab[8 * kq + j]];
      return __builtin_bit_cast(bf16x8, make_uint4(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16),
                                                   h[4] | ((uint32_t)h[5] << 16), h[6] | ((uint32_t)h[7] << 16)));
    }
  };

  // (gate / skip-gradient / keep-byte loads through buffer descriptors: a lane without an output voxel, a voxel outside the
  // skip-gradient window or an absent tensor is an out-of-range offset = zeros.  As conditional plain loads the merge of
  // loaded value and default sat right behind the load -- s_waitcnt vmcnt(0) IN FRONT of the MFMA chain, a full memory
  // latency per tile pair at 16x the fp32 matrix rate.)
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)ep.gate, 0, ep.gate ? ep.gbytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void *)ep.add, 0, ep.add ? ep.abytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void *)ep.keep_mask, 0, ep.keep_mode == 2 ? ep.mbytes : 0, 0x00020000);
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  struct Prep { int oy, ox; bool valid; uint2 g, a; uint32_t kb; };
  auto prep = [&](int t) -> Prep {
    Prep q;
    const int v = t * 16 + ti;
    const int r = fdiv(v, p.TX, p.magicTX), x = v - r * p.TX;
    q.oy = oy0 + r; q.ox = ox0 + x;
    q.valid = v < L && x < TXo && eco < CO;
    const bool vf = q.valid && first;
    int goff = vf ? (n * ep.gN + oz * ep.gD + q.oy * ep.gH + q.ox * ep.gW + eco) * 2 : (int)0x80000000;
    asm volatile("" : "+v"(goff));
    const u32x2 g = __builtin_amdgcn_raw_buffer_load_b64(grs, goff, 0, 0);
    q.g = make_uint2(g.x, g.y);
    const int az = oz - ep.aoz, ay = q.oy - ep.aoy, ax = q.ox - ep.aox;
    const bool ain = vf && (unsigned)az < (unsigned)ep.aDd && (unsigned)ay < (unsigned)ep.aHh && (unsigned)ax < (unsigned)ep.aWw;
    int aoff = ain ? (n * ep.aN + az * ep.aD + ay * ep.aH + ax * ep.aW + eco) * 2 : (int)0x80000000;
    asm volatile("" : "+v"(aoff));
    const u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(ars, aoff, 0, 0);
    q.a = make_uint2(a.x, a.y);
    const uint32_t e3 = (uint32_t)((((((uint64_t)n * ep.dD + (oz + ep.doz)) * ep.dH + (q.oy + ep.doy)) * ep.dW + (q.ox + ep.dox)) * (uint64_t)p.CO0 + eco) >> 3);
    int moff = vf ? (int)e3 : (int)0x80000000;
    asm volatile("" : "+v"(moff));
    q.kb = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(mrs, moff, 0, 0);
    return q;
  };
  auto finish = [&](const f32x4 &acc, const Prep &q) {
#pragma unroll
    for (int r = 0; r < 4; ++r) tp[(kq * 4 + r) * TPITCH + m] = acc[r];
    __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): this wave's own LDS writes have landed
    const float4 v4 = *reinterpret_cast<const float4 *>(tp + ti * TPITCH + tcq * 4);
    float vv[4] = {v4.x, v4.y, v4.z, v4.w};
    const int oy = q.oy, ox = q.ox;
    const bool valid = q.valid;
    if (first) {
      if (ep.bias) {
#pragma unroll
        for (int c = 0; c < 4; ++c) vv[c] += (eco + c < CO) ? ep.bias[eco + c] : 0.f;
      }
      vv[0] += bf2f((u16)(q.a.x & 0xffffu)); vv[1] += bf2f((u16)(q.a.x >> 16));
      vv[2] += bf2f((u16)(q.a.y & 0xffffu)); vv[3] += bf2f((u16)(q.a.y >> 16));
      if (ep.gate) {
        vv[0] = bf2f((u16)(q.g.x & 0xffffu)) > 0.f ? vv[0] : ep.gate_slope * vv[0];
        vv[1] = bf2f((u16)(q.g.x >> 16)) > 0.f ? vv[1] : ep.gate_slope * vv[1];
        vv[2] = bf2f((u16)(q.g.y & 0xffffu)) > 0.f ? vv[2] : ep.gate_slope * vv[2];
        vv[3] = bf2f((u16)(q.g.y >> 16)) > 0.f ? vv[3] : ep.gate_slope * vv[3];
      }
      if (ep.dropout) {                                    // kernel-uniform; C_out0 a multiple of 8 (host)
        const uint64_t e = ((((uint64_t)n * ep.dD + (oz + ep.doz)) * ep.dH + (oy + ep.doy)) * ep.dW + (ox + ep.dox)) * (uint64_t)p.CO0 + eco;
        uint32_t bits;
        if (ep.keep_mode == 2) {
          bits = (q.kb >> (uint32_t)(e & 4u)) & 15u;          // (fetched by prep; zero for lanes without a voxel)
        } else {
          const Philox128 ph = ds.block(e >> 7);
          const uint32_t eb = (uint32_t)(e & 127);
          bits = 0;
#pragma unroll
          for (int c = 0; c < 4; ++c) bits |= (DropoutStream::bit(ph, eb + c) ? 1u : 0u) << c;
          if (ep.keep_mode == 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)bits, 1, 64);
            if (valid && !(tcq & 1)) ep.keep_mask[e >> 3] = (uint8_t)(bits | (other << 4));
          }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) vv[c] = ((bits >> c) & 1u) ? 2.f * vv[c] : 0.f;
      }
      if (ep.slope != 1.f) {
#pragma unroll
        for (int c = 0; c < 4; ++c) vv[c] = vv[c] > 0.f ? vv[c] : ep.slope * vv[c];
      }
    }
    if (valid) {
      u16 *o = first ? p.out0 + (n * p.o0N + oz * p.o0D + oy * p.o0H + ox * p.o0W + eco)
                     : p.out1 + (n * p.o1N + oz * p.o1D + oy * p.o1H + ox * p.o1W + (eco - p.CO0));
      if constexpr (CO % 4 == 0) {
        *reinterpret_cast<uint2 *>(o) = make_uint2(f2bf(vv[0]) | ((uint32_t)f2bf(vv[1]) << 16),
                                                   f2bf(vv[2]) | ((uint32_t)f2bf(vv[3]) << 16));
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (eco + c < CO) o[c] = f2bf(vv[c]);
      }
    }
  };

  const u16 *Bw = Bl + (size_t)(nt * 64 + lane) * 8;       // + s * NT * 64 * 8
  for (int t = wave / NT; t < ntiles; t += 2 * WPN) {     // wave-uniform
    const int t2 = t + WPN;
    const bool two = t2 < ntiles;
    const int b0 = a_base(t), b1 = a_base(two ? t2 : t);
    const Prep q0 = prep(t), q1 = prep(two ? t2 : t);
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (!BLDS && CI == 32) {
      // 32 input channels, kernel read from L2: k-step s IS tap s, so neither the patch offset nor the kernel offset needs the
      // LDS table (a dependent LDS read in front of every gather and every kernel load), and the kernel fragments of 16 taps
      // are requested in one go, one batch ahead of the matrix work.  [As a table-driven loop unrolled by 4 the 64 k-steps
      // were 16 dependent L2 round trips: 27 us for the 8^3 x 32 output of the discriminators' last strided layer, and
      // the same 28 us for the 29^3 one.]
      constexpr int DEPTH = 16, NB = (NSTEP + DEPTH - 1) / DEPTH;
      static_assert(NSTEP == NTAP, "one k-step per tap");
      const int cols32 = p.cols * PITCH, plane32 = plane * PITCH;
      const u16 *wl = p.w + (nt * 16 + m) * CI + 8 * kq;   // + tap * CO * CI   (C_out a multiple of 16: no padded column)
      uint4 bq[2][DEPTH];
      auto load_b = [&](uint4 (&dst)[DEPTH], int s0) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
          const int tap = p.flip ? NTAP - 1 - (s0 + i) : s0 + i;
          if (s0 + i < NSTEP) dst[i] = *reinterpret_cast<const uint4 *>(wl + tap * (CO * CI));
        }
      };
      load_b(bq[0], 0);
#pragma unroll
      for (int bt = 0; bt < NB; ++bt) {
        if (bt + 1 < NB) load_b(bq[(bt + 1) & 1], (bt + 1) * DEPTH);
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
          const int tap = bt * DEPTH + i;
          if (tap >= NSTEP) continue;
          const int dz = tap / (K * K), dy = (tap / K) % K, dx = tap % K;
          const int off = dz * plane32 + dy * cols32 + dx * PITCH + 8 * kq;
          const bf16x8 a0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(patch + b0 + off));
          const bf16x8 a1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(patch + b1 + off));
          const bf16x8 bf = __builtin_bit_cast(bf16x8, bq[bt & 1][i]);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bf, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bf, acc1, 0, 0, 0);
        }
      }
    } else
#pragma unroll 4
    for (int s = 0; s < NSTEP; ++s) {
      bf16x8 bf;
      if constexpr (BLDS) {
        bf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(Bw + (size_t)s * (NT * 64 * 8)));
      } else {
        const int wo = tab[((NSTEP * 4 + 3) & ~3) + s * 4 + kq];
        const int co = nt * 16 + m;
        bf = __builtin_bit_cast(bf16x8, (wo >= 0 && co < CO) ? *reinterpret_cast<const uint4 *>(p.w + wo + co * CI)
                                                              : make_uint4(0u, 0u, 0u, 0u));
      }
      const bf16x8 a0 = gather(b0, s), a1 = gather(b1, s);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bf, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bf, acc1, 0, 0, 0);
    }
    finish(acc0, q0);
    if (two) finish(acc1, q1);
  }
}

// ------------------------------------------------------------------------------------------ host
static uint32_t magic_for(int d) { return d <= 1 ? 0u : (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d); }

static bool fits32(const tem_view &v) {
  int64_t span = (int64_t)(v.N - 1) * v.sN + (int64_t)(v.D - 1) * v.sD + (int64_t)(v.H - 1) * v.sH +
                 (int64_t)(v.W - 1) * v.sW + v.C;
  return span < ((int64_t)1 << 31);
}

static thread_local char *g_name = nullptr;
static thread_local int g_name_len = 0;

template <int CI, int CO, int K, int S, int PF, bool BLDS>
int run(Dev p, int N, hipStream_t st, bool dry) {
  constexpr int NTAP = K * K * K, NSTEP = (NTAP * CI + 31) / 32, NT = (CO + 15) / 16;
  constexpr int PITCH = CI >= 8 ? CI : 1, CPV = CI >= 8 ? CI / 8 : 1;
  constexpr size_t B_BYTES = BLDS ? (size_t)NSTEP * NT * 64 * 16 : 0;
  constexpr size_t TAB_BYTES = (CI == 1 ? 32 : (BLDS ? 1 : 2) * ((NSTEP * 4 + 3) & ~3)) * 4;
  constexpr size_t FIXED = B_BYTES + TAB_BYTES + 4 * 16 * 20 * 4;
  // output patch (TX x TY): the largest that fits ~64 KB of LDS and the loader's registers, preferring wide patches
  // (halo share) and few wasted lanes in the last tile
  double best = -1.0;
  size_t best_bytes = 0;
  for (int TY = 1; TY <= 16 && TY <= p.OH; ++TY) {
    for (int nbx = 1; nbx <= 8; ++nbx) {
      const int TX = (p.OW + nbx - 1) / nbx;
      const int cols = (TX - 1) * S + K, rows = (TY - 1) * S + K;
      const size_t chunks = (size_t)K * rows * cols * CPV;
      const size_t bytes = ((((size_t)K * rows * cols * PITCH * 2) + 15) & ~(size_t)15) + FIXED;
      if (chunks > (size_t)PF * 256 || bytes > 64 * 1024) continue;
      const int nby = (p.OH + TY - 1) / TY;
      const double halo = (double)(TX * TY * S * S) / ((double)rows * cols);           // useful share of the loaded patch
      const int tiles = (TX * TY + 15) / 16;
      const double lanes = (double)(TX * TY) / (tiles * 16.0);
      const double work = tiles >= 8 ? 1.0 : tiles / 8.0;                            // enough tiles to amortise the B staging
      const double score = halo * lanes * work;
      if (score > best) { best = score; p.TX = TX; p.TY = TY; p.nbx = nbx; p.nby = nby; p.cols = cols; p.rows = rows; best_bytes = bytes; }
    }
  }
  if (best < 0) return TEM_EUNSUPPORTED;
  p.magicCols = magic_for(p.cols);
  p.magicPlane = magic_for(p.rows * p.cols);
  p.magicTX = magic_for(p.TX);
  if (dry) {
    if (g_name) snprintf(g_name, g_name_len, "conv_bf16_k<%d, %d, %d, %d, %d, %s>", CI, CO, K, S, PF, BLDS ? "true" : "false");
    return TEM_OK;
  }
  static int dbg = -1;
  if (dbg < 0) dbg = tem_env_int("TEM_DEBUG_FLAGS", 0);
  const int nblocks = N * p.nby * p.nbx * p.OD;
  if (dbg & 8)
    fprintf(stderr, "conv_bf16<%d,%d,%d,%d> O=%dx%dx%d P=%d: TX=%d TY=%d patch=%dx%d blocks=%d lds=%zu\n", CI, CO, K, S, p.OD,
            p.OH, p.OW, p.P, p.TX, p.TY, p.rows, p.cols, nblocks, best_bytes);
  auto kern = conv_bf16_k<CI, CO, K, S, PF, BLDS>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(256), best_bytes, st, p);
  TEM_CHECK_LAUNCH();
  return TEM_OK;
}

int dispatch(const tem_conv_args *a, hipStream_t st, bool dry) {
  const tem_view &i0 = a->in0, &o0 = a->out0;
  if (!(a->kd == a->kh && a->kh == a->kw && a->sd == a->sh && a->sh == a->sw && a->pd == a->ph && a->ph == a->pw))
    return TEM_EUNSUPPORTED;
  if (o0.N != i0.N) return TEM_ESHAPE;
  if (!fits32(i0) || !fits32(o0)) return TEM_EUNSUPPORTED;
  Dev p{};
  auto U = [](const float *q) { return reinterpret_cast<const u16 *>(q); };
  auto al16 = [](const tem_view &v) {       // 16-byte channel chunks: 8 bf16
    return v.C % 8 != 0 || (((uintptr_t)v.ptr & 15) == 0 && v.sW % 8 == 0 && v.sH % 8 == 0 && v.sD % 8 == 0 && v.sN % 8 == 0);
  };
  auto al8 = [](const tem_view &v) {        // 8-byte accesses of 4 bf16 (stores, gate, add)
    return v.C % 4 != 0 || (((uintptr_t)v.ptr & 7) == 0 && v.sW % 4 == 0 && v.sH % 4 == 0 && v.sD % 4 == 0 && v.sN % 4 == 0);
  };
  p.in0 = U(i0.ptr); p.i0N = (int)i0.sN; p.i0D = (int)i0.sD; p.i0H = (int)i0.sH; p.i0W = (int)i0.sW; p.C0 = i0.C;
  p.in1 = p.in0; p.i1N = p.i0N; p.i1D = p.i0D; p.i1H = p.i0H; p.i1W = p.i0W;
  int CI = i0.C;
  if (!al16(i0)) return TEM_EUNSUPPORTED;
  if (a->in1.ptr) {
    const tem_view &i1 = a->in1;
    if (i1.N != i0.N || i1.D != i0.D || i1.H != i0.H || i1.W != i0.W) return TEM_ESHAPE;
    if (!fits32(i1) || !al16(i1) || i0.C % 8 || i1.C % 8) return TEM_EUNSUPPORTED;
    p.in1 = U(i1.ptr); p.i1N = (int)i1.sN; p.i1D = (int)i1.sD; p.i1H = (int)i1.sH; p.i1W = (int)i1.sW;
    CI += i1.C;
  }
  p.D = i0.D; p.H = i0.H; p.W = i0.W;
  p.w = U(a->w); p.flip = a->w_layout == TEM_W_FLIP_CO_CI;
  p.out0 = const_cast<u16 *>(U(o0.ptr)); p.o0N = (int)o0.sN; p.o0D = (int)o0.sD; p.o0H = (int)o0.sH; p.o0W = (int)o0.sW;
  p.CO0 = o0.C;
  int CO = o0.C;
  if (!al8(o0)) return TEM_EUNSUPPORTED;
  if (a->out1.ptr) {
    const tem_view &o1 = a->out1;
    if (o1.N != o0.N || o1.D != o0.D || o1.H != o0.H || o1.W != o0.W) return TEM_ESHAPE;
    if (!fits32(o1) || !al8(o1) || o0.C % 4 || o1.C % 4) return TEM_EUNSUPPORTED;
    p.out1 = const_cast<u16 *>(U(o1.ptr)); p.o1N = (int)o1.sN; p.o1D = (int)o1.sD; p.o1H = (int)o1.sH; p.o1W = (int)o1.sW;
    CO += o1.C;
  }
  p.OD = o0.D; p.OH = o0.H; p.OW = o0.W; p.P = a->pd;
  const tem_epilogue &e = a->ep;
  Ep &q = p.ep;
  q.bias = e.bias; q.slope = e.slope; q.gate_slope = e.gate_slope;
  if (e.gate.ptr) {
    const tem_view &g = e.gate;
    if (g.N != o0.N || g.D != o0.D || g.H != o0.H || g.W != o0.W || g.C < o0.C) return TEM_ESHAPE;
    if (!fits32(g) || !al8(g) || o0.C % 4) return TEM_EUNSUPPORTED;
    q.gate = U(g.ptr); q.gN = (int)g.sN; q.gD = (int)g.sD; q.gH = (int)g.sH; q.gW = (int)g.sW;
  }
  if (e.add.ptr) {
    const tem_view &ad = e.add;
    if (ad.C < o0.C || ad.N != o0.N) return TEM_ESHAPE;
    if (!fits32(ad) || !al8(ad) || o0.C % 4) return TEM_EUNSUPPORTED;
    q.add = U(ad.ptr); q.aN = (int)ad.sN; q.aD = (int)ad.sD; q.aH = (int)ad.sH; q.aW = (int)ad.sW;
    q.aoz = e.add_off[0]; q.aoy = e.add_off[1]; q.aox = e.add_off[2];
    q.aDd = ad.D; q.aHh = ad.H; q.aWw = ad.W;
  }
  q.dropout = e.dropout;
  if (e.dropout && o0.C % 8) return TEM_EUNSUPPORTED;
  q.ds.k0 = (uint32_t)e.seed; q.ds.k1 = (uint32_t)(e.seed >> 32); q.ds.site = e.site; q.ds.step = e.step;
  q.step_dev = e.step_dev;
  q.keep_mask = (e.dropout && e.keep_mask) ? e.keep_mask : nullptr;
  q.keep_mode = q.keep_mask ? e.keep_mode : 0;
  q.doz = e.drop_org[0]; q.doy = e.drop_org[1]; q.dox = e.drop_org[2];
  q.dD = e.drop_dims[0] ? e.drop_dims[0] : o0.D; q.dH = e.drop_dims[0] ? e.drop_dims[1] : o0.H;
  q.dW = e.drop_dims[0] ? e.drop_dims[2] : o0.W;
  {
    auto span = [](const tem_view &v) {
      return (int64_t)(v.N - 1) * v.sN + (int64_t)(v.D - 1) * v.sD + (int64_t)(v.H - 1) * v.sH + (int64_t)(v.W - 1) * v.sW + v.C;
    };
    const int64_t melems = (int64_t)o0.N * q.dD * q.dH * q.dW * o0.C;
    if (melems >= ((int64_t)1 << 33)) return TEM_EUNSUPPORTED;
    q.mbytes = (int)((melems + 7) / 8);
    if ((e.gate.ptr && span(e.gate) >= ((int64_t)1 << 30)) || (e.add.ptr && span(e.add) >= ((int64_t)1 << 30)))
      return TEM_EUNSUPPORTED;                     // byte offsets of the epilogue's buffer loads stay below 2^31
    q.gbytes = e.gate.ptr ? (int)(span(e.gate) * 2) : 0;
    q.abytes = e.add.ptr ? (int)(span(e.add) * 2) : 0;
  }
  const int K = a->kd, S = a->sd, N = i0.N;
#define CB(ci, co, k, s, pf) if (CI == ci && CO == co && K == k && S == s) return run<ci, co, k, s, pf, true>(p, N, st, dry);
#define CBG(ci, co, k, s, pf) if (CI == ci && CO == co && K == k && S == s) return run<ci, co, k, s, pf, false>(p, N, st, dry);
  // k3 s1: forward layers and (flip) their input-gradients
  CB(1, 8, 3, 1, 12) CB(8, 1, 3, 1, 12) CB(1, 16, 3, 1, 12) CB(16, 1, 3, 1, 12)
  CB(8, 8, 3, 1, 12) CB(8, 16, 3, 1, 12) CB(16, 8, 3, 1, 12) CB(16, 16, 3, 1, 12)
  CB(16, 32, 3, 1, 12) CB(32, 16, 3, 1, 12) CBG(32, 32, 3, 1, 12)     // 32 -> 32: 54 KB of fragments, read from L2 instead
  // k4 s2: strided forward layers and the input-gradients of the transposed convolutions
  CB(8, 8, 4, 2, 12) CB(16, 16, 4, 2, 12) CBG(32, 32, 4, 2, 12) CB(8, 16, 4, 2, 12) CBG(16, 32, 4, 2, 12)
  // 1x1 head of the discriminator and its gradients
  CB(32, 32, 1, 1, 12) CB(32, 1, 1, 1, 12) CB(1, 32, 1, 1, 12)
#undef CB
#undef CBG
  return TEM_EUNSUPPORTED;
}

}  // namespace conv_bf16

namespace conv3_bf16 { int dispatch(const tem_conv_args *a, hipStream_t st, bool dry, char *name, int name_len); }
int tem_conv_c1_bf16_try(const tem_conv_args *a, hipStream_t st, bool dry, char *name, int name_len);     // stencil_c1.hip
int tem_conv_c1out_bf16_try(const tem_conv_args *a, hipStream_t st, bool dry, char *name, int name_len);  // c1out_mfma.hip

// bf16 mode: activations, gate / add views and the packed kernel are bf16 (the float* fields of tem_conv_args carry
// bf16 pointers, strides in elements); slope / bias / dropout as in tem_conv.
extern "C" int tem_conv_bf16(const tem_conv_args *a, tem_stream_t stream) {
  TEM_CLEAR_ERR();
  if (!a || !tem_view_ok(a->in0) || !tem_view_ok(a->out0) || !a->w) return TEM_EINVAL;
  const int rc1 = tem_conv_c1_bf16_try(a, (hipStream_t)stream, false, nullptr, 0);    // one input channel, 3x3x3
  if (rc1 != TEM_EUNSUPPORTED) return rc1;
  const int rc2 = tem_conv_c1out_bf16_try(a, (hipStream_t)stream, false, nullptr, 0); // one output channel, 3x3x3
  if (rc2 != TEM_EUNSUPPORTED) return rc2;
  const int rc = conv3_bf16::dispatch(a, (hipStream_t)stream, false, nullptr, 0);     // 3x3x3 stride 1, 8..32 channels
  if (rc != TEM_EUNSUPPORTED) return rc;
  return conv_bf16::dispatch(a, (hipStream_t)stream, false);
}

extern "C" int tem_conv_bf16_describe(const tem_conv_args *a, char *buf, int32_t len) {
  if (!a || !tem_view_ok(a->in0) || !tem_view_ok(a->out0) || !a->w) return TEM_EINVAL;
  const int rc1 = tem_conv_c1_bf16_try(a, nullptr, true, buf, len);
  if (rc1 != TEM_EUNSUPPORTED) return rc1;
  const int rc2 = tem_conv_c1out_bf16_try(a, nullptr, true, buf, len);
  if (rc2 != TEM_EUNSUPPORTED) return rc2;
  const int rc3 = conv3_bf16::dispatch(a, nullptr, true, buf, len);
  if (rc3 != TEM_EUNSUPPORTED) return rc3;
  conv_bf16::g_name = buf; conv_bf16::g_name_len = len;
  int rc = conv_bf16::dispatch(a, nullptr, true);
  conv_bf16::g_name = nullptr;
  return rc;
}
