// Shared device helpers for libtem_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tem_hip.h"

// Tuning / ablation knobs (TEM_DEBUG_FLAGS, stamp buffers, kernel on/off switches) are read from the environment only in
// builds with -DTEM_DEBUG_KNOBS (the microbenchmarks under tests/tools: TEM_BUILD_FLAGS=-DTEM_DEBUG_KNOBS python -m
// transfer_em_amd.build --force).  The shipped library ignores the environment: no result-changing debug bit and no
// device address can reach a kernel from outside the C ABI.
#ifdef TEM_DEBUG_KNOBS
#include <stdlib.h>
static inline int tem_env_int(const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; }
static inline unsigned long long tem_env_hex(const char *name) { const char *v = getenv(name); return v ? strtoull(v, nullptr, 16) : 0ull; }
#else
static inline int tem_env_int(const char *, int dflt) { return dflt; }
static inline unsigned long long tem_env_hex(const char *) { return 0ull; }
#endif

#define TEM_CHECK_LAUNCH()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

// hipGetLastError() reports the last error of ANY earlier runtime call on this host thread
// (e.g. a probe inside another library); drop it so that the check after our launch sees only ours.
#define TEM_CLEAR_ERR() ((void)hipGetLastError())

static inline bool tem_view_ok(const tem_view &v) {
  return v.ptr != nullptr && v.N > 0 && v.D > 0 && v.H > 0 && v.W > 0 && v.C > 0;
}

// ---------------------------------------------------------------- Philox4x32-10
// Same stream definition as oracle/tem_oracle.c:orc_dropout_mask.
struct Philox128 { uint32_t r[4]; };

__device__ __forceinline__ Philox128 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                  uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    if (r) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
  }
  return Philox128{{c0, c1, c2, c3}};
}

// keep-bit of dense element index e (stream = seed/site/step)
struct DropoutStream {
  uint32_t k0, k1, site, step;
  __device__ __forceinline__ Philox128 block(uint64_t blk) const {
    return philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), site, step, k0, k1);
  }
  __device__ __forceinline__ static bool bit(const Philox128 &p, uint32_t e_in_block) {
    // select chain, not p.r[i]: a runtime-indexed register array would be placed in scratch memory
    const uint32_t i = (e_in_block >> 5) & 3u;
    const uint32_t w = i == 0 ? p.r[0] : (i == 1 ? p.r[1] : (i == 2 ? p.r[2] : p.r[3]));
    return (w >> (e_in_block & 31)) & 1u;
  }
};

// ---------------------------------------------------------------- epilogue (device copy)
struct EpilogueDev {
  const float *bias;
  float slope;
  const float *gate; int64_t gN, gD, gH, gW; float gate_slope;
  const float *add;  int64_t aN, aD, aH, aW; int32_t aoz, aoy, aox, aDd, aHh, aWw;
  int32_t dropout;
  DropoutStream ds;
  const uint32_t *step_dev;
  int32_t doz, doy, dox, dD, dH, dW;     // dropout frame: origin of out0 inside the full tensor, full extents (dD == 0: none)
  uint8_t *keep_mask;                    // tem_epilogue.keep_mask / keep_mode
  int32_t keep_mode;
};

static inline EpilogueDev make_epilogue(const tem_epilogue &e) {
  EpilogueDev d{};
  d.bias = e.bias;
  d.slope = e.slope;
  d.gate = e.gate.ptr; d.gN = e.gate.sN; d.gD = e.gate.sD; d.gH = e.gate.sH; d.gW = e.gate.sW;
  d.gate_slope = e.gate_slope;
  d.add = e.add.ptr; d.aN = e.add.sN; d.aD = e.add.sD; d.aH = e.add.sH; d.aW = e.add.sW;
  d.aoz = e.add_off[0]; d.aoy = e.add_off[1]; d.aox = e.add_off[2];
  d.aDd = e.add.D; d.aHh = e.add.H; d.aWw = e.add.W;
  d.dropout = e.dropout;
  d.ds.k0 = (uint32_t)e.seed; d.ds.k1 = (uint32_t)(e.seed >> 32);
  d.ds.site = e.site; d.ds.step = e.step;
  d.step_dev = e.step_dev;
  d.doz = e.drop_org[0]; d.doy = e.drop_org[1]; d.dox = e.drop_org[2];
  d.dD = e.drop_dims[0]; d.dH = e.drop_dims[1]; d.dW = e.drop_dims[2];
  d.keep_mask = e.keep_mask; d.keep_mode = e.keep_mask ? e.keep_mode : 0;
  return d;
}

// Apply the epilogue to NC consecutive channels [c0, c0+NC) of output voxel (n,z,y,x) of out0
// (dense extents D,H,W,C).  NC must divide 128 and c0 % NC == 0 so that all NC dropout bits
// lie in one Philox block.
template <int NC>
__device__ __forceinline__ void apply_epilogue(const EpilogueDev &ep, float (&v)[NC], int n, int z, int y, int x,
                                               int c0, int D, int H, int W, int C) {
  if (ep.bias) {
#pragma unroll
    for (int i = 0; i < NC; ++i) v[i] += ep.bias[c0 + i];
  }
  if (ep.add) {
    int az = z - ep.aoz, ay = y - ep.aoy, ax = x - ep.aox;
    if (az >= 0 && az < ep.aDd && ay >= 0 && ay < ep.aHh && ax >= 0 && ax < ep.aWw) {
      const float *ap = ep.add + n * ep.aN + az * ep.aD + ay * ep.aH + ax * ep.aW + c0;
#pragma unroll
      for (int i = 0; i < NC; ++i) v[i] += ap[i];
    }
  }
  if (ep.gate) {
    const float *gp = ep.gate + n * ep.gN + z * ep.gD + y * ep.gH + x * ep.gW + c0;
#pragma unroll
    for (int i = 0; i < NC; ++i) v[i] = gp[i] > 0.f ? v[i] : ep.gate_slope * v[i];
  }
  if (ep.dropout) {
    DropoutStream ds = ep.ds;
    if (ep.step_dev) ds.step = *ep.step_dev;
    uint64_t e = ep.dD ? ((((uint64_t)n * ep.dD + (z + ep.doz)) * ep.dH + (y + ep.doy)) * ep.dW + (x + ep.dox)) * (uint64_t)C + c0
                       : ((((uint64_t)n * D + z) * H + y) * W + x) * (uint64_t)C + c0;
    if (NC % 8 == 0 && ep.keep_mode == 2) {              // bits drawn by the forward pass
#pragma unroll
      for (int b = 0; b < NC / 8; ++b) {
        const uint32_t bits = ep.keep_mask[(e >> 3) + b];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[8 * b + i] = ((bits >> i) & 1u) ? 2.f * v[8 * b + i] : 0.f;
      }
    } else {
      Philox128 p = ds.block(e >> 7);
      uint32_t eb = (uint32_t)(e & 127);
      if (NC % 8 == 0 && ep.keep_mode == 1) {
#pragma unroll
        for (int b = 0; b < NC / 8; ++b) {
          uint32_t bits = 0;
#pragma unroll
          for (int i = 0; i < 8; ++i) bits |= (DropoutStream::bit(p, eb + 8 * b + i) ? 1u : 0u) << i;
          ep.keep_mask[(e >> 3) + b] = (uint8_t)bits;
        }
      }
#pragma unroll
      for (int i = 0; i < NC; ++i) v[i] = DropoutStream::bit(p, eb + i) ? 2.f * v[i] : 0.f;
    }
  }
  if (ep.slope != 1.f) {
#pragma unroll
    for (int i = 0; i < NC; ++i) v[i] = v[i] > 0.f ? v[i] : ep.slope * v[i];
  }
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an XCD and its 4 MB L2;
// observed behaviour, used for speed only).  Map the hardware block id to a LOGICAL block id such that
// each XCD works through one contiguous range of logical blocks: neighbouring tiles -- whose halos
// overlap -- then meet in one L2 instead of being fetched over the fabric by up to eight of them.
__device__ __forceinline__ unsigned xcd_contiguous_block(unsigned b, unsigned nb) {
  const unsigned q = nb >> 3, r = nb & 7u, k = b & 7u;
  return k * q + (k < r ? k : r) + (b >> 3);
}

// wave-level sum (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Host-side memo of launch plans.  The planners of the z-marching kernels search a few thousand (tile, z-run) candidates per
// call -- 10-50 us of host time per launch (1.7 of the 2.8 ms the launching thread spent per fp32 step, tests/tools/
// host_per_launch.py) for a result that depends on the geometry alone.
#include <array>
#include <map>
#include <mutex>
template <size_t NK, typename R> struct tem_plan_cache {
  std::mutex mu;
  std::map<std::array<int, NK>, R> m;
  bool get(const std::array<int, NK> &k, R &r) {
    std::lock_guard<std::mutex> g(mu);
    auto it = m.find(k);
    if (it == m.end()) return false;
    r = it->second;
    return true;
  }
  void put(const std::array<int, NK> &k, const R &r) {
    std::lock_guard<std::mutex> g(mu);
    if (m.size() < 4096) m[k] = r;
  }
};
