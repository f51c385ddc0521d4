// Shared pieces of the Winograd-form kernels (wino.hip: forward / input-gradient; wino_bww.hip: kernel gradient).
#pragma once
#include "tem_common.h"

namespace wino {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t fdiv(uint32_t x, uint32_t d, uint32_t magic) { return d == 1 ? x : __umulhi(x, magic); }

// Single-instruction fp32 adds: beside MFMAs a packed v_pk_add_f32 costs the matrix pipe ~13 cycles each
// (MI355X_MICROARCH.md, "price of one filler beside MFMAs": packed f32 VALU is an anti-lever there, and plain -O3 SLP-packs
// adjacent adds into it), a plain v_add_f32 / v_sub_f32 issues in the MFMA's shadow.  The asm keeps them un-packed.
__device__ __forceinline__ float sadd(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float ssub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// 1-D transforms of F(2,3)
template <typename T> __device__ __forceinline__ void bt4(T &a, T &b, T &c, T &d) {   // B^T
  const T v0 = a - c, v1 = b + c, v2 = c - b, v3 = b - d;
  a = v0; b = v1; c = v2; d = v3;
}
#ifdef TEM_WINO_SCALAR_ADDS   // measured (round 3): f1 fwd 101.3 -> 104.4 us with the scalar form -- twice the issue slots outweigh the pipe sharing
template <> __device__ __forceinline__ void bt4<f32x2>(f32x2 &a, f32x2 &b, f32x2 &c, f32x2 &d) {
  const f32x2 v0 = {ssub(a.x, c.x), ssub(a.y, c.y)}, v1 = {sadd(b.x, c.x), sadd(b.y, c.y)};
  const f32x2 v2 = {ssub(c.x, b.x), ssub(c.y, b.y)}, v3 = {ssub(b.x, d.x), ssub(b.y, d.y)};
  a = v0; b = v1; c = v2; d = v3;
}
#endif
template <typename T> __device__ __forceinline__ void at4(const T &a, const T &b, const T &c, const T &d, T &y0, T &y1) {   // A^T
  const T s = b + c, t = b - c;
  y0 = a + s; y1 = t - d;
}

// LDS image of one input plane (YR = 2 BY + 2 rows of 2 E voxels): one sub-image per 8 input channels (= per channel-pair
// half h of the k loop; a concat input's two sources land in different sub-images), rows split into their even-x and
// odd-x voxels:
//   16-byte chunk slot of (row yr, x = 2 e + o, chunk c of the 8 channels) = ((yr * 2 + o) * E + e) * 2 + (c ^ swz(e, yr)),
//   swz(e, yr) = ((e >> 3) + ((yr >> 1) & 1)) & 1
// The 16 tiles of an MFMA row block are 16 consecutive e (stride-2 voxels of the dense row are consecutive here) and the
// XOR spreads them over the sixteen 16-byte bank slots: the ds_read_b64 of a (tile, channel-pair) fragment is conflict
// free within a tile row and nearly so across a row wrap.  The image is filled by LDS-DMA (buffer_load_dwordx4 ... lds:
// no staging registers, no ds_write pass); the swizzle is on the source address, the LDS side is lane-linear.
__device__ __forceinline__ int swz(int e, int yr) { return ((e >> 3) + ((yr >> 1) & 1)) & 1; }

// One 8-channel sub-image of one input plane: wave-instruction j = wave + 8 i moves chunk slots 64 j .. 64 j + 63 (1 KB,
// lane-linear) from the plane at `base`; offsets outside [0, span) arrive as zeros.
// ASM: the fetch is issued as inline assembly.  For the builtin, hipcc cannot tell the ring slots apart and puts
// s_waitcnt vmcnt(0) in front of the first LDS read that follows it in program order -- the fetch latency inside the step,
// on every wave; with ASM the kernel's own waits (before the barrier that hands the slots over) are the only ones, and they
// MUST be there: the compiler no longer knows that anything is in flight.
template <int NI, bool ASM = false>
__device__ __forceinline__ void dma_subimage(const float *base, int span, const int *voff, char *dst, int wave, int ndma) {
  if constexpr (ASM) {
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t r = u32x4_t{(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)base),
                              (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)((uintptr_t)base >> 32) & 0xffffu)), (uint32_t)span, 0x00020000u};
    const uint32_t d0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)dst;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int j = wave + 8 * i;
      if (j < ndma)                                          // wave-uniform
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(d0 + j * 1024), "v"(voff[i]), "s"(r) : "memory");
    }
  } else {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, span, 0x00020000);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int j = wave + 8 * i;
      if (j < ndma)                                          // wave-uniform
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, voff[i], 0, 0, 0);
    }
  }
}

static uint32_t magic_for(int d) { return d <= 1 ? 0u : (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d); }

static bool fits32(const tem_view &v) {
  int64_t span = (int64_t)(v.N - 1) * v.sN + (int64_t)(v.D - 1) * v.sD + (int64_t)(v.H - 1) * v.sH +
                 (int64_t)(v.W - 1) * v.sW + v.C;
  return span < (int64_t)1 << 31 && v.sN < ((int64_t)1 << 31);
}

}  // namespace wino
