// Shared device helpers + host-side error plumbing for libmi355q (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>

#include "mi355q.h"

namespace mi355q {

// ---------------------------------------------------------------- host side
void set_error(const char* fmt, ...);
int32_t fail(mi355q_status st, const char* fmt, ...);
void clear_error();

#define MI355Q_CHECK_LAUNCH(what)                                            \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess)                                                   \
      return ::mi355q::fail(MI355Q_HIP_ERROR, "%s: %s", what, hipGetErrorString(e__)); \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// NumPy's pairwise summation splits n > 128 at n/2 rounded down to a multiple of 8. When
// n = leaf << depth with leaf <= 128 a multiple of 8, every split is exact, so the chunk is
// 2^depth equal leaves at offsets i * leaf under the complete binary tree. True for the sizes
// real layers have (4096 = 128 << 5, 11008 - 8192 = 88 << 5, 14336 - 8192 = 96 << 6, ...).
static inline bool balanced_chunk(int n, int* leaf, int* depth) {
  int d = 0;
  while ((n >> d) > 128) ++d;
  const int l = n >> d;
  if ((l << d) != n || l < 8 || (l % 8) != 0) return false;
  *leaf = l;
  *depth = d;
  return true;
}

// -------------------------------------------------------------- device side
constexpr int kWave = 64;  // gfx950 wavefront

__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

// |x| as an unsigned pattern: ordering of these patterns is the ordering of |x|,
// and every NaN sorts above +inf, so an integer max propagates NaN the way
// np.max(np.abs(x)) does.
__device__ __forceinline__ uint32_t abs_bits(float x) { return f2u(x) & 0x7FFFFFFFu; }

// Butterfly max over the `width` lanes (power of two <= 64) that share a group.
template <int WIDTH>
__device__ __forceinline__ uint32_t group_max_u32(uint32_t v) {
#pragma unroll
  for (int off = WIDTH / 2; off > 0; off >>= 1) {
    uint32_t o = static_cast<uint32_t>(__shfl_xor(static_cast<int>(v), off, kWave));
    v = o > v ? o : v;
  }
  return v;
}

// float32 -> bfloat16 round-to-nearest-even, returned as float32.
// ref: uniform_quantize_tensor.py:580 (`.astype(ml_dtypes.bfloat16)`).
__device__ __forceinline__ float round_bf16(float x) {
  uint32_t b = f2u(x);
  if ((b & 0x7FFFFFFFu) > 0x7F800000u) return x;  // NaN stays NaN
  b = (b + 0x7FFFu + ((b >> 16) & 1u)) & 0xFFFF0000u;
  return u2f(b);
}

// scale -> bf16 -> f16 (RNE, subnormals kept, overflow -> inf) -> f32.
// ref: uniform_quantize_tensor.py:577-581. Also yields the stored half pattern
// (ref: transformations/quantize_tensor.py:129-137).
__device__ __forceinline__ float round_scale_blockwise(float s, uint16_t* half_bits) {
  _Float16 h = static_cast<_Float16>(round_bf16(s));
  *half_bits = __builtin_bit_cast(uint16_t, h);
  return static_cast<float>(h);
}

// A workgroup barrier for loops whose LDS buffers are filled by global_load ... lds SEVERAL tiles ahead. __syncthreads() is a
// release fence + s_barrier, and for the fence the compiler drains the vector-memory counter (s_waitcnt vmcnt(0)) -- it waits
// for the pieces of LATER tiles as well, so a ring of N buffers never had more than one tile in flight across a barrier
// whatever the s_waitcnt vmcnt(n) in front of it said (found in the ISA in round 5; every xtx kernel had it). Here the
// caller has waited for what it needs (its own pieces of the tile about to be read: s_waitcnt vmcnt(n)); LDS reads of the
// buffer about to be refilled have been consumed by the MFMAs that used them. The "memory" clobber keeps the compiler from
// moving LDS accesses across it. MI355Q_FENCED_BARRIER=1 at compile time restores __syncthreads() (A / B timing).
__device__ __forceinline__ void barrier_loads_in_flight() {
#if defined(MI355Q_FENCED_BARRIER)
  __syncthreads();
#else
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

template <int BITS>
struct QRange {
  static constexpr float qmax = static_cast<float>((1 << (BITS - 1)) - 1);
  static constexpr float qmin = -static_cast<float>(1 << (BITS - 1));
  // narrow range only for symmetric >= 8 bit (ref: uniform_quantize_tensor.py:313-315)
  static constexpr float lo_sym = BITS >= 8 ? qmin + 1.0f : qmin;
};

// clip(rint(v)) -> int; NaN -> 0 (NumPy: clip keeps NaN, the C cast of NaN to an
// 8-bit int yields 0 on the reference's x86 hosts).
__device__ __forceinline__ int round_clip(float v, float lo, float hi) {
  float r = __builtin_rintf(v);  // v_rndne_f32: ties to even == np.rint
  r = fminf(fmaxf(r, lo), hi);
  return (v != v) ? 0 : static_cast<int>(r);
}

// Symmetric quantize of one value given the group scale (zero point 0).
// x / s is the IEEE correctly rounded quotient (never reciprocal-multiply).
template <int BITS>
__device__ __forceinline__ int quant_sym(float x, float s) {
  return round_clip(x / s, QRange<BITS>::lo_sym, QRange<BITS>::qmax);
}

}  // namespace mi355q
