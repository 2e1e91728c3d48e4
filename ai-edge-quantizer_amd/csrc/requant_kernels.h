// Kernel templates of the fused symmetric requantization (K1+K2+K3+K4); included
// by requant.hip (the product entry points) and by tools/kbench (variant timing).
// See requant.hip for the design notes.
#pragma once

#include "common.h"

namespace mi355q {
namespace requant {

struct RequantArgs {
  // Direct pointers (single tensor) or device tables of pointers (batched).
  const void* x;
  void* q;
  void* packed;
  void* scale;
  void* scale_f16;
  const float* clip;  // single-tensor only
  int64_t rows;
  int64_t cols;
  int32_t block;  // 0 = one scale per row
};

template <bool BATCHED, typename T>
__device__ __forceinline__ T* pick(const void* p, int t) {
  if constexpr (BATCHED) {
    auto tab = reinterpret_cast<T* const*>(p);
    return tab ? tab[t] : nullptr;
  } else {
    return reinterpret_cast<T*>(const_cast<void*>(p));
  }
}

// The batched form with its pointer tables IN the kernel arguments (up to kInlineTensors buffers per launch): the
// tables of a group of equally shaped weights travel with the dispatch packet instead of through a host-to-device copy
// that the launch has to wait for on its stream (the copy's completion signal, not its 512 bytes, is what costs:
// ~10 us of stream time per launch against 14 us of kernel per tensor). blockIdx.y is uniform, so x[t] is one scalar
// load from the kernarg segment.
constexpr int kInlineTensors = 16;
struct PtrTable {
  const void* p[kInlineTensors];
};
struct RequantInlineArgs {
  PtrTable x, q, packed, scale, scale_f16;
  const float* clip;  // always null (the batched forms take no clip)
  int64_t rows;
  int64_t cols;
  int32_t block;
};

template <bool BATCHED, typename T>
__device__ __forceinline__ T* pick(const PtrTable& tab, int t) {
  return reinterpret_cast<T*>(const_cast<void*>(tab.p[t]));
}

// Streaming access helpers: x is read exactly once and q is never re-read by this
// kernel, so both can bypass cache retention (`nt`), see kbench for the effect.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ float4 load4(const float4* p) {
  if constexpr (NT) {
    const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  } else {
    return *p;
  }
}
template <bool NT>
__device__ __forceinline__ void store1(uint32_t* p, uint32_t a) {
  if constexpr (NT) __builtin_nontemporal_store(a, p); else *p = a;
}
template <bool NT>
__device__ __forceinline__ void store2(uint32_t* p, uint32_t a, uint32_t b) {
  if constexpr (NT) { u32x2_t v = {a, b}; __builtin_nontemporal_store(v, reinterpret_cast<u32x2_t*>(p)); }
  else *reinterpret_cast<uint2*>(p) = make_uint2(a, b);
}
template <bool NT>
__device__ __forceinline__ void store4(uint32_t* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  if constexpr (NT) { u32x4_t v = {a, b, c, d}; __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(p)); }
  else *reinterpret_cast<uint4*>(p) = make_uint4(a, b, c, d);
}

// bound -> scale (K2). ref: uniform_quantize_tensor.py:552-563, 577-581.
template <int BITS, bool BLOCKWISE>
__device__ __forceinline__ float make_scale(uint32_t absmax_bits, const float* clip,
                                            int64_t g, uint16_t* half_bits) {
  float bound = fmaxf(u2f(absmax_bits), 1e-9f);
  if ((absmax_bits & 0x7FFFFFFFu) > 0x7F800000u) bound = u2f(absmax_bits);  // NaN
  if (clip != nullptr) {
    float pos = clip[g], neg = -clip[g];
    const bool clip_nan = pos != pos;   // np.minimum / np.maximum with the f16 cap keep a NaN clip NaN
    if constexpr (BLOCKWISE) {
      // f16 scale range cap (ref :529-550): +65280*(2^bits-1), -65280*2^bits
      pos = fminf(pos, 65280.0f * static_cast<float>((1 << BITS) - 1));
      neg = fmaxf(neg, -65280.0f * static_cast<float>(1 << BITS));
    }
    // np.clip(bound, neg, pos) = minimum(maximum(bound, neg), pos), both NaN-propagating:
    // a NaN bound stays NaN (fminf / fmaxf would return the other operand), a NaN clip makes it NaN
    if (bound == bound) bound = fminf(fmaxf(bound, neg), pos);
    if (clip_nan) bound = clip[g];
  }
  float s = bound / QRange<BITS>::qmax;
  if constexpr (BLOCKWISE) s = round_scale_blockwise(s, half_bits);
  return s;
}

// ------------------------------------------------------------------------
// Exact quantization of 4 values with ONE true division per group.
//
// rint(RN(x/s)) only depends on RN(x/s) near the half-integers. With r = RN(1/s)
// and t = RN(x*r):  |t - RN(x/s)| <= 1.5 * 2^-23 * |x/s|. For |t| <= qmax + 2 that
// is < 2.4e-5, so whenever t is farther than kGuard = 3.2e-5 from every
// half-integer, rint(t) == rint(RN(x/s)); when |t| > qmax + 2 both clip to the
// same bound. Anything else (incl. NaN / inf, where the test is false) takes the
// IEEE division. The slow path fires for ~6e-5 of the elements, decided per wave.
// ------------------------------------------------------------------------
template <int BITS>
struct Quant4 {
  int a, b, c, d;
};

template <int BITS, bool FAST>
__device__ __forceinline__ Quant4<BITS> quant4(float4 v, float s, float r) {
  constexpr float lo = QRange<BITS>::lo_sym, hi = QRange<BITS>::qmax;
  Quant4<BITS> o;
  if constexpr (!FAST) {
    o.a = quant_sym<BITS>(v.x, s); o.b = quant_sym<BITS>(v.y, s);
    o.c = quant_sym<BITS>(v.z, s); o.d = quant_sym<BITS>(v.w, s);
    return o;
  } else {
    constexpr float kGuard = 3.2e-5f;
    // r must be a normal number: 1/s overflowed, underflowed or is NaN otherwise
    const bool r_ok = fabsf(r) >= 1.17549435e-38f && fabsf(r) <= 3.40282347e+38f;
    const float t0 = v.x * r, t1 = v.y * r, t2 = v.z * r, t3 = v.w * r;
    const float r0 = __builtin_rintf(t0), r1 = __builtin_rintf(t1);
    const float r2 = __builtin_rintf(t2), r3 = __builtin_rintf(t3);
    const bool ok0 = fabsf(t0 - r0) < 0.5f - kGuard || fabsf(t0) > hi + 2.0f;
    const bool ok1 = fabsf(t1 - r1) < 0.5f - kGuard || fabsf(t1) > hi + 2.0f;
    const bool ok2 = fabsf(t2 - r2) < 0.5f - kGuard || fabsf(t2) > hi + 2.0f;
    const bool ok3 = fabsf(t3 - r3) < 0.5f - kGuard || fabsf(t3) > hi + 2.0f;
    if (__builtin_expect(__any(!(r_ok && ok0 && ok1 && ok2 && ok3)), 0)) {
      o.a = quant_sym<BITS>(v.x, s); o.b = quant_sym<BITS>(v.y, s);
      o.c = quant_sym<BITS>(v.z, s); o.d = quant_sym<BITS>(v.w, s);
    } else {
      o.a = static_cast<int>(fminf(fmaxf(r0, lo), hi));
      o.b = static_cast<int>(fminf(fmaxf(r1, lo), hi));
      o.c = static_cast<int>(fminf(fmaxf(r2, lo), hi));
      o.d = static_cast<int>(fminf(fmaxf(r3, lo), hi));
    }
    return o;
  }
}

template <int BITS>
__device__ __forceinline__ uint32_t pack_i8(const Quant4<BITS>& o) {
  return (o.a & 0xFF) | ((o.b & 0xFF) << 8) | ((o.c & 0xFF) << 16) |
         (static_cast<uint32_t>(o.d & 0xFF) << 24);
}
template <int BITS>
__device__ __forceinline__ uint32_t pack_sub(const Quant4<BITS>& o) {  // 4 values -> 4*BITS bits
  constexpr int M = (1 << BITS) - 1;
  return (o.a & M) | ((o.b & M) << BITS) | ((o.c & M) << (2 * BITS)) | ((o.d & M) << (3 * BITS));
}

// Quantize CL consecutive float4 (idx4 = index of the first, in float4 units) and emit
// them in the requested containers with the widest store the alignment allows.
template <int BITS, int CL, bool FAST, bool NT = false>
__device__ __forceinline__ void emit(const float4 (&v)[CL], float s, int64_t idx4, int8_t* q,
                                     uint8_t* packed) {
  const float r = FAST ? 1.0f / s : 0.f;  // one IEEE division per call
  uint32_t w8[CL];    // 4 values as int8 containers
  uint32_t subw[CL];  // the same 4 values packed to 4*BITS bits
#pragma unroll
  for (int c = 0; c < CL; ++c) {
    const Quant4<BITS> o = quant4<BITS, FAST>(v[c], s, r);
    w8[c] = pack_i8<BITS>(o);
    subw[c] = pack_sub<BITS>(o);
  }
  const bool same = BITS == 8 && reinterpret_cast<int8_t*>(packed) == q;
  if (q != nullptr) {
    uint32_t* dst = reinterpret_cast<uint32_t*>(q) + idx4;
    if constexpr (CL == 4) store4<NT>(dst, w8[0], w8[1], w8[2], w8[3]);
    else if constexpr (CL == 2) store2<NT>(dst, w8[0], w8[1]);
    else store1<NT>(dst, w8[0]);
  }
  if (packed != nullptr && !same) {
    if constexpr (BITS == 8) {
      uint32_t* dst = reinterpret_cast<uint32_t*>(packed) + idx4;
      if constexpr (CL == 4) store4<NT>(dst, w8[0], w8[1], w8[2], w8[3]);
      else if constexpr (CL == 2) store2<NT>(dst, w8[0], w8[1]);
      else store1<NT>(dst, w8[0]);
    } else if constexpr (BITS == 4) {  // 16 bits per float4
      uint16_t* dst = reinterpret_cast<uint16_t*>(packed) + idx4;
      if constexpr (CL == 4)
        store2<NT>(reinterpret_cast<uint32_t*>(dst), subw[0] | (subw[1] << 16), subw[2] | (subw[3] << 16));
      else if constexpr (CL == 2) store1<NT>(reinterpret_cast<uint32_t*>(dst), subw[0] | (subw[1] << 16));
      else dst[0] = static_cast<uint16_t>(subw[0]);
    } else {  // 2 bit: 8 bits per float4
      uint8_t* dst = packed + idx4;
      if constexpr (CL == 4)
        *reinterpret_cast<uint32_t*>(dst) = subw[0] | (subw[1] << 8) | (subw[2] << 16) | (subw[3] << 24);
      else if constexpr (CL == 2) *reinterpret_cast<uint16_t*>(dst) = static_cast<uint16_t>(subw[0] | (subw[1] << 8));
      else dst[0] = static_cast<uint8_t>(subw[0]);
    }
  }
}

__device__ __forceinline__ uint32_t absmax4(float4 v) {
  return max(max(abs_bits(v.x), abs_bits(v.y)), max(abs_bits(v.z), abs_bits(v.w)));
}

// ------------------------------------------------------------------------
// (A) small groups: BLOCKWISE_32/64/128/256 -> G4 = 8/16/32/64 float4 per group.
// The tensor is a flat run of groups. Every lane owns CL consecutive float4
// (G4/CL lanes share a group); a 256-thread block streams U tiles of 256*CL float4.
// ------------------------------------------------------------------------
template <int BITS, int G4, int U, int CL, bool FAST, bool BATCHED, bool NT = false, typename ARGS = RequantArgs>
__global__ __launch_bounds__(256) void requant_groups_kernel(ARGS a) {
  static_assert(G4 % CL == 0 && G4 / CL >= 1, "group must be a multiple of the lane piece");
  constexpr int LPG = G4 / CL;  // lanes per group
  const int t = BATCHED ? blockIdx.y : 0;
  const float4* __restrict__ x = pick<BATCHED, const float4>(a.x, t);
  int8_t* q = pick<BATCHED, int8_t>(a.q, t);
  uint8_t* packed = pick<BATCHED, uint8_t>(a.packed, t);
  float* scale = pick<BATCHED, float>(a.scale, t);
  uint16_t* scale_f16 = pick<BATCHED, uint16_t>(a.scale_f16, t);
  const float* clip = BATCHED ? nullptr : a.clip;

  const int64_t n4 = a.rows * a.cols / 4;
  const int64_t base = (static_cast<int64_t>(blockIdx.x) * U * 256 + threadIdx.x) * CL;

  float4 v[U][CL];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = base + static_cast<int64_t>(u) * 256 * CL;
#pragma unroll
    for (int c = 0; c < CL; ++c)
      v[u][c] = i < n4 ? load4<NT>(x + i + c) : make_float4(0.f, 0.f, 0.f, 0.f);  // n4 % CL == 0
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = base + static_cast<int64_t>(u) * 256 * CL;
    uint32_t m = 0;
#pragma unroll
    for (int c = 0; c < CL; ++c) m = max(m, absmax4(v[u][c]));
    m = group_max_u32<LPG>(m);
    if (i < n4) {  // groups never straddle n4 (cols % block == 0)
      const int64_t g = i / G4;
      uint16_t hb = 0;
      const float s = make_scale<BITS, true>(m, clip, g, &hb);
      if ((threadIdx.x & (LPG - 1)) == 0) {
        scale[g] = s;
        if (scale_f16 != nullptr) scale_f16[g] = hb;
      }
      emit<BITS, CL, FAST, NT>(v[u], s, i, q, packed);
    }
  }
}

// ------------------------------------------------------------------------
// (B) one scale per row, row held in registers: TPR threads x R float4 cover a
// row of cols4 <= TPR*R float4. TPR = 64 -> a wave owns the row (no LDS);
// TPR = 256 -> the block owns the row (one LDS exchange).
// ------------------------------------------------------------------------
template <int BITS, int TPR, int R, bool FAST, bool BATCHED, bool NT = false, typename ARGS = RequantArgs>
__global__ __launch_bounds__(256) void requant_rows_kernel(ARGS a) {
  static_assert(TPR == 256 || (TPR <= kWave && (TPR & (TPR - 1)) == 0),
                "a row is owned by part of a wave, one wave, or the whole 256-thread block");
  constexpr int RPB = 256 / TPR;  // rows per block
  const int t = BATCHED ? blockIdx.y : 0;
  const float4* __restrict__ x = pick<BATCHED, const float4>(a.x, t);
  int8_t* q = pick<BATCHED, int8_t>(a.q, t);
  uint8_t* packed = pick<BATCHED, uint8_t>(a.packed, t);
  float* scale = pick<BATCHED, float>(a.scale, t);
  const float* clip = BATCHED ? nullptr : a.clip;

  const int lane = threadIdx.x % TPR;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * RPB + threadIdx.x / TPR;
  const int cols4 = static_cast<int>(a.cols / 4);
  const bool live = row < a.rows;
  const int64_t row4 = row * cols4;

  float4 v[R][1];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int c = j * TPR + lane;
    v[j][0] = (live && c < cols4) ? load4<NT>(x + row4 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  uint32_t m = 0;
#pragma unroll
  for (int j = 0; j < R; ++j) m = max(m, absmax4(v[j][0]));
  m = group_max_u32<(TPR < kWave ? TPR : kWave)>(m);
  if constexpr (TPR > kWave) {
    __shared__ uint32_t part[256 / kWave];
    if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = m;
    __syncthreads();
    m = max(max(part[0], part[1]), max(part[2], part[3]));
  }
  if (!live) return;
  uint16_t hb;
  const float s = make_scale<BITS, false>(m, clip, row, &hb);
  if (lane == 0) scale[row] = s;
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int c = j * TPR + lane;
    if (c < cols4) emit<BITS, 1, FAST, NT>(v[j], s, row4 + c, q, packed);
  }
}

// ------------------------------------------------------------------------
// (C) generic fallback: any cols (also cols % 4 != 0), any group length. One
// block per group, two sweeps (the second one hits L2). Packed output is not
// produced here (the host entry refuses ragged packing; use mi355q_pack_bits).
// ------------------------------------------------------------------------
template <int BITS, bool BLOCKWISE, bool BATCHED, typename ARGS = RequantArgs>
__global__ __launch_bounds__(256) void requant_generic_kernel(ARGS a) {
  const int t = BATCHED ? blockIdx.y : 0;
  const float* __restrict__ x = pick<BATCHED, const float>(a.x, t);
  int8_t* q = pick<BATCHED, int8_t>(a.q, t);
  float* scale = pick<BATCHED, float>(a.scale, t);
  uint16_t* scale_f16 = pick<BATCHED, uint16_t>(a.scale_f16, t);
  const float* clip = BATCHED ? nullptr : a.clip;

  const int64_t glen = a.block > 0 ? a.block : a.cols;
  const int64_t g = blockIdx.x;
  const float* xg = x + g * glen;
  uint32_t m = 0;
  for (int64_t i = threadIdx.x; i < glen; i += 256) m = max(m, abs_bits(xg[i]));
  m = group_max_u32<kWave>(m);
  __shared__ uint32_t part[256 / kWave];
  if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = m;
  __syncthreads();
  m = max(max(part[0], part[1]), max(part[2], part[3]));
  uint16_t hb = 0;
  const float s = make_scale<BITS, BLOCKWISE>(m, clip, g, &hb);
  if (threadIdx.x == 0) {
    scale[g] = s;
    if (BLOCKWISE && scale_f16 != nullptr) scale_f16[g] = hb;
  }
  if (q != nullptr) {
    int8_t* qg = q + g * glen;
    for (int64_t i = threadIdx.x; i < glen; i += 256)
      qg[i] = static_cast<int8_t>(quant_sym<BITS>(xg[i], s));
  }
}

}  // namespace requant
}  // namespace mi355q
