// Fused symmetric min/max requantization (K1+K2+K3+K4) for gfx950.
//
// Replaces, for one weight buffer and in ONE pass over HBM:
//   ref: algorithms/uniform_quantize/common_quantize.py:1311-1359  (min / max)
//   ref: algorithms/uniform_quantize/uniform_quantize_tensor.py:492-586 (scale)
//   ref: algorithms/uniform_quantize/uniform_quantize_tensor.py:273-362 (quantize)
//   ref: transformations/transformation_utils.py:293-353 (pack_data)
//
// The path is HBM-bound (5 B of traffic and ~25 VALU ops per element), so the
// design is about bytes in flight and coalescing, not MFMA:
//   * every lane owns 16-byte (float4) pieces; a wave instruction moves 1 KiB
//     contiguous; the values stay in VGPRs between the |x|-max reduction and the
//     divide/round/clip, so x is read from HBM exactly once;
//   * reductions are wave64 butterflies (__shfl_xor -> DPP / ds_bpermute), one LDS
//     round trip only when a row spans several waves;
//   * int8 results leave as packed dwords, int4 as packed 16-bit pairs.
#include "common.h"

namespace mi355q {
namespace {

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

// bound -> scale (K2). ref: uniform_quantize_tensor.py:552-563, 577-581.
template <int BITS, bool BLOCKWISE>
__device__ __forceinline__ float make_scale(uint32_t absmax_bits, const float* clip,
                                            int64_t g, uint16_t* half_bits) {
  float bound = fmaxf(u2f(absmax_bits), 1e-9f);
  if ((absmax_bits & 0x7FFFFFFFu) > 0x7F800000u) bound = u2f(absmax_bits);  // NaN
  if (clip != nullptr) {
    float pos = clip[g], neg = -clip[g];
    if constexpr (BLOCKWISE) {
      // f16 scale range cap (ref :529-550): +65280*(2^bits-1), -65280*2^bits
      pos = fminf(pos, 65280.0f * static_cast<float>((1 << BITS) - 1));
      neg = fmaxf(neg, -65280.0f * static_cast<float>(1 << BITS));
    }
    bound = fminf(fmaxf(bound, neg), pos);  // np.clip(bound, neg, pos)
  }
  float s = bound / QRange<BITS>::qmax;
  if constexpr (BLOCKWISE) s = round_scale_blockwise(s, half_bits);
  return s;
}

// Quantize one float4 and emit it in the requested containers.
template <int BITS>
__device__ __forceinline__ void emit4(float4 v, float s, int64_t idx4, int8_t* q,
                                      uint8_t* packed) {
  const int a = quant_sym<BITS>(v.x, s), b = quant_sym<BITS>(v.y, s);
  const int c = quant_sym<BITS>(v.z, s), d = quant_sym<BITS>(v.w, s);
  if (q != nullptr) {
    const uint32_t w = (a & 0xFF) | ((b & 0xFF) << 8) | ((c & 0xFF) << 16) |
                       (static_cast<uint32_t>(d & 0xFF) << 24);
    reinterpret_cast<uint32_t*>(q)[idx4] = w;
  }
  if (packed != nullptr) {
    if constexpr (BITS == 8) {
      if (reinterpret_cast<int8_t*>(packed) != q) {
        const uint32_t w = (a & 0xFF) | ((b & 0xFF) << 8) | ((c & 0xFF) << 16) |
                           (static_cast<uint32_t>(d & 0xFF) << 24);
        reinterpret_cast<uint32_t*>(packed)[idx4] = w;
      }
    } else if constexpr (BITS == 4) {
      const uint16_t w = static_cast<uint16_t>((a & 0xF) | ((b & 0xF) << 4) |
                                               ((c & 0xF) << 8) | ((d & 0xF) << 12));
      reinterpret_cast<uint16_t*>(packed)[idx4] = w;
    } else {  // 2 bit
      packed[idx4] = static_cast<uint8_t>((a & 3) | ((b & 3) << 2) | ((c & 3) << 4) |
                                          ((d & 3) << 6));
    }
  }
}

// ------------------------------------------------------------------------
// (A) small groups: BLOCKWISE_32/64/128/256 -> G4 = 8/16/32/64 float4 per group.
// The tensor is a flat run of groups; a 256-thread block streams a tile of
// U*256 float4 (U independent 1 KiB wave loads in flight per wave).
// ------------------------------------------------------------------------
template <int BITS, int G4, int U, bool BATCHED>
__global__ __launch_bounds__(256) void requant_groups_kernel(RequantArgs a) {
  const int t = BATCHED ? blockIdx.y : 0;
  const float4* __restrict__ x = pick<BATCHED, const float4>(a.x, t);
  int8_t* q = pick<BATCHED, int8_t>(a.q, t);
  uint8_t* packed = pick<BATCHED, uint8_t>(a.packed, t);
  float* scale = pick<BATCHED, float>(a.scale, t);
  uint16_t* scale_f16 = pick<BATCHED, uint16_t>(a.scale_f16, t);
  const float* clip = BATCHED ? nullptr : a.clip;

  const int64_t n4 = a.rows * a.cols / 4;
  const int64_t base = static_cast<int64_t>(blockIdx.x) * (256 * U) + threadIdx.x;

  float4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = base + u * 256;
    v[u] = i < n4 ? x[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = base + u * 256;
    uint32_t m = max(max(abs_bits(v[u].x), abs_bits(v[u].y)),
                     max(abs_bits(v[u].z), abs_bits(v[u].w)));
    m = group_max_u32<G4>(m);
    if (i < n4) {  // groups never straddle n4 (cols % block == 0)
      const int64_t g = i / G4;
      uint16_t hb = 0;
      const float s = make_scale<BITS, true>(m, clip, g, &hb);
      if ((threadIdx.x & (G4 - 1)) == 0) {
        scale[g] = s;
        if (scale_f16 != nullptr) scale_f16[g] = hb;
      }
      emit4<BITS>(v[u], s, i, q, packed);
    }
  }
}

// ------------------------------------------------------------------------
// (B) one scale per row, row held in registers: TPR threads x R float4 cover a
// row of cols4 <= TPR*R float4. TPR = 64 -> a wave owns the row (no LDS);
// TPR = 256 -> the block owns the row (one LDS exchange).
// ------------------------------------------------------------------------
template <int BITS, int TPR, int R, bool BATCHED>
__global__ __launch_bounds__(256) void requant_rows_kernel(RequantArgs a) {
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

  float4 v[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int c = j * TPR + lane;
    v[j] = (live && c < cols4) ? x[row4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  uint32_t m = 0;
#pragma unroll
  for (int j = 0; j < R; ++j) {
    m = max(m, max(max(abs_bits(v[j].x), abs_bits(v[j].y)),
                   max(abs_bits(v[j].z), abs_bits(v[j].w))));
  }
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
    if (c < cols4) emit4<BITS>(v[j], s, row4 + c, q, packed);
  }
}

// ------------------------------------------------------------------------
// (C) generic fallback: any cols (also cols % 4 != 0), any group length. One
// block per group, two sweeps (the second one hits L2). Packed output is not
// produced here (the host entry refuses ragged packing; use mi355q_pack_bits).
// ------------------------------------------------------------------------
template <int BITS, bool BLOCKWISE, bool BATCHED>
__global__ __launch_bounds__(256) void requant_generic_kernel(RequantArgs a) {
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

template <int BITS, bool BATCHED>
int32_t launch_bits(const RequantArgs& a, int count, bool aligned16, hipStream_t st) {
  const int64_t rows = a.rows, cols = a.cols;
  const dim3 blk(256);
  const unsigned gy = static_cast<unsigned>(count);
  const bool vec_ok = aligned16 && cols % 4 == 0;
  if (a.block > 0) {
    const int g4 = a.block / 4;
    const int64_t n4 = rows * cols / 4;
    if (vec_ok && (g4 == 8 || g4 == 16 || g4 == 32 || g4 == 64)) {
      constexpr int U = 4;
      const dim3 grid(static_cast<unsigned>((n4 + 256 * U - 1) / (256 * U)), gy);
      switch (g4) {
        case 8: hipLaunchKernelGGL((requant_groups_kernel<BITS, 8, U, BATCHED>), grid, blk, 0, st, a); break;
        case 16: hipLaunchKernelGGL((requant_groups_kernel<BITS, 16, U, BATCHED>), grid, blk, 0, st, a); break;
        case 32: hipLaunchKernelGGL((requant_groups_kernel<BITS, 32, U, BATCHED>), grid, blk, 0, st, a); break;
        default: hipLaunchKernelGGL((requant_groups_kernel<BITS, 64, U, BATCHED>), grid, blk, 0, st, a); break;
      }
    } else {
      if (a.packed != nullptr && BITS != 8)
        return fail(MI355Q_UNSUPPORTED, "packed output needs 16-byte aligned buffers and a block size in {32,64,128,256}");
      const dim3 grid(static_cast<unsigned>(rows * (cols / a.block)), gy);
      hipLaunchKernelGGL((requant_generic_kernel<BITS, true, BATCHED>), grid, blk, 0, st, a);
    }
  } else {
    const int64_t cols4 = cols / 4;
    if (vec_ok && cols4 <= 256 * 16) {
#define MI355Q_ROWS(TPR, R)                                                            \
  hipLaunchKernelGGL((requant_rows_kernel<BITS, TPR, R, BATCHED>),                      \
                     dim3(static_cast<unsigned>((rows + (256 / TPR) - 1) / (256 / TPR)), gy), \
                     blk, 0, st, a)
      if (cols4 <= 64) MI355Q_ROWS(64, 1);
      else if (cols4 <= 128) MI355Q_ROWS(64, 2);
      else if (cols4 <= 256) MI355Q_ROWS(64, 4);
      else if (cols4 <= 512) MI355Q_ROWS(256, 2);
      else if (cols4 <= 1024) MI355Q_ROWS(256, 4);
      else if (cols4 <= 2048) MI355Q_ROWS(256, 8);
      else MI355Q_ROWS(256, 16);
#undef MI355Q_ROWS
    } else {
      if (a.packed != nullptr && BITS != 8)
        return fail(MI355Q_UNSUPPORTED, "packed output needs cols %% 4 == 0, 16-byte aligned buffers and cols <= 16384");
      const dim3 grid(static_cast<unsigned>(rows), gy);
      hipLaunchKernelGGL((requant_generic_kernel<BITS, false, BATCHED>), grid, blk, 0, st, a);
    }
  }
  MI355Q_CHECK_LAUNCH("requant_sym launch");
  return MI355Q_OK;
}

template <bool BATCHED>
int32_t launch(const RequantArgs& a, int bits, int count, bool aligned16, hipStream_t st) {
  switch (bits) {
    case 8: return launch_bits<8, BATCHED>(a, count, aligned16, st);
    case 4: return launch_bits<4, BATCHED>(a, count, aligned16, st);
    case 2: return launch_bits<2, BATCHED>(a, count, aligned16, st);
    default: return fail(MI355Q_UNSUPPORTED, "bits must be 8, 4 or 2 (got %d)", bits);
  }
}

int32_t check_shape(int64_t rows, int64_t cols, int32_t block, int32_t bits,
                    bool want_packed) {
  if (rows < 0 || cols < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  if (block < 0) return fail(MI355Q_BAD_ARG, "negative block size");
  if (block > 0 && cols % block != 0)
    return fail(MI355Q_BAD_SHAPE,
                "Quantized dimension %lld is not divisible by block size %d.",
                static_cast<long long>(cols), block);
  if (rows > 0x7FFFFFFFLL) return fail(MI355Q_UNSUPPORTED, "rows > 2^31-1");
  if (want_packed && bits < 8 && (rows * cols) % (8 / bits) != 0)
    return fail(MI355Q_BAD_SHAPE, "packed output needs numel %% %d == 0", 8 / bits);
  return MI355Q_OK;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace
}  // namespace mi355q

using namespace mi355q;

extern "C" int32_t mi355q_requant_sym_f32(const float* x, int64_t rows, int64_t cols,
                                          int32_t block, int32_t bits, const float* clip,
                                          int8_t* q_out, uint8_t* packed_out,
                                          float* scale_out, uint16_t* scale_f16_out,
                                          void* stream) {
  clear_error();
  if (int32_t st = check_shape(rows, cols, block, bits, packed_out != nullptr)) return st;
  if (rows == 0 || cols == 0) return MI355Q_OK;
  if (x == nullptr || scale_out == nullptr)
    return fail(MI355Q_BAD_ARG, "x and scale_out must not be null");
  RequantArgs a{x, q_out, packed_out, scale_out, scale_f16_out, clip, rows, cols, block};
  const bool aligned = al16(x) && (q_out == nullptr || al16(q_out)) &&
                       (packed_out == nullptr || al16(packed_out));
  return launch<false>(a, bits, 1, aligned, as_stream(stream));
}

extern "C" int32_t mi355q_requant_sym_f32_batched(
    const float* const* x_ptrs, int32_t count, int64_t rows, int64_t cols, int32_t block,
    int32_t bits, int8_t* const* q_ptrs, uint8_t* const* packed_ptrs,
    float* const* scale_ptrs, uint16_t* const* scale_f16_ptrs, void* stream) {
  clear_error();
  if (count < 0 || count > 65535) return fail(MI355Q_BAD_ARG, "count must be in [0, 65535]");
  if (int32_t st = check_shape(rows, cols, block, bits, packed_ptrs != nullptr)) return st;
  if (count == 0 || rows == 0 || cols == 0) return MI355Q_OK;
  if (x_ptrs == nullptr || scale_ptrs == nullptr)
    return fail(MI355Q_BAD_ARG, "x_ptrs and scale_ptrs must not be null");
  RequantArgs a{x_ptrs, const_cast<int8_t**>(q_ptrs), const_cast<uint8_t**>(packed_ptrs),
                const_cast<float**>(scale_ptrs), const_cast<uint16_t**>(scale_f16_ptrs),
                nullptr, rows, cols, block};
  // The pointed-to buffers are required to be 16-byte aligned in the batched form
  // (hipMalloc / torch allocations are 256-byte aligned).
  return launch<true>(a, bits, count, true, as_stream(stream));
}
