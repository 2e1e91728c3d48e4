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
#include "requant_kernels.h"

namespace mi355q {
namespace {

using namespace requant;

// Tuning choices; tools/kbench/kbench.hip times the alternatives on MI355X with
// interleaved rounds (profiles/r01_kbench_variants.txt):
//   * rows kernel: non-temporal loads/stores (x is read once, q never re-read):
//     72.7 % -> 76.6 % of 8 TB/s on C2;
//   * blockwise kernel: every lane owns 8 consecutive floats (CL = 2) so packed
//     int4 leaves as full dwords (55 % -> 72 %); for sub-byte outputs the exact
//     reciprocal path (requant_kernels.h, one IEEE division per block instead of
//     one per element) adds another ~4 points; nt does not help there;
//   * int8 blockwise output: CL = 2 with two tiles in flight.
constexpr bool kRowsNT = true;
template <int BITS> constexpr bool kGroupsFast = BITS < 8;
template <int BITS> constexpr int kGroupsU = BITS == 8 ? 2 : 1;
template <int BITS> constexpr int kGroupsCL = 2;

inline bool wants_packed(const RequantArgs& a) { return a.packed != nullptr; }
inline bool wants_packed(const RequantInlineArgs& a) { return a.packed.p[0] != nullptr; }

template <int BITS, bool BATCHED, typename ARGS>
int32_t launch_bits(const ARGS& a, int count, bool aligned16, hipStream_t st) {
  const int64_t rows = a.rows, cols = a.cols;
  const dim3 blk(256);
  const unsigned gy = static_cast<unsigned>(count);
  const bool vec_ok = aligned16 && cols % 4 == 0;
  if (a.block > 0) {
    const int g4 = a.block / 4;
    const int64_t n4 = rows * cols / 4;
    if (vec_ok && (g4 == 8 || g4 == 16 || g4 == 32 || g4 == 64)) {
      constexpr int U = kGroupsU<BITS>, CL = kGroupsCL<BITS>;
      const dim3 grid(static_cast<unsigned>((n4 + 256 * U * CL - 1) / (256 * U * CL)), gy);
      switch (g4) {
        case 8: hipLaunchKernelGGL((requant_groups_kernel<BITS, 8, U, CL, kGroupsFast<BITS>, BATCHED, false, ARGS>), grid, blk, 0, st, a); break;
        case 16: hipLaunchKernelGGL((requant_groups_kernel<BITS, 16, U, CL, kGroupsFast<BITS>, BATCHED, false, ARGS>), grid, blk, 0, st, a); break;
        case 32: hipLaunchKernelGGL((requant_groups_kernel<BITS, 32, U, CL, kGroupsFast<BITS>, BATCHED, false, ARGS>), grid, blk, 0, st, a); break;
        default: hipLaunchKernelGGL((requant_groups_kernel<BITS, 64, U, CL, kGroupsFast<BITS>, BATCHED, false, ARGS>), grid, blk, 0, st, a); break;
      }
    } else {
      if (wants_packed(a) && BITS != 8)
        return fail(MI355Q_UNSUPPORTED, "packed output needs 16-byte aligned buffers and a block size in {32,64,128,256}");
      const dim3 grid(static_cast<unsigned>(rows * (cols / a.block)), gy);
      hipLaunchKernelGGL((requant_generic_kernel<BITS, true, BATCHED, ARGS>), grid, blk, 0, st, a);
    }
  } else {
    const int64_t cols4 = cols / 4;
    if (vec_ok && cols4 <= 256 * 16) {
#define MI355Q_ROWS(TPR, R)                                                            \
  hipLaunchKernelGGL((requant_rows_kernel<BITS, TPR, R, false, BATCHED, kRowsNT, ARGS>),                      \
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
      if (wants_packed(a) && BITS != 8)
        return fail(MI355Q_UNSUPPORTED, "packed output needs cols %% 4 == 0, 16-byte aligned buffers and cols <= 16384");
      const dim3 grid(static_cast<unsigned>(rows), gy);
      hipLaunchKernelGGL((requant_generic_kernel<BITS, false, BATCHED, ARGS>), grid, blk, 0, st, a);
    }
  }
  MI355Q_CHECK_LAUNCH("requant_sym launch");
  return MI355Q_OK;
}

template <bool BATCHED, typename ARGS>
int32_t launch(const ARGS& a, int bits, int count, bool aligned16, hipStream_t st) {
  switch (bits) {
    case 8: return launch_bits<8, BATCHED, ARGS>(a, count, aligned16, st);
    case 4: return launch_bits<4, BATCHED, ARGS>(a, count, aligned16, st);
    case 2: return launch_bits<2, BATCHED, ARGS>(a, count, aligned16, st);
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

extern "C" int32_t mi355q_requant_sym_f32_batched_hostptrs(
    const float* const* x_ptrs_host, int32_t count, int64_t rows, int64_t cols, int32_t block,
    int32_t bits, int8_t* const* q_ptrs_host, uint8_t* const* packed_ptrs_host,
    float* const* scale_ptrs_host, uint16_t* const* scale_f16_ptrs_host, void* stream) {
  clear_error();
  if (count < 0) return fail(MI355Q_BAD_ARG, "negative count");
  if (int32_t st = check_shape(rows, cols, block, bits, packed_ptrs_host != nullptr)) return st;
  if (count == 0 || rows == 0 || cols == 0) return MI355Q_OK;
  if (x_ptrs_host == nullptr || scale_ptrs_host == nullptr)
    return fail(MI355Q_BAD_ARG, "x_ptrs_host and scale_ptrs_host must not be null");
  for (int32_t i = 0; i < count; ++i) {
    if (!x_ptrs_host[i] || !scale_ptrs_host[i]) return fail(MI355Q_BAD_ARG, "null buffer pointer in a table (entry %d)", i);
    // (inputs are read as float4; int8 values leave as dwords, packed sub-byte values as dwords: 4-byte aligned outputs,
    // which equally shaped slices of one allocation are whenever cols % 4 == 0 -- the vector kernels' own condition)
    if (!al16(x_ptrs_host[i]))
      return fail(MI355Q_BAD_ARG, "the batched forms take 16-byte aligned inputs (entry %d)", i);
  }
  for (int32_t first = 0; first < count; first += kInlineTensors) {
    const int n = count - first < kInlineTensors ? count - first : kInlineTensors;
    RequantInlineArgs a{};
    for (int i = 0; i < n; ++i) {
      a.x.p[i] = x_ptrs_host[first + i];
      a.q.p[i] = q_ptrs_host ? q_ptrs_host[first + i] : nullptr;
      a.packed.p[i] = packed_ptrs_host ? packed_ptrs_host[first + i] : nullptr;
      a.scale.p[i] = scale_ptrs_host[first + i];
      a.scale_f16.p[i] = scale_f16_ptrs_host ? scale_f16_ptrs_host[first + i] : nullptr;
    }
    a.clip = nullptr;
    a.rows = rows; a.cols = cols; a.block = block;
    if (int32_t st = launch<true>(a, bits, n, true, as_stream(stream))) return st;
  }
  return MI355Q_OK;
}
