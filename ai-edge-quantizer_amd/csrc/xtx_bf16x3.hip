// GPTQ Hessian X^T X on the bf16 matrix cores with float32-class accuracy (gfx950).
//
//   ref: algorithms/uniform_quantize/gptq.py:100-107  (2.0 / num_samples) * x.T.dot(x), float32 sgemm
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 matrix rate (157 vs 2500 TFLOP/s), the FP32
// product holds the socket at its power limit (profiles/r02_gptq_mfma_util.txt), and the d = 16384
// Hessian of a Gemma down_proj (65 536 tokens: 17.6 TFLOP on the triangle) is the largest single
// cost of a GPTQ layer. Every float32 is the exact sum of three bfloat16 numbers
// (x = x1 + x2 + x3, 8 significant bits each; the residuals x - x1 and x - x1 - x2 are exact in
// float32), and a product of two bfloat16 numbers is exact in float32, so
//   x * y = x1 y1 + (x1 y2 + x2 y1) + (x1 y3 + x2 y2 + x3 y1) + O(2^-25 |x y|)
// with every kept term exact: six v_mfma_f32_32x32x16_bf16 products accumulated in float32 give
// the float32 dot product to within the rounding noise of its own accumulation (what is dropped
// is below half an ulp of x y). Like the sgemm it replaces, the result depends on the order of the
// float32 additions (tolerance class T2). An infinite activation gives +-inf where x.T.dot(x) does
// (the x1 y1 sum decides; the NaN of an inf * 0 cross term is dropped); the damped Cholesky refuses it.
//
// Two kernels per slab of <= 16384 tokens:
//   split   X [n, d] float32 -> P[k tile of 16 tokens][plane 0..2][row i < d][32 bytes]: the 16
//           tokens of row i as two 16-byte chunks, swapped where (i >> 3) & 1. A 128-row
//           operand tile of one plane and k tile is 4 KB of contiguous memory in exactly the image
//           LDS needs, so it is staged by global_load_lds (no registers, no ds_write), and the
//           swizzle makes every 16-lane group of the ds_read_b128 fragment reads touch all 64 banks.
//   xtx     lower-triangular grid of 128 x 128 output tiles (8 x 8 patches of tiles per XCD so that
//           what shares an L2 shares its operand panels); per k tile of 16 tokens the three planes of
//           both operand tiles (24 KB, double-buffered) go to LDS and every wave issues 24 MFMAs on
//           its 64 x 64 quadrant from 12 fragment reads -- six products per pair of fragments, so
//           the L2 -> LDS traffic per MFMA is half that of a plain bf16 GEMM with this tile.
//           Accumulation: the five cross terms have FP32 accumulators of their own (they do not round
//           against the large x1 y1 sums), and the x1 y1 accumulators are folded into a third set
//           every 32 k tiles, so no chain of float32 additions is longer than 32 + 32 per slab:
//           2e-7 of the largest entry against the FP64 product at d = 16384 x 16384 tokens, where
//           the FP32-MFMA product (one chain of 8192 additions) is at 4e-6.
#include "common.h"

namespace mi355q {
namespace {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

constexpr int kTile = 128;                  // output tile edge
constexpr int kBK = 16;                     // tokens per k tile (one MFMA K step)
constexpr int kRowB = 2 * kBK;              // bytes of one row of one plane in one k tile
constexpr int kPlaneTileB = kTile * kRowB;  // 8 KB
constexpr int kOperandB = 3 * kPlaneTileB;  // 24 KB
constexpr int kSuper = 8;                   // tiles per side of an XCD patch
constexpr int kSlabTokens = 16384;
constexpr int kFold = 32;                   // k tiles per first-level accumulation chain

__device__ __forceinline__ unsigned bf16_rne_bits(float x) {
  unsigned b = __float_as_uint(x);
  if ((b & 0x7FFFFFFFu) > 0x7F800000u) return (b >> 16) | 0x40u;   // NaN stays NaN
  const unsigned r = (b + 0x7FFFu + ((b >> 16) & 1u)) >> 16;
  return ((r & 0x7F80u) == 0x7F80u && (b & 0x7F800000u) != 0x7F800000u) ? b >> 16 : r;   // never round a finite value to inf
}

// grid (d / 64, pairs of k tiles), 256 threads; tokens [k0, k_end) of x, zero beyond
__global__ __launch_bounds__(256) void xtx_split_kernel(const float* __restrict__ x, int d, long long k0,
                                                       long long k_end, int kt_total, unsigned char* __restrict__ planes) {
  __shared__ float tile[32][65];
  const int i0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 32; r += 4) {
    const long long k = k0 + static_cast<long long>(blockIdx.y) * 32 + r;
    tile[r][tx] = k < k_end ? x[k * d + i0 + tx] : 0.f;
  }
  __syncthreads();
  const int ii = threadIdx.x >> 2, c = threadIdx.x & 3, i = i0 + ii;
  const int kt = 2 * blockIdx.y + (c >> 1);
  if (kt >= kt_total) return;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = tile[8 * c + j][ii];
  const int cs = (c & 1) ^ ((i >> 3) & 1);
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool finite = (__float_as_uint(v[j]) & 0x7F800000u) != 0x7F800000u;
      const unsigned bits = bf16_rne_bits(v[j]);
      w[j >> 1] |= bits << (16 * (j & 1));
      v[j] = finite ? v[j] - __uint_as_float(bits << 16) : 0.f;   // exact
    }
    unsigned char* dst = planes + ((static_cast<long long>(kt) * 3 + p) * d + i) * kRowB + cs * 16;
    *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

struct XtxArgs {
  const unsigned char* planes;   // [k tile][3][d][64 B]
  int d, tiles;                  // tiles = d / 128
  int kt_total, kt_per_split;
  float* c;                      // [d, d] float32 (lower-triangular tiles), or partials [split][d][d]
  int accumulate;                // c += product (direct mode)
  int partial;                   // write split z's product to c + z d d
  int patches;                   // 1: 8 x 8 patches of tiles dealt to XCDs; 0: plain triangular list
  int tri;                       // X is lower triangular (X[k][i] == 0 for k < i): tile row ti starts at k = 128 ti
};

__global__ __launch_bounds__(256) void xtx_bf16x3_kernel(XtxArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  int ti, tj;
  if (a.patches) {
    const int b = blockIdx.x, xcd = b & 7, local = b >> 3;
    const int sup = (local / (kSuper * kSuper)) * 8 + xcd, within = local % (kSuper * kSuper);
    int si = static_cast<int>((__builtin_sqrtf(8.0f * static_cast<float>(sup) + 1.0f) - 1.0f) * 0.5f);
    while ((si + 1) * (si + 2) / 2 <= sup) ++si;
    while (si * (si + 1) / 2 > sup) --si;
    const int sj = sup - si * (si + 1) / 2;
    ti = si * kSuper + within / kSuper;
    tj = sj * kSuper + within % kSuper;
    if (ti >= a.tiles || tj > ti) return;
  } else {
    const int b = blockIdx.x;
    ti = static_cast<int>((__builtin_sqrtf(8.0f * static_cast<float>(b) + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= b) ++ti;
    while (ti * (ti + 1) / 2 > b) --ti;
    tj = b - ti * (ti + 1) / 2;
  }
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wave >> 1, wc = wave & 1;                 // this wave's 64 x 64 quadrant
  const int kt0 = a.tri ? ti * (kTile / kBK) : blockIdx.y * a.kt_per_split;
  const int kt1 = a.tri ? a.kt_total : min(a.kt_total, kt0 + a.kt_per_split);

  // x1 y1, the five cross terms (the small ones do not round against the large sum), and the sum of the
  // x1 y1 accumulators folded away every kFold k tiles: chains of 32 + 32 additions instead of 1024
  f32x16 acc[2][2], lo[2][2], top[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = lo[i][j][r] = top[i][j][r] = 0.f;

  // fragment addresses: row = quadrant + 32 i + (lane & 31); chunk (lane >> 5) ^ ((row >> 3) & 1)
  const int frow = lane & 31;
  const int fch = ((lane >> 5) ^ ((frow >> 3) & 1)) * 16;
  const int offA = (wr * 64 + frow) * kRowB + fch, offB = kOperandB + (wc * 64 + frow) * kRowB + fch;

  const long long row_stride = static_cast<long long>(a.d) * kRowB;   // one plane of one k tile
  const unsigned char* gA = a.planes + static_cast<long long>(ti) * kPlaneTileB + lane * 16;
  const unsigned char* gB = a.planes + static_cast<long long>(tj) * kPlaneTileB + lane * 16;

  // 24 wave-wide 1 KB pieces per k tile: operand (A, B) x plane x 4 pieces of 32 rows; wave w takes w, w + 4, ...
  auto stage = [&](int kt, int buf) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int piece = q * 4 + wave;                 // 0 .. 23
      const int op = piece / 12, r = piece % 12, p = r >> 2, seg = r & 3;
      const unsigned char* src = (op ? gB : gA) + (static_cast<long long>(kt) * 3 + p) * row_stride + seg * 1024;
      unsigned char* dst = lds + buf * (2 * kOperandB) + op * kOperandB + p * kPlaneTileB + seg * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  if (kt0 < kt1) stage(kt0, 0);
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();          // this k tile has landed; every wave is done with the other buffer
    if (kt + 1 < kt1) stage(kt + 1, buf ^ 1);
    const unsigned char* img = lds + buf * (2 * kOperandB);
    bf16x8 fa[2][3], fb[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fa[i][p] = *reinterpret_cast<const bf16x8*>(img + offA + p * kPlaneTileB + i * 32 * kRowB);
        fb[i][p] = *reinterpret_cast<const bf16x8*>(img + offB + p * kPlaneTileB + i * 32 * kRowB);
      }
    // the five cross terms (x1 y3 + x2 y2 + x3 y1) + (x1 y2 + x2 y1) and x1 y1 have accumulators of their own
#define MI355Q_TERM(ACC, PA, PB)                                                                \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)    \
      ACC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA], fb[j][PB], ACC[i][j], 0, 0, 0)
    MI355Q_TERM(lo, 0, 2);
    MI355Q_TERM(lo, 1, 1);
    MI355Q_TERM(lo, 2, 0);
    MI355Q_TERM(lo, 0, 1);
    MI355Q_TERM(lo, 1, 0);
    MI355Q_TERM(acc, 0, 0);
#undef MI355Q_TERM
    if (((kt - kt0) & (kFold - 1)) == kFold - 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            top[i][j][r] = top[i][j][r] + acc[i][j][r];
            acc[i][j][r] = 0.f;
          }
    }
  }

  // C/D layout of the 32 x 32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  float* c = a.c + (a.partial ? static_cast<long long>(blockIdx.y) * a.d * a.d : 0);
  float* base = c + static_cast<long long>(ti * kTile + wr * 64 + 4 * (lane >> 5)) * a.d + tj * kTile + wc * 64 + (lane & 31);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    // (a later slab adds to the product so far: the 32 loads of a lane go out together, then the
    // stores -- one at a time, every store waited for its own load)
    float old[2][16];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        old[j][r] = a.accumulate ? base[static_cast<long long>(i * 32 + (r & 3) + 8 * (r >> 2)) * a.d + j * 32] : 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // (an infinite x1 y1 sum is the product's value: the cross terms of a non-finite operand are
        // inf * 0 = NaN whenever the other side's residual planes are zero, where x.T.dot(x) has +-inf)
        const float big = top[i][j][r] + acc[i][j][r];
        const float v = __builtin_isinf(big) ? big : big + lo[i][j][r];
        base[static_cast<long long>(i * 32 + (r & 3) + 8 * (r >> 2)) * a.d + j * 32] = a.accumulate ? old[j][r] + v : v;
      }
  }
}

// The same product on 128 x 256 output tiles, eight waves per workgroup (2 x 4 quadrants of 64 x 64; a wave's
// accumulators, fragment reads and 24 MFMAs per k tile unchanged), two k tiles per stage: a stage is 72 KB for 384
// MFMAs where the kernel above stages 24 KB for 96 -- a quarter fewer bytes out of the L2 per MFMA and half the
// barriers -- double-buffered (144 KB of LDS, one workgroup per CU, two waves per SIMD). Round 4: the exact split is
// the default Hessian product again (parity), so what xtx_f16x2.hip gained from wide tiles is taken here too. An XCD's
// patch is 8 x 4 of these tiles = its 32 CUs, resident together. Tile (ti, tj) covers rows 128 ti.. and columns
// 256 tj..; it is needed when 2 tj <= ti (the tile on the diagonal of an even tile row also computes 128 columns
// above the diagonal: valid entries of the symmetric product, never read). Same products, same accumulator
// structure, same k order per output as the narrow kernel: the same bits (tests/test_gpu_gptq.py).
constexpr int kWideA = 3 * kPlaneTileB;            // 12 KB: three planes of 128 rows, one k tile
constexpr int kWideB = 6 * kPlaneTileB;            // 24 KB: three planes of 256 rows
constexpr int kWideKt = kWideA + kWideB;           // 36 KB per k tile
constexpr int kWideStageB = 2 * kWideKt;           // 72 KB: two k tiles

__global__ __launch_bounds__(512) void xtx_bf16x3_wide_kernel(XtxArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int b = blockIdx.x, xcd = b & 7, local = b >> 3;
  const int sup = (local / 32) * 8 + xcd, within = local % 32;
  int si = static_cast<int>((__builtin_sqrtf(8.0f * static_cast<float>(sup) + 1.0f) - 1.0f) * 0.5f);
  while ((si + 1) * (si + 2) / 2 <= sup) ++si;
  while (si * (si + 1) / 2 > sup) --si;
  const int sj = sup - si * (si + 1) / 2;
  const int ti = si * kSuper + within / 4;          // 128-row tile
  const int tj = sj * (kSuper / 2) + within % 4;    // 256-column tile
  if (ti >= a.tiles || 2 * tj > ti) return;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wave >> 2, wc = wave & 3;           // this wave's 64 x 64 quadrant of the 128 x 256 tile
  const int kt0 = 0, kt1 = a.kt_total;               // (no split-K: wide tiles are for d >= 4096)

  f32x16 acc[2][2], lo[2][2], top[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = lo[i][j][r] = top[i][j][r] = 0.f;

  const int frow = lane & 31;
  const int fch = ((lane >> 5) ^ ((frow >> 3) & 1)) * 16;
  const int offA = (wr * 64 + frow) * kRowB + fch, offB = kWideA + (wc * 64 + frow) * kRowB + fch;

  const long long row_stride = static_cast<long long>(a.d) * kRowB;   // one plane of one k tile
  const unsigned char* gA = a.planes + static_cast<long long>(ti) * kPlaneTileB + lane * 16;
  const unsigned char* gB = a.planes + static_cast<long long>(tj) * (2 * kPlaneTileB) + lane * 16;

  // 36 wave-wide 1 KB pieces per k tile: 12 of A (plane x 4 pieces of 32 rows) and 24 of B (plane x 8); 72 per stage,
  // nine per wave
  auto stage = [&](int kt, int buf, int count) {
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int piece = q * 8 + wave;                 // 0 .. 71
      const int kk = piece / 36, r = piece % 36;
      const bool isB = r >= 12;
      const int pl = isB ? (r - 12) >> 3 : r >> 2, seg = isB ? (r - 12) & 7 : r & 3;
      const unsigned char* src = (isB ? gB : gA) + (static_cast<long long>(kt + (kk < count ? kk : 0)) * 3 + pl) * row_stride + seg * 1024;
      unsigned char* dst = lds + buf * kWideStageB + kk * kWideKt + (isB ? kWideA + pl * (2 * kPlaneTileB) : pl * kPlaneTileB) + seg * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  const int nst = (kt1 - kt0 + 1) / 2;               // (an odd last k tile: the stage's second half is loaded again from the first and skipped)
  if (nst > 0) stage(kt0, 0, kt1 - kt0 >= 2 ? 2 : 1);
  for (int s = 0; s < nst; ++s) {
    const int buf = s & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier_loads_in_flight();   // this stage has landed; every wave is done with the other buffer
    if (s + 1 < nst) stage(kt0 + 2 * (s + 1), buf ^ 1, kt1 - (kt0 + 2 * (s + 1)) >= 2 ? 2 : 1);
    const int here = kt1 - (kt0 + 2 * s) >= 2 ? 2 : 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (kk < here) {
        const unsigned char* img = lds + buf * kWideStageB + kk * kWideKt;
        bf16x8 fa[2][3], fb[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            fa[i][pl] = *reinterpret_cast<const bf16x8*>(img + offA + pl * kPlaneTileB + i * 32 * kRowB);
            fb[i][pl] = *reinterpret_cast<const bf16x8*>(img + offB + pl * (2 * kPlaneTileB) + i * 32 * kRowB);
          }
#define MI355Q_TERM(ACC, PA, PB)                                                                \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)    \
      ACC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA], fb[j][PB], ACC[i][j], 0, 0, 0)
        MI355Q_TERM(lo, 0, 2);
        MI355Q_TERM(lo, 1, 1);
        MI355Q_TERM(lo, 2, 0);
        MI355Q_TERM(lo, 0, 1);
        MI355Q_TERM(lo, 1, 0);
        MI355Q_TERM(acc, 0, 0);
#undef MI355Q_TERM
        if (((2 * s + kk) & (kFold - 1)) == kFold - 1) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                top[i][j][r] = top[i][j][r] + acc[i][j][r];
                acc[i][j][r] = 0.f;
              }
        }
      }
    }
  }

  float* base = a.c + static_cast<long long>(ti * kTile + wr * 64 + 4 * (lane >> 5)) * a.d + tj * (2 * kTile) + wc * 64 + (lane & 31);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float old[16];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        old[r] = a.accumulate ? base[static_cast<long long>(i * 32 + (r & 3) + 8 * (r >> 2)) * a.d + j * 32] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float big = top[i][j][r] + acc[i][j][r];
        const float v = __builtin_isinf(big) ? big : big + lo[i][j][r];
        base[static_cast<long long>(i * 32 + (r & 3) + 8 * (r >> 2)) * a.d + j * 32] = a.accumulate ? old[r] + v : v;
      }
    }
  }
}

// The wide kernel with FOUR one-k-tile buffers instead of two two-k-tile stages (the same 144 KB of LDS): see the staging
// comment inside. SQ counters of the kernel above (profiles/r05_xtx_bound.txt): its waves spend 32 % of their cycles in
// s_waitcnt / s_barrier -- all eight at the same time -- and the MFMA pipe is 57 % busy. Same products, same accumulator
// structure, same k order per output: the same bits (tests/test_gpu_gptq.py).
template <int ALT>
__global__ __launch_bounds__(512) void xtx_bf16x3_deep_kernel(XtxArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int b = blockIdx.x, xcd = b & 7, local = b >> 3;
  const int sup = (local / 32) * 8 + xcd, within = local % 32;
  int si = static_cast<int>((__builtin_sqrtf(8.0f * static_cast<float>(sup) + 1.0f) - 1.0f) * 0.5f);
  while ((si + 1) * (si + 2) / 2 <= sup) ++si;
  while (si * (si + 1) / 2 > sup) --si;
  const int sj = sup - si * (si + 1) / 2;
  const int ti = si * kSuper + within / 4;          // 128-row tile
  const int tj = sj * (kSuper / 2) + within % 4;    // 256-column tile
  if (ti >= a.tiles || 2 * tj > ti) return;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wave >> 2, wc = wave & 3;           // this wave's 64 x 64 quadrant of the 128 x 256 tile
  const int kt0 = 0, kt1 = a.kt_total;               // (no split-K: wide tiles are for d >= 4096)

  f32x16 acc[2][2], lo[2][2], top[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = lo[i][j][r] = top[i][j][r] = 0.f;

  const int frow = lane & 31;
  const int fch = ((lane >> 5) ^ ((frow >> 3) & 1)) * 16;
  const int offA = (wr * 64 + frow) * kRowB + fch, offB = kWideA + (wc * 64 + frow) * kRowB + fch;

  const long long row_stride = static_cast<long long>(a.d) * kRowB;   // one plane of one k tile
  const unsigned char* gA = a.planes + static_cast<long long>(ti) * kPlaneTileB + lane * 16;
  const unsigned char* gB = a.planes + static_cast<long long>(tj) * (2 * kPlaneTileB) + lane * 16;

  // 36 wave-wide 1 KB pieces per k tile: 12 of A (plane x 4 pieces of 32 rows) and 24 of B (plane x 8); wave w takes pieces
  // w, w + 8, w + 16, w + 24 and -- waves 0..3 -- w + 32. FOUR buffers of one k tile: the pieces of k tile t + 3 go out
  // behind the barrier that opens k tile t, so a piece has three k tiles of MFMAs (2300 cycles per SIMD at full rate) to land
  // where the two-k-tile stages of the kernel above leave it one stage, and a wave waits for ITS OWN pieces of k tile t only
  // (s_waitcnt vmcnt(n): those of t + 1 and t + 2 stay in flight).
  // HALF of the waves stage a k tile, nine pieces each: waves 0-3 the even k tiles, waves 4-7 the odd ones (ALT == 2: even /
  // odd waves). A piece costs its wave ~100 cycles of issue in which it issues no MFMA; when both waves of a SIMD stage behind
  // the same barrier the SIMD's MFMA pipe idles for as long, when one does its partner's 24 MFMAs run underneath.
  // (ALT == 0: every wave stages 4-5 pieces of every k tile. Same box, alternating runs: ALT 0 / 1 / 2 = 23.4 / 22.7 / 25.6 ms.)
  const int team = ALT == 2 ? (wave & 1) : (wave >> 2);
  const int slot = ALT == 2 ? (wave >> 1) : (wave & 3);          // 0 .. 3 within the team
  auto stage = [&](int kt, int buf) {
    if constexpr (ALT != 0) {
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const int r = q * 4 + slot;                   // 0 .. 35
        const bool isB = r >= 12;
        const int pl = isB ? (r - 12) >> 3 : r >> 2, seg = isB ? (r - 12) & 7 : r & 3;
        const unsigned char* src = (isB ? gB : gA) + (static_cast<long long>(kt) * 3 + pl) * row_stride + seg * 1024;
        unsigned char* dst = lds + buf * kWideKt + (isB ? kWideA + pl * (2 * kPlaneTileB) : pl * kPlaneTileB) + seg * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const int r = q * 8 + wave;                     // 0 .. 35
        if (q == 4 && wave >= 4) break;
        const bool isB = r >= 12;
        const int pl = isB ? (r - 12) >> 3 : r >> 2, seg = isB ? (r - 12) & 7 : r & 3;
        const unsigned char* src = (isB ? gB : gA) + (static_cast<long long>(kt) * 3 + pl) * row_stride + seg * 1024;
        unsigned char* dst = lds + buf * kWideKt + (isB ? kWideA + pl * (2 * kPlaneTileB) : pl * kPlaneTileB) + seg * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
  };
  auto stages = [&](int kt) { return ALT == 0 || ((kt - kt0) & 1) == team; };
#pragma unroll
  for (int pre = 0; pre < 3; ++pre)
    if (kt0 + pre < kt1 && stages(kt0 + pre)) stage(kt0 + pre, pre);
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 3;
    // this wave's pieces of k tile kt have landed (the pieces of later k tiles it issued since may still be in flight)
    if constexpr (ALT != 0) {
      if (stages(kt)) {                 // ... it also staged kt + 2, nothing in between
        if (kt + 2 < kt1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else {
      const int later = (kt + 1 < kt1 ? 1 : 0) + (kt + 2 < kt1 ? 1 : 0);
      if (wave < 4) {
        if (later == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        if (later == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    barrier_loads_in_flight();   // every wave's pieces of this k tile have landed; every wave is done with k tile kt - 1's buffer
    if (kt + 3 < kt1 && stages(kt + 3)) stage(kt + 3, (buf + 3) & 3);
    {
      const unsigned char* img = lds + buf * kWideKt;
      bf16x8 fa[2][3], fb[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          fa[i][pl] = *reinterpret_cast<const bf16x8*>(img + offA + pl * kPlaneTileB + i * 32 * kRowB);
          fb[i][pl] = *reinterpret_cast<const bf16x8*>(img + offB + pl * (2 * kPlaneTileB) + i * 32 * kRowB);
        }
#define MI355Q_TERM(ACC, PA, PB)                                                                \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)    \
      ACC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA], fb[j][PB], ACC[i][j], 0, 0, 0)
      MI355Q_TERM(lo, 0, 2);
      MI355Q_TERM(lo, 1, 1);
      MI355Q_TERM(lo, 2, 0);
      MI355Q_TERM(lo, 0, 1);
      MI355Q_TERM(lo, 1, 0);
      MI355Q_TERM(acc, 0, 0);
#undef MI355Q_TERM
      if (((kt - kt0) & (kFold - 1)) == kFold - 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              top[i][j][r] = top[i][j][r] + acc[i][j][r];
              acc[i][j][r] = 0.f;
            }
      }
    }
  }

  float* base = a.c + static_cast<long long>(ti * kTile + wr * 64 + 4 * (lane >> 5)) * a.d + tj * (2 * kTile) + wc * 64 + (lane & 31);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float old[16];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        old[r] = a.accumulate ? base[static_cast<long long>(i * 32 + (r & 3) + 8 * (r >> 2)) * a.d + j * 32] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float big = top[i][j][r] + acc[i][j][r];
        const float v = __builtin_isinf(big) ? big : big + lo[i][j][r];
        base[static_cast<long long>(i * 32 + (r & 3) + 8 * (r >> 2)) * a.d + j * 32] = a.accumulate ? old[r] + v : v;
      }
    }
  }
}

// ---- the same exact split for GPTQ's update behind a group of columns (gptq.hip):
//   W[:, g1:] -= E @ Hinv[g0:g1, g1:],  E = the group's errors [rows, kk] float32 (kk <= 256).
// Hinv's planes are made once per call by xtx_split_kernel (x = Hinv: "token" = row k of Hinv,
// "feature" = column j), E's per group by upd_split_err_kernel; upd_bf16x3_kernel is the product
// kernel above with a rectangular grid, 16 k tiles and a read-modify-write epilogue.
// grid (rows / 8), 256 threads: thread -> (row, chunk of 8 k); e[row][k] with leading dimension ld
__global__ __launch_bounds__(256) void upd_split_err_kernel(const float* __restrict__ e, int rows, int ld, int kk,
                                                           unsigned char* __restrict__ planes) {
  const int chunks = kk / 8;                       // per row
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int row = idx / chunks, ch = idx % chunks;
  if (row >= rows) return;
  const float4 lo4 = *reinterpret_cast<const float4*>(e + static_cast<long long>(row) * ld + ch * 8);
  const float4 hi4 = *reinterpret_cast<const float4*>(e + static_cast<long long>(row) * ld + ch * 8 + 4);
  float v[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
  const int kt = ch >> 1, cs = (ch & 1) ^ ((row >> 3) & 1);
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool finite = (__float_as_uint(v[j]) & 0x7F800000u) != 0x7F800000u;
      const unsigned bits = bf16_rne_bits(v[j]);
      w[j >> 1] |= bits << (16 * (j & 1));
      v[j] = finite ? v[j] - __uint_as_float(bits << 16) : 0.f;
    }
    unsigned char* dst = planes + ((static_cast<long long>(kt) * 3 + p) * rows + row) * kRowB + cs * 16;
    *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

struct UpdArgs {
  const unsigned char* eplanes;   // [k tile][3][rows][32 B]
  const unsigned char* hplanes;   // [k tile of Hinv rows][3][d][32 B]
  int rows, d;
  int kt_count;                   // k tiles of the group
  int h_kt0;                      // first k tile of the group in Hinv (g0 / 16)
  int col0;                       // first output column (g1)
  float* w;                       // [rows, d]
};

// grid (column tiles, row tiles)
__global__ __launch_bounds__(256) void upd_bf16x3_kernel(UpdArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int ti = blockIdx.y, tj = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  f32x16 acc[2][2], lo[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = lo[i][j][r] = 0.f;
  const int frow = lane & 31;
  const int fch = ((lane >> 5) ^ ((frow >> 3) & 1)) * 16;
  const int offA = (wr * 64 + frow) * kRowB + fch, offB = kOperandB + (wc * 64 + frow) * kRowB + fch;
  const long long strideA = static_cast<long long>(a.rows) * kRowB, strideB = static_cast<long long>(a.d) * kRowB;
  const unsigned char* gA = a.eplanes + static_cast<long long>(ti) * kPlaneTileB + lane * 16;
  const unsigned char* gB = a.hplanes + (static_cast<long long>(a.col0) + static_cast<long long>(tj) * kTile) * kRowB + lane * 16;
  auto stage = [&](int kt, int buf) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int piece = q * 4 + wave;
      const int op = piece / 12, r = piece % 12, p = r >> 2, seg = r & 3;
      const unsigned char* src = op ? gB + (static_cast<long long>(a.h_kt0 + kt) * 3 + p) * strideB + seg * 1024
                                    : gA + (static_cast<long long>(kt) * 3 + p) * strideA + seg * 1024;
      unsigned char* dst = lds + buf * (2 * kOperandB) + op * kOperandB + p * kPlaneTileB + seg * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  stage(0, 0);
  for (int kt = 0; kt < a.kt_count; ++kt) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < a.kt_count) stage(kt + 1, buf ^ 1);
    const unsigned char* img = lds + buf * (2 * kOperandB);
    bf16x8 fa[2][3], fb[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fa[i][p] = *reinterpret_cast<const bf16x8*>(img + offA + p * kPlaneTileB + i * 32 * kRowB);
        fb[i][p] = *reinterpret_cast<const bf16x8*>(img + offB + p * kPlaneTileB + i * 32 * kRowB);
      }
#define MI355Q_TERM(ACC, PA, PB)                                                                \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)    \
      ACC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA], fb[j][PB], ACC[i][j], 0, 0, 0)
    MI355Q_TERM(lo, 0, 2);
    MI355Q_TERM(lo, 1, 1);
    MI355Q_TERM(lo, 2, 0);
    MI355Q_TERM(lo, 0, 1);
    MI355Q_TERM(lo, 1, 0);
    MI355Q_TERM(acc, 0, 0);
#undef MI355Q_TERM
  }
  // read-modify-write of the tile in two phases (all 64 loads of a lane in flight, then the stores:
  // one at a time, every store waited for its own load)
  float* base = a.w + static_cast<long long>(ti * kTile + wr * 64 + 4 * (lane >> 5)) * a.d + a.col0 + tj * kTile + wc * 64 + (lane & 31);
  float old[2][2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        old[i][j][r] = __builtin_nontemporal_load(base + static_cast<long long>(i * 32 + (r & 3) + 8 * (r >> 2)) * a.d + j * 32);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __builtin_nontemporal_store(old[i][j][r] - (acc[i][j][r] + lo[i][j][r]),
                                    base + static_cast<long long>(i * 32 + (r & 3) + 8 * (r >> 2)) * a.d + j * 32);
}

// Two column tiles per workgroup: the E fragments are read once for both, a k tile carries 48 MFMAs
// per wave between two barriers instead of 24, and the prologue is paid once per 128 x 256 outputs.
// All six terms go to one accumulator per tile (16 k tiles: the cross terms' rounding against the
// x1 y1 sums is far below the float32 update's own).
__global__ __launch_bounds__(256) void upd2_bf16x3_kernel(UpdArgs a, int ntj) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  constexpr int kImage = 3 * kOperandB;            // A, B0, B1
  const int ti = blockIdx.y, tj0 = 2 * blockIdx.x;
  const bool two = tj0 + 1 < ntj;
  const int tj1 = two ? tj0 + 1 : tj0;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  f32x16 acc[2][2][2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][i][j][r] = 0.f;
  const int frow = lane & 31;
  const int fch = ((lane >> 5) ^ ((frow >> 3) & 1)) * 16;
  const int offA = (wr * 64 + frow) * kRowB + fch, offB = kOperandB + (wc * 64 + frow) * kRowB + fch;
  const long long strideA = static_cast<long long>(a.rows) * kRowB, strideB = static_cast<long long>(a.d) * kRowB;
  const unsigned char* gA = a.eplanes + static_cast<long long>(ti) * kPlaneTileB + lane * 16;
  const unsigned char* gB0 = a.hplanes + (static_cast<long long>(a.col0) + static_cast<long long>(tj0) * kTile) * kRowB + lane * 16;
  const unsigned char* gB1 = a.hplanes + (static_cast<long long>(a.col0) + static_cast<long long>(tj1) * kTile) * kRowB + lane * 16;
  auto stage = [&](int kt, int buf) {
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int piece = q * 4 + wave;                 // 0 .. 35
      const int op = piece / 12, r = piece % 12, p = r >> 2, seg = r & 3;
      const unsigned char* src = op == 0 ? gA + (static_cast<long long>(kt) * 3 + p) * strideA + seg * 1024
                                         : (op == 1 ? gB0 : gB1) + (static_cast<long long>(a.h_kt0 + kt) * 3 + p) * strideB + seg * 1024;
      unsigned char* dst = lds + buf * kImage + op * kOperandB + p * kPlaneTileB + seg * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  stage(0, 0);
  for (int kt = 0; kt < a.kt_count; ++kt) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < a.kt_count) stage(kt + 1, buf ^ 1);
    const unsigned char* img = lds + buf * kImage;
    bf16x8 fa[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) fa[i][p] = *reinterpret_cast<const bf16x8*>(img + offA + p * kPlaneTileB + i * 32 * kRowB);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      bf16x8 fb[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          fb[i][p] = *reinterpret_cast<const bf16x8*>(img + b * kOperandB + offB + p * kPlaneTileB + i * 32 * kRowB);
#define MI355Q_TERM(PA, PB)                                                                     \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)    \
      acc[b][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA], fb[j][PB], acc[b][i][j], 0, 0, 0)
      MI355Q_TERM(0, 2);
      MI355Q_TERM(1, 1);
      MI355Q_TERM(2, 0);
      MI355Q_TERM(0, 1);
      MI355Q_TERM(1, 0);
      MI355Q_TERM(0, 0);
#undef MI355Q_TERM
    }
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    if (b == 1 && !two) break;
    float* base = a.w + static_cast<long long>(ti * kTile + wr * 64 + 4 * (lane >> 5)) * a.d + a.col0 +
                  (b ? tj1 : tj0) * kTile + wc * 64 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i) {      // 32 loads of a lane in flight, then their stores
      float old[2][16];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          old[j][r] = __builtin_nontemporal_load(base + static_cast<long long>(i * 32 + (r & 3) + 8 * (r >> 2)) * a.d + j * 32);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          __builtin_nontemporal_store(old[j][r] - acc[b][i][j][r],
                                      base + static_cast<long long>(i * 32 + (r & 3) + 8 * (r >> 2)) * a.d + j * 32);
    }
  }
}

// ---- the same split for the GEMMs of the Hessian inverse (gptq.hip): operands are FP64 blocks of
// the factor; every element is rounded to float32 first (the reference runs these steps in single
// precision: scipy's strtri and a float32 einsum, ref gptq.py:121-128), then split exactly.
//
// B-type operand: element (k, j) = src[k * ld + j] -> planes[kt][p][row0 + j]; grid (cols / 64, pairs of
// k tiles, items); with `lower` blocks that only hold elements above the diagonal (k < j) are skipped
// (they are never read: the products start at the diagonal).
template <typename T>
__global__ __launch_bounds__(256) void split_cols_kernel(const T* __restrict__ src, long long ld, long long item_stride,
                                                        int kk, int kt_total, int rows_per_item, int plane_rows,
                                                        int lower, unsigned char* __restrict__ planes) {
  __shared__ float tile[32][65];
  const int i0 = blockIdx.x * 64;
  const int kb = blockIdx.y * 32;
  if (lower && kb + 31 < (i0 / kTile) * kTile) return;   // (the products start at the 128-aligned diagonal tile)
  const T* x = src + static_cast<long long>(blockIdx.z) * item_stride;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 32; r += 4) {
    const int k = kb + r;
    tile[r][tx] = k < kk ? static_cast<float>(x[static_cast<long long>(k) * ld + i0 + tx]) : 0.f;
  }
  __syncthreads();
  const int ii = threadIdx.x >> 2, c = threadIdx.x & 3, i = i0 + ii;
  const int kt = 2 * blockIdx.y + (c >> 1);
  if (kt >= kt_total) return;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = tile[8 * c + j][ii];
  const int prow = blockIdx.z * rows_per_item + i;
  const int cs = (c & 1) ^ ((prow >> 3) & 1);
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool finite = (__float_as_uint(v[j]) & 0x7F800000u) != 0x7F800000u;
      const unsigned bits = bf16_rne_bits(v[j]);
      w[j >> 1] |= bits << (16 * (j & 1));
      v[j] = finite ? v[j] - __uint_as_float(bits << 16) : 0.f;   // exact
    }
    unsigned char* dst = planes + ((static_cast<long long>(kt) * 3 + p) * plane_rows + prow) * kRowB + cs * 16;
    *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// A-type operand: element (r, k) = src[r * ld + k] -> planes[kt][p][row0 + r]; thread -> (row, chunk of 8 k)
template <typename T>
__global__ __launch_bounds__(256) void split_rows_kernel(const T* __restrict__ src, long long ld, long long item_stride,
                                                        int rows, int kk, int plane_rows, unsigned char* __restrict__ planes) {
  const int chunks = kk / 8;
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  const int row = static_cast<int>(idx / chunks), ch = static_cast<int>(idx % chunks);
  if (row >= rows) return;
  const T* e = src + static_cast<long long>(blockIdx.z) * item_stride + static_cast<long long>(row) * ld + ch * 8;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = static_cast<float>(e[j]);
  const int prow = blockIdx.z * rows + row;
  const int kt = ch >> 1, cs = (ch & 1) ^ ((prow >> 3) & 1);
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool finite = (__float_as_uint(v[j]) & 0x7F800000u) != 0x7F800000u;
      const unsigned bits = bf16_rne_bits(v[j]);
      w[j >> 1] |= bits << (16 * (j & 1));
      v[j] = finite ? v[j] - __uint_as_float(bits << 16) : 0.f;
    }
    unsigned char* dst = planes + ((static_cast<long long>(kt) * 3 + p) * plane_rows + prow) * kRowB + cs * 16;
    *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

struct Gemm3Args {
  const unsigned char* aplanes;   // [k tile][3][a_rows][32 B]: item z owns rows [z m, (z + 1) m)
  const unsigned char* bplanes;   // [k tile][3][b_rows][32 B]: item z owns rows [z n, (z + 1) n)
  int a_rows, b_rows;
  int m_tiles, n_tiles;           // output tiles of one item
  int kt_total;
  int k_mode;                     // 0: all k; 1: A lower triangular (k < 128 (ti + 1)); 3: B lower triangular (k >= 128 tj)
  void* c;                        // item z at c + z c_item (elements), row stride ldc
  long long ldc, c_item;
  float alpha;
  int out_f64;
};

// C = alpha A B, one 128 x 128 tile per workgroup; the product kernel above with two plane sets, a
// k range per tile and a plain store. grid (tiles of an item, items).
__global__ __launch_bounds__(256) void gemm3_bf16x3_kernel(Gemm3Args a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  int ti, tj;
  {
    const int b = blockIdx.x;
    if (a.k_mode == 3) { tj = b / a.m_tiles; ti = b % a.m_tiles; }                       // long columns first
    else if (a.k_mode == 1) { ti = a.m_tiles - 1 - b / a.n_tiles; tj = b % a.n_tiles; }  // long rows first
    else { ti = b / a.n_tiles; tj = b % a.n_tiles; }
  }
  const int z = blockIdx.y;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int kt0 = a.k_mode == 3 ? tj * (kTile / kBK) : 0;
  const int kt1 = a.k_mode == 1 ? min(a.kt_total, (ti + 1) * (kTile / kBK)) : a.kt_total;
  f32x16 acc[2][2], lo[2][2], top[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = lo[i][j][r] = top[i][j][r] = 0.f;
  const int frow = lane & 31;
  const int fch = ((lane >> 5) ^ ((frow >> 3) & 1)) * 16;
  const int offA = (wr * 64 + frow) * kRowB + fch, offB = kOperandB + (wc * 64 + frow) * kRowB + fch;
  const long long strideA = static_cast<long long>(a.a_rows) * kRowB, strideB = static_cast<long long>(a.b_rows) * kRowB;
  const unsigned char* gA = a.aplanes + (static_cast<long long>(z) * a.m_tiles + ti) * kPlaneTileB + lane * 16;
  const unsigned char* gB = a.bplanes + (static_cast<long long>(z) * a.n_tiles + tj) * kPlaneTileB + lane * 16;
  auto stage = [&](int kt, int buf) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int piece = q * 4 + wave;
      const int op = piece / 12, r = piece % 12, p = r >> 2, seg = r & 3;
      const unsigned char* src = op ? gB + (static_cast<long long>(kt) * 3 + p) * strideB + seg * 1024
                                    : gA + (static_cast<long long>(kt) * 3 + p) * strideA + seg * 1024;
      unsigned char* dst = lds + buf * (2 * kOperandB) + op * kOperandB + p * kPlaneTileB + seg * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  if (kt0 < kt1) stage(kt0, 0);
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < kt1) stage(kt + 1, buf ^ 1);
    const unsigned char* img = lds + buf * (2 * kOperandB);
    bf16x8 fa[2][3], fb[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fa[i][p] = *reinterpret_cast<const bf16x8*>(img + offA + p * kPlaneTileB + i * 32 * kRowB);
        fb[i][p] = *reinterpret_cast<const bf16x8*>(img + offB + p * kPlaneTileB + i * 32 * kRowB);
      }
#define MI355Q_TERM(ACC, PA, PB)                                                                \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)    \
      ACC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA], fb[j][PB], ACC[i][j], 0, 0, 0)
    MI355Q_TERM(lo, 0, 2);
    MI355Q_TERM(lo, 1, 1);
    MI355Q_TERM(lo, 2, 0);
    MI355Q_TERM(lo, 0, 1);
    MI355Q_TERM(lo, 1, 0);
    MI355Q_TERM(acc, 0, 0);
#undef MI355Q_TERM
    if (((kt - kt0) & (kFold - 1)) == kFold - 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            top[i][j][r] = top[i][j][r] + acc[i][j][r];
            acc[i][j][r] = 0.f;
          }
    }
  }
  const long long row0 = static_cast<long long>(ti) * kTile + wr * 64 + 4 * (lane >> 5);
  const long long col0 = static_cast<long long>(tj) * kTile + wc * 64 + (lane & 31);
  const long long base = static_cast<long long>(z) * a.c_item + row0 * a.ldc + col0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = a.alpha * ((top[i][j][r] + acc[i][j][r]) + lo[i][j][r]);
        const long long at = base + static_cast<long long>(i * 32 + (r & 3) + 8 * (r >> 2)) * a.ldc + j * 32;
        if (a.out_f64) static_cast<double*>(a.c)[at] = static_cast<double>(v);
        else static_cast<float*>(a.c)[at] = v;
      }
}

// c (+)= partial[0] + partial[1] + ... (slices added in order) over the lower-triangular tiles
__global__ __launch_bounds__(256) void xtx_reduce_kernel(const float* __restrict__ partial, int splits, int d,
                                                        int accumulate, float* __restrict__ c) {
  const long long n = static_cast<long long>(d) * d;
  const long long stride = static_cast<long long>(gridDim.x) * 256;
  for (long long e = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; e < n; e += stride) {
    const long long i = e / d, j = e % d;
    if (j / kTile > i / kTile) continue;
    float s = partial[e];
    for (int z = 1; z < splits; ++z) s = s + partial[z * n + e];
    c[e] = accumulate ? c[e] + s : s;
  }
}

int xtx_splits(int64_t d, int64_t kt) {
  const int64_t tiles = (d / kTile) * (d / kTile + 1) / 2;
#ifndef MI355Q_XTX_SPLIT_BELOW
#define MI355Q_XTX_SPLIT_BELOW 512
#endif
  if (tiles >= MI355Q_XTX_SPLIT_BELOW) return 1;
  int64_t s = (768 + tiles - 1) / tiles;
  if (s > 16) s = 16;
  if (s > kt / 64) s = kt / 64;        // >= 1024 tokens per split
  return s < 1 ? 1 : static_cast<int>(s);
}

}  // namespace

bool xtx_bf16x3_usable(int64_t n, int64_t d) {
  return d % kTile == 0 && d >= 256 && n >= 1024 && getenv("MI355Q_XTX_FP32_MFMA") == nullptr;
}

size_t xtx_bf16x3_workspace_bytes(int64_t n, int64_t d) {
  const int64_t ks = n < kSlabTokens ? n : kSlabTokens;
  const int64_t kt = (ks + kBK - 1) / kBK;
  const int splits = xtx_splits(d, kt);
  return 1024 + static_cast<size_t>(kt) * 3 * d * kRowB +
         (splits > 1 ? static_cast<size_t>(splits) * d * d * sizeof(float) : 0);
}

// p (float32 [d, d], lower-triangular 128 x 128 tiles valid) = X^T X for X float32 [n, d]
// (accumulate_first: p += X^T X, the product of earlier calls).
int32_t xtx_bf16x3(const float* x, int64_t n, int64_t d, float* p, void* workspace, hipStream_t st, bool accumulate_first) {
  unsigned char* planes = reinterpret_cast<unsigned char*>(
      (reinterpret_cast<uintptr_t>(workspace) + 1023) & ~static_cast<uintptr_t>(1023));
  const int tiles = static_cast<int>(d / kTile);
  const int64_t ks_max = n < kSlabTokens ? n : kSlabTokens;
  const int64_t kt_max = (ks_max + kBK - 1) / kBK;
  float* partial = reinterpret_cast<float*>(planes + static_cast<size_t>(kt_max) * 3 * d * kRowB);
  for (int64_t k0 = 0; k0 < n; k0 += kSlabTokens) {
    const int64_t ks = n - k0 < kSlabTokens ? n - k0 : kSlabTokens;
    const int kt = static_cast<int>((ks + kBK - 1) / kBK);
    hipLaunchKernelGGL(xtx_split_kernel, dim3(static_cast<unsigned>(d / 64), static_cast<unsigned>((kt + 1) / 2)), dim3(256), 0, st,
                       x, static_cast<int>(d), static_cast<long long>(k0), static_cast<long long>(k0 + ks), kt, planes);
    const int splits = xtx_splits(d, kt);
    XtxArgs a{};
    a.planes = planes; a.d = static_cast<int>(d); a.tiles = tiles; a.kt_total = kt;
    a.kt_per_split = (kt + splits - 1) / splits;
    a.partial = splits > 1 ? 1 : 0;
    a.c = splits > 1 ? partial : p;
    a.accumulate = (splits == 1 && (k0 > 0 || accumulate_first)) ? 1 : 0;
    a.patches = tiles >= 4 * kSuper ? 1 : 0;
    unsigned gx;
    if (a.patches) {
      const int sside = (tiles + kSuper - 1) / kSuper, nsup = sside * (sside + 1) / 2;
      gx = static_cast<unsigned>((nsup + 7) / 8 * 8 * kSuper * kSuper);
    } else {
      gx = static_cast<unsigned>(tiles * (tiles + 1) / 2);
    }
    static const bool wide_ok = [] { const char* e = getenv("MI355Q_XTX_NARROW"); return e == nullptr || *e == 0; }();
    if (wide_ok && a.patches && d % (2 * kTile) == 0 && splits == 1) {
      // 128 x 256 tiles: the same 8-XCD patch list, 32 workgroups per patch
      static const bool deep = [] { const char* e = getenv("MI355Q_XTX_DEEP"); return e == nullptr || atoi(e) != 0; }();
      static const int alt = [] { const char* e = getenv("MI355Q_XTX_ALT"); return e == nullptr ? 1 : atoi(e); }();
      const void* fn = !deep ? reinterpret_cast<const void*>(xtx_bf16x3_wide_kernel)
                       : alt == 2 ? reinterpret_cast<const void*>(xtx_bf16x3_deep_kernel<2>)
                       : alt == 1 ? reinterpret_cast<const void*>(xtx_bf16x3_deep_kernel<1>) : reinterpret_cast<const void*>(xtx_bf16x3_deep_kernel<0>);
      if (hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kWideStageB))
        return fail(MI355Q_HIP_ERROR, "xtx bf16x3 LDS attribute: %s", hipGetErrorString(e));
      if (deep && alt == 2) hipLaunchKernelGGL(xtx_bf16x3_deep_kernel<2>, dim3(gx / 2, 1), dim3(512), 2 * kWideStageB, st, a);
      else if (deep && alt == 1) hipLaunchKernelGGL(xtx_bf16x3_deep_kernel<1>, dim3(gx / 2, 1), dim3(512), 2 * kWideStageB, st, a);
      else if (deep) hipLaunchKernelGGL(xtx_bf16x3_deep_kernel<0>, dim3(gx / 2, 1), dim3(512), 2 * kWideStageB, st, a);
      else hipLaunchKernelGGL(xtx_bf16x3_wide_kernel, dim3(gx / 2, 1), dim3(512), 2 * kWideStageB, st, a);
    } else {
      hipLaunchKernelGGL(xtx_bf16x3_kernel, dim3(gx, static_cast<unsigned>(splits)), dim3(256), 4 * kOperandB, st, a);
    }
    if (splits > 1)
      hipLaunchKernelGGL(xtx_reduce_kernel, dim3(2048), dim3(256), 0, st, partial, splits, static_cast<int>(d),
                         (k0 > 0 || accumulate_first) ? 1 : 0, p);
  }
  MI355Q_CHECK_LAUNCH("xtx bf16x3 launch");
  return MI355Q_OK;
}

// ---- Hessian inverse on the split (gptq.hip) ----
// One merge level of the triangular inverse: for `items` pairs of adjacent inverted s x s blocks
// (pair z at a + z hop, hop = 2 s (d + 1)):  L21 <- -L22^-1 (L21 L11^-1).  scratch: 4 d s bytes... see hinv_split_scratch_bytes.
size_t hinv_split_scratch_bytes(int64_t d) { return static_cast<size_t>(d) * d * 6; }

int32_t trtri_level_bf16x3(double* a, int64_t d, int64_t s, int64_t items, void* scratch, hipStream_t st) {
  const long long hop = 2 * s * (d + 1);
  const int kt = static_cast<int>(s / kBK);
  const int rows = static_cast<int>(items * s);            // plane rows of every operand set
  unsigned char* base = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(scratch) + 1023) & ~static_cast<uintptr_t>(1023));
  float* tmat = reinterpret_cast<float*>(base);                               // items x [s, s] float32
  unsigned char* pa = base + static_cast<size_t>(items) * s * s * sizeof(float);
  pa = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(pa) + 1023) & ~static_cast<uintptr_t>(1023));
  unsigned char* pb = pa + static_cast<size_t>(kt) * 3 * rows * kRowB;
  const unsigned tiles = static_cast<unsigned>(s / kTile);
  // T = L21 L11^-1 (B lower triangular: k >= column)
  hipLaunchKernelGGL((split_rows_kernel<double>), dim3(static_cast<unsigned>((s * (s / 8) + 255) / 256), 1, static_cast<unsigned>(items)),
                     dim3(256), 0, st, a + s * d, static_cast<long long>(d), hop, static_cast<int>(s), static_cast<int>(s), rows, pa);
  hipLaunchKernelGGL((split_cols_kernel<double>), dim3(static_cast<unsigned>(s / 64), static_cast<unsigned>((kt + 1) / 2), static_cast<unsigned>(items)),
                     dim3(256), 0, st, a, static_cast<long long>(d), hop, static_cast<int>(s), kt, static_cast<int>(s), rows, 1, pb);
  Gemm3Args g1{pa, pb, rows, rows, static_cast<int>(tiles), static_cast<int>(tiles), kt, 3, tmat, s, s * s, 1.0f, 0};
  hipLaunchKernelGGL(gemm3_bf16x3_kernel, dim3(tiles * tiles, static_cast<unsigned>(items)), dim3(256), 4 * kOperandB, st, g1);
  // L21 = -L22^-1 T (A lower triangular: k <= row)
  hipLaunchKernelGGL((split_rows_kernel<double>), dim3(static_cast<unsigned>((s * (s / 8) + 255) / 256), 1, static_cast<unsigned>(items)),
                     dim3(256), 0, st, a + s * d + s, static_cast<long long>(d), hop, static_cast<int>(s), static_cast<int>(s), rows, pa);
  hipLaunchKernelGGL((split_cols_kernel<float>), dim3(static_cast<unsigned>(s / 64), static_cast<unsigned>((kt + 1) / 2), static_cast<unsigned>(items)),
                     dim3(256), 0, st, tmat, static_cast<long long>(s), static_cast<long long>(s * s), static_cast<int>(s), kt,
                     static_cast<int>(s), rows, 0, pb);
  Gemm3Args g2{pa, pb, rows, rows, static_cast<int>(tiles), static_cast<int>(tiles), kt, 1, a + s * d, d, hop, -1.0f, 1};
  hipLaunchKernelGGL(gemm3_bf16x3_kernel, dim3(tiles * tiles, static_cast<unsigned>(items)), dim3(256), 4 * kOperandB, st, g2);
  MI355Q_CHECK_LAUNCH("trtri level (bf16 split) launch");
  return MI355Q_OK;
}

// hinv (float32 [d, d], lower-triangular tiles) = Linv^T Linv for the lower-triangular FP64 Linv [d, d]
// (d a multiple of 128, <= 16384: one slab).
int32_t ltl_bf16x3(const double* linv, int64_t d, float* hinv, void* scratch, hipStream_t st) {
  unsigned char* planes = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(scratch) + 1023) & ~static_cast<uintptr_t>(1023));
  const int kt = static_cast<int>(d / kBK);
  hipLaunchKernelGGL((split_cols_kernel<double>), dim3(static_cast<unsigned>(d / 64), static_cast<unsigned>((kt + 1) / 2), 1), dim3(256), 0, st,
                     linv, static_cast<long long>(d), 0LL, static_cast<int>(d), kt, static_cast<int>(d), static_cast<int>(d), 1, planes);
  const int tiles = static_cast<int>(d / kTile);
  XtxArgs a{};
  a.planes = planes; a.d = static_cast<int>(d); a.tiles = tiles; a.kt_total = kt; a.kt_per_split = kt;
  a.c = hinv; a.accumulate = 0; a.partial = 0; a.tri = 1;
  a.patches = tiles >= 4 * kSuper ? 1 : 0;
  unsigned gx;
  if (a.patches) {
    const int sside = (tiles + kSuper - 1) / kSuper, nsup = sside * (sside + 1) / 2;
    gx = static_cast<unsigned>((nsup + 7) / 8 * 8 * kSuper * kSuper);
  } else {
    gx = static_cast<unsigned>(tiles * (tiles + 1) / 2);
  }
  hipLaunchKernelGGL(xtx_bf16x3_kernel, dim3(gx, 1), dim3(256), 4 * kOperandB, st, a);
  MI355Q_CHECK_LAUNCH("hinv product (bf16 split) launch");
  return MI355Q_OK;
}

// ---- GPTQ update behind a group (see upd_bf16x3_kernel) ----
bool upd_bf16x3_usable(int64_t rows, int64_t d) {
  return rows % kTile == 0 && d % kTile == 0 && (d >= 4096 || (d >= 1024 && rows >= 8192)) && getenv("MI355Q_UPD_FP32_MFMA") == nullptr;
}

size_t upd_bf16x3_workspace_bytes(int64_t rows, int64_t d, int64_t kk_max) {
  return 2048 + static_cast<size_t>(d / kBK) * 3 * d * kRowB + static_cast<size_t>(kk_max / kBK) * 3 * rows * kRowB;
}

static unsigned char* upd_hplanes(void* workspace) {
  return reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(workspace) + 1023) & ~static_cast<uintptr_t>(1023));
}

// once per call: the planes of Hinv [d, d]
int32_t upd_bf16x3_prepare(const float* hinv, int64_t d, void* workspace, hipStream_t st) {
  const int kt = static_cast<int>(d / kBK);
  hipLaunchKernelGGL(xtx_split_kernel, dim3(static_cast<unsigned>(d / 64), static_cast<unsigned>((kt + 1) / 2)), dim3(256), 0, st,
                     hinv, static_cast<int>(d), 0LL, static_cast<long long>(d), kt, upd_hplanes(workspace));
  MI355Q_CHECK_LAUNCH("gptq update planes launch");
  return MI355Q_OK;
}

// w[:, g1:] -= err[:, 0:kk] @ hinv[g0:g0+kk, g1:]   (kk a multiple of 16, g0 of 16, g1 and d - g1 of 128)
int32_t upd_bf16x3(const float* err, int64_t ld, int64_t rows, int64_t d, int64_t g0, int64_t kk, int64_t g1, float* w,
                   void* workspace, hipStream_t st) {
  unsigned char* hplanes = upd_hplanes(workspace);
  unsigned char* eplanes = hplanes + static_cast<size_t>(d / kBK) * 3 * d * kRowB;
  const int chunks = static_cast<int>(kk / 8);
  hipLaunchKernelGGL(upd_split_err_kernel, dim3(static_cast<unsigned>((rows * chunks + 255) / 256)), dim3(256), 0, st, err,
                     static_cast<int>(rows), static_cast<int>(ld), static_cast<int>(kk), eplanes);
  UpdArgs a{eplanes, hplanes, static_cast<int>(rows), static_cast<int>(d), static_cast<int>(kk / kBK),
            static_cast<int>(g0 / kBK), static_cast<int>(g1), w};
  const int ntj = static_cast<int>((d - g1) / kTile);
  static const bool two_tiles = getenv("MI355Q_UPD_ONE_TILE") == nullptr;
  if (two_tiles && ntj >= 8) {
    // 72 KB of dynamic LDS has to be asked for; the attribute belongs to the current device and a
    // process may drive several, so it is set on every call (a host-side table write)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(upd2_bf16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            6 * kOperandB) != hipSuccess)
      return fail(MI355Q_HIP_ERROR, "hipFuncSetAttribute failed");
    hipLaunchKernelGGL(upd2_bf16x3_kernel, dim3(static_cast<unsigned>((ntj + 1) / 2), static_cast<unsigned>(rows / kTile)),
                       dim3(256), 6 * kOperandB, st, a, ntj);
  } else {
    hipLaunchKernelGGL(upd_bf16x3_kernel, dim3(static_cast<unsigned>(ntj), static_cast<unsigned>(rows / kTile)),
                       dim3(256), 4 * kOperandB, st, a);
  }
  MI355Q_CHECK_LAUNCH("gptq update launch");
  return MI355Q_OK;
}

}  // namespace mi355q
