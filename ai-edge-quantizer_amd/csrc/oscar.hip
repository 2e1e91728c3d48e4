// OSCAR (activation-aware channel scaling + optimal clipping) for FULLY_CONNECTED weights.
// ref: algorithms/uniform_quantize/oscar.py (all arithmetic FP64 there; reproduced bit for bit).
//
// The O(in_ch) vector algebra (geometric-mean normalisation, log/exp, clamps) stays on the host in
// NumPy exactly as the reference does it; everything that touches the [out_ch, in_ch] matrix is
// here:
//   col_sumsq      per-column sum of squares, rows added in order (NumPy's axis-0 order)
//   group_terms    per (row, group) max of |w|*s with first-index argmax, then per group the
//                  NumPy-order (8192-chunk, pairwise) sum over rows of the squared maxima
//   winner_energy  the np.add.at accumulation of the winners' squares, per column, rows in order
//   clip_bounds    per (row, group): stable descending sort of |w|*s carrying the masses (LDS
//                  bitonic tiles, 8192-element runs + merge passes beyond that), three
//                  sequential running sums, the closed-form candidate of every segment, first
//                  minimum
//   quantize       clip(rint((w*s) / scale)) with FP64 product and quotient
// Compiled with -ffp-contract=off: none of the FP64 expressions may be fused.
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace mi355q {
namespace {

// ------------------------------------------------------------------ col_sumsq ---
// One lane per column, rows strictly in order. UNROLL row loads are issued before the adds so
// that a wave keeps UNROLL * 256 B in flight.
template <int UNROLL>
__global__ __launch_bounds__(64) void col_sumsq_kernel(const float* __restrict__ x, int64_t rows,
                                                       int64_t d, int32_t mean,
                                                       double* __restrict__ out) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * 64 + threadIdx.x;
  if (j >= d) return;
  const float* p = x + j;
  double acc = 0.0;
  int64_t r = 0;
  for (; r + UNROLL <= rows; r += UNROLL) {
    float v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(p + (r + u) * d);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const double t = static_cast<double>(v[u]);
      acc = acc + t * t;
    }
  }
  for (; r < rows; ++r) {
    const double t = static_cast<double>(p[r * d]);
    acc = acc + t * t;
  }
  out[j] = mean ? acc / static_cast<double>(rows) : acc;
}

// ---------------------------------------------------------------- group_terms ---
struct Top {
  double v;     // |w| * s
  int32_t j;    // column
  float w;      // the weight itself (its square feeds winner_energy)
};

__device__ __forceinline__ bool beats(const Top& a, const Top& b) {   // np.argmax: first maximum
  return a.v > b.v || (a.v == b.v && a.j < b.j);
}

__device__ __forceinline__ Top shfl_xor_top(const Top& t, int off) {
  Top o;
  o.v = __shfl_xor(t.v, off, kWave);
  o.j = __shfl_xor(t.j, off, kWave);
  o.w = __shfl_xor(t.w, off, kWave);
  return o;
}

// Blockwise groups (g in {32, 64, 128, 256}, d % g == 0): a lane owns 4 consecutive columns,
// g/4 lanes form a group. top2 / winner / wsq are laid out [G][n]: the per-group sum and the
// winners' accumulation both walk one group's rows as a contiguous vector.
template <int LANES>
__global__ __launch_bounds__(256) void group_top_block_kernel(
    const float* __restrict__ w, const double* __restrict__ s, int64_t n, int64_t d, int32_t g,
    double* __restrict__ top2, int32_t* __restrict__ winner, double* __restrict__ wsq) {
  const int64_t quad = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t quads = n * d / 4;
  const bool live = quad < quads;
  const int64_t e = (live ? quad : quads - 1) * 4;
  const int64_t r = e / d;
  const int32_t j0 = static_cast<int32_t>(e - r * d);
  const float4 v = *reinterpret_cast<const float4*>(w + e);
  const double2 s01 = *reinterpret_cast<const double2*>(s + j0);
  const double2 s23 = *reinterpret_cast<const double2*>(s + j0 + 2);
  const float wv[4] = {v.x, v.y, v.z, v.w};
  const double sv[4] = {s01.x, s01.y, s23.x, s23.y};
  Top best{fabs(static_cast<double>(wv[0])) * sv[0], j0, wv[0]};
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    const Top c{fabs(static_cast<double>(wv[k])) * sv[k], j0 + k, wv[k]};
    if (beats(c, best)) best = c;
  }
#pragma unroll
  for (int off = 1; off < LANES; off <<= 1) {
    const Top o = shfl_xor_top(best, off);
    if (beats(o, best)) best = o;
  }
  if (live && (threadIdx.x & (LANES - 1)) == 0) {
    const int64_t k = j0 / g;
    top2[k * n + r] = best.v * best.v;
    winner[k * n + r] = best.j;
    const double t = static_cast<double>(best.w);
    wsq[k * n + r] = t * t;
  }
}

// One group per row (tensor-/channel-wise, any d): a 256-thread block per row.
__global__ __launch_bounds__(256) void group_top_row_kernel(
    const float* __restrict__ w, const double* __restrict__ s, int64_t n, int64_t d,
    double* __restrict__ top2, int32_t* __restrict__ winner, double* __restrict__ wsq) {
  const int64_t r = blockIdx.x;
  const float* row = w + r * d;
  Top best{-1.0, 0x7FFFFFFF, 0.0f};
  for (int64_t j = threadIdx.x; j < d; j += 256) {
    const float x = row[j];
    const Top c{fabs(static_cast<double>(x)) * s[j], static_cast<int32_t>(j), x};
    if (beats(c, best)) best = c;
  }
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const Top o = shfl_xor_top(best, off);
    if (beats(o, best)) best = o;
  }
  __shared__ double sv[4];
  __shared__ int32_t sj[4];
  __shared__ float sw[4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sv[wave] = best.v;
    sj[wave] = best.j;
    sw[wave] = best.w;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) {
      const Top o{sv[k], sj[k], sw[k]};
      if (beats(o, best)) best = o;
    }
    top2[r] = best.v * best.v;
    winner[r] = best.j;
    const double t = static_cast<double>(best.w);
    wsq[r] = t * t;
  }
}

// NumPy's pairwise_sum (blocks of <= 128 with 8 accumulators, halves rounded to multiples of 8).
__device__ double pairwise_f64(const double* a, int64_t n) {
  if (n < 8) {
    double res = 0.0;
    for (int64_t i = 0; i < n; ++i) res = res + a[i];
    return res;
  }
  if (n <= 128) {
    double r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = a[k];
    int64_t i = 8;
    for (; i + 8 <= n; i += 8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) r[k] = r[k] + a[i + k];
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res = res + a[i];
    return res;
  }
  int64_t n2 = n / 2;
  n2 -= n2 % 8;
  return pairwise_f64(a, n2) + pairwise_f64(a + n2, n - n2);
}

// Leaves of pairwise_f64's recursion over [lo, lo + n), in order.
__device__ void pairwise_leaves(int32_t lo, int32_t n, int32_t* leaf_lo, int32_t* leaf_n,
                                int32_t* count) {
  if (n <= 128) {
    leaf_lo[*count] = lo;
    leaf_n[*count] = n;
    ++*count;
    return;
  }
  int32_t n2 = n / 2;
  n2 -= n2 % 8;
  pairwise_leaves(lo, n2, leaf_lo, leaf_n, count);
  pairwise_leaves(lo + n2, n - n2, leaf_lo, leaf_n, count);
}

// The recursion again, with the leaves' sums already known (consumed in order).
__device__ double pairwise_combine(int32_t n, const double* leaf_sum, int32_t* next) {
  if (n <= 128) return leaf_sum[(*next)++];
  int32_t n2 = n / 2;
  n2 -= n2 % 8;
  const double left = pairwise_combine(n2, leaf_sum, next);
  return left + pairwise_combine(n - n2, leaf_sum, next);
}

// np.sum of a contiguous FP64 vector: 8192-element chunks added in order, each pairwise. One
// wave per vector: lane 0 lists the <= 128 leaves of a chunk; 8 lanes share a leaf, one per
// accumulator of NumPy's unrolled loop (so a wave reads 8 x 64 B runs per step), the
// accumulators are folded pairwise by an xor butterfly, which is the ((r0+r1)+(r2+r3))+... tree;
// lane 0 folds the leaf sums in the recursion's order.
__global__ __launch_bounds__(64) void group_sum_kernel(const double* __restrict__ top2, int64_t n,
                                                       double* __restrict__ sums) {
  __shared__ int32_t leaf_lo[128], leaf_n[128], leaves;
  __shared__ double leaf_sum[128];
  const double* a = top2 + static_cast<int64_t>(blockIdx.x) * n;
  const int lane = threadIdx.x, k = lane & 7, slot = lane >> 3;
  double total = 0.0;
  for (int64_t lo = 0; lo < n; lo += 8192) {
    const int32_t len = static_cast<int32_t>((n - lo < 8192) ? n - lo : 8192);
    if (lane == 0) {
      int32_t c = 0;
      pairwise_leaves(0, len, leaf_lo, leaf_n, &c);
      leaves = c;
    }
    __syncthreads();
    for (int32_t base = 0; base < leaves; base += 8) {
      const int32_t leaf = base + slot;
      const bool live = leaf < leaves;
      const double* p = a + lo + (live ? leaf_lo[leaf] : 0);
      const int32_t ln = live ? leaf_n[leaf] : 0;
      double res = 0.0;
      if (ln < 8) {                                  // short leaf: plain left-to-right sum
        for (int32_t i = 0; i < ln; ++i) res = res + p[i];
      } else {
        double v[16];                                // all loads first, then the ordered adds
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = (8 * (u + 1) <= ln) ? p[8 * u + k] : 0.0;
        double r = v[0];
#pragma unroll
        for (int u = 1; u < 16; ++u)
          if (8 * (u + 1) <= ln) r = r + v[u];
        int32_t i = ln & ~7;
        r = r + __shfl_xor(r, 1, kWave);             // IEEE addition commutes: both partners agree
        r = r + __shfl_xor(r, 2, kWave);
        r = r + __shfl_xor(r, 4, kWave);
        for (; i < ln; ++i) r = r + p[i];
        res = r;
      }
      if (live && k == 0) leaf_sum[leaf] = res;
    }
    __syncthreads();
    if (lane == 0) {
      int32_t next = 0;
      const double part = pairwise_combine(len, leaf_sum, &next);
      total = (lo == 0) ? part : total + part;
    }
    __syncthreads();
  }
  if (lane == 0) sums[blockIdx.x] = total;
}

// The same sum when every chunk's tree is complete (balanced_chunk, common.h; n = 2048, 4096,
// 16384, 11008, ... rows): leaf index = 8 * step + (lane >> 3), so the accumulators and the
// three lowest tree levels are lane butterflies and the levels above fold whole steps like a
// binary counter (unrolled: straight-line code). No leaf tables, no LDS, no serial fold -- with
// one block per group (a single block for channelwise scales) that serial part was the kernel.
__global__ __launch_bounds__(64) void group_sum_balanced_kernel(const double* __restrict__ top2, int64_t n,
                                                                int last_leaf, int last_depth,
                                                                double* __restrict__ sums) {
  const double* a = top2 + static_cast<int64_t>(blockIdx.x) * n;
  const int lane = threadIdx.x, k = lane & 7, slot = lane >> 3;
  double total = 0.0;
  for (int64_t lo = 0; lo < n; lo += 8192) {
    const bool last = lo + 8192 >= n;
    const int leaf_len = last ? last_leaf : 128, depth = last ? last_depth : 6;
    const int count = 1 << depth;
    const int steps = count > 8 ? count >> 3 : 1;
    double lvl0 = 0.0, lvl1 = 0.0, lvl2 = 0.0, x = 0.0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < steps) {
        const int leaf = s * 8 + slot;
        const double* p = a + lo + static_cast<int64_t>(leaf < count ? leaf : 0) * leaf_len + k;
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = (8 * (q + 1) <= leaf_len) ? p[8 * q] : 0.0;
        double r = v[0];
#pragma unroll
        for (int q = 1; q < 16; ++q)
          if (8 * (q + 1) <= leaf_len) r = r + v[q];
        r = r + __shfl_xor(r, 1, kWave);
        r = r + __shfl_xor(r, 2, kWave);
        r = r + __shfl_xor(r, 4, kWave);
        if (count > 1) r = r + __shfl_xor(r, 8, kWave);
        if (count > 2) r = r + __shfl_xor(r, 16, kWave);
        if (count > 4) r = r + __shfl_xor(r, 32, kWave);
        x = r;
        if (s & 1) {
          x = lvl0 + x;
          if (s & 2) {
            x = lvl1 + x;
            if (s & 4) x = lvl2 + x; else lvl2 = x;
          } else {
            lvl1 = x;
          }
        } else {
          lvl0 = x;
        }
      }
    }
    total = (lo == 0) ? x : total + x;
  }
  if (lane == 0) sums[blockIdx.x] = total;
}

// -------------------------------------------------------------- winner_energy ---
// eff[j] = sum over rows, in order, of wsq[group(j)][r] where winner[group(j)][r] == j.
// blockIdx.x = group, blockIdx.y = 256-column slice of it: the block stages 1024 rows of the
// group's winner / wsq vectors in LDS (coalesced), then every column walks them in order; all
// lanes read the same LDS word (broadcast).
__global__ __launch_bounds__(256) void winner_energy_kernel(const int32_t* __restrict__ winner,
                                                            const double* __restrict__ wsq,
                                                            int64_t n, int64_t d, int32_t g,
                                                            double* __restrict__ eff) {
  constexpr int ROWS = 1024;
  __shared__ int32_t win[ROWS];
  __shared__ double sq[ROWS];
  const int64_t k = blockIdx.x;
  const int32_t jg = static_cast<int32_t>(blockIdx.y) * 256 + threadIdx.x;   // column within group
  const int32_t j = static_cast<int32_t>(k * g) + jg;
  const int32_t* wk = winner + k * n;
  const double* sk = wsq + k * n;
  double acc = 0.0;
  for (int64_t base = 0; base < n; base += ROWS) {
    const int32_t here = static_cast<int32_t>((n - base < ROWS) ? n - base : ROWS);
    __syncthreads();
    for (int32_t i = threadIdx.x; i < here; i += 256) {
      win[i] = wk[base + i];
      sq[i] = sk[base + i];
    }
    __syncthreads();
    if (jg < g) {
      // branch-free: adding +0.0 leaves a non-negative running sum unchanged, bit for bit
      int32_t i = 0;
      for (; i + 8 <= here; i += 8) {
        int32_t wi[8];
        double si[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          wi[u] = win[i + u];
          si[u] = sq[i + u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = acc + (wi[u] == j ? si[u] : 0.0);
      }
      for (; i < here; ++i) acc = acc + (win[i] == j ? sq[i] : 0.0);
    }
  }
  if (jg < g) eff[j] = acc;
}

// ---------------------------------------------------------------- clip_bounds ---
// Stable descending sort of every segment of g <= P (P a power of two, 32..8192) elements in
// LDS: bitonic network on (key, position) pairs -- the position breaks ties, so the result is
// the stable order. A 256-thread block sorts a tile of TILE / P consecutive segments; keys are
// computed on the fly (|w| * s), the sorted keys and the masses of their columns are written
// out in the natural [segment][rank] layout, or rank-major ([rank][segment]) for short segments
// so that the scan kernel's lanes (one per segment) read consecutive addresses.
constexpr int kSortThreadsMax = 1024;

// +1 every 8 and every 32 elements: the 8-element register blocks (stride 8) and the rank-major
// store (stride P >= 32) both spread over the banks
__host__ __device__ __forceinline__ int lds_pad(int i) { return i + (i >> 3) + (i >> 5); }

// (key, position) pairs are distinct, so this is a strict total order: exactly one of
// sorts_before(a, b), sorts_before(b, a) holds. Bitwise operators: no short-circuit branches.
__device__ __forceinline__ bool sorts_before(double ka, uint32_t ia, double kb, uint32_t ib) {
  return (ka > kb) | ((ka == kb) & (ia < ib));
}

// LEVELS consecutive compare-exchange levels (distances j, j/2, ...) of the bitonic network on
// 2^LEVELS elements held in registers: one LDS read and one write per element instead of LEVELS.
template <int LEVELS>
__device__ __forceinline__ void sort_levels(double* key, uint16_t* pos, int tile, int P, int k,
                                            int j) {
  constexpr int N = 1 << LEVELS;
  const int low = j >> (LEVELS - 1);                 // smallest distance of this round
  for (int t = threadIdx.x; t < tile / N; t += blockDim.x) {
    // spread t around LEVELS zero bits at the positions of the distances
    const int base = ((t & ~(low - 1)) << LEVELS) | (t & (low - 1));
    const bool desc = ((base & (P - 1)) & k) == 0;
    double kv[N];
    uint32_t pv[N];
#pragma unroll
    for (int c = 0; c < N; ++c) {
      const int at = lds_pad(base + c * low);
      kv[c] = key[at];
      pv[c] = pos[at];
    }
#pragma unroll
    for (int lvl = LEVELS - 1; lvl >= 0; --lvl) {    // distance low << lvl
#pragma unroll
      for (int c = 0; c < N; ++c) {
        if ((c >> lvl) & 1) continue;
        const int o = c | (1 << lvl);
        const bool swap = sorts_before(kv[o], pv[o], kv[c], pv[c]) == desc;
        const double tk = swap ? kv[o] : kv[c];
        kv[o] = swap ? kv[c] : kv[o];
        kv[c] = tk;
        const uint32_t tp = swap ? pv[o] : pv[c];
        pv[o] = swap ? pv[c] : pv[o];
        pv[c] = tp;
      }
    }
#pragma unroll
    for (int c = 0; c < N; ++c) {
      const int at = lds_pad(base + c * low);
      key[at] = kv[c];
      pos[at] = static_cast<uint16_t>(pv[c]);
    }
  }
}

// Segments longer than the tile are sorted as runs: `runs_per_seg` > 1 makes unit u of the grid
// the run u % runs_per_seg of segment u / runs_per_seg, i.e. elements [run * g, run * g + g) of
// that `seg_len`-element segment, cut at its end (the last run of a segment is shorter);
// merge_runs_kernel then merges neighbouring runs. With runs_per_seg == 1 a unit is a whole segment
// of g == seg_len elements.
__global__ __launch_bounds__(kSortThreadsMax) void sort_tile_kernel(
    const float* __restrict__ w, const double* __restrict__ s, const double* __restrict__ m,
    int64_t segments, int32_t g, int32_t P, int32_t tile, int64_t d, int32_t transposed,
    int64_t seg_len, int32_t runs_per_seg, double* __restrict__ keys_out, double* __restrict__ vals_out,
    const uint8_t* __restrict__ todo) {
  extern __shared__ unsigned char lds_raw[];
  double* key = reinterpret_cast<double*>(lds_raw);
  uint16_t* pos = reinterpret_cast<uint16_t*>(key + lds_pad(tile) + 1);
  const int segs_per_tile = tile / P;
  const int64_t seg0 = static_cast<int64_t>(blockIdx.x) * segs_per_tile;
  const int64_t segs_here = (segments - seg0 < segs_per_tile) ? segments - seg0 : segs_per_tile;
  if (todo) {                        // the prefix kernel has answered most rows: a tile none of whose segments is left returns
    bool any = false;
    for (int64_t sl = 0; sl < segs_here; ++sl) any |= todo[(seg0 + sl) / runs_per_seg] != 0;
    if (!any) return;
  }

  for (int e = threadIdx.x; e < tile; e += blockDim.x) {
    const int sl = e / P, i = e - sl * P;
    double k = -1.0;                 // padding sorts behind every real magnitude
    if (sl < segs_here) {
      const int64_t unit = seg0 + sl, run = unit % runs_per_seg;
      const int64_t off = run * g + i;                       // within the segment
      if (i < g && off < seg_len) {
        const int64_t ge = (unit / runs_per_seg) * seg_len + off;
        k = fabs(static_cast<double>(w[ge]));
        if (s) k = k * s[ge % d];
      }
    }
    key[lds_pad(e)] = k;
    pos[lds_pad(e)] = static_cast<uint16_t>(i);
  }
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) {
    int j = k >> 1;
    while (j > 0) {        // up to three consecutive levels per LDS round trip
      if (j >= 4) {
        sort_levels<3>(key, pos, tile, P, k, j);
        j >>= 3;
      } else if (j == 2) {
        sort_levels<2>(key, pos, tile, P, k, j);
        j = 0;
      } else {
        sort_levels<1>(key, pos, tile, P, k, j);
        j = 0;
      }
      __syncthreads();
    }
  }
  if (transposed) {
    // consecutive threads -> consecutive segments of the tile, same rank
    for (int e = threadIdx.x; e < segs_per_tile * g; e += blockDim.x) {
      const int i = e / segs_per_tile, sl = e - i * segs_per_tile;
      if (sl < segs_here) {
        const int at = lds_pad(sl * P + i);
        const int64_t seg = seg0 + sl;
        keys_out[static_cast<int64_t>(i) * segments + seg] = key[at];
        if (m) vals_out[static_cast<int64_t>(i) * segments + seg] = m[(seg * g + pos[at]) % d];
      }
    }
  } else {
    for (int e = threadIdx.x; e < tile; e += blockDim.x) {
      const int sl = e / P, i = e - sl * P;
      if (sl < segs_here && i < g) {
        const int at = lds_pad(e);
        const int64_t unit = seg0 + sl, run = unit % runs_per_seg;
        const int64_t first = (unit / runs_per_seg) * seg_len + run * g;   // the unit's first element
        if (run * g + i < seg_len) {
          keys_out[first + i] = key[at];
          if (m) vals_out[first + i] = m[(first + pos[at]) % d];
        }
      }
    }
  }
}

// Stable merge of neighbouring sorted runs of `run` elements inside every g-element segment
// (descending; on equal keys the left run -- the lower original positions -- goes first): every
// element finds its rank by a binary search in the sibling run. The last run of a segment may be
// shorter, a run without a sibling is copied. log2(runs) passes sort segments of any length.
__global__ __launch_bounds__(256) void merge_runs_kernel(const double* __restrict__ keys,
                                                         const double* __restrict__ vals,
                                                         int64_t total, int64_t g, int64_t run,
                                                         double* __restrict__ keys_out,
                                                         double* __restrict__ vals_out,
                                                         const uint8_t* __restrict__ todo) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= total) return;
  const int64_t seg = e / g, off = e - seg * g;
  if (todo && !todo[seg]) return;
  const int64_t pair0 = off / (2 * run) * (2 * run);       // first element of the pair of runs
  const int64_t mid = pair0 + run < g ? pair0 + run : g;   // end of the left run
  const int64_t end = pair0 + 2 * run < g ? pair0 + 2 * run : g;
  const double k = keys[e];
  const bool left = off < mid;
  const double* other = keys + seg * g + (left ? mid : pair0);
  int64_t lo = 0, hi = left ? end - mid : mid - pair0;     // count of the sibling's elements that go before k
  while (lo < hi) {
    const int64_t m2 = (lo + hi) >> 1;
    const bool before = left ? other[m2] > k : other[m2] >= k;
    if (before) lo = m2 + 1; else hi = m2;
  }
  const int64_t at = seg * g + pair0 + (left ? off - pair0 : off - mid) + lo;
  keys_out[at] = k;
  if (vals) vals_out[at] = vals[e];
}

// One lane per sorted segment: running sums in order, the candidate of every breakpoint
// interval, first minimum. u = M / (6 qmax^2), noise = M / (12 qmax^2) per group (host FP64).
// Element i of segment seg lives at seg * seg_stride + i * elem_stride (natural layout: g, 1;
// transposed layout of the tile sort for short segments: 1, segments -> coalesced lanes).
template <int BATCH>
__global__ __launch_bounds__(64) void clip_scan_kernel(const double* __restrict__ keys,
                                                       const double* __restrict__ vals,
                                                       int64_t segments, int64_t g, int64_t G,
                                                       int64_t seg_stride, int64_t elem_stride,
                                                       const double* __restrict__ u,
                                                       const double* __restrict__ noise,
                                                       double qmax, int32_t blockwise,
                                                       double* __restrict__ bounds,
                                                       double* __restrict__ scale,
                                                       const uint8_t* __restrict__ todo) {
  const int64_t seg = static_cast<int64_t>(blockIdx.x) * 64 + threadIdx.x;
  if (seg >= segments) return;
  if (todo && !todo[seg]) return;
  const double* a = keys + seg * seg_stride;
  const double* m = vals + seg * seg_stride;
  const double uk = u[seg % G], nk = noise[seg % G];
  double run_m = 0.0, run_am = 0.0, run_a2m = 0.0;
  const double a0 = a[0];
  double best_c = a0, best_e = (a0 * a0) * nk;
  double cur = a0;
  for (int64_t i = 0; i < g; i += BATCH) {
    double ab[BATCH + 1], mb[BATCH];
#pragma unroll
    for (int t = 0; t < BATCH; ++t) {
      mb[t] = (i + t < g) ? m[(i + t) * elem_stride] : 0.0;
      ab[t + 1] = (i + t + 1 < g) ? a[(i + t + 1) * elem_stride] : 0.0;
    }
    ab[0] = cur;
#pragma unroll
    for (int t = 0; t < BATCH; ++t) {
      if (i + t < g) {
        const double ai = ab[t], mi = mb[t], lower = ab[t + 1];
        run_m = run_m + mi;
        run_am = run_am + ai * mi;
        run_a2m = run_a2m + (ai * ai) * mi;
        double c = (2.0 * run_am) / (uk + 2.0 * run_m);
        c = fmin(fmax(c, lower), ai);
        const double c2 = c * c;
        const double e = ((c2 * nk + run_a2m) - (2.0 * c) * run_am) + c2 * run_m;
        if (e < best_e) {
          best_e = e;
          best_c = c;
        }
      }
    }
    cur = ab[BATCH];
  }
  if (bounds) bounds[seg] = best_c;
  if (scale) {
    // tensor_zp_scale_from_min_max(-c, c, symmetric): bound = max(c, 1e-9), scale = bound / qmax;
    // blockwise scales go FP64 -> float32 -> bfloat16 -> float16 (ref uniform_quantize_tensor.py
    // :553-581; no clipping values on this path)
    double sc = fmax(best_c, 1e-9) / qmax;
    if (blockwise) {
      uint16_t half_bits;
      sc = static_cast<double>(round_scale_blockwise(static_cast<float>(sc), &half_bits));
    }
    scale[seg] = sc;
  }
}

// The same scan with one WAVE per segment, for long segments (a lane per segment leaves most of
// the chip idle and walks memory with a stride). 64 elements at a time: the lanes put the three
// addends of their element in LDS; lanes 0..2 each own one running sum, read its 64 addends
// into registers, add them in order (the carry never leaves the lane) and write the 64 prefix
// values back; every lane then evaluates the candidate of its element, and the first minimum is
// an (error, index) butterfly.
__global__ __launch_bounds__(256) void clip_scan_wave_kernel(
    const double* __restrict__ keys, const double* __restrict__ vals, int64_t segments, int64_t g,
    int64_t G, const double* __restrict__ u, const double* __restrict__ noise, double qmax,
    int32_t blockwise, double* __restrict__ bounds, double* __restrict__ scale,
    const uint8_t* __restrict__ todo) {
  const int64_t seg = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (seg >= segments) return;                       // whole waves leave together
  if (todo && !todo[seg]) return;
  const int lane = threadIdx.x & 63;
  const double* a = keys + seg * g;
  const double* m = vals + seg * g;
  const double uk = u[seg % G], nk = noise[seg % G];
  const double a0 = a[0];
  double best_c = a0, best_e = (a0 * a0) * nk;       // the "clip nothing" candidate, index -1
  int64_t best_i = -1;
  __shared__ double addend[4][3][64], prefix[4][3][64];
  double (*add_w)[64] = addend[threadIdx.x >> 6];
  double (*pre_w)[64] = prefix[threadIdx.x >> 6];
  const int chain = lane < 3 ? lane : 0;             // lanes >= 3 shadow chain 0, results unused
  double acc = 0.0;                                  // this chain's running sum across chunks
  for (int64_t base = 0; base < g; base += 64) {
    const int64_t i = base + lane;
    const bool live = i < g;
    const double ai = live ? a[i] : 0.0;
    const double mi = live ? m[i] : 0.0;
    const double lower = (i + 1 < g) ? a[i + 1] : 0.0;
    add_w[0][lane] = mi;
    add_w[1][lane] = ai * mi;
    add_w[2][lane] = (ai * ai) * mi;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): the wave's LDS writes landed
    double v[64];
#pragma unroll
    for (int t = 0; t < 64; ++t) v[t] = add_w[chain][t];
#pragma unroll
    for (int t = 0; t < 64; ++t) {
      acc = acc + v[t];
      v[t] = acc;
    }
    if (lane < 3) {
#pragma unroll
      for (int t = 0; t < 64; ++t) pre_w[chain][t] = v[t];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const double run_m = pre_w[0][lane], run_am = pre_w[1][lane], run_a2m = pre_w[2][lane];
    if (live) {
      double c = (2.0 * run_am) / (uk + 2.0 * run_m);
      c = fmin(fmax(c, lower), ai);
      const double c2 = c * c;
      const double e = ((c2 * nk + run_a2m) - (2.0 * c) * run_am) + c2 * run_m;
      if (e < best_e) {
        best_e = e;
        best_c = c;
        best_i = i;
      }
    }
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double oe = __shfl_xor(best_e, off, kWave);
    const double oc = __shfl_xor(best_c, off, kWave);
    const int64_t oi = __shfl_xor(best_i, off, kWave);
    if (oe < best_e || (oe == best_e && oi < best_i)) {
      best_e = oe;
      best_c = oc;
      best_i = oi;
    }
  }
  if (lane == 0) {
    if (bounds) bounds[seg] = best_c;
    if (scale) {
      double sc = fmax(best_c, 1e-9) / qmax;
      if (blockwise) {
        uint16_t half_bits;
        sc = static_cast<double>(round_scale_blockwise(static_cast<float>(sc), &half_bits));
      }
      scale[seg] = sc;
    }
  }
}

// -------------------------------------------------------- the prefix of the scan ---
// CHANNELWISE rows (g == d): the first minimum of the breakpoint scan almost always lies among the few largest
// magnitudes -- with qmax >= 7 the optimal bound clips a percent of a row -- and everything the scan does up to index k
// depends on the k largest elements only. So a row is answered from a PREFIX of its sorted order:
//   1. every thread keeps the upper halves of the float32 patterns of its elements' keys in registers (monotone in
//      the key, 128 steps per binade); a bisection
//      over thresholds one eighth of a binade apart below the row's largest key (seven counts over the row) finds the
//      threshold with at least `target` elements at or above it; those P elements ARE the first P of the stable
//      descending order (equal keys have equal patterns), and the largest key below the threshold is element P + 1
//      (`a_next`, recomputed in FP64);
//   2. the P elements are sorted in LDS as 64-bit composites -- the upper 48 bits of the FP64 key's pattern (monotone)
//      over 0xFFFF - position (equal keys: lower position first, the full sort's tie rule): ONE integer comparison
//      per exchange instead of two FP64 and one integer comparison on a 10-byte pair. Keys that agree in their upper
//      48 bits and differ below are ordered by position, which may be wrong: the scan compares every neighbouring pair
//      with the exact keys and hands the row to the full route if one is out of order (1e-5 of ordinary rows);
//   3. one wave runs the three sequential FP64 sums and the candidates 1..P exactly as clip_scan_wave_kernel does --
//      the same operations on the same operands (keys recomputed as |w| s from the positions) in the same order, hence
//      the same bits -- with a_next as the lower end of the last interval;
//   4. the row is answered only if candidates P+1..g PROVABLY cannot win. E(c) = noise c^2 + sum_j m_j max(a_j - c, 0)^2
//      is convex and candidate k is E at a point of [a_(k+1), a_k] evaluated in floating point. If the stationary
//      point of interval P, 2 S_am / (u + 2 S_m), lies above a_next (with 1e-6 to spare), E decreases towards
//      a_next from below, so every later candidate's exact value is >= E(a_next). A computed candidate is within
//      B = 64 g 2^-53 (a_1^2 noise + 4 S_a2m(g)) of its exact value (g positive addends per sum, a handful of
//      roundings in the closed form, every intermediate <= the bracket; S_a2m(g) <= S_a2m(P) + a_next^2 M), so
//      E(a_next) computed > best + 4 B implies that every later computed candidate exceeds the best of the prefix:
//      np.argmin's first minimum is the prefix's. Otherwise -- or when a key is not finite, or more than `cap`
//      elements share the top eighths -- the row is flagged and the full sort + scan answers it (todo[row] = 1).
// Rows of up to 4096 columns are ONE WAVE's work (WAVES = 1: 16 / 32 / 64 elements per lane): no workgroup barrier
// anywhere, twenty rows in flight per CU. The first form of this kernel gave every row a 256-thread workgroup: 110 000
// cycles per row, half of them in the 45 barriers of the sort with six workgroups per CU taking turns
// (profiles/r06_oscar_prefix.txt). Longer rows (to 16384) take four waves with 32 / 64 elements per thread (sixteen
// waves with 16 each left a CU one row at a time: 433 us for 2048 x 16384).
// Ref oscar.py:62-104. The answer is the reference's, bit for bit, on either route.

// (float32 pattern of a key, monotone in the key for finite keys >= 0; anything else -- NaN, inf, a negative product --
// maps to 0x7F800000 or above and sends the row to the full route)
__device__ __forceinline__ uint32_t key_bits(float w, double s) {
  return f2u(static_cast<float>(fabs(static_cast<double>(w)) * s));
}

__device__ __forceinline__ uint64_t key_composite(double key, uint32_t position) {
  return (static_cast<uint64_t>(__double_as_longlong(key)) & ~0xFFFFull) | (0xFFFFu - position);
}

// LEVELS consecutive compare-exchange levels of the descending bitonic network on 64-bit composites in LDS
// (see sort_levels: the same index arithmetic, one register per element).
template <int LEVELS>
__device__ __forceinline__ void sort_levels_u64(uint64_t* a, int tile, int k, int j) {
  constexpr int N = 1 << LEVELS;
  const int low = j >> (LEVELS - 1);
  for (int t = threadIdx.x; t < tile / N; t += blockDim.x) {
    const int base = ((t & ~(low - 1)) << LEVELS) | (t & (low - 1));
    const bool desc = (base & k) == 0;
    uint64_t v[N];
#pragma unroll
    for (int c = 0; c < N; ++c) v[c] = a[lds_pad(base + c * low)];
#pragma unroll
    for (int lvl = LEVELS - 1; lvl >= 0; --lvl) {
#pragma unroll
      for (int c = 0; c < N; ++c) {
        if ((c >> lvl) & 1) continue;
        const int o = c | (1 << lvl);
        const bool swap = (v[o] > v[c]) == desc;
        const uint64_t t0 = swap ? v[o] : v[c];
        v[o] = swap ? v[c] : v[o];
        v[c] = t0;
      }
    }
#pragma unroll
    for (int c = 0; c < N; ++c) a[lds_pad(base + c * low)] = v[c];
  }
}

#if defined(MI355Q_PREFIX_PROF)
#define MI355Q_PREFIX_STAMP(k) do { if (row == 1000 && lane == 0 && wv == 0) stamp[k] = __builtin_readcyclecounter(); } while (0)
#else
#define MI355Q_PREFIX_STAMP(k) do {} while (0)
#endif

// One row per workgroup of WAVES waves, EPT elements per thread (element tid + 64 WAVES k).
template <int EPT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES == 1 ? 4 : 1) void clip_prefix_kernel(      // (one-wave form: four waves per SIMD, 128 VGPRs)
    const float* __restrict__ w, const double* __restrict__ s, const double* __restrict__ m, int64_t n,
    int32_t g, int32_t target, const double* __restrict__ u, const double* __restrict__ noise, double qmax,
    int32_t blockwise, double* __restrict__ bounds, double* __restrict__ scale, uint8_t* __restrict__ todo) {
  constexpr int THREADS = 64 * WAVES;
  constexpr int CAP = WAVES == 1 ? 512 : 1024;
  __shared__ uint64_t comp[CAP + CAP / 8 + CAP / 32 + 2];
  __shared__ double stage[2][3][64];                 // the scan's addends and running sums; before it, the selected positions
  double (&addend)[3][64] = stage[0];
  double (&prefix)[3][64] = stage[1];
  __shared__ uint32_t cnt_slot[WAVES == 1 ? 1 : 10][WAVES];     // one set of per-wave figures per row-wide reduction
  __shared__ double max_slot[WAVES];
  __shared__ uint32_t filled;
  const int64_t row = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* wr = w + row * g;
#if defined(MI355Q_PREFIX_PROF)
  unsigned long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  if (WAVES > 1 && tid == 0) filled = 0;
  MI355Q_PREFIX_STAMP(0);
  // a reduction over the row: the wave's butterfly, then (WAVES > 1) the waves' figures through LDS -- one barrier
  auto over_row = [&](auto v, auto* slot, auto op) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v = op(v, __shfl_xor(v, off, kWave));
    if constexpr (WAVES > 1) {
      if (lane == 0) slot[wv] = v;
      __syncthreads();
      v = slot[0];
#pragma unroll
      for (int k = 1; k < WAVES; ++k) v = op(v, slot[k]);
    }
    return v;
  };
  auto umax = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
  auto uadd = [](uint32_t a, uint32_t b) { return a + b; };
  auto dmax = [](double a, double b) { return fmax(a, b); };
  // ---- 1. the upper halves of the keys' float32 patterns, two per register (EPT = 64 would not fit otherwise:
  // 128 VGPRs is what four waves per SIMD leave); sixteen (weight, scale) pairs in flight per thread
  uint32_t kp[EPT / 2];
#pragma unroll
  for (int k0 = 0; k0 < EPT; k0 += 16) {
    float wv_[16];
    double sv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      // (unsigned 32-bit element index: base in SGPRs + a 32-bit lane offset; with a signed index every load keeps a
      // 64-bit address of its own -- 2 x 128 VGPRs of addresses for EPT = 64, spilled)
      const uint32_t i = static_cast<uint32_t>(tid) + static_cast<uint32_t>(THREADS * (k0 + k));
      const uint32_t ic = i < static_cast<uint32_t>(g) ? i : 0u;
      wv_[k] = wr[ic];
      sv[k] = s[ic];
    }
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      const uint32_t i = static_cast<uint32_t>(tid) + static_cast<uint32_t>(THREADS * (k0 + k));
      const uint32_t b0 = i < static_cast<uint32_t>(g) ? key_bits(wv_[k], sv[k]) >> 16 : 0u;
      const uint32_t b1 = i + THREADS < static_cast<uint32_t>(g) ? key_bits(wv_[k + 1], sv[k + 1]) & 0xFFFF0000u : 0u;
      kp[(k0 + k) / 2] = b0 | b1;
    }
    asm volatile("" ::: "memory");                   // (the next sixteen pairs are not loaded before these are packed: with all
  }                                                  //  EPT pairs in flight the one-wave form for 4096 columns spills)
  auto pattern = [&](int k) { return (k & 1) ? kp[k / 2] >> 16 : kp[k / 2] & 0xFFFFu; };
  uint32_t top = 0;
#pragma unroll
  for (int k = 0; k < EPT; ++k) top = pattern(k) > top ? pattern(k) : top;
  top = over_row(top, cnt_slot[0], umax);
  MI355Q_PREFIX_STAMP(1);
  if (top >= 0x7F80u || top == 0u) {                 // a key that is not a finite non-negative number; or a row of (almost) zeros
    if (tid == 0) todo[row] = 1;
    return;
  }
  // ---- the threshold: the half patterns step 16 per eighth of a binade; the smallest j in [1, 64] whose threshold
  // top - 16 j has at least `target` elements at or above it (64 = eight binades below the row's largest: take those)
  auto count_at = [&](uint32_t tb, uint32_t* slot) {
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) c += pattern(k) >= tb ? 1u : 0u;
    return over_row(c, slot, uadd);
  };
  auto threshold = [&](int j) {
    const uint32_t step = static_cast<uint32_t>(j) << 4;
    return top > step ? top - step : 1u;             // (never 0: absent elements and zeros carry pattern 0)
  };
  constexpr int kSlots = WAVES == 1 ? 1 : 10;        // (a single wave needs no slots: index 0 throughout)
  int lo = 1, hi = 64;                               // the answer lies in [lo, hi]
  uint32_t P = count_at(threshold(64), cnt_slot[1 % kSlots]);
  if (P >= static_cast<uint32_t>(target)) {
#pragma unroll 1
    for (int r = 0; r < 6; ++r) {
      const int mid = (lo + hi) >> 1;
      const uint32_t c = count_at(threshold(mid), cnt_slot[(2 + r) % kSlots]);
      if (c >= static_cast<uint32_t>(target)) {
        hi = mid;
        P = c;
      } else {
        lo = mid + 1;
      }
      if (lo >= hi) break;
    }
  }
  int j = hi;
  if (P > static_cast<uint32_t>(CAP) && j > 1) {       // a crowded eighth: stop above it if that leaves a prefix worth scanning
    const uint32_t c = count_at(threshold(j - 1), cnt_slot[8 % kSlots]);
    if (c >= 32u) {
      P = c;
      j -= 1;
    }
  }
  if (P > static_cast<uint32_t>(CAP) || P < 1u) {
    if (tid == 0) todo[row] = 1;
    return;
  }
  const uint32_t tb = threshold(j);
  MI355Q_PREFIX_STAMP(2);
  // ---- 2. the selected elements and the largest key among the others. Positions first -- a ballot per element
  // slot, no memory traffic -- then every thread fetches (weight, scale) of the slots lane, lane + THREADS, ... and
  // writes their composites: a handful of independent gathers per thread. (Fetching inside the ballot loop made every
  // one of its EPT iterations wait for its own pair of loads: 120 000 of a row's 200 000 cycles.)
  uint16_t* sel_pos = reinterpret_cast<uint16_t*>(&stage[0][0][0]);   // (CAP positions: 1 / 2 KiB; the scan's staging area is free until then)
  static_assert(sizeof(stage) >= 2 * CAP, "position list");
  uint32_t below = 0;
  uint32_t placed = 0;                               // where this wave's next selected elements go
  if constexpr (WAVES > 1) {
    // the wave's stretch of the list: ONE atomic per wave (an atomic per ballot made each of the EPT iterations wait
    // for its return)
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) mine += static_cast<uint32_t>(__builtin_popcountll(__ballot(pattern(k) >= tb)));
    if (lane == 0) placed = atomicAdd(&filled, mine);
    placed = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(placed)));
  }
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const uint32_t pk = pattern(k);
    const bool sel = pk >= tb;
    const unsigned long long mask = __ballot(sel);
    if (sel)
      sel_pos[placed + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = static_cast<uint16_t>(tid + THREADS * k);
    placed += static_cast<uint32_t>(__builtin_popcountll(mask));
    if (!sel) below = pk > below ? pk : below;
  }
  below = over_row(below, cnt_slot[9 % kSlots], umax);
  // exact FP64 key of element P + 1: the largest key among the elements that share the largest unselected half pattern
  // (a dozen of a 4096-column row). Their positions are listed like the selected ones', then fetched side by side.
  constexpr int kNextCap = 256;
  uint16_t* next_pos = sel_pos + CAP;
  static_assert(sizeof(stage) >= 2 * (CAP + kNextCap), "position lists");
  uint32_t listed = 0;
  if (below != 0u) {
    if constexpr (WAVES > 1) {
      if (tid == 0) filled = 0;                      // (every wave has its stretch of the selected list: the counter is free)
      __syncthreads();
      uint32_t mine = 0;
#pragma unroll
      for (int k = 0; k < EPT; ++k) mine += static_cast<uint32_t>(__builtin_popcountll(__ballot(pattern(k) == below)));
      if (lane == 0) listed = atomicAdd(&filled, mine);
      listed = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(listed)));
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const bool hit = pattern(k) == below;
      const unsigned long long mask = __ballot(hit);
      const uint32_t at = listed + static_cast<uint32_t>(__builtin_popcountll(mask & ((1ull << lane) - 1ull)));
      if (hit && at < static_cast<uint32_t>(kNextCap)) next_pos[at] = static_cast<uint16_t>(tid + THREADS * k);
      listed += static_cast<uint32_t>(__builtin_popcountll(mask));
    }
  }
  uint32_t n_next = listed;                          // WAVES == 1: the count; WAVES > 1: read back below
  if constexpr (WAVES > 1) {
    __syncthreads();
    n_next = filled;
  } else {
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
  }
  if (n_next > static_cast<uint32_t>(kNextCap)) {    // hundreds of elements in one 128th of a binade: the full route sorts them
    if (tid == 0) todo[row] = 1;
    return;
  }
  double a_next = 0.0;
  for (uint32_t q = tid; q < n_next; q += THREADS) {
    const uint32_t at = next_pos[q];
    a_next = fmax(a_next, fabs(static_cast<double>(wr[at])) * s[at]);
  }
  a_next = over_row(a_next, max_slot, dmax);         // (WAVES > 1: its barrier also publishes the positions)
  int P2 = 64;
  while (P2 < static_cast<int>(P)) P2 <<= 1;
  if constexpr (WAVES == 1) {
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
  }
  for (int q0 = tid; q0 < P2; q0 += 4 * THREADS) {    // four gathers in flight per thread
    uint32_t at[4];
    float wq[4];
    double sq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int slot = q0 + q * THREADS;
      at[q] = slot < static_cast<int>(P) ? sel_pos[slot] : 0u;
      wq[q] = wr[at[q]];
      sq[q] = s[at[q]];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int slot = q0 + q * THREADS;
      if (slot < P2)                                   // padding (composite 0) sorts behind every element
        comp[lds_pad(slot)] = slot < static_cast<int>(P) ? key_composite(fabs(static_cast<double>(wq[q])) * sq[q], at[q]) : 0ull;
    }
  }
  auto row_sync = [&]() {
    if constexpr (WAVES > 1) {
      __syncthreads();
    } else {                                         // one wave: its LDS operations complete in order
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);
    }
  };
  row_sync();
  MI355Q_PREFIX_STAMP(3);
  {
    const int per_thread = P2 / THREADS;             // elements per thread with every thread at work
    for (int k = 2; k <= P2; k <<= 1) {
      int jj = k >> 1;
      while (jj > 0) {
        if (jj >= 4 && per_thread >= 8) {
          sort_levels_u64<3>(comp, P2, k, jj);
          jj >>= 3;
        } else if (jj >= 2 && per_thread >= 4) {
          sort_levels_u64<2>(comp, P2, k, jj);
          jj >>= 2;
        } else {
          sort_levels_u64<1>(comp, P2, k, jj);
          jj >>= 1;
        }
        row_sync();
      }
    }
  }
  MI355Q_PREFIX_STAMP(4);
  if (wv) return;
  // ---- 3. the scan of candidates 1..P (one wave; see clip_scan_wave_kernel)
  const double uk = u[0], nk = noise[0];
  const int Pn = static_cast<int>(P);
  auto exact_key = [&](int i, uint32_t* position) {  // (key, position) of the i-th element of the sorted prefix
    const uint32_t at = 0xFFFFu - static_cast<uint32_t>(comp[lds_pad(i)] & 0xFFFFull);
    *position = at;
    return fabs(static_cast<double>(wr[at])) * s[at];
  };
  uint32_t p0;
  const double a0 = exact_key(0, &p0);
  double best_c = a0, best_e = (a0 * a0) * nk;
  int32_t best_i = -1;
  const int chain = lane < 3 ? lane : 0;
  double acc = 0.0;
  bool out_of_order = false;
  // (key, mass, position) of this lane's element of a chunk; the NEXT chunk's are fetched while this one's sums run
  uint32_t pi_n = 0;
  double ai_n = lane < Pn ? exact_key(lane, &pi_n) : 0.0;
  double mi_n = lane < Pn ? m[pi_n] : 0.0;
  for (int base = 0; base < Pn; base += 64) {
    const int i = base + lane;
    const bool live = i < Pn;
    const uint32_t pi = pi_n;
    const double ai = ai_n, mi = mi_n;
    if (i + 64 < Pn) {
      ai_n = exact_key(i + 64, &pi_n);
      mi_n = m[pi_n];
    } else {
      ai_n = mi_n = 0.0;
      pi_n = 0;
    }
    double lower = __shfl_down(ai, 1, kWave);        // the next element's key: the neighbouring lane's
    uint32_t pn = static_cast<uint32_t>(__shfl_down(static_cast<int>(pi), 1, kWave));
    const double first_of_next = __shfl(ai_n, 0, kWave);                       // lane 63's neighbour opens the next chunk
    const uint32_t first_pos = static_cast<uint32_t>(__shfl(static_cast<int>(pi_n), 0, kWave));
    if (lane == 63) {
      lower = first_of_next;
      pn = first_pos;
    }
    const bool has_next = i + 1 < Pn;
    if (!has_next) lower = a_next;
    // the composites order equal upper-48-bit keys by position: the exact keys must agree with that order
    if (live && has_next && !((ai > lower) | ((ai == lower) & (pi < pn)))) out_of_order = true;
    addend[0][lane] = mi;
    addend[1][lane] = ai * mi;
    addend[2][lane] = (ai * ai) * mi;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // (sixteen addends at a time: the whole chunk in registers, as clip_scan_wave_kernel keeps it, is 128 VGPRs)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double v[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) v[t] = addend[chain][16 * q + t];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        acc = acc + v[t];
        v[t] = acc;
      }
      if (lane < 3) {
#pragma unroll
        for (int t = 0; t < 16; ++t) prefix[chain][16 * q + t] = v[t];
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const double run_m = prefix[0][lane], run_am = prefix[1][lane], run_a2m = prefix[2][lane];
    if (live) {
      double cc = (2.0 * run_am) / (uk + 2.0 * run_m);
      cc = fmin(fmax(cc, lower), ai);
      const double c2 = cc * cc;
      const double e = ((c2 * nk + run_a2m) - (2.0 * cc) * run_am) + c2 * run_m;
      if (e < best_e) {
        best_e = e;
        best_c = cc;
        best_i = i;
      }
    }
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double oe = __shfl_xor(best_e, off, kWave);
    const double oc = __shfl_xor(best_c, off, kWave);
    const int32_t oi = __shfl_xor(best_i, off, kWave);
    if (oe < best_e || (oe == best_e && oi < best_i)) {
      best_e = oe;
      best_c = oc;
      best_i = oi;
    }
  }
  MI355Q_PREFIX_STAMP(5);
#if defined(MI355Q_PREFIX_PROF)
  if (row == 1000 && lane == 0)
    printf("prefix row 1000: P %u P2 %d | load+max %llu  search %llu  compact+next %llu  sort %llu  scan %llu cycles\n", P, P2,
           stamp[1] - stamp[0], stamp[2] - stamp[1], stamp[3] - stamp[2], stamp[4] - stamp[3], stamp[5] - stamp[4]);
#endif
  // ---- 4. may the rest of the row be left out?
  const double s_m = __shfl(acc, 0, kWave), s_am = __shfl(acc, 1, kWave), s_a2m = __shfl(acc, 2, kWave);
  bool done = Pn == g;
  if (!done) {
    const double mass = (12.0 * qmax * qmax) * nk;                        // M (+ 1e-12), as the host formed noise from it
    const double t_max = (a0 * a0) * nk + 4.0 * (s_a2m + (a_next * a_next) * mass);
    const double slack = 64.0 * static_cast<double>(g) * 1.1102230246251565e-16 * t_max;
    const double stationary = (2.0 * s_am) / (uk + 2.0 * s_m);
    const double an2 = a_next * a_next;
    const double e_next = ((an2 * nk + s_a2m) - (2.0 * a_next) * s_am) + an2 * s_m;
    // (every comparison is false when one of its operands is NaN, and inf fails `t_max < inf`)
    done = stationary >= a_next * (1.0 + 1e-6) && e_next > best_e + 4.0 * slack && t_max < __builtin_inf() && best_e >= 0.0;
  }
  if (__ballot(out_of_order) != 0) done = false;
  if (lane == 0) {
    todo[row] = done ? 0 : 1;
    if (done) {
      if (bounds) bounds[row] = best_c;
      if (scale) {
        double sc = fmax(best_c, 1e-9) / qmax;
        if (blockwise) {
          uint16_t half_bits;
          sc = static_cast<double>(round_scale_blockwise(static_cast<float>(sc), &half_bits));
        }
        scale[row] = sc;
      }
    }
  }
}

// ------------------------------------------------- dequantized weight recovery ---
// ref: algorithms/uniform_quantize/dequantized_weight_recovery.py:48-61, 118-186. The scale of a
// group of fake-quantized weights is the smallest positive step between its sorted magnitudes
// (with 0 appended). The minimum is order independent, so after the segment sort every element
// compares with its successor and the per-segment minimum is an atomic on the (positive) bit
// pattern. rounded != 0: the reference works in float32 there (steps rounded to float32, only
// steps > float32(1e-9) count, floor float32(1e-9)); rounded == 0 (TENSORWISE): float64, every
// positive step counts, floor 1e-9.
__global__ __launch_bounds__(256) void gap_init_kernel(uint64_t* __restrict__ best, int64_t segments) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < segments) best[i] = 0x7FF0000000000000ull;   // +inf
}

__global__ __launch_bounds__(256) void gap_kernel(const double* __restrict__ keys, int64_t total,
                                                  int64_t g, int32_t rounded,
                                                  uint64_t* __restrict__ best) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= total) return;
  const int64_t seg = e / g, i = e - seg * g;
  const double a = keys[e];
  const double b = (i + 1 < g) ? keys[e + 1] : 0.0;     // descending order; the appended 0 is last
  double step = a - b;                                   // exact: both are float32 magnitudes
  bool counts;
  if (rounded) {
    const float f = static_cast<float>(step);
    counts = f > 1e-9f;
    step = static_cast<double>(f);
  } else {
    counts = step > 0.0;
  }
  if (counts) {
    const uint64_t bits = __builtin_bit_cast(uint64_t, step);
    if (bits < best[seg]) atomicMin(reinterpret_cast<unsigned long long*>(best + seg),
                                    static_cast<unsigned long long>(bits));
  }
}

__global__ __launch_bounds__(256) void gap_final_kernel(uint64_t* __restrict__ best, int64_t segments,
                                                        int32_t rounded) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= segments) return;
  const double v = __builtin_bit_cast(double, best[i]);
  const double floor_v = rounded ? static_cast<double>(1e-9f) : 1e-9;
  const double out = (best[i] == 0x7FF0000000000000ull) ? floor_v : fmax(v, floor_v);
  best[i] = __builtin_bit_cast(uint64_t, out);
}

// max over all elements of |q * scale - w| in FP64 (int32 * float32 promotes to float64 in
// NumPy); NaN wins, as np.max propagates it.
__global__ __launch_bounds__(256) void recover_error_kernel(const float* __restrict__ w,
                                                            const int8_t* __restrict__ q,
                                                            const double* __restrict__ scale,
                                                            int64_t total, int64_t g,
                                                            uint64_t* __restrict__ worst) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  double diff = 0.0;
  if (e < total) diff = fabs(static_cast<double>(q[e]) * scale[e / g] - static_cast<double>(w[e]));
  uint64_t bits = __builtin_bit_cast(uint64_t, diff) & 0x7FFFFFFFFFFFFFFFull;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const uint64_t o = __shfl_xor(static_cast<unsigned long long>(bits), off, kWave);
    bits = o > bits ? o : bits;
  }
  if ((threadIdx.x & 63) == 0 && bits > *worst)
    atomicMax(reinterpret_cast<unsigned long long*>(worst), static_cast<unsigned long long>(bits));
}

// ------------------------------------------------------------------- quantize ---
__global__ __launch_bounds__(256) void quantize_kernel(const float* __restrict__ w,
                                                       const double* __restrict__ s,
                                                       const double* __restrict__ scale,
                                                       int64_t total, int64_t d, int64_t g,
                                                       double qlo, double qhi,
                                                       int8_t* __restrict__ out) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= total) return;
  const int64_t j = e % d;
  const double v = (static_cast<double>(w[e]) * s[j]) / scale[e / g];
  double r = __builtin_rint(v);
  r = fmin(fmax(r, qlo), qhi);
  out[e] = (v != v) ? 0 : static_cast<int8_t>(static_cast<int>(r));
}

bool pow2_block(int32_t g) { return g == 32 || g == 64 || g == 128 || g == 256; }

}  // namespace
}  // namespace mi355q

using namespace mi355q;

extern "C" int32_t mi355q_oscar_col_sumsq_f32(const float* x, int64_t rows, int64_t d, int32_t mean,
                                              double* out, void* stream) {
  clear_error();
  if (!x || !out) return fail(MI355Q_BAD_ARG, "oscar_col_sumsq: null pointer");
  if (rows <= 0 || d <= 0) return fail(MI355Q_BAD_SHAPE, "oscar_col_sumsq: rows=%lld d=%lld",
                                       (long long)rows, (long long)d);
  const unsigned blocks = static_cast<unsigned>((d + 63) / 64);
  hipLaunchKernelGGL(col_sumsq_kernel<64>, dim3(blocks), dim3(64), 0, as_stream(stream), x, rows, d,
                     mean, out);
  MI355Q_CHECK_LAUNCH("oscar_col_sumsq");
  return MI355Q_OK;
}

extern "C" int32_t mi355q_oscar_group_terms_f32(const float* w, const double* s, int64_t n,
                                                int64_t d, int32_t g, double* top2_workspace,
                                                int32_t* winner_out, double* wsq_out,
                                                double* sums_out, void* stream) {
  clear_error();
  if (!w || !s || !top2_workspace || !winner_out || !wsq_out || !sums_out)
    return fail(MI355Q_BAD_ARG, "oscar_group_terms: null pointer");
  if (n <= 0 || d <= 0 || g <= 0 || d % g)
    return fail(MI355Q_BAD_SHAPE, "oscar_group_terms: n=%lld d=%lld g=%d", (long long)n,
                (long long)d, g);
  if (d > 0x7FFFFFFF) return fail(MI355Q_UNSUPPORTED, "oscar_group_terms: d too large");
  hipStream_t st = as_stream(stream);
  const int64_t G = d / g;
  if (g == d) {
    hipLaunchKernelGGL(group_top_row_kernel, dim3(static_cast<unsigned>(n)), dim3(256), 0, st, w, s,
                       n, d, top2_workspace, winner_out, wsq_out);
  } else {
    if (!pow2_block(g))
      return fail(MI355Q_UNSUPPORTED, "oscar_group_terms: block size %d (32/64/128/256)", g);
    const unsigned blocks = static_cast<unsigned>((n * d / 4 + 255) / 256);
#define MI355Q_LAUNCH_TOP(L)                                                                    \
  hipLaunchKernelGGL(group_top_block_kernel<L>, dim3(blocks), dim3(256), 0, st, w, s, n, d, g, \
                     top2_workspace, winner_out, wsq_out)
    switch (g) {
      case 32: MI355Q_LAUNCH_TOP(8); break;
      case 64: MI355Q_LAUNCH_TOP(16); break;
      case 128: MI355Q_LAUNCH_TOP(32); break;
      default: MI355Q_LAUNCH_TOP(64); break;
    }
#undef MI355Q_LAUNCH_TOP
  }
  MI355Q_CHECK_LAUNCH("oscar_group_top");
  int leaf = 0, depth = 0;
  if (balanced_chunk(static_cast<int>(n - (n - 1) / 8192 * 8192), &leaf, &depth))
    hipLaunchKernelGGL(group_sum_balanced_kernel, dim3(static_cast<unsigned>(G)), dim3(64), 0, st,
                       top2_workspace, n, leaf, depth, sums_out);
  else
    hipLaunchKernelGGL(group_sum_kernel, dim3(static_cast<unsigned>(G)), dim3(64), 0, st,
                       top2_workspace, n, sums_out);
  MI355Q_CHECK_LAUNCH("oscar_group_sum");
  return MI355Q_OK;
}

extern "C" int32_t mi355q_oscar_winner_energy_f64(const int32_t* winner, const double* wsq,
                                                  int64_t n, int64_t d, int32_t g, double* eff_out,
                                                  void* stream) {
  clear_error();
  if (!winner || !wsq || !eff_out) return fail(MI355Q_BAD_ARG, "oscar_winner_energy: null pointer");
  if (n <= 0 || d <= 0 || g <= 0 || d % g)
    return fail(MI355Q_BAD_SHAPE, "oscar_winner_energy: n=%lld d=%lld g=%d", (long long)n,
                (long long)d, g);
  hipLaunchKernelGGL(winner_energy_kernel,
                     dim3(static_cast<unsigned>(d / g), static_cast<unsigned>((g + 255) / 256)),
                     dim3(256), 0, as_stream(stream), winner, wsq, n, d, g, eff_out);
  MI355Q_CHECK_LAUNCH("oscar_winner_energy");
  return MI355Q_OK;
}

namespace {
size_t align256(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }
}  // namespace

extern "C" int32_t mi355q_oscar_clip_workspace_bytes(int64_t n, int64_t d, int64_t g,
                                                     size_t* bytes_out) {
  clear_error();
  if (!bytes_out) return fail(MI355Q_BAD_ARG, "oscar_clip_workspace_bytes: null pointer");
  if (n <= 0 || d <= 0 || g <= 0 || (n * d) % g)
    return fail(MI355Q_BAD_SHAPE, "oscar_clip_workspace_bytes: n=%lld d=%lld g=%lld", (long long)n,
                (long long)d, (long long)g);
  const int64_t total = n * d;
  if (total > 0xFFFFFFFFll - g) return fail(MI355Q_UNSUPPORTED, "oscar clip: more than 2^32 weights");
  // two (key, value) slabs: the run merges ping-pong between them; behind them one byte per segment (the rows the prefix
  // kernel left to the full sort, clip_prefix_kernel)
  *bytes_out = 4 * align256(static_cast<size_t>(total) * sizeof(double)) + align256(static_cast<size_t>(total / g));
  return MI355Q_OK;
}

namespace {
struct Sorted {
  const double* keys;
  const double* vals;
  int64_t seg_stride, elem_stride;
};

// Stable descending sort of every g-element segment of |w| * s (s == NULL: |w|) carrying m[column]
// (m == NULL: nothing): segments of up to 8192 elements are sorted whole in LDS (bitonic network on
// (key, position) pairs); longer ones -- rows beyond 8192 columns, odd lengths, the whole tensor
// for TENSORWISE -- as 8192-element runs in LDS followed by log2(runs) stable merge passes.
constexpr int64_t kSortRun = 8192;

int32_t sort_segments(const float* w, const double* s, const double* m, int64_t n, int64_t d,
                      int64_t g, bool allow_rank_major, void* workspace, size_t need, hipStream_t st,
                      Sorted* out, const uint8_t* todo = nullptr) {
  (void)need;
  const int64_t total = n * d;
  const size_t slab = align256(static_cast<size_t>(total) * sizeof(double));
  char* base = static_cast<char*>(workspace);
  double* keys_a = reinterpret_cast<double*>(base);
  double* vals_a = reinterpret_cast<double*>(base + slab);
  double* keys_b = reinterpret_cast<double*>(base + 2 * slab);
  double* vals_b = reinterpret_cast<double*>(base + 3 * slab);
  int64_t seg_stride = g, elem_stride = 1;
  const bool runs = g > kSortRun;
  const int64_t unit_len = runs ? kSortRun : g;                       // what one LDS sort covers
  const int64_t runs_per_seg = runs ? (g + kSortRun - 1) / kSortRun : 1;
  const int64_t units = (total / g) * runs_per_seg;
  if (runs) allow_rank_major = false;
  int32_t P = 32;
  while (P < unit_len) P <<= 1;
  const int32_t tile = P > 4096 ? P : 4096;
  const int32_t transposed = (P <= 256 && allow_rank_major) ? 1 : 0;
  const size_t lds = (static_cast<size_t>(lds_pad(tile)) + 2) * (sizeof(double) + sizeof(uint16_t));
  const int64_t tiles = (units + tile / P - 1) / (tile / P);
  if (tiles > 0x7FFFFFFFLL) return fail(MI355Q_UNSUPPORTED, "oscar sort: too many segments");
  if (lds > 64 * 1024) {
    hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(sort_tile_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                        static_cast<int>(lds));
    if (ea != hipSuccess) return fail(MI355Q_HIP_ERROR, "oscar sort LDS: %s", hipGetErrorString(ea));
  }
  const int threads = tile >= 8192 ? 1024 : 512;   // measured best of {256, 512, 1024} per tile size
  hipLaunchKernelGGL(sort_tile_kernel, dim3(static_cast<unsigned>(tiles)), dim3(threads), lds, st,
                     w, s, m, units, static_cast<int32_t>(unit_len), P, tile, d, transposed, g,
                     static_cast<int32_t>(runs_per_seg), keys_a, vals_a, todo);
  MI355Q_CHECK_LAUNCH("oscar_sort_tile");
  const double *keys = keys_a, *vals = vals_a;
  double *keys_o = keys_b, *vals_o = vals_b;
  for (int64_t run = kSortRun; run < g; run *= 2) {
    hipLaunchKernelGGL(merge_runs_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256),
                       0, st, keys, m ? vals : nullptr, total, g, run, keys_o, vals_o, todo);
    MI355Q_CHECK_LAUNCH("oscar_merge_runs");
    const double* tk = keys; keys = keys_o; keys_o = const_cast<double*>(tk);
    const double* tv = vals; vals = vals_o; vals_o = const_cast<double*>(tv);
  }
  if (transposed) {
    seg_stride = 1;
    elem_stride = total / g;
  }
  out->keys = keys;
  out->vals = vals;
  out->seg_stride = seg_stride;
  out->elem_stride = elem_stride;
  return MI355Q_OK;
}
}  // namespace

extern "C" int32_t mi355q_oscar_clip_bounds_f32(const float* w, const double* s, const double* m,
                                                int64_t n, int64_t d, int64_t g, const double* u,
                                                const double* noise, int32_t qmax,
                                                int32_t blockwise_scale, double* bounds_out,
                                                double* scale_out, void* workspace,
                                                size_t workspace_bytes, void* stream) {
  clear_error();
  if (!w || !s || !m || !u || !noise || (!bounds_out && !scale_out) || !workspace)
    return fail(MI355Q_BAD_ARG, "oscar_clip_bounds: null pointer");
  if (qmax <= 0) return fail(MI355Q_BAD_ARG, "oscar_clip_bounds: qmax=%d", qmax);
  size_t need = 0;
  int32_t st_code = mi355q_oscar_clip_workspace_bytes(n, d, g, &need);
  if (st_code != MI355Q_OK) return st_code;
  if (workspace_bytes < need)
    return fail(MI355Q_BAD_ARG, "oscar_clip_bounds: workspace %zu < %zu", workspace_bytes, need);
  if (g != n * d && d % g)
    return fail(MI355Q_BAD_SHAPE, "oscar_clip_bounds: g=%lld does not divide d=%lld", (long long)g,
                (long long)d);
  hipStream_t st = as_stream(stream);
  const int64_t total = n * d, segments = total / g;
  const int64_t G = (g == total) ? 1 : d / g;
  // CHANNELWISE rows are answered from a prefix of their sorted order where that provably gives the full scan's
  // answer (clip_prefix_kernel); the rows it flags -- and every other layout -- go through the full sort + scan.
  // MI355Q_OSCAR_PREFIX=0: full sort + scan for every row (A / B timing, and the equality test of the two routes).
  const uint8_t* todo = nullptr;
  const char* prefix_env = getenv("MI355Q_OSCAR_PREFIX");          // (read per call: the tests switch routes in one process)
  const bool prefix_on = !(prefix_env && prefix_env[0] == '0');
  // (rows of 384 .. 1023 columns since late round 6: 2^24 elements in rows of 768: 0.68 -> 0.20 ms, of 384: 0.53 -> 0.37 ms;
  // at 256 columns the prefix -- 128 elements, half the row -- costs more than the full sort)
#ifndef MI355Q_OSCAR_PREFIX_MIN
#define MI355Q_OSCAR_PREFIX_MIN 384
#endif
  if (prefix_on && g == d && n > 0 && g >= MI355Q_OSCAR_PREFIX_MIN && g <= 16384 && qmax >= 7) {
    uint8_t* flags = static_cast<uint8_t*>(workspace) + 4 * align256(static_cast<size_t>(total) * sizeof(double));
    // elements asked for: a sixteenth of the row within [128, 256] for the one-wave form (it sorts at most 512), a
    // thirty-second within [256, 512] beyond (at most 1024 sorted)
    int64_t target = g / 16;
    if (g <= 4096) target = target < 128 ? 128 : (target > 256 ? 256 : target);
    else target = g / 32 < 256 ? 256 : (g / 32 > 512 ? 512 : g / 32);
    if (const char* e = getenv("MI355Q_OSCAR_PREFIX_TARGET")) {           // A / B: how long a prefix is worth sorting
      const long v = atol(e);
      if (v >= 32 && v <= (g <= 4096 ? 512 : 1024)) target = v;
    }
#define MI355Q_LAUNCH_PREFIX(EPT, WAVES)                                                                                \
  hipLaunchKernelGGL((clip_prefix_kernel<EPT, WAVES>), dim3(static_cast<unsigned>(n)), dim3(64 * WAVES), 0, st, w, s, m, n, \
                     static_cast<int32_t>(g), static_cast<int32_t>(target), u, noise, static_cast<double>(qmax),       \
                     blockwise_scale, bounds_out, scale_out, flags)
    if (g <= 1024) MI355Q_LAUNCH_PREFIX(16, 1);
    else if (g <= 2048) MI355Q_LAUNCH_PREFIX(32, 1);
    else if (g <= 4096) MI355Q_LAUNCH_PREFIX(64, 1);
    else if (g <= 8192) MI355Q_LAUNCH_PREFIX(32, 4);
    else MI355Q_LAUNCH_PREFIX(64, 4);
#undef MI355Q_LAUNCH_PREFIX
    MI355Q_CHECK_LAUNCH("oscar_clip_prefix");
    todo = flags;
  }
  Sorted sorted;
  st_code = sort_segments(w, s, m, n, d, g, true, workspace, need, st, &sorted, todo);
  if (st_code != MI355Q_OK) return st_code;
  const double* keys_out = sorted.keys;
  const double* vals_out = sorted.vals;
  const int64_t seg_stride = sorted.seg_stride, elem_stride = sorted.elem_stride;
  if (elem_stride == 1 && g >= 256 && (segments < 12288 || todo)) {
    // the wave form costs ~40 SIMD cycles per element instead of ~6, but runs on `segments` (not
    // segments / 64) waves: measured faster up to ~12k long segments (4096 x 4096: 0.73 vs 1.27 ms)
    // -- and always when only the rows the prefix kernel flagged are left
    hipLaunchKernelGGL(clip_scan_wave_kernel, dim3(static_cast<unsigned>((segments + 3) / 4)), dim3(256),
                       0, st, keys_out, vals_out, segments, g, G, u, noise, static_cast<double>(qmax),
                       blockwise_scale, bounds_out, scale_out, todo);
  } else {
    hipLaunchKernelGGL(clip_scan_kernel<8>, dim3(static_cast<unsigned>((segments + 63) / 64)), dim3(64),
                       0, st, keys_out, vals_out, segments, g, G, seg_stride, elem_stride, u, noise,
                       static_cast<double>(qmax), blockwise_scale, bounds_out, scale_out, todo);
  }
  MI355Q_CHECK_LAUNCH("oscar_clip_scan");
  return MI355Q_OK;
}

extern "C" int32_t mi355q_oscar_quantize_f32(const float* w, const double* s, const double* scale,
                                             int64_t n, int64_t d, int64_t g, int32_t qlo,
                                             int32_t qhi, int8_t* out, void* stream) {
  clear_error();
  if (!w || !s || !scale || !out) return fail(MI355Q_BAD_ARG, "oscar_quantize: null pointer");
  if (n <= 0 || d <= 0 || g <= 0 || (n * d) % g || qlo < -128 || qhi > 127 || qlo > qhi)
    return fail(MI355Q_BAD_SHAPE, "oscar_quantize: n=%lld d=%lld g=%lld range [%d, %d]",
                (long long)n, (long long)d, (long long)g, qlo, qhi);
  const int64_t total = n * d;
  hipLaunchKernelGGL(quantize_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                     as_stream(stream), w, s, scale, total, d, g, static_cast<double>(qlo),
                     static_cast<double>(qhi), out);
  MI355Q_CHECK_LAUNCH("oscar_quantize");
  return MI355Q_OK;
}

extern "C" int32_t mi355q_dwr_scales_f32(const float* w, int64_t n, int64_t d, int64_t g,
                                         int32_t rounded, double* scale_out, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  clear_error();
  if (!w || !scale_out || !workspace) return fail(MI355Q_BAD_ARG, "dwr_scales: null pointer");
  size_t need = 0;
  int32_t code = mi355q_oscar_clip_workspace_bytes(n, d, g, &need);
  if (code != MI355Q_OK) return code;
  if (workspace_bytes < need)
    return fail(MI355Q_BAD_ARG, "dwr_scales: workspace %zu < %zu", workspace_bytes, need);
  hipStream_t st = as_stream(stream);
  const int64_t total = n * d, segments = total / g;
  Sorted sorted;
  code = sort_segments(w, nullptr, nullptr, n, d, g, false, workspace, need, st, &sorted);
  if (code != MI355Q_OK) return code;
  uint64_t* best = reinterpret_cast<uint64_t*>(scale_out);
  hipLaunchKernelGGL(gap_init_kernel, dim3(static_cast<unsigned>((segments + 255) / 256)), dim3(256), 0,
                     st, best, segments);
  hipLaunchKernelGGL(gap_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st,
                     sorted.keys, total, g, rounded, best);
  hipLaunchKernelGGL(gap_final_kernel, dim3(static_cast<unsigned>((segments + 255) / 256)), dim3(256), 0,
                     st, best, segments, rounded);
  MI355Q_CHECK_LAUNCH("dwr_scales");
  return MI355Q_OK;
}

extern "C" int32_t mi355q_dwr_max_error_f32(const float* w, const int8_t* q, const double* scale,
                                            int64_t total, int64_t g, double* max_out,
                                            void* stream) {
  clear_error();
  if (!w || !q || !scale || !max_out) return fail(MI355Q_BAD_ARG, "dwr_max_error: null pointer");
  if (total <= 0 || g <= 0 || total % g)
    return fail(MI355Q_BAD_SHAPE, "dwr_max_error: total=%lld g=%lld", (long long)total, (long long)g);
  hipStream_t st = as_stream(stream);
  hipError_t e = hipMemsetAsync(max_out, 0, sizeof(double), st);
  if (e != hipSuccess) return fail(MI355Q_HIP_ERROR, "dwr_max_error: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(recover_error_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256),
                     0, st, w, q, scale, total, g, reinterpret_cast<uint64_t*>(max_out));
  MI355Q_CHECK_LAUNCH("dwr_max_error");
  return MI355Q_OK;
}
