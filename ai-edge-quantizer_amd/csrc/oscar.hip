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
//   clip_bounds    per (row, group): stable descending sort of |w|*s carrying the masses, three
//                  sequential running sums, the closed-form candidate of every segment, first
//                  minimum
//   quantize       clip(rint((w*s) / scale)) with FP64 product and quotient
// Compiled with -ffp-contract=off: none of the FP64 expressions may be fused.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

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
// g/4 lanes form a group. top2 is laid out [G][n] so that the per-group sum runs over a
// contiguous vector; winner / wsq are [n][G].
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
    const int64_t G = d / g;
    const int64_t k = j0 / g;
    top2[k * n + r] = best.v * best.v;
    winner[r * G + k] = best.j;
    const double t = static_cast<double>(best.w);
    wsq[r * G + k] = t * t;
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

// np.sum of a contiguous FP64 vector: 8192-element chunks added in order, each pairwise. The
// chunks of one vector are summed by different lanes and combined in order by lane 0.
__global__ __launch_bounds__(64) void group_sum_kernel(const double* __restrict__ top2, int64_t n,
                                                       double* __restrict__ sums) {
  const double* a = top2 + static_cast<int64_t>(blockIdx.x) * n;
  const int64_t chunks = (n + 8191) / 8192;
  double total = 0.0;
  for (int64_t base = 0; base < chunks; base += 64) {
    const int64_t c = base + threadIdx.x;
    double part = 0.0;
    if (c < chunks) {
      const int64_t lo = c * 8192;
      part = pairwise_f64(a + lo, (n - lo < 8192) ? n - lo : 8192);
    }
    const int64_t here = (chunks - base < 64) ? chunks - base : 64;
    for (int64_t k = 0; k < here; ++k) {
      const double p = __shfl(part, static_cast<int>(k), kWave);
      total = (base == 0 && k == 0) ? p : total + p;
    }
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// -------------------------------------------------------------- winner_energy ---
// eff[j] = sum over rows, in order, of wsq[r][group(j)] where winner[r][group(j)] == j.
__global__ __launch_bounds__(64) void winner_energy_kernel(const int32_t* __restrict__ winner,
                                                           const double* __restrict__ wsq,
                                                           int64_t n, int64_t d, int32_t g,
                                                           double* __restrict__ eff) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * 64 + threadIdx.x;
  if (j >= d) return;
  const int64_t G = d / g;
  const int64_t k = j / g;
  double acc = 0.0;
  int64_t r = 0;
  for (; r + 8 <= n; r += 8) {
    int32_t wj[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) wj[u] = winner[(r + u) * G + k];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (wj[u] == j) acc = acc + wsq[(r + u) * G + k];
  }
  for (; r < n; ++r)
    if (winner[r * G + k] == j) acc = acc + wsq[r * G + k];
  eff[j] = acc;
}

// ---------------------------------------------------------------- clip_bounds ---
__global__ __launch_bounds__(256) void sort_keys_kernel(const float* __restrict__ w,
                                                        const double* __restrict__ s,
                                                        const double* __restrict__ m, int64_t total,
                                                        int64_t d, double* __restrict__ keys,
                                                        double* __restrict__ vals) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= total) return;
  const int64_t j = e % d;
  keys[e] = fabs(static_cast<double>(w[e])) * s[j];
  vals[e] = m[j];
}

struct SegmentStart {
  uint32_t g;
  __host__ __device__ uint32_t operator()(uint32_t i) const { return i * g; }
};

// One lane per sorted segment: running sums in order, the candidate of every breakpoint
// interval, first minimum. u = M / (6 qmax^2), noise = M / (12 qmax^2) per group (host FP64).
template <int BATCH>
__global__ __launch_bounds__(64) void clip_scan_kernel(const double* __restrict__ keys,
                                                       const double* __restrict__ vals,
                                                       int64_t segments, int64_t g, int64_t G,
                                                       const double* __restrict__ u,
                                                       const double* __restrict__ noise,
                                                       double* __restrict__ bounds) {
  const int64_t seg = static_cast<int64_t>(blockIdx.x) * 64 + threadIdx.x;
  if (seg >= segments) return;
  const double* a = keys + seg * g;
  const double* m = vals + seg * g;
  const double uk = u[seg % G], nk = noise[seg % G];
  double run_m = 0.0, run_am = 0.0, run_a2m = 0.0;
  const double a0 = a[0];
  double best_c = a0, best_e = (a0 * a0) * nk;
  double cur = a0;
  for (int64_t i = 0; i < g; i += BATCH) {
    double ab[BATCH + 1], mb[BATCH];
#pragma unroll
    for (int t = 0; t < BATCH; ++t) {
      mb[t] = (i + t < g) ? m[i + t] : 0.0;
      ab[t + 1] = (i + t + 1 < g) ? a[i + t + 1] : 0.0;
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
  bounds[seg] = best_c;
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
  hipLaunchKernelGGL(col_sumsq_kernel<16>, dim3(blocks), dim3(64), 0, as_stream(stream), x, rows, d,
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
  hipLaunchKernelGGL(winner_energy_kernel, dim3(static_cast<unsigned>((d + 63) / 64)), dim3(64), 0,
                     as_stream(stream), winner, wsq, n, d, g, eff_out);
  MI355Q_CHECK_LAUNCH("oscar_winner_energy");
  return MI355Q_OK;
}

namespace {
// Temporary storage of the library sort for `total` (key, value) pairs in `segments` segments.
hipError_t sort_storage(size_t* bytes, int64_t total, int64_t segments, int64_t g) {
  double* nil = nullptr;
  if (segments == 1)
    return rocprim::radix_sort_pairs_desc(nullptr, *bytes, nil, nil, nil, nil,
                                          static_cast<size_t>(total), 0, 64, hipStream_t(0));
  auto begin = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>(0),
                                                SegmentStart{static_cast<uint32_t>(g)});
  return rocprim::segmented_radix_sort_pairs_desc(nullptr, *bytes, nil, nil, nil, nil,
                                                  static_cast<size_t>(total),
                                                  static_cast<unsigned>(segments), begin, begin + 1,
                                                  0, 64, hipStream_t(0));
}
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
  size_t tmp = 0;
  hipError_t e = sort_storage(&tmp, total, total / g, g);
  if (e != hipSuccess) return fail(MI355Q_HIP_ERROR, "oscar sort sizing: %s", hipGetErrorString(e));
  *bytes_out = 4 * align256(static_cast<size_t>(total) * sizeof(double)) + align256(tmp);
  return MI355Q_OK;
}

extern "C" int32_t mi355q_oscar_clip_bounds_f32(const float* w, const double* s, const double* m,
                                                int64_t n, int64_t d, int64_t g, const double* u,
                                                const double* noise, double* bounds_out,
                                                void* workspace, size_t workspace_bytes,
                                                void* stream) {
  clear_error();
  if (!w || !s || !m || !u || !noise || !bounds_out || !workspace)
    return fail(MI355Q_BAD_ARG, "oscar_clip_bounds: null pointer");
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
  const size_t slab = align256(static_cast<size_t>(total) * sizeof(double));
  char* base = static_cast<char*>(workspace);
  double* keys_in = reinterpret_cast<double*>(base);
  double* vals_in = reinterpret_cast<double*>(base + slab);
  double* keys_out = reinterpret_cast<double*>(base + 2 * slab);
  double* vals_out = reinterpret_cast<double*>(base + 3 * slab);
  void* tmp = base + 4 * slab;
  size_t tmp_bytes = need - 4 * slab;
  hipLaunchKernelGGL(sort_keys_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                     st, w, s, m, total, d, keys_in, vals_in);
  MI355Q_CHECK_LAUNCH("oscar_sort_keys");
  hipError_t e;
  if (segments == 1) {
    e = rocprim::radix_sort_pairs_desc(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out,
                                       static_cast<size_t>(total), 0, 64, st);
  } else {
    auto begin = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>(0),
                                                  SegmentStart{static_cast<uint32_t>(g)});
    e = rocprim::segmented_radix_sort_pairs_desc(tmp, tmp_bytes, keys_in, keys_out, vals_in,
                                                 vals_out, static_cast<size_t>(total),
                                                 static_cast<unsigned>(segments), begin, begin + 1,
                                                 0, 64, st);
  }
  if (e != hipSuccess) return fail(MI355Q_HIP_ERROR, "oscar sort: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(clip_scan_kernel<8>, dim3(static_cast<unsigned>((segments + 63) / 64)), dim3(64),
                     0, st, keys_out, vals_out, segments, g, G, u, noise, bounds_out);
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
