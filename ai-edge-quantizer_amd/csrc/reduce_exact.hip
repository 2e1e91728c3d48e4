// Order-exact reductions: OCTAV clipping search (K5) and MSE scale (a14).
//
//   ref: algorithms/uniform_quantize/octav.py:30-112  (_guess_clipping_with_octav)
//   ref: algorithms/uniform_quantize/mse.py:100-109   (k * sqrt(mean(x^2)))
//
// Both reductions are float32 sums whose value depends on the order NumPy adds
// in. That order is deterministic (verified against NumPy 2.2 in
// tests/test_numpy_sum_model.py) and is reproduced here so the clipping
// constants / scales are BIT-IDENTICAL to the reference, not merely close:
//   * a reduction unit (row, block, or the whole tensor) is consumed in chunks
//     of 8192 elements (the nditer buffer);
//   * inside a chunk every maximal run of consecutive selected elements is summed
//     with NumPy's pairwise routine (n < 8: left to right; n <= 128: eight strided
//     accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) then the tail;
//     larger: split at n/2 rounded down to a multiple of 8) and the run sum is
//     added to the running total:  acc = acc + pairwise(run).
//
// One wave owns one unit. The unit is staged once into LDS (one HBM read for all
// 10 Newton iterations); selections are wave ballots, runs are found with scalar
// bit scans, and short runs are summed from registers with v_readlane.
#include <cstdlib>

#include "common.h"

namespace mi355q {
namespace {

constexpr int kChunk = 8192;  // NumPy nditer buffer size (elements)

__device__ __forceinline__ float lane_bcast(float v, int src_lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// lane L receives lane L-1's value, lane 0 receives 0 (v_mov_b32_dpp wave_shr:1)
__device__ __forceinline__ float wave_shr1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, false));
}

template <bool SQUARE>
__device__ __forceinline__ float tr(float v) {
  if constexpr (SQUARE) return v * v;
  return v;
}

// NumPy pairwise leaf, 8 <= m <= 128: lanes 0..7 are the eight accumulators.
template <bool SQUARE>
__device__ __forceinline__ float leaf_sum(const float* a, int m, int lane) {
  const int full = m & ~7;
  float r = 0.f;
  if (lane < 8) {
    r = tr<SQUARE>(a[lane]);
    for (int i = 8; i < full; i += 8) r = r + tr<SQUARE>(a[i + lane]);
  }
  const float r0 = lane_bcast(r, 0), r1 = lane_bcast(r, 1), r2 = lane_bcast(r, 2),
              r3 = lane_bcast(r, 3), r4 = lane_bcast(r, 4), r5 = lane_bcast(r, 5),
              r6 = lane_bcast(r, 6), r7 = lane_bcast(r, 7);
  float res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (int i = full; i < m; ++i) res = res + tr<SQUARE>(a[i]);
  return res;
}

// pairwise(a[0:n]) for n <= 8192; recursion depth is bounded by DEPTH.
template <bool SQUARE, int DEPTH>
__device__ __noinline__ float pairwise_sum(const float* a, int n, int lane) {
  if (n < 8) {
    float res = 0.f;
    for (int i = 0; i < n; ++i) res = res + tr<SQUARE>(a[i]);
    return res;
  }
  if (n <= 128) return leaf_sum<SQUARE>(a, n, lane);
  if constexpr (DEPTH > 0) {
    int n2 = n / 2;
    n2 -= n2 % 8;
    const float left = pairwise_sum<SQUARE, DEPTH - 1>(a, n2, lane);
    const float right = pairwise_sum<SQUARE, DEPTH - 1>(a + n2, n - n2, lane);
    return left + right;
  } else {
    return leaf_sum<SQUARE>(a, n, lane);  // unreachable for n <= 8192
  }
}

// Running masked sum in NumPy order. All members are wave-uniform.
struct RunSum {
  float acc = 0.f;
  unsigned count = 0;
  int pend_start = 0;    // a run that touches the end of the last batch and may continue
  int pend_len = 0;
  float pend_sum = 0.f;  // its left-to-right sum so far; meaningful while pend_len < 8

  // A run shorter than 8 is summed left to right from 0, so the part seen so far is a valid
  // prefix of that chain as long as the run stays shorter than 8; a run that reaches 8 switches
  // to the eight-accumulator scheme and is summed from memory once its end is known.
  __device__ __forceinline__ void flush(const float* a, int lane) {
    if (pend_len > 0) {
      acc = acc + (pend_len < 8 ? pend_sum : pairwise_sum<false, 8>(a + pend_start, pend_len, lane));
      pend_len = 0;
    }
  }

  // m: ballot of the selected lanes of the batch starting at element `base`;
  // v: this lane's element (lane l <-> element base + l); a: the unit.
  //
  // Fast path (no run of 8+ lanes in the batch; a run pending from the previous batch either
  // ended there or continues here and still stays shorter than 8): all runs are summed at once,
  // lane-parallel, by rounds of "left neighbour's partial sum + mine" (v_mov_b32_dpp
  // wave_shr:1; lane 0 continues the pending chain), then the run totals are added to the
  // running total in lane order. A run that touches the batch end stays pending with its partial
  // sum. Everything else takes the general per-run path.
  __device__ __forceinline__ void feed(unsigned long long m, int base, float v, const float* a,
                                       int lane) {
    if ((base % kChunk) == 0 || (m & 1ull) == 0) flush(a, lane);
    count += static_cast<unsigned>(__builtin_popcountll(m));
    if (m == 0) return;
    const unsigned long long m4 = m & (m >> 1) & (m >> 2) & (m >> 3);
    if ((m4 & (m4 >> 4)) != 0 || (pend_len > 0 && pend_len + __builtin_ctzll(~m) >= 8)) {
      feed_runs(m, base, v, a, lane);
      return;
    }
    const float carry = pend_len > 0 ? pend_sum : 0.f;  // lane 0 continues the pending chain
    pend_len = 0;
    const bool sel = ((m >> lane) & 1ull) != 0;
    float partial = sel ? (lane == 0 ? carry : 0.f) + v : 0.f;
    const unsigned long long m2 = m & (m >> 1);
    if (m2 != 0) {  // some run is longer than one element
      // every lane repeats "left neighbour's partial + mine" (run starts keep their first sum): a
      // lane at position p of its run is final after p rounds and a further round recomputes
      // the same value, so the round count only has to reach the longest run
      const bool cont = (((m2 << 1) >> lane) & 1ull) != 0;  // my left neighbour is in my run
      const int rounds = (m2 & (m >> 2)) == 0 ? 1 : (m4 == 0 ? 2 : 6);
      for (int t = 0; t < rounds; ++t) {
        const float left = wave_shr1(partial);
        if (cont) partial = left + v;
      }
    }
    unsigned long long mf = m;
    if (m >> 63) {  // trailing run (1..7 lanes): it may continue in the next batch
      const int t = __builtin_clzll(~m);
      mf = m & (~0ull >> t);
      pend_start = base + 64 - t;
      pend_len = t;
      pend_sum = lane_bcast(partial, 63);
    }
    unsigned long long ends = mf & ~(mf >> 1);  // last lane of every finished run
    const int k = __builtin_popcountll(ends);
    if (k <= 3) {
      while (ends != 0) {
        acc = acc + lane_bcast(partial, __builtin_ctzll(ends));
        ends &= ends - 1ull;
      }
      return;
    }
    // acc = (..((acc + R0) + R1)..) + R(k-1) without a scalar loop over set bits: the run totals
    // are compacted to lanes 0..k-1 (one ds_permute; the other lanes park theirs above k), then
    // every lane repeats x = left neighbour's x + R (lane 0's neighbour is acc). Lane j is final
    // after j + 1 rounds and stays final, so k rounds (rounded up to the unroll) leave the
    // total in lane k - 1.
    const bool is_end = ((ends >> lane) & 1ull) != 0;
    const int rank = __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(ends >> 32),
                                               __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(ends), 0));
    const int dest = is_end ? rank : k + lane - rank;
    const float r = __int_as_float(__builtin_amdgcn_ds_permute(dest << 2, __float_as_int(partial)));
    float x = r;
    for (int t = 0; t < k; t += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float left = __int_as_float(__builtin_amdgcn_update_dpp(
            __float_as_int(acc), __float_as_int(x), 0x138, 0xF, 0xF, false));
        x = left + r;
      }
    }
    acc = lane_bcast(x, k - 1);
  }

  // left-to-right sum of lanes s .. s+len-1 on top of `start` (len < 8)
  __device__ __forceinline__ float chain(float start, float v, int s, int len) {
    float rs = start;
    for (int i = 0; i < len; ++i) rs = rs + lane_bcast(v, s + i);
    return rs;
  }

  // General path: runs one by one (runs of 8+, runs that grow to 8+ across batches).
  __device__ __forceinline__ void feed_runs(unsigned long long m, int base, float v, const float* a,
                                            int lane) {
    while (m != 0) {
      const int s = __builtin_ctzll(m);
      const unsigned long long t = m >> s;
      const int len = (~t == 0) ? 64 - s : __builtin_ctzll(~t);
      if (s + len == 64) {  // touches the batch end: may continue in the next batch
        if (pend_len > 0) {  // (only when s == 0: the whole batch belongs to the pending run)
          pend_len += len;
        } else {
          pend_start = base + s;
          pend_len = len;
          if (len < 8) pend_sum = chain(0.f, v, s, len);
        }
        return;
      }
      if (pend_len > 0) {  // run that started in an earlier batch ends here (s == 0)
        if (pend_len + len < 8) {
          acc = acc + chain(pend_sum, v, 0, len);
          pend_len = 0;
        } else {
          pend_len += len;
          flush(a, lane);
        }
      } else if (len < 8) {  // sum straight from registers
        acc = acc + chain(0.f, v, s, len);
      } else {
        acc = acc + pairwise_sum<false, 8>(a + base + s, len, lane);
      }
      m &= ~(((1ull << len) - 1ull) << s);  // len < 64 here
    }
  }
};

// Copy a unit into this wave's LDS slice (or return the global pointer).
template <bool USE_LDS>
__device__ __forceinline__ const float* stage_unit(const float* g, int len, float* lds, int lane) {
  if constexpr (!USE_LDS) return g;
  if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
    const int n4 = len / 4;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* l4 = reinterpret_cast<float4*>(lds);
    for (int i = lane; i < n4; i += kWave) l4[i] = g4[i];
    for (int i = n4 * 4 + lane; i < len; i += kWave) lds[i] = g[i];
  } else {
    for (int i = lane; i < len; i += kWave) lds[i] = g[i];
  }
  return lds;
}

struct OctavArgs {
  const float* x;
  long long units;
  int len;          // elements per unit
  int lds_stride;   // floats per wave slice (multiple of 4)
  int max_iter;
  int count_is_f64; // axis given: s * N is evaluated in float64 (N is np.int64)
  float s;          // float32(4^-bits / divisor)
  float* hist;      // [max_iter][units] guesses
  unsigned long long* moving;  // bit `it` set: some unit's guess still moved in iteration `it`
  // octav_rows_kernel -> octav_tail_kernel hand-over (null: the rows kernel runs every iteration itself)
  struct TailState* tail;      // [units]
  float* tail_values;          // [units][2][tail_cap]: the candidates' values, positive mask first
  unsigned short* tail_pos;    // [units][2][tail_cap]: their positions in the row
  int tail_cap;
};

// What a row's workgroup leaves for the wave that finishes the row (octav_tail_kernel).
struct TailState {
  int next_it;          // first iteration the tail runs; 0: the rows kernel finished the row itself
  int n[2];             // candidates per mask
  float guess;          // the iterate to continue from
  float cand_guess;     // every listed candidate satisfies |x| >= cand_guess
  unsigned moved_lo, moved_hi;   // iterations (bit mask) in which the guess still moved so far
  int pad;
};

struct OctavStep {
  float next;
  bool close;
};

// One Newton update from the two masked sums and their element counts
// (ref octav.py:76-110), with NumPy's promotions spelled out.
__device__ __forceinline__ OctavStep octav_step(float guess, float pos_sum, float neg_sum,
                                                long long pos_count, long long neg_count,
                                                long long len, float s, int count_is_f64) {
  const float num = pos_sum - neg_sum;
  float den = static_cast<float>(pos_count);
  den = static_cast<float>(static_cast<double>(den) + static_cast<double>(neg_count));
  den = den * (1.0f - s);
  if (count_is_f64)
    den = static_cast<float>(static_cast<double>(den) +
                             static_cast<double>(s) * static_cast<double>(len));
  else
    den = den + s * static_cast<float>(len);
  const float next = num / den;
  // np.allclose(old, new): |old - new| <= atol + rtol * |new|, all float32
  const float tol = 1e-8f + 1e-5f * fabsf(next);
  const bool close = (fabsf(guess - next) <= tol && __builtin_isfinite(next)) || guess == next;
  return {next, close};
}

// The reference's early stop only asks whether *any* unit still moved in an iteration, so the
// units publish a bit mask instead of counting: one atomic per wave at most, and none when the
// bits are already there (a stale read only costs a redundant atomic). With 131 072 block units
// the per-iteration counters were ~1.3 M same-address atomics = 6.5 of the 7.3 ms.
__device__ __forceinline__ void publish_moving(unsigned long long* flags, unsigned long long moved) {
  if (moved == 0) return;
  const unsigned long long seen = __atomic_load_n(flags, __ATOMIC_RELAXED);
  if ((seen & moved) != moved) atomicOr(flags, moved);
}

// A Newton update that returns its own input (same value, hence the same two masks, the same
// sums and the same update again) has reached an exact fixed point: every later iteration of
// this unit would recompute the identical number and find it "close". The masks stop changing
// after a handful of iterations on real weights, so the remaining passes over the unit are
// skipped and their history entries filled in.
__device__ __forceinline__ bool reached_fixed_point(float guess, float next) { return next == guess; }

__device__ __forceinline__ void repeat_iterate(const OctavArgs& a, int it, long long unit, float value) {
  for (int r = it + 1; r < a.max_iter; ++r) a.hist[static_cast<long long>(r) * a.units + unit] = value;
}

template <bool USE_LDS>
__global__ void octav_kernel(OctavArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const long long unit = static_cast<long long>(blockIdx.x) * (blockDim.x / kWave) + wave;
  const bool live = unit < a.units;
  const float* u = nullptr;
  if (live)
    u = stage_unit<USE_LDS>(a.x + unit * a.len, a.len, smem + static_cast<size_t>(wave) * a.lds_stride, lane);
  if constexpr (USE_LDS) __syncthreads();
  if (!live) return;

  const int len = a.len;
  const float qnan = __builtin_nanf("");
  float guess = 1.0f;
  unsigned long long moved = 0;  // iterations in which this unit's guess still moved
  for (int it = 0; it < a.max_iter; ++it) {
    RunSum pos, neg;
    const float hi = guess, lo = -guess;
    for (int base = 0; base < len; base += kWave) {
      const int i = base + lane;
      const float v = i < len ? u[i] : qnan;
      const unsigned long long mp = __ballot(v >= hi), mn = __ballot(v <= lo);
      // nothing selected and no run waiting for its end: the batch changes nothing
      if ((mp | mn) == 0 && (pos.pend_len | neg.pend_len) == 0) continue;
      pos.feed(mp, base, v, u, lane);
      neg.feed(mn, base, v, u, lane);
    }
    pos.flush(u, lane);
    neg.flush(u, lane);
    const OctavStep st = octav_step(guess, pos.acc, neg.acc, pos.count, neg.count, len, a.s,
                                    a.count_is_f64);
    if (lane == 0) a.hist[static_cast<long long>(it) * a.units + unit] = st.next;
    if (!st.close) moved |= 1ull << it;
    if (reached_fixed_point(guess, st.next)) {
      if (lane == 0) repeat_iterate(a, it, unit, st.next);
      break;
    }
    guess = st.next;
  }
  if (lane == 0) publish_moving(a.moving, moved);
}

// ---- long contiguous units (1024 <= len <= 8192): one workgroup per unit, lanes own 16 elements ----
//
// Where the time went in octav_kernel on rows of 4096 weights (profiles/r01_octav_iterations...):
// with the guess at 0 each of the two masks selects every other element at random, a row has
// ~1000 runs per mask, and the running total `acc = acc + pairwise(run)` is a chain of ~2000
// dependent additions per row that ONE wave walked 64 elements at a time, with a scalar bit scan,
// DPP rounds and a ds_permute per batch (328 us for one iteration over 4096 rows), at one or two
// waves per SIMD because a 16 KB row sits behind every wave.
//
// Here a 256-thread workgroup owns a row (staged once in LDS for all iterations) and the work is
// split by what can be parallel:
//   masks   every wave compares its 64-element batches; thread t keeps the 16 mask bits of "its"
//           piece, elements [16 t, 16 t + 16). When the guess did not decrease (it never does after
//           the first update on real weights) the new selection is a subset of the old one, so
//           every thread just re-tests the set bits of its own old word: the cost follows the
//           number of selected elements;
//   runs    every thread walks the runs that START in its piece (bit scans on its own word; a run
//           that reaches the end of the piece continues through the following words, read from
//           LDS, up to NumPy's 8192-element chunk boundary) and sums each with NumPy's pairwise
//           scheme from LDS: left to right below 8 elements, eight strided accumulators up to 128,
//           runs longer than that (dense rows) by a whole wave. The sums land, in run order, in a
//           list in LDS (a workgroup prefix sum over the per-thread run counts gives the slots);
//   chain   acc = acc + R_j over the list: the only serial part. The positive and the negative
//           mask's chains run on two different waves at the same time (one wave issues one
//           addition per four cycles however the operands arrive); the lists are read with
//           broadcast 16-byte LDS loads, 32 entries ahead of the additions.
// The row is stored with one float of padding per 16 (thread-owned pieces start on distinct banks).
// (round 6: from 129 elements on -- 1024 until then. Units of 384 / 640 / 768 / 896 elements, the rows of small transformers'
// projections, took the one-wave-per-unit kernel: 0.63 ms per 2^24 elements against 0.37 / 0.26 / 0.23 / 0.21 ms here; 512
// took the groups kernel: 0.32 against 0.29 ms; 144 / 200 / 250: 0.78 / 0.72 / 0.66 against 0.75 / 0.58 / 0.48 ms. Below 128
// the one-wave kernel is the faster of the two; 32 / 64 / 128 / 256 elements are a lane's: octav_unit_lanes_kernel.)
#ifndef MI355Q_ROWS_MIN_LEN
#define MI355Q_ROWS_MIN_LEN 129
#endif
constexpr int kRowsMinLen = MI355Q_ROWS_MIN_LEN, kRowsMaxLen = 2 * kChunk;   // up to 4 pieces per thread
constexpr int kPiece = 16;

__device__ __forceinline__ int pidx(int e) { return e + (e >> 4); }


// Wave-wide integer prefix sum on the DPP network (no LDS round trips): Hillis-Steele inside each
// row of 16 lanes (row_shr 1, 2, 4, 8; lanes without a source add 0), then lane 15 of rows 0 / 2
// into rows 1 / 3 (row_bcast:15) and lane 31 into rows 2 and 3 (row_bcast:31).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add(int x) {
  return x + __builtin_amdgcn_update_dpp(0, x, CTRL, ROW_MASK, 0xF, true);
}

__device__ __forceinline__ int wave_incl_scan(int x) {
  x = dpp_add<0x111, 0xF>(x);
  x = dpp_add<0x112, 0xF>(x);
  x = dpp_add<0x114, 0xF>(x);
  x = dpp_add<0x118, 0xF>(x);
  x = x + __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
  x = x + __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
  return x;
}

// NumPy's pairwise sum of row[e0 .. e0 + n) (padded addressing), computed by ONE lane.
// n <= 128 (longer runs go to pairwise_wave below).
__device__ __forceinline__ float pairwise_lane(const float* row, int e0, int n) {
  if (n < 8) {
    float res = 0.f;
    for (int i = 0; i < n; ++i) res = res + row[pidx(e0 + i)];
    return res;
  }
  float r[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) r[k] = row[pidx(e0 + k)];
  const int full = n & ~7;
  for (int i = 8; i < full; i += 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = r[k] + row[pidx(e0 + i + k)];
  }
  float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (int i = full; i < n; ++i) res = res + row[pidx(e0 + i)];
  return res;
}

// The same for n > 128, by a whole wave (wave-uniform arguments): lanes 0..7 are the eight
// accumulators of a leaf, the recursion splits as NumPy does.
__device__ __forceinline__ float leaf_wave(const float* row, int e0, int m, int lane) {
  const int full = m & ~7;
  float r = 0.f;
  if (lane < 8) {
    r = row[pidx(e0 + lane)];
    for (int i = 8; i < full; i += 8) r = r + row[pidx(e0 + i + lane)];
  }
  const float r0 = lane_bcast(r, 0), r1 = lane_bcast(r, 1), r2 = lane_bcast(r, 2),
              r3 = lane_bcast(r, 3), r4 = lane_bcast(r, 4), r5 = lane_bcast(r, 5),
              r6 = lane_bcast(r, 6), r7 = lane_bcast(r, 7);
  float res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (int i = full; i < m; ++i) res = res + row[pidx(e0 + i)];
  return res;
}

template <int DEPTH>
__device__ __noinline__ float pairwise_wave(const float* row, int e0, int n, int lane) {
  if (n <= 128) return leaf_wave(row, e0, n, lane);   // (n >= 8 here: halves of n > 128 are >= 64)
  if constexpr (DEPTH > 0) {
    int n2 = n / 2;
    n2 -= n2 % 8;
    const float left = pairwise_wave<DEPTH - 1>(row, e0, n2, lane);
    const float right = pairwise_wave<DEPTH - 1>(row, e0 + n2, n - n2, lane);
    return left + right;
  } else {
    return leaf_wave(row, e0, n, lane);  // unreachable for n <= 8192
  }
}

template <int W>
struct RowsShared {      // small per-workgroup exchange area (in front of the row in LDS)
  int wave_runs[2][4][W];   // [mask][slot][wave]: runs starting in that wave's pieces
  int wave_count[2][W];     // selected elements per wave
  int wave_changed[W];      // some word of the wave differs from the previous iteration's
  float sum[2];             // the two chain totals
  float wave_amax[W];       // largest |x| of the wave's pieces (written once, area 0)
};
// floats per exchange area (there are two, by iteration parity)
constexpr int rows_xchg_floats(int threads) { return threads <= 256 ? 64 : 256; }

// acc = (..((acc + R0) + R1)..) over the lanes set in `ends`, in lane order (R = that lane's `partial`).
// Few ends: a scalar loop over the set bits. Otherwise the run totals are compacted to lanes 0..k-1
// (one ds_permute; the other lanes park theirs above k), then every lane repeats x = left
// neighbour's x + R (lane 0's neighbour is acc): lane j is final after j + 1 rounds and stays
// final, so k rounds (rounded up to the unroll) leave the total in lane k - 1.
__device__ __forceinline__ float chain_ends(float acc, float partial, unsigned long long ends, int lane) {
  const int k = __builtin_popcountll(ends);
  if (k <= 3) {
    while (ends != 0) {
      acc = acc + lane_bcast(partial, __builtin_ctzll(ends));
      ends &= ends - 1ull;
    }
    return acc;
  }
  const bool is_end = ((ends >> lane) & 1ull) != 0;
  const int rank = __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(ends >> 32),
                                             __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(ends), 0));
  const int dest = is_end ? rank : k + lane - rank;
  const float r = __int_as_float(__builtin_amdgcn_ds_permute(dest << 2, __float_as_int(partial)));
  float x = r;
  for (int t = 0; t < k; t += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float left = __int_as_float(__builtin_amdgcn_update_dpp(
          __float_as_int(acc), __float_as_int(x), 0x138, 0xF, 0xF, false));
      x = left + r;
    }
  }
  return lane_bcast(x, k - 1);
}

// One mask of a sparsely selected row, by ONE wave. `cv` / `cp` list the n elements that passed an
// earlier, lower guess (values and positions in the row, ascending): while the guess keeps
// growing, what it selects is a subset of them, so an iteration costs what it selects instead of a
// pass over the row. A run is a stretch of selected candidates at consecutive positions inside
// one 8192-chunk; lane l holds candidate base + l and its successor, so the link "my successor
// continues my run" needs no neighbour exchange. Runs shorter than 8 are summed left to right
// from +0.0 (NumPy's n < 8 loop) by rounds of "left neighbour's partial sum + mine", the run
// totals join the running total in order (chain_ends). A run of 8 or more candidates needs the
// eight-accumulator scheme: the caller is told (false) and repeats the iteration the dense way.
template <bool NEG>
__device__ __forceinline__ bool sparse_mask_sum(float* cv, unsigned short* cp, int n, float thr, float keep, int lane,
                                                float* sum_out, int* count_out, int* kept_out) {
  float acc = 0.f;
  int count = 0;
  int kept = 0;                 // the list shrinks as the guess grows: candidates beyond `keep` move to the front
  bool carry_link = false;      // lane 0 continues the run the previous batch ended with
  float carry_partial = 0.f;
  int carry_len = 0;
  for (int base = 0; base < n; base += kWave) {
    const int i = base + lane;
    const bool in = i < n, in1 = i + 1 < n;
    const float v = in ? cv[i] : 0.f, vn = in1 ? cv[i + 1] : 0.f;
    const int p = in ? static_cast<int>(cp[i]) : -1, pn = in1 ? static_cast<int>(cp[i + 1]) : -3;
    const bool sel = in && (NEG ? v <= thr : v >= thr);
    const bool seln = in1 && (NEG ? vn <= thr : vn >= thr);
    const bool linkn = sel && seln && pn == p + 1 && (pn & (kChunk - 1)) != 0;
    const unsigned long long S = __ballot(sel), Ln = __ballot(linkn);
    count += __builtin_popcountll(S);
    {
      // (in place: everything this batch reads is in registers, and what is kept lands at or below
      // the batch's own start)
      const bool stay = in && (NEG ? v <= keep : v >= keep);
      const unsigned long long K = __ballot(stay);
      const int at = kept + __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(K >> 32),
                                                     __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(K), 0));
      if (stay) { cv[at] = v; cp[at] = static_cast<unsigned short>(p); }
      kept += __builtin_popcountll(K);
    }
    const unsigned long long L = (Ln << 1) | (carry_link ? 1ull : 0ull);   // bit l: lane l continues lane l - 1's run
    unsigned long long c7 = L & (L << 1);
    c7 &= c7 << 2;
    c7 &= c7 << 3;                                                         // seven links in a row: a run of 8
    const int lead = carry_link ? __builtin_ctzll(~L) : 0;                 // elements of the carried run in this batch
    if (c7 != 0 || carry_len + lead >= 8) return false;
    float partial = sel ? v : 0.f;
    if (lane == 0 && carry_link) partial = carry_partial + v;
    const bool cont = lane != 0 && ((L >> lane) & 1ull) != 0;
    for (unsigned long long c = L & ~1ull; c != 0; c &= c << 1) {
      const float left = wave_shr1(partial);
      if (cont) partial = left + v;
    }
    const unsigned long long ends = S & ~Ln;
    if (ends != 0) acc = chain_ends(acc, partial, ends, lane);
    carry_link = (Ln >> 63) != 0;
    if (carry_link) {     // the open run started in this batch (fewer than seven links end at lane 63)
      carry_partial = lane_bcast(partial, 63);
      carry_len = __builtin_clzll(~L) + 1;
    }
  }
  *sum_out = acc;
  *count_out = count;
  *kept_out = kept;
  return true;
}

// One mask of one piece: `word` = the 16 selection bits of elements x[0..16) (element e0 = 16 pc),
// `carry` = the element before the piece is selected too (same chunk). Writes the sums of the runs
// that START in this piece to list[j0 ...] in order.
//   * runs shorter than 8 that end inside the piece: a fixed 16-step pass over the registers --
//     cur = cur + (selected ? x : +0.0), flushed to the list where a run ends (no data-dependent
//     control flow; a step that ends no run stores to a per-thread dummy slot);
//   * a run that reaches the piece's end continues through the following words (LDS); shorter
//     than 8 in total: the left-to-right chain simply goes on over the row in LDS;
//   * runs of 8 .. 128 elements: eight strided accumulators (pairwise_lane, from LDS);
//   * longer runs (dense rows): reported back, summed by the whole wave afterwards.
__device__ __forceinline__ void piece_runs(const float (&x)[kPiece], unsigned word, unsigned carry, int pc,
                                           int npieces, const unsigned short* words, const float* row,
                                           float* list, float* dummy, int j0, int* long_e0, int* long_n,
                                           int* long_j, int qmask = kChunk / kPiece - 1) {
  // leading elements that continue a run started in an earlier piece are not this thread's business
  const unsigned lead = carry ? ((word + 1u) & ~word) - 1u : 0u;   // the low run of ones (if bit 0 is set)
  const unsigned w = word & ~lead & 0xFFFFu;
  const unsigned ends = w & ~(w >> 1) & 0x7FFFu;   // runs ending inside the piece (bit 15 = open end)
  float cur = 0.f;
  int j = j0;
#pragma unroll
  for (int i = 0; i < kPiece; ++i) {
    const int sel = static_cast<int>(w << (31 - i)) >> 31;      // 0 / -1
    cur = cur + __int_as_float(__float_as_int(x[i]) & sel);
    if (i < kPiece - 1) {
      const int fin = static_cast<int>(ends << (31 - i)) >> 31;
      float* dst = fin ? list + j : dummy;
      *dst = cur;
      j -= fin;
      cur = __int_as_float(__float_as_int(cur) & ~fin);
    }
  }
  // runs of 8+ elements that lie inside the piece (rare): redo them with the eight accumulators
  unsigned m8 = w & (w >> 1); m8 &= m8 >> 2; m8 &= m8 >> 4;      // bit i: ones at i .. i+7
  m8 &= ~(m8 << 1);                                              // ... and i is where they start
  const unsigned starts = w & ~(w << 1);
  while (m8 != 0) {
    const int i = __builtin_ctz(m8);
    m8 &= m8 - 1u;
    const int n = __builtin_ctz(~(w >> i));
    if (i + n < kPiece)    // (a run that reaches the end is handled below)
      list[j0 + __builtin_popcount(starts & ((1u << i) - 1u))] = pairwise_lane(row, kPiece * pc + i, n);
  }
  if (w >> (kPiece - 1)) {   // the last run is open: follow it
    const unsigned zeros = ~w & 0xFFFFu;
    const int i = zeros ? 32 - __builtin_clz(zeros) : 0;     // its first bit (0 when w is all ones)
    int n = kPiece - i;
    int q = pc + 1;
    while (q < npieces && (q & qmask) != 0) {   // (a run ends where NumPy's chunk -- or the unit -- does)
      const unsigned nx = words[q];
      if (nx == 0xFFFFu) { n += kPiece; ++q; continue; }
      n += __builtin_ctz(~nx);
      break;
    }
    const int e0 = kPiece * pc + i;
    float res = cur;
    if (n < 8) {
      for (int k = kPiece - i; k < n; ++k) res = res + row[pidx(e0 + k)];
    } else if (n <= 128) {
      res = pairwise_lane(row, e0, n);
    } else {
      *long_e0 = e0; *long_n = n; *long_j = j;
    }
    list[j] = res;
  }
}

// The same for a sparsely selected piece (late iterations select ~1 % of a row): a loop over
// the runs that start here instead of the fixed pass over all 16 elements -- the cost follows the
// number of runs (the caller takes this form when no lane of the wave has more than a few).
__device__ __forceinline__ void piece_runs_sparse(unsigned word, unsigned carry, int pc, int npieces,
                                                  const unsigned short* words, const float* row, float* list,
                                                  int j0, int* long_e0, int* long_n, int* long_j,
                                                  int qmask = kChunk / kPiece - 1) {
  const unsigned lead = carry ? ((word + 1u) & ~word) - 1u : 0u;
  const unsigned w = word & ~lead & 0xFFFFu;
  unsigned st = w & ~(w << 1);
  int j = j0;
  while (st != 0) {
    const int i = __builtin_ctz(st);
    st &= st - 1u;
    int n = __builtin_ctz(~(w >> i));            // bits above 15 - i read as "run ended"
    if (i + n == kPiece) {                       // reaches the end of the piece: follow it
      int q = pc + 1;
      while (q < npieces && (q & qmask) != 0) {
        const unsigned nx = words[q];
        if (nx == 0xFFFFu) { n += kPiece; ++q; continue; }
        n += __builtin_ctz(~nx);
        break;
      }
    }
    const int e0 = kPiece * pc + i;
    float res = 0.f;
    if (n == 1) {
      res = 0.f + row[pidx(e0)];
    } else if (n <= 128) {
      res = pairwise_lane(row, e0, n);
    } else {
      *long_e0 = e0; *long_n = n; *long_j = j;
    }
    list[j] = res;
    ++j;
  }
}

// THREADS = 64 / 128 / 256: rows of up to 1024 / 2048 elements leave half or three quarters of a
// 256-thread workgroup without a piece, so they get smaller workgroups (more rows per CU).
// THREADS = 512 / 1024 (one piece per thread) for rows of up to 8192 / 16384 elements: with 256
// threads and two to four pieces each, such a row (70 - 140 KB of LDS) left its CU with one wave
// per SIMD walking four pieces one after the other.
// 16-byte list loads the chain wave keeps in flight per half step: 8 (32 entries ahead) cost the whole workgroup 64
// VGPRs and left 26 - 29 spilled at the 128 the occupancy allows; 4 leave 4 - 8 spilled and are 2 - 4 % faster; 2 are slower
#ifndef MI355Q_OCTAV_CHAIN_Q
#define MI355Q_OCTAV_CHAIN_Q 4
#endif
template <int SLOTS, int THREADS>
__global__ __launch_bounds__(THREADS, THREADS >= 512 ? 4 : (SLOTS == 1 ? 4 : (SLOTS == 2 ? 2 : 1))) void octav_rows_kernel(OctavArgs a) {
  constexpr int kRowsThreads = THREADS, kWaves = THREADS / kWave;
  constexpr int kWavesAlloc = kWaves < 4 ? 4 : kWaves;      // (entries of absent waves stay 0)
  constexpr int kX = rows_xchg_floats(THREADS);
  using Shared = RowsShared<kWavesAlloc>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const long long unit = blockIdx.x;
  const int len = a.len;
  const int npieces = (len + kPiece - 1) / kPiece;
  static_assert(sizeof(Shared) <= kX * sizeof(float), "exchange area");
  float* dummy = smem + 2 * kX + tid;                      // (two exchange areas in front)
  float* row = smem + 2 * kX + kRowsThreads;
  const int row_floats = (pidx(len) + 4) & ~3;
  float* list_pos = row + row_floats;                       // run sums of the two masks, in run order
  const int cap = ((len / 2 + 2 + 63) & ~63) + 64;         // (+ one per 8192-chunk) + the chain's read-ahead
  float* list_neg = list_pos + cap;
  unsigned short* words_pos = reinterpret_cast<unsigned short*>(list_neg + cap);
  unsigned short* words_neg = words_pos + ((npieces + 1) & ~1);

  for (int i = tid; i < 2 * kX; i += THREADS) smem[i] = 0.f;   // both exchange areas: waves this workgroup does not have count as 0
  __syncthreads();
  // ---- stage the unit: every thread keeps its pieces in registers for all iterations (one HBM
  // read), the row also goes to LDS for the runs that leave a piece
  const float qnan = __builtin_nanf("");
  float x[SLOTS][kPiece];
  float before[SLOTS];      // the element in front of each piece (same chunk), else NaN
  {
    const float* g = a.x + unit * len;
    const bool al = (reinterpret_cast<uintptr_t>(g) & 15) == 0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int e0 = kPiece * (tid + kRowsThreads * s);
      if (al && e0 + kPiece <= len) {
        const float4* g4 = reinterpret_cast<const float4*>(g + e0);
        const float4 v0 = g4[0], v1 = g4[1], v2 = g4[2], v3 = g4[3];
        x[s][0] = v0.x; x[s][1] = v0.y; x[s][2] = v0.z; x[s][3] = v0.w;
        x[s][4] = v1.x; x[s][5] = v1.y; x[s][6] = v1.z; x[s][7] = v1.w;
        x[s][8] = v2.x; x[s][9] = v2.y; x[s][10] = v2.z; x[s][11] = v2.w;
        x[s][12] = v3.x; x[s][13] = v3.y; x[s][14] = v3.z; x[s][15] = v3.w;
      } else {
#pragma unroll
        for (int i = 0; i < kPiece; ++i) x[s][i] = e0 + i < len ? g[e0 + i] : qnan;
      }
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int e0 = kPiece * (tid + kRowsThreads * s);
      const int p = pidx(e0);
#pragma unroll
      for (int i = 0; i < kPiece; ++i)
        if (e0 + i < len) row[p + i] = x[s][i];
    }
  }
  {
    // largest |x| of the row (NaN ignored: a NaN is never selected): a guess above it selects nothing
    float am = 0.f;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
#pragma unroll
      for (int i = 0; i < kPiece; ++i) am = fmaxf(am, fabsf(x[s][i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off));
    if (lane == 0) reinterpret_cast<Shared*>(smem)->wave_amax[wave] = am;
  }
  __syncthreads();
  float row_amax = 0.f;
#pragma unroll
  for (int k = 0; k < kWavesAlloc; ++k) row_amax = fmaxf(row_amax, reinterpret_cast<Shared*>(smem)->wave_amax[k]);
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int pc = tid + kRowsThreads * s;
    before[s] = (pc > 0 && pc < npieces && (pc & (kChunk / kPiece - 1)) != 0) ? row[pidx(kPiece * pc - 1)] : qnan;
  }

  unsigned wp[SLOTS], wn[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) wp[s] = wn[s] = 0u;
  float guess = 1.0f;
  float pos_sum = 0.f, neg_sum = 0.f;
  int cp = 0, cn = 0;
  unsigned long long moved = 0;
  // the serial chain runs on one wave per workgroup: spread it over the SIMDs of the CU
  // (workgroups u, u + 256, u + 512, ... tend to be co-resident)
  const int chain_wave = static_cast<int>((unit + (unit >> 8)) % kWaves);
  // late iterations select a few per cent of a row: the candidates are listed once and a single wave
  // per row (octav_tail_kernel: no row in LDS, no barriers, dozens of rows per CU) finishes the row
  constexpr bool kCanHandOver = SLOTS == 1;
  const int cand_cap = a.tail_cap;
  bool force = true, handed_over = false;
  int nit = 0;      // iterations that exchanged something (the exchange areas alternate with it)
  int scan_p = 0, scan_n = 0;   // the last dense iteration's inclusive scans of selected elements inside the wave
  for (int it = 0; it < a.max_iter; ++it) {
#if !defined(MI355Q_OCTAV_HOISTED)
    // The per-lane LDS addresses of an iteration (this thread's mask words, its dummy slot, its wave's exchange slots)
    // are functions of the thread index alone: hoisted out of the loop they are eight VGPRs that live across every
    // iteration -- and at the 128 VGPRs four rows per CU allow, eight scratch dwords per lane (33 MB of dirty lines
    // per 4096 x 4096 call, profiles/r05_octav_exact_writes.txt). Laundering the index makes them this iteration's
    // values: a dozen integer instructions per iteration instead, no scratch in any instantiation, and 4096 x 4096
    // went from 187 to 175 us (profiles/r06_octav_remat.txt; -DMI355Q_OCTAV_HOISTED restores the hoisted form).
    int tid_now = threadIdx.x;
    asm volatile("" : "+v"(tid_now));
    const int tid = tid_now, lane = tid_now & (kWave - 1), wave = tid_now >> 6;
    float* dummy = smem + 2 * kX + tid;
#endif
    const float hi = guess, lo = -guess;
    bool have_sums = false;
    if (guess > row_amax) {
      // nothing reaches the guess (the reference's first guess 1.0 on ordinary weights): empty masks
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) wp[s] = wn[s] = 0u;
      pos_sum = neg_sum = 0.f;
      cp = cn = 0;
      force = true;
      have_sums = true;
    }
    if (!have_sums) {
    // (two exchange areas, alternating: an iteration whose masks did not change has only
    // one barrier, so a fast wave may already publish the next iteration's counts while a slow one
    // still reads this one's)
    Shared* sh = reinterpret_cast<Shared*>(smem + kX * (nit & 1));
    ++nit;
    // ---- masks of the thread's own pieces, from registers
    unsigned changed = 0;
    unsigned sp[SLOTS], sn[SLOTS];   // run starts
    int pre_p[SLOTS], pre_n[SLOTS];
    int tp = 0, tn = 0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      unsigned np_ = 0, nn_ = 0;
#pragma unroll
      for (int i = 0; i < kPiece; ++i) {
        np_ |= x[s][i] >= hi ? 1u << i : 0u;
        nn_ |= x[s][i] <= lo ? 1u << i : 0u;
      }
      changed |= (np_ ^ wp[s]) | (nn_ ^ wn[s]);
      wp[s] = np_; wn[s] = nn_;
      const int pc = tid + kRowsThreads * s;
      if (pc < npieces) {
        words_pos[pc] = static_cast<unsigned short>(np_);
        words_neg[pc] = static_cast<unsigned short>(nn_);
      }
      tp += __builtin_popcount(np_);
      tn += __builtin_popcount(nn_);
      const unsigned cpos = before[s] >= hi ? 1u : 0u, cneg = before[s] <= lo ? 1u : 0u;
      sp[s] = np_ & ~((np_ << 1) | cpos) & 0xFFFFu;
      sn[s] = nn_ & ~((nn_ << 1) | cneg) & 0xFFFFu;
      const int ip = wave_incl_scan(__builtin_popcount(sp[s])), in = wave_incl_scan(__builtin_popcount(sn[s]));
      pre_p[s] = ip - __builtin_popcount(sp[s]);
      pre_n[s] = in - __builtin_popcount(sn[s]);
      if (lane == kWave - 1) { sh->wave_runs[0][s][wave] = ip; sh->wave_runs[1][s][wave] = in; }
    }
    tp = wave_incl_scan(tp);
    tn = wave_incl_scan(tn);
    scan_p = tp; scan_n = tn;
    const bool wave_changed = __ballot(changed != 0) != 0;
    if (lane == kWave - 1) {
      sh->wave_count[0][wave] = tp; sh->wave_count[1][wave] = tn;
      sh->wave_changed[wave] = wave_changed ? 1 : 0;
    }
    __syncthreads();
    bool any_changed;
    if constexpr (kWavesAlloc > 4) {
      any_changed = __ballot(lane < kWavesAlloc && sh->wave_changed[lane] != 0) != 0;
    } else {
      any_changed = (sh->wave_changed[0] | sh->wave_changed[1] | sh->wave_changed[2] | sh->wave_changed[3]) != 0;
    }
    if (any_changed || force) {
      force = false;
      // (the same masks give the same sums and counts: late iterations mostly skip all of this)
      int npos = 0, nneg = 0;
      if constexpr (kWavesAlloc > 4) {
        // many waves, one slot: lane k holds wave k's figures, one scan per figure on the DPP network
        static_assert(SLOTS == 1, "wide workgroups own one piece per thread");
        const int uw = __builtin_amdgcn_readfirstlane(wave);
        const bool in = lane < kWavesAlloc;
        const int ip = wave_incl_scan(in ? sh->wave_runs[0][0][lane] : 0);
        const int inn = wave_incl_scan(in ? sh->wave_runs[1][0][lane] : 0);
        const int icp = wave_incl_scan(in ? sh->wave_count[0][lane] : 0);
        const int icn = wave_incl_scan(in ? sh->wave_count[1][lane] : 0);
        npos = __builtin_amdgcn_readlane(ip, kWavesAlloc - 1);
        nneg = __builtin_amdgcn_readlane(inn, kWavesAlloc - 1);
        cp = __builtin_amdgcn_readlane(icp, kWavesAlloc - 1);
        cn = __builtin_amdgcn_readlane(icn, kWavesAlloc - 1);
        if (uw > 0) {
          pre_p[0] += __builtin_amdgcn_readlane(ip, uw - 1);
          pre_n[0] += __builtin_amdgcn_readlane(inn, uw - 1);
        }
      } else {
        cp = cn = 0;
#pragma unroll
        for (int k = 0; k < kWavesAlloc; ++k) { cp += sh->wave_count[0][k]; cn += sh->wave_count[1][k]; }
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
          int bp = npos, bn = nneg;   // runs of earlier slots, then of earlier waves of this slot
#pragma unroll
          for (int k = 0; k < kWavesAlloc; ++k) {
            const int c0 = sh->wave_runs[0][s][k], c1 = sh->wave_runs[1][s][k];
            if (k < wave) { bp += c0; bn += c1; }
            npos += c0; nneg += c1;
          }
          pre_p[s] += bp; pre_n[s] += bn;
        }
      }
      // ---- run sums, in run order
      int long_e0[2 * SLOTS], long_n[2 * SLOTS], long_j[2 * SLOTS];
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        const int pc = tid + kRowsThreads * s;
        long_n[2 * s] = long_n[2 * s + 1] = 0;
        long_e0[2 * s] = long_e0[2 * s + 1] = long_j[2 * s] = long_j[2 * s + 1] = 0;
        // (wave-uniform choice per mask: the fixed 16-step pass costs ~240 instructions whatever
        // is selected, the run loop ~30 per run of the busiest lane)
        const bool sparse_p = __ballot(__builtin_popcount(sp[s]) > 3) == 0;
        const bool sparse_n = __ballot(__builtin_popcount(sn[s]) > 3) == 0;
        if (pc < npieces) {
          if (sparse_p)
            piece_runs_sparse(wp[s], before[s] >= hi ? 1u : 0u, pc, npieces, words_pos, row, list_pos,
                              pre_p[s], &long_e0[2 * s], &long_n[2 * s], &long_j[2 * s]);
          else
            piece_runs(x[s], wp[s], before[s] >= hi ? 1u : 0u, pc, npieces, words_pos, row, list_pos, dummy,
                       pre_p[s], &long_e0[2 * s], &long_n[2 * s], &long_j[2 * s]);
          if (sparse_n)
            piece_runs_sparse(wn[s], before[s] <= lo ? 1u : 0u, pc, npieces, words_neg, row, list_neg,
                              pre_n[s], &long_e0[2 * s + 1], &long_n[2 * s + 1], &long_j[2 * s + 1]);
          else
            piece_runs(x[s], wn[s], before[s] <= lo ? 1u : 0u, pc, npieces, words_neg, row, list_neg, dummy,
                       pre_n[s], &long_e0[2 * s + 1], &long_n[2 * s + 1], &long_j[2 * s + 1]);
        }
      }
      // dense rows: runs longer than 128 elements, one at a time by the wave that found them
#pragma unroll
      for (int k = 0; k < 2 * SLOTS; ++k) {
        unsigned long long pending = __ballot(long_n[k] > 0);
        while (pending != 0) {
          const int src = __builtin_ctzll(pending);
          pending &= pending - 1ull;
          const int e0 = __builtin_amdgcn_readlane(long_e0[k], src);
          const int n = __builtin_amdgcn_readlane(long_n[k], src);
          const int j = __builtin_amdgcn_readlane(long_j[k], src);
          const float res = pairwise_wave<8>(row, e0, n, lane);
          if (lane == 0) ((k & 1) ? list_neg : list_pos)[j] = res;
        }
      }
      // whole blocks of 64 entries, both lists padded to the longer one: no tail code in the chain
      // (adding +0.0 padding is exact: a running total that started at +0.0 is never -0.0)
      const int kp = (npos + 63) & ~63, kn = (nneg + 63) & ~63;
      const int kmax = kp > kn ? kp : kn;
      for (int j = npos + tid; j < kmax; j += kRowsThreads) list_pos[j] = 0.f;
      for (int j = nneg + tid; j < kmax; j += kRowsThreads) list_neg[j] = 0.f;
      __syncthreads();
      // ---- the chains: acc = acc + R_j in run order. A wave issues one dependent addition per four
      // cycles however many lanes take part, so ONE wave walks both lists at once: even lanes the
      // positive mask's, odd lanes the negative's (per-lane LDS addresses, two distinct ones per
      // load). Software pipelined by hand: the eight 16-byte loads of the NEXT 32 entries are issued,
      // then the 32 dependent additions of the current ones run while those loads are in flight
      // (the scheduling barriers keep the compiler from sinking the loads back behind the additions;
      // it still places the s_waitcnt itself). The lists are over-allocated by one block: what the
      // last read-ahead fetches is never added.
      if (wave == chain_wave) {
        const float4* l4 = reinterpret_cast<const float4*>((lane & 1) ? list_neg : list_pos);
        constexpr int Q = MI355Q_OCTAV_CHAIN_Q;      // 16-byte loads in flight per half step
        const int npair = kmax / (8 * Q);
        float acc = 0.f;
        float4 qa[Q], qb[Q];
#pragma unroll
        for (int k = 0; k < Q; ++k) qa[k] = l4[k];
        for (int pr = 0; pr < npair; ++pr) {
#pragma unroll
          for (int k = 0; k < Q; ++k) qb[k] = l4[(2 * pr + 1) * Q + k];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < Q; ++k) { acc = acc + qa[k].x; acc = acc + qa[k].y; acc = acc + qa[k].z; acc = acc + qa[k].w; }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < Q; ++k) qa[k] = l4[(2 * pr + 2) * Q + k];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < Q; ++k) { acc = acc + qb[k].x; acc = acc + qb[k].y; acc = acc + qb[k].z; acc = acc + qb[k].w; }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (lane < 2) sh->sum[lane] = acc;
      }
      __syncthreads();   // totals visible; the list may be rewritten by the next iteration
      pos_sum = sh->sum[0];
      neg_sum = sh->sum[1];
    }
    }  // dense body
    const OctavStep st = octav_step(guess, pos_sum, neg_sum, cp, cn, len, a.s, a.count_is_f64);
    if (tid == 0) a.hist[static_cast<long long>(it) * a.units + unit] = st.next;
    if (!st.close) moved |= 1ull << it;
    if (reached_fixed_point(guess, st.next)) {
      if (tid == 0) repeat_iterate(a, it, unit, st.next);
      break;
    }
    if constexpr (kCanHandOver) {
      if (a.tail != nullptr && !have_sums && it + 2 < a.max_iter && guess > 0.f && st.next >= guess && cp <= cand_cap && cn <= cand_cap) {
        // few elements selected and the guess growing: every later selection is a subset of this one.
        // List it (value, position; row order) for the tail. scan_p / scan_n are this iteration's
        // inclusive scans of the per-thread counts inside the wave, the exchange area has the waves' totals.
        const Shared* shc = reinterpret_cast<const Shared*>(smem + kX * ((nit - 1) & 1));
        int bp = scan_p - __builtin_popcount(wp[0]), bn = scan_n - __builtin_popcount(wn[0]);
        for (int k = 0; k < wave; ++k) { bp += shc->wave_count[0][k]; bn += shc->wave_count[1][k]; }
        // (each thread stores the few candidates of its own piece straight to global memory. Staging them
        // in LDS and copying whole lines out was measured: the partial-line stores here cost 4 x the row in
        // HBM traffic (profiles/r03_pmc_traffic.txt) but the kernel is bound by instruction issue, and the
        // two extra barriers + copy loops made it 10 % slower: 200 against 182 us at 4096 x 4096)
        float* vp = a.tail_values + (static_cast<long long>(unit) * 2) * cand_cap;
        float* vn = vp + cand_cap;
        unsigned short* pp = a.tail_pos + (static_cast<long long>(unit) * 2) * cand_cap;
        unsigned short* pn = pp + cand_cap;
        const int e0 = kPiece * tid;
#pragma unroll
        for (int i = 0; i < kPiece; ++i) {
          if ((wp[0] >> i) & 1u) { vp[bp] = x[0][i]; pp[bp] = static_cast<unsigned short>(e0 + i); ++bp; }
          if ((wn[0] >> i) & 1u) { vn[bn] = x[0][i]; pn[bn] = static_cast<unsigned short>(e0 + i); ++bn; }
        }
        if (tid == 0) {
          TailState ts;
          ts.next_it = it + 1; ts.n[0] = cp; ts.n[1] = cn; ts.guess = st.next; ts.cand_guess = guess;
          ts.moved_lo = static_cast<unsigned>(moved); ts.moved_hi = static_cast<unsigned>(moved >> 32); ts.pad = 0;
          a.tail[unit] = ts;
        }
        handed_over = true;
        break;
      }
    }
    guess = st.next;
  }
  if (tid == 0 && !handed_over) {
    publish_moving(a.moving, moved);
    if (a.tail != nullptr) a.tail[unit].next_it = 0;
  }
}

// The rest of a row's iterations on ONE wave: the candidates the row's workgroup listed are staged in
// LDS (a few KB) and re-tested against every new guess (sparse_mask_sum; the lists shrink as the
// guess grows). Should an iterate step below what the candidates passed, or a run of 8+ candidates
// show up, the wave falls back to scanning the row itself from global memory (the generic one-wave
// scheme of octav_kernel) for the iterations that are left -- correct whatever the data does.
__global__ __launch_bounds__(kWave) void octav_tail_kernel(OctavArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const long long unit = blockIdx.x;
  const int lane = threadIdx.x;
  const TailState ts = a.tail[unit];
  if (ts.next_it <= 0 || ts.next_it >= a.max_iter) return;
  const int cap = a.tail_cap, len = a.len;
  float* cv[2] = {smem, smem + cap};
  unsigned short* cp[2] = {reinterpret_cast<unsigned short*>(smem + 2 * cap), reinterpret_cast<unsigned short*>(smem + 2 * cap) + cap};
  int n[2] = {ts.n[0], ts.n[1]};
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const float* gv = a.tail_values + (unit * 2 + m) * cap;
    const unsigned short* gp = a.tail_pos + (unit * 2 + m) * cap;
    for (int i = lane; i < n[m]; i += kWave) { cv[m][i] = gv[i]; cp[m][i] = gp[i]; }
  }
  const float* u = a.x + unit * len;
  const float qnan = __builtin_nanf("");
  float guess = ts.guess, cand_guess = ts.cand_guess;
  unsigned long long moved = static_cast<unsigned long long>(ts.moved_lo) | (static_cast<unsigned long long>(ts.moved_hi) << 32);
  bool lists_ok = true;
  for (int it = ts.next_it; it < a.max_iter; ++it) {
    const float hi = guess, lo = -guess;
    float pos_sum = 0.f, neg_sum = 0.f;
    int cpos = 0, cneg = 0;
    bool have = false;
    if (lists_ok && guess >= cand_guess) {
      // candidates a little below the guess stay listed: an iterate that settles may step back by an ulp
      const float keep = hi * 0.998046875f;     // 1 - 2^-9
      int l0 = 0, l1 = 0;
      have = sparse_mask_sum<false>(cv[0], cp[0], n[0], hi, keep, lane, &pos_sum, &cpos, &l0) &&
             sparse_mask_sum<true>(cv[1], cp[1], n[1], lo, -keep, lane, &neg_sum, &cneg, &l1);
      if (have) { n[0] = l0; n[1] = l1; cand_guess = keep; }
    }
    if (!have) {
      lists_ok = false;          // (a failed pass leaves its list half compacted)
      RunSum pos, neg;
      for (int base = 0; base < len; base += kWave) {
        const int i = base + lane;
        const float v = i < len ? u[i] : qnan;
        const unsigned long long mp = __ballot(v >= hi), mn = __ballot(v <= lo);
        if ((mp | mn) == 0 && (pos.pend_len | neg.pend_len) == 0) continue;
        pos.feed(mp, base, v, u, lane);
        neg.feed(mn, base, v, u, lane);
      }
      pos.flush(u, lane);
      neg.flush(u, lane);
      pos_sum = pos.acc; neg_sum = neg.acc;
      cpos = static_cast<int>(pos.count); cneg = static_cast<int>(neg.count);
    }
    const OctavStep st = octav_step(guess, pos_sum, neg_sum, cpos, cneg, len, a.s, a.count_is_f64);
    if (lane == 0) a.hist[static_cast<long long>(it) * a.units + unit] = st.next;
    if (!st.close) moved |= 1ull << it;
    if (reached_fixed_point(guess, st.next)) {
      if (lane == 0) repeat_iterate(a, it, unit, st.next);
      break;
    }
    guess = st.next;
  }
  if (lane == 0) publish_moving(a.moving, moved);
}

// ---- short units (blockwise recipes: 32 .. 512 elements): a workgroup takes kGroupLen contiguous
// elements = 8 .. 128 whole units through the same machinery ----------------------------------
// One wave per 128-element unit (octav_kernel) spends ~400 cycles per unit, mask and iteration on
// ballots, scalar bit scans and lane broadcasts with most lanes idle: 0.43 ms for 4096 x 4096 in
// blocks of 128. Here thread t owns piece t (16 elements in registers) of the group, a unit is
// ppu = unit_len / 16 consecutive pieces, and everything that is per row in octav_rows_kernel is
// per unit: the guess (LDS, one float per unit), where a run may continue (not past the unit's last
// piece), the slice of the run-sum list a chain walks, the selected-element counts (differences of
// the wave's inclusive scan at the unit's ends: units never straddle waves). Thread u < units runs
// unit u's two short chains and its Newton update; the group leaves the loop when every unit has
// reached a fixed point.
constexpr int kGroupLen = 4096, kGroupThreads = kGroupLen / kPiece;

struct GroupsShared {          // per iteration parity, like RowsShared
  int wave_runs[2][4];         // [mask][wave]: runs starting in that wave's pieces
  int wave_changed[4];
};

__global__ __launch_bounds__(kGroupThreads, 2) void octav_groups_kernel(OctavArgs a, int ulen) {
  constexpr int kWaves = kGroupThreads / kWave;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const int ppu = ulen / kPiece, qmask = ppu - 1, upg = kGroupLen / ulen;   // pieces per unit (a power of two), units per group
  const long long unit0 = static_cast<long long>(blockIdx.x) * upg;
  constexpr int npieces = kGroupThreads;
  float* dummy = smem + 128 + tid;
  float* row = smem + 128 + kGroupThreads;
  constexpr int row_floats = (kGroupLen + (kGroupLen >> 4) + 4) & ~3;
  float* list_pos = row + row_floats;
  constexpr int cap = ((kGroupLen / 2 + 2 + 63) & ~63) + 64;
  float* list_neg = list_pos + cap;
  unsigned short* words_pos = reinterpret_cast<unsigned short*>(list_neg + cap);
  unsigned short* words_neg = words_pos + npieces;
  int* run_pre = reinterpret_cast<int*>(words_neg + npieces);   // [2][threads]: index of the thread's first run in the list
  int* cnt_incl = run_pre + 2 * kGroupThreads;                   // [2][threads]: selected elements up to and including the thread's piece (per wave)
  float* ug = reinterpret_cast<float*>(cnt_incl + 2 * kGroupThreads);   // [upg]: the units' current guesses
  int* totals = reinterpret_cast<int*>(ug + kGroupThreads);      // [2]: runs in the whole group

  for (int i = tid; i < 128; i += kGroupThreads) smem[i] = 0.f;
  if (tid < upg) ug[tid] = 1.0f;
  const float qnan = __builtin_nanf("");
  float x[kPiece];
  {
    const float4* g4 = reinterpret_cast<const float4*>(a.x + unit0 * ulen + kPiece * tid);   // (16-byte aligned: see the dispatch)
    const float4 v0 = g4[0], v1 = g4[1], v2 = g4[2], v3 = g4[3];
    x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w;
    x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
    x[8] = v2.x; x[9] = v2.y; x[10] = v2.z; x[11] = v2.w;
    x[12] = v3.x; x[13] = v3.y; x[14] = v3.z; x[15] = v3.w;
    const int p = pidx(kPiece * tid);
#pragma unroll
    for (int i = 0; i < kPiece; ++i) row[p + i] = x[i];
  }
  __syncthreads();
  const float before = (tid & qmask) != 0 ? row[pidx(kPiece * tid - 1)] : qnan;   // the element in front of the piece, same unit
  const int my_unit = tid / ppu;

  unsigned wp = 0u, wn = 0u;
  // thread u < upg carries unit u's state
  float pos_sum = 0.f, neg_sum = 0.f;
  int cp = 0, cn = 0;
  unsigned long long moved = 0;
  bool done = tid >= upg;
  for (int it = 0; it < a.max_iter; ++it) {
    const float hi = ug[my_unit], lo = -hi;
    GroupsShared* sh = reinterpret_cast<GroupsShared*>(smem + 64 * (it & 1));
    unsigned np_ = 0, nn_ = 0;
#pragma unroll
    for (int i = 0; i < kPiece; ++i) {
      np_ |= x[i] >= hi ? 1u << i : 0u;
      nn_ |= x[i] <= lo ? 1u << i : 0u;
    }
    const unsigned changed = (np_ ^ wp) | (nn_ ^ wn);
    wp = np_; wn = nn_;
    words_pos[tid] = static_cast<unsigned short>(np_);
    words_neg[tid] = static_cast<unsigned short>(nn_);
    const unsigned cpos = before >= hi ? 1u : 0u, cneg = before <= lo ? 1u : 0u;
    const unsigned sp = np_ & ~((np_ << 1) | cpos) & 0xFFFFu, sn = nn_ & ~((nn_ << 1) | cneg) & 0xFFFFu;   // run starts
    const int ip = wave_incl_scan(__builtin_popcount(sp)), in = wave_incl_scan(__builtin_popcount(sn));
    int pre_p = ip - __builtin_popcount(sp), pre_n = in - __builtin_popcount(sn);
    cnt_incl[tid] = wave_incl_scan(__builtin_popcount(np_));
    cnt_incl[kGroupThreads + tid] = wave_incl_scan(__builtin_popcount(nn_));
    const bool wave_changed = __ballot(changed != 0) != 0;
    if (lane == kWave - 1) {
      sh->wave_runs[0][wave] = ip; sh->wave_runs[1][wave] = in;
      sh->wave_changed[wave] = wave_changed ? 1 : 0;
    }
    __syncthreads();
    const bool any_changed = (sh->wave_changed[0] | sh->wave_changed[1] | sh->wave_changed[2] | sh->wave_changed[3]) != 0;
    if (any_changed || it == 0) {
      int npos = 0, nneg = 0;
#pragma unroll
      for (int k = 0; k < kWaves; ++k) {
        const int c0 = sh->wave_runs[0][k], c1 = sh->wave_runs[1][k];
        if (k < wave) { pre_p += c0; pre_n += c1; }
        npos += c0; nneg += c1;
      }
      run_pre[tid] = pre_p;
      run_pre[kGroupThreads + tid] = pre_n;
      if (tid == 0) { totals[0] = npos; totals[1] = nneg; }
      int long_e0[2] = {0, 0}, long_n[2] = {0, 0}, long_j[2] = {0, 0};
      const bool sparse_p = __ballot(__builtin_popcount(sp) > 3) == 0;
      const bool sparse_n = __ballot(__builtin_popcount(sn) > 3) == 0;
      if (sparse_p)
        piece_runs_sparse(wp, cpos, tid, npieces, words_pos, row, list_pos, pre_p, &long_e0[0], &long_n[0], &long_j[0], qmask);
      else
        piece_runs(x, wp, cpos, tid, npieces, words_pos, row, list_pos, dummy, pre_p, &long_e0[0], &long_n[0], &long_j[0], qmask);
      if (sparse_n)
        piece_runs_sparse(wn, cneg, tid, npieces, words_neg, row, list_neg, pre_n, &long_e0[1], &long_n[1], &long_j[1], qmask);
      else
        piece_runs(x, wn, cneg, tid, npieces, words_neg, row, list_neg, dummy, pre_n, &long_e0[1], &long_n[1], &long_j[1], qmask);
      // units of more than 128 elements only: runs longer than 128, one at a time by the wave that found them
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        unsigned long long pending = __ballot(long_n[k] > 0);
        while (pending != 0) {
          const int src = __builtin_ctzll(pending);
          pending &= pending - 1ull;
          const int e0 = __builtin_amdgcn_readlane(long_e0[k], src);
          const int n = __builtin_amdgcn_readlane(long_n[k], src);
          const int j = __builtin_amdgcn_readlane(long_j[k], src);
          const float res = pairwise_wave<8>(row, e0, n, lane);
          if (lane == 0) (k ? list_neg : list_pos)[j] = res;
        }
      }
      __syncthreads();
      if (!done) {
        // ---- unit tid: acc = acc + R_j over its slice of each list, its counts
        const int first = tid * ppu, last = first + ppu - 1;
        const bool at_wave_start = (first & (kWave - 1)) == 0;
        {
          const int j0 = run_pre[first], j1 = tid + 1 < upg ? run_pre[first + ppu] : totals[0];
          float acc = 0.f;
          for (int j = j0; j < j1; ++j) acc = acc + list_pos[j];
          pos_sum = acc;
          cp = cnt_incl[last] - (at_wave_start ? 0 : cnt_incl[first - 1]);
        }
        {
          const int j0 = run_pre[kGroupThreads + first], j1 = tid + 1 < upg ? run_pre[kGroupThreads + first + ppu] : totals[1];
          float acc = 0.f;
          for (int j = j0; j < j1; ++j) acc = acc + list_neg[j];
          neg_sum = acc;
          cn = cnt_incl[kGroupThreads + last] - (at_wave_start ? 0 : cnt_incl[kGroupThreads + first - 1]);
        }
      }
    }
    if (!done) {
      const float guess = ug[tid];
      const OctavStep st = octav_step(guess, pos_sum, neg_sum, cp, cn, ulen, a.s, a.count_is_f64);
      a.hist[static_cast<long long>(it) * a.units + unit0 + tid] = st.next;
      if (!st.close) moved |= 1ull << it;
      if (reached_fixed_point(guess, st.next)) {
        repeat_iterate(a, it, unit0 + tid, st.next);
        done = true;
      }
      ug[tid] = st.next;
    }
    if (__syncthreads_and(done ? 1 : 0)) break;   // (also: guesses visible, lists free for the next iteration)
  }
  if (tid < upg) publish_moving(a.moving, moved);
}

// ---- short units on LANES (the four blockwise granularities: 32 / 64 / 128 / 256 elements per unit) ----
//
// octav_groups_kernel above treats a stretch of 4096 elements like a row: pieces of 16 on threads, run sums listed in
// LDS through a workgroup prefix sum, chains on a few lanes, two barriers per iteration (~15 vector instructions per
// element and mask, 0.039 of one read at 4096 x 4096 in blocks of 128). A unit this short needs none of that: LANE l of
// a wave owns unit 64 b + l, keeps its elements in REGISTERS for all iterations (16-register tuples read with a
// wave-uniform index, so the walks are loops) and walks them left to right as NumPy's masked reduction does -- nothing is
// exchanged between lanes, no run sum is listed, there is no barrier. An iteration is one of
//   the fast step (lane_step_pair), valid while every run is shorter than 8: seq = 0 + a0 + a1 + ... is NumPy's n < 8
//       loop, and acc = acc + seq where the run ends. Branch-free: every step adds (selected ? +0.0 : seq) to acc --
//       adding +0.0 is exact, a total that started at +0.0 is never -0.0 -- and sets seq = selected ? seq + x : +0.0, both
//       masks side by side, selections as sign bits of x - g and (-x) - g (no compare, no scalar register);
//   the long-aware step (lane_step_pair_long) wherever a run of 8+ is met -- the tuple in which a run reaches 8 is walked
//       again from the state it began with (walk_unit) -- and throughout at guess 0, the second iterate of a weight
//       tensor, where half of every mask is selected: the same step, and where a run of 8+ ENDS, NumPy's eight-accumulator
//       leaf over that run (from memory; only the lanes concerned) takes the place of seq;
//   the step by compares (lane_step_exact) for units that hold a NaN or an infinity and for iterates that are not finite;
//   and, once every live unit of the wave selects a handful of elements with growing iterates, a walk over per-lane
//       candidate lists in LDS (walk_candidates) instead of the unit.
// A lane that reached its fixed point is masked off; the wave leaves when all have. Bit-identical to octav_kernel /
// octav_groups_kernel (tests/test_gpu_octav_unit_lanes.py); how it got here, measured: profiles/r06_octav_unit_lanes.txt.
__device__ __forceinline__ float unit_leaf(const float* a, int n) {   // NumPy's leaf, 8 <= n <= 128
  float r[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) r[k] = a[k];
  const int full = n & ~7;
  if (full == 8) {
    // runs of 8 .. 15 (nearly all there are): the up to seven elements behind the accumulators are asked for together with
    // them -- one round trip to memory, not one per element of the tail
    float t[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) t[k] = 8 + k < n ? a[8 + k] : 0.f;
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
    for (int k = 0; k < 7; ++k) res = 8 + k < n ? res + t[k] : res;
    return res;
  }
  for (int i = 8; i < full; i += 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = r[k] + a[i + k];
  }
  float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (int i = full; i < n; ++i) res = res + a[i];
  return res;
}

// NumPy's pairwise sum of a run of 8 .. 256 elements: a leaf up to 128, beyond that the two halves (the first rounded
// down to a multiple of 8), each a leaf -- a unit of 256 needs one level of the recursion, no more.
template <bool WIDE>      // WIDE: units of 256 (shorter units never see the second case: one branch less at every site)
__device__ __forceinline__ float unit_long_run(const float* a, int n) {
  if (!WIDE || n <= 128) return unit_leaf(a, n);
  int n2 = n / 2;
  n2 -= n2 % 8;
  const float left = unit_leaf(a, n2);
  return left + unit_leaf(a + n2, n - n2);
}

struct LaneMask {
  float acc, seq;
  int cnt;
};

// One element, one mask, any run: compares, the run length kept, NumPy's leaf where a run of 8+ ends (the exact walk;
// `here` = the element's address).
template <bool NEG, bool WIDE>
__device__ __forceinline__ void lane_step_exact(LaneMask& m, int& len, float v, float thr, const float* here) {
  const bool sel = NEG ? v <= thr : v >= thr;
  float add = sel ? 0.f : m.seq;
  if (!sel && len >= 8) add = unit_long_run<WIDE>(here - len, len);
  m.acc = m.acc + add;
  m.seq = sel ? m.seq + v : 0.f;
  len = sel ? len + 1 : 0;
  m.cnt += sel ? 1 : 0;
}

// The fast walk's state: both masks, no compare and no scalar register anywhere (a v_cmp result crosses to the scalar
// file and back before a v_cndmask can use it: with the running totals depending on every select, that round trip was
// most of a step's time). For a guess g > 0 and a finite x:   x >= g  <=>  x - g is not negative,   x <= -g  <=>
// (-x) - g is not negative, and neither difference can be -0.0 (x - x = +0.0), so "not selected" is the difference's
// sign bit smeared over the word (one subtraction, one shift) and the selects are bit operations.
// The run lengths of both masks share a register (16 bits each) and so do the counts of unselected elements.
struct LanePair {
  float acc_p, seq_p, acc_n, seq_n;
  unsigned len2, unsel2, long2;     // [15:0] positive mask, [31:16] negative mask; long2 = OR of every len2 seen
};

// (as an instruction the compiler cannot look into: it rewrites (d >> 31) & y as d < 0 ? y : 0 -- the compare and select
// through the scalar file this walk exists to avoid)
__device__ __forceinline__ unsigned sign_smear(float d) {
  unsigned m;
  asm("v_ashrrev_i32 %0, 31, %1" : "=v"(m) : "v"(d));
  return m;
}

// (the additions as single instructions too: seen as C++, the two masks' additions are paired into v_pk_add_f32, whose
// second operand must be a register PAIR holding x twice -- a copy of the whole unit, spilled)
__device__ __forceinline__ float add_f32(float a_, float b_) {
  float r;
  asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a_), "v"(b_));
  return r;
}

__device__ __forceinline__ void lane_step_pair(LanePair& s, float v, float g) {
  const unsigned np_ = sign_smear(v - g);       // all ones: NOT selected
  const unsigned nn_ = sign_smear((-v) - g);
  s.acc_p = add_f32(s.acc_p, __uint_as_float(__float_as_uint(s.seq_p) & np_));   // + (selected ? +0.0 : seq)
  s.acc_n = add_f32(s.acc_n, __uint_as_float(__float_as_uint(s.seq_n) & nn_));
  s.seq_p = __uint_as_float(__float_as_uint(add_f32(s.seq_p, v)) & ~np_);        // selected ? seq + x : +0.0
  s.seq_n = __uint_as_float(__float_as_uint(add_f32(s.seq_n, v)) & ~nn_);
  const unsigned both = (np_ & 0xFFFFu) | (nn_ & 0xFFFF0000u);
  s.len2 = (s.len2 + 0x00010001u) & ~both;
  asm("v_or_b32 %0, %1, %2" : "=v"(s.long2) : "v"(s.long2), "v"(s.len2));   // (s.long2 |= s.len2 as C++ is turned into a
  s.unsel2 += both & 0x00010001u;                                           //  tree over all the unit's run lengths, kept live and spilled)
}

// (the second bound is waves per SIMD: 1 / 2 / 3 / 4 = 512 / 256 / 168 / 128 registers)
//
// The unit lives in 32-float register TUPLES read with a wave-uniform index (s_set_gpr_idx_on): the walks are loops of
// four steps, not LEN unrolled steps. Unrolled, the fast walk was 18 KB of straight-line code that every wave of the
// chip entered at the same moment: its first execution cost 20-25 us of instruction-cache misses, more than the walk.
typedef float v16f __attribute__((ext_vector_type(16)));

// (eight named 16-register tuples, not an array of them and not four of 32: the array went through scratch on its way
// into registers, and 32-register tuples were spilled whole when a second one had to be placed)
struct UnitRegs {
  v16f t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11, t12, t13, t14, t15;      // (units of 256: all sixteen)
};

// (tuples are handed around BY VALUE: a reference to a member plus a run-time element index is an address computation, and the
// struct then lives in scratch)
template <int T>
__device__ __forceinline__ v16f unit_tuple(const UnitRegs& r) {
  if constexpr (T == 0) return r.t0;
  else if constexpr (T == 1) return r.t1;
  else if constexpr (T == 2) return r.t2;
  else if constexpr (T == 3) return r.t3;
  else if constexpr (T == 4) return r.t4;
  else if constexpr (T == 5) return r.t5;
  else if constexpr (T == 6) return r.t6;
  else if constexpr (T == 7) return r.t7;
  else if constexpr (T == 8) return r.t8;
  else if constexpr (T == 9) return r.t9;
  else if constexpr (T == 10) return r.t10;
  else if constexpr (T == 11) return r.t11;
  else if constexpr (T == 12) return r.t12;
  else if constexpr (T == 13) return r.t13;
  else if constexpr (T == 14) return r.t14;
  else return r.t15;
}

template <int T>
__device__ __forceinline__ void set_unit_tuple(UnitRegs& r, v16f t) {
  if constexpr (T == 0) r.t0 = t;
  else if constexpr (T == 1) r.t1 = t;
  else if constexpr (T == 2) r.t2 = t;
  else if constexpr (T == 3) r.t3 = t;
  else if constexpr (T == 4) r.t4 = t;
  else if constexpr (T == 5) r.t5 = t;
  else if constexpr (T == 6) r.t6 = t;
  else if constexpr (T == 7) r.t7 = t;
  else if constexpr (T == 8) r.t8 = t;
  else if constexpr (T == 9) r.t9 = t;
  else if constexpr (T == 10) r.t10 = t;
  else if constexpr (T == 11) r.t11 = t;
  else if constexpr (T == 12) r.t12 = t;
  else if constexpr (T == 13) r.t13 = t;
  else if constexpr (T == 14) r.t14 = t;
  else r.t15 = t;
}

__device__ __forceinline__ float tuple_element(v16f t, int jj) { return t[jj]; }

template <int J0, int N, typename F>
__device__ __forceinline__ void each_slot(const UnitRegs& r, F&& f) {   // f(value) for positions J0 .. J0 + N - 1
  if constexpr (N > 0) {
    f(tuple_element(unit_tuple<J0 / 16>(r), J0 % 16));
    each_slot<J0 + 1, N - 1>(r, f);
  }
}

__device__ __forceinline__ void walk16(LanePair& s, v16f t, float guess) {
#pragma unroll 1
  for (int j = 0; j < 16; j += 4) {
    lane_step_pair(s, t[j], guess);
    lane_step_pair(s, t[j + 1], guess);
    lane_step_pair(s, t[j + 2], guess);
    lane_step_pair(s, t[j + 3], guess);
  }
}

// The same step for ANY run length and for guess 0 as well (finite data, finite guess >= 0): the differences get + 0.0 --
// nothing for a guess above zero, and at guess +0.0 it turns the one wrong sign, (-0.0) - (+0.0) = -0.0, into +0.0 (x >= +0.0
// holds for -0.0) -- and where a run of 8 or more ends, NumPy's leaf over that run (from memory, only the lanes concerned)
// takes the place of the left-to-right sum. Costs a compare and a branch per step more than lane_step_pair: this walk runs
// where the fast one reported a long run, and at guess 0, where nearly every unit has one.
template <bool WIDE>
__device__ __forceinline__ void lane_step_pair_long(LanePair& s, float v, float g, const float* at) {
  const unsigned np_ = sign_smear((v - g) + 0.f);
  const unsigned nn_ = sign_smear(((-v) - g) + 0.f);
  const unsigned both = (np_ & 0xFFFFu) | (nn_ & 0xFFFF0000u);
  float add_p = __uint_as_float(__float_as_uint(s.seq_p) & np_);
  float add_n = __uint_as_float(__float_as_uint(s.seq_n) & nn_);
  const unsigned ends_long = s.len2 & 0xFFF8FFF8u & both;       // a half that holds a length >= 8 and is not selected here
  if (ends_long != 0) {
    // (at most one of the two masks ends a long run at a given element: both would need the element before it in both
    // masks, which only a zero at guess 0 is -- and then this element, being finite, is in one of them. One site of the leaf.)
    const bool pos = (ends_long & 0xFFFFu) != 0;
    const int len = static_cast<int>(pos ? s.len2 & 0xFFFFu : s.len2 >> 16);
    const float sum = unit_long_run<WIDE>(at - len, len);
    add_p = pos ? sum : add_p;
    add_n = pos ? add_n : sum;
  }
  s.acc_p = add_f32(s.acc_p, add_p);
  s.acc_n = add_f32(s.acc_n, add_n);
  s.seq_p = __uint_as_float(__float_as_uint(add_f32(s.seq_p, v)) & ~np_);
  s.seq_n = __uint_as_float(__float_as_uint(add_f32(s.seq_n, v)) & ~nn_);
  s.len2 = (s.len2 + 0x00010001u) & ~both;
  s.unsel2 += both & 0x00010001u;
}

template <bool WIDE>
__device__ __forceinline__ void walk16_long(LanePair& s, v16f t, float guess, const float* at) {
#pragma unroll 1
  for (int j = 0; j < 16; ++j) lane_step_pair_long<WIDE>(s, t[j], guess, at + j);
}

// The full walk of the ordinary iterations, a tuple at a time: the fast steps, unless a run of 8+ is open where the tuple
// begins (then the long-aware steps at once) or reaches 8 inside it (then the tuple is walked AGAIN, from the state it
// began with, by the long-aware steps). A rare long run costs its wave sixteen slower steps -- walking the whole unit
// again cost that wave another iteration, and the kernel ends with its slowest wave.
template <int TUPLES, int T = 0>
__device__ __forceinline__ void walk_unit(LanePair& s, const UnitRegs& r, float guess, const float* u) {
  if constexpr (T < TUPLES) {
    bool long_steps = __ballot((s.len2 & 0xFFF8FFF8u) != 0) != 0;      // (wave-uniform)
    if (!long_steps) {
      const LanePair began = s;
      s.long2 = 0;
      walk16(s, unit_tuple<T>(r), guess);
      if (__ballot((s.long2 & 0x00F800F8u) != 0) != 0) {
        s = began;
        long_steps = true;
      }
    }
    if (long_steps) walk16_long<(TUPLES > 8)>(s, unit_tuple<T>(r), guess, u + 16 * T);
    walk_unit<TUPLES, T + 1>(s, r, guess, u);
  }
}

template <int TUPLES, int T = 0>
__device__ __forceinline__ void walk_unit_long(LanePair& s, const UnitRegs& r, float guess, const float* u) {
  if constexpr (T < TUPLES) {
    walk16_long<(TUPLES > 8)>(s, unit_tuple<T>(r), guess, u + 16 * T);
    walk_unit_long<TUPLES, T + 1>(s, r, guess, u);
  }
}

// ---- late iterations on a few candidates. Once an iterate grows, what it selects is a subset of what the previous one
// selected; the last iterations of a unit select a handful of elements and would still walk all of it. So when every live
// unit of the wave selected at most cand_capacity<LEN>() elements and its iterate grew, the next fast walk also LISTS what it selects
// (value and position per lane, in LDS: slot k of lane l at k * 64 + l), and the iterations after that walk the lists:
// the same step with one addition -- a gap between two candidates' positions ends the runs in front of it, exactly what
// walking over the unselected elements in between does (acc + seq, then seq = +0.0). An iterate below the one the list was
// made for (never seen on weights; the iteration does not forbid it) sends the wave back to full walks.
// capacity: 16 / 24 / 32 / 32 entries per lane for units of 32 / 64 / 128 / 256 (+ one slot that takes the unconditional store of a
// step that lists nothing). Measured at 4096 x 4096 (us per call): units of 128: 16 -> 107.7, 24 -> 93.4, 32 -> 91.1; of 64: 99.6 / 97.2 / 98.2;
// of 32: 81.5 / 95.2 / 111.5 (longer lists are walked in full by every later iteration).
template <int LEN>
constexpr int cand_capacity() { return LEN <= 32 ? 16 : (LEN <= 64 ? 24 : 32); }
constexpr int kCandNoPos = -4;     // position of an empty slot: never adjacent to anything

struct CandList {
  float* v;      // [capacity + 1][64]
  int* pos;      // [capacity + 1][64]
};

__device__ __forceinline__ void lane_step_pair_collect(LanePair& s, float v, float g, const CandList& c, int lane,
                                                       int position, int& count) {
  const unsigned np_ = sign_smear(v - g);
  const unsigned nn_ = sign_smear((-v) - g);
  c.v[count * kWave + lane] = v;                       // (overwritten by the next step unless this one is selected)
  c.pos[count * kWave + lane] = position;
  count += static_cast<int>(~(np_ & nn_) & 1u);        // selected by either mask
  s.acc_p = add_f32(s.acc_p, __uint_as_float(__float_as_uint(s.seq_p) & np_));
  s.acc_n = add_f32(s.acc_n, __uint_as_float(__float_as_uint(s.seq_n) & nn_));
  s.seq_p = __uint_as_float(__float_as_uint(add_f32(s.seq_p, v)) & ~np_);
  s.seq_n = __uint_as_float(__float_as_uint(add_f32(s.seq_n, v)) & ~nn_);
  const unsigned both = (np_ & 0xFFFFu) | (nn_ & 0xFFFF0000u);
  s.len2 = (s.len2 + 0x00010001u) & ~both;
  asm("v_or_b32 %0, %1, %2" : "=v"(s.long2) : "v"(s.long2), "v"(s.len2));
  s.unsel2 += both & 0x00010001u;
}

__device__ __forceinline__ void walk16_collect(LanePair& s, v16f t, float guess, const CandList& c, int lane, int base,
                                               int& count) {
#pragma unroll 1
  for (int j = 0; j < 16; j += 2) {
    lane_step_pair_collect(s, t[j], guess, c, lane, base + j, count);
    lane_step_pair_collect(s, t[j + 1], guess, c, lane, base + j + 1, count);
  }
}

template <int TUPLES, int T = 0>
__device__ __forceinline__ void walk_unit_collect(LanePair& s, const UnitRegs& r, float guess, const CandList& c, int lane,
                                                  int& count) {
  if constexpr (T < TUPLES) {
    walk16_collect(s, unit_tuple<T>(r), guess, c, lane, 16 * T, count);
    walk_unit_collect<TUPLES, T + 1>(s, r, guess, c, lane, count);
  }
}

// One pass over the first `slots` candidates (wave-uniform; a lane with fewer has empty slots: value 0, never selected
// by a guess above 0, position never adjacent). s.unsel2 counts the candidates NOT selected.
__device__ __forceinline__ void walk_candidates(LanePair& s, const CandList& c, int lane, int slots, float g) {
  int prev = kCandNoPos;
#pragma unroll 1
  for (int k = 0; k < slots; ++k) {
    const float v = c.v[k * kWave + lane];
    const int position = c.pos[k * kWave + lane];
    const unsigned gap = static_cast<unsigned>((prev + 1 - position) >> 31);     // positions ascend: all ones unless adjacent
    prev = position;
    const unsigned np_ = sign_smear(v - g);
    const unsigned nn_ = sign_smear((-v) - g);
    const unsigned ep = gap | np_, en = gap | nn_;      // the run in front of this element ends (or there is none)
    s.acc_p = add_f32(s.acc_p, __uint_as_float(__float_as_uint(s.seq_p) & ep));
    s.acc_n = add_f32(s.acc_n, __uint_as_float(__float_as_uint(s.seq_n) & en));
    s.seq_p = __uint_as_float(__float_as_uint(add_f32(__uint_as_float(__float_as_uint(s.seq_p) & ~ep), v)) & ~np_);
    s.seq_n = __uint_as_float(__float_as_uint(add_f32(__uint_as_float(__float_as_uint(s.seq_n) & ~en), v)) & ~nn_);
    const unsigned both = (np_ & 0xFFFFu) | (nn_ & 0xFFFF0000u);
    s.len2 = ((s.len2 & ~gap) + 0x00010001u) & ~both;
    asm("v_or_b32 %0, %1, %2" : "=v"(s.long2) : "v"(s.long2), "v"(s.len2));
    s.unsel2 += both & 0x00010001u;
  }
}

// the exact walk, tuple by tuple like the fast one (ONE loop over the unit with the tuple chosen by the position made
// the compiler keep a second copy of the unit in scratch and read it from there, a dependent load per step)
struct ExactWalk {
  LaneMask p, n;
  int lp, ln;
};

template <bool WIDE>
__device__ __forceinline__ void exact16(ExactWalk& w, v16f t, float hi, float lo, const float* at) {
#pragma unroll 1
  for (int j = 0; j < 16; ++j) {
    const float v = t[j];
    lane_step_exact<false, WIDE>(w.p, w.lp, v, hi, at + j);
    lane_step_exact<true, WIDE>(w.n, w.ln, v, lo, at + j);
  }
}

template <int TUPLES, int T = 0>
__device__ __forceinline__ void exact_unit(ExactWalk& w, const UnitRegs& r, float hi, float lo, const float* u) {
  if constexpr (T < TUPLES) {
    exact16<(TUPLES > 8)>(w, unit_tuple<T>(r), hi, lo, u + 16 * T);
    exact_unit<TUPLES, T + 1>(w, r, hi, lo, u);
  }
}

// tuples T0 .. T0 + N - 1 from this lane's LDS row (16 floats each)
template <int T0, int N>
__device__ __forceinline__ void tuples_from_lds(UnitRegs& r, const float* row) {
  if constexpr (N > 0) {
    v16f t;
#pragma unroll
    for (int p_ = 0; p_ < 16; ++p_) t[p_] = row[p_];
    set_unit_tuple<T0>(r, t);
    tuples_from_lds<T0 + 1, N - 1>(r, row + 16);
  }
}

// The load: positions H * STAGE .. (H + 1) * STAGE - 1 of the wave's 64 units per stage, whole lines from HBM (16 lanes per
// 256-byte stretch of a unit) turned through LDS so that lane l ends up with unit l. A stage's lines are asked for BEFORE the stage
// in front of it is turned (its trip through LDS then hides under their round trip to memory).
template <int LEN, int STAGE>
struct StageLines {
  float4 v[STAGE / 4];
};

template <int LEN, int STAGE, int H>
__device__ __forceinline__ StageLines<LEN, STAGE> ask_unit_stage(const float* base, long long wave_floats, int lane) {
  constexpr int kPer = STAGE / 4;              // float4 per unit and stage
  StageLines<LEN, STAGE> out;
#pragma unroll
  for (int q = 0; q < STAGE / 4; ++q) {        // 64 units x kPer float4 = kPer instructions of 64 lanes
    const int f = q * kWave + lane;
    const int uu = f / kPer, p4 = f % kPer;
    const long long e = static_cast<long long>(uu) * LEN + H * STAGE + p4 * 4;
    out.v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < wave_floats) out.v[q] = *reinterpret_cast<const float4*>(base + e);
  }
  return out;
}

template <int LEN, int STAGE, int STRIDE, int H>
__device__ __forceinline__ void load_unit_stages(UnitRegs& xr, const float* base, long long wave_floats, float* lds, int lane,
                                                 const StageLines<LEN, STAGE>& mine) {
  constexpr int kPer = STAGE / 4;
  StageLines<LEN, STAGE> next;
  if constexpr (H + 1 < LEN / STAGE) next = ask_unit_stage<LEN, STAGE, H + 1>(base, wave_floats, lane);
#pragma unroll
  for (int q = 0; q < STAGE / 4; ++q) {
    const int f = q * kWave + lane;
    const int uu = f / kPer, p4 = f % kPer;
    float* d = lds + uu * STRIDE + p4 * 4;
    d[0] = mine.v[q].x; d[1] = mine.v[q].y; d[2] = mine.v[q].z; d[3] = mine.v[q].w;
  }
  __syncthreads();
  tuples_from_lds<H * STAGE / 16, STAGE / 16>(xr, lds + lane * STRIDE);
  __syncthreads();
  if constexpr (H + 1 < LEN / STAGE) load_unit_stages<LEN, STAGE, STRIDE, H + 1>(xr, base, wave_floats, lds, lane, next);
}

template <int LEN>
__global__ __launch_bounds__(kWave, LEN >= 256 ? 1 : (LEN >= 128 ? 2 : (LEN >= 64 ? 3 : 4))) void octav_unit_lanes_kernel(OctavArgs a) {
  constexpr int kStage = LEN < 64 ? LEN : 64;       // positions per trip through LDS
  constexpr int kStride = kStage + 1;               // floats per unit in LDS: lane l reads bank (l + p) % 32
  constexpr int kTuples = LEN / 16;
  constexpr int kCand = cand_capacity<LEN>();
  constexpr int kLdsFloats = kWave * kStride > 2 * (kCand + 1) * kWave ? kWave * kStride : 2 * (kCand + 1) * kWave;
  __shared__ float lds[kLdsFloats];    // the transposition's staging area, then the candidate lists
  const int lane = threadIdx.x;
  const long long unit0 = static_cast<long long>(blockIdx.x) * kWave;
  const long long unit = unit0 + lane;
  const bool live = unit < a.units;
  const float* u = a.x + (live ? unit : 0) * LEN;
  UnitRegs xr;
  // (a lane reading its own 16-byte pieces instead cost 25 us at 4096 x 4096: 64 lines per load instruction)
  {
    const float* base = a.x + unit0 * LEN;
    const long long wave_floats = (a.units - unit0 < kWave ? a.units - unit0 : kWave) * LEN;
    load_unit_stages<LEN, kStage, kStride, 0>(xr, base, wave_floats, lds, lane, ask_unit_stage<LEN, kStage, 0>(base, wave_floats, lane));
  }
  float amax = 0.f;     // NaN is never selected and never the maximum
  float poison = 0.f;   // x * 0 summed: NaN as soon as the unit holds a NaN or an infinity (those units: the exact walk only)
  each_slot<0, LEN>(xr, [&](float v) {
    amax = fmaxf(amax, fabsf(v));
    poison = __builtin_fmaf(v, 0.f, poison);
  });
  const bool special = poison != poison;
  float guess = 1.0f;
  bool done = !live;
  unsigned long long moved = 0;     // wave-uniform: iterations in which some unit of this wave still moved
  const CandList cand{lds, reinterpret_cast<int*>(lds + (kCand + 1) * kWave)};
  int mode = 0;                     // wave-uniform: 0 full walks, 1 the next walk lists what it selects, 2 the lists are walked
  int cand_slots = 0;               // wave-uniform: the longest list
  int cand_count = 0;               // this lane's list
  float cand_guess = 0.f;           // the iterate this lane's list was made for
  for (int it = 0; it < a.max_iter; ++it) {
    bool still_moving = false;
    bool may_list = false;          // after this iteration: few elements selected and the iterate grew
    if (mode == 2 && __ballot(!done && !(guess >= cand_guess)) != 0) mode = 0;      // (an iterate fell below its list)
    if (mode == 1) {                // (also for a unit this iterate selects nothing of: it walks nothing)
      cand_count = 0;
      cand_guess = guess;
    }
    if (!done) {
      LaneMask p{0.f, 0.f, 0}, n{0.f, 0.f, 0};
      if (guess <= amax) {      // (a guess above the unit's largest |x| selects nothing: the reference's first guess 1.0)
        // three walks. Units with special values and guesses that are not finite: compares (exact_unit). Guess 0 (a weight
        // tensor's second iterate: x >= 0 / x <= -0, a run of 8+ in nearly every unit) and the units whose fast walk met a
        // run of 8+: the sign-bit walk with the leaf where such a run ends (walk_unit_long). Everything else: the fast walk.
        const bool by_compares = special || !(guess >= 0.f) || !(guess < __builtin_inff());
        bool long_runs = !by_compares && !(guess > 0.f);
        if (__ballot(!by_compares && !long_runs) != 0) {
          LanePair s{0.f, 0.f, 0.f, 0.f, 0u, 0u, 0u};
          int walked = LEN;
          if (mode == 2) {
            walk_candidates(s, cand, lane, cand_slots, guess);
            walked = cand_slots;
          } else if (mode == 1) {
            walk_unit_collect<kTuples>(s, xr, guess, cand, lane, cand_count);
          } else {
            walk_unit<kTuples>(s, xr, guess, u);
            s.long2 = 0;         // (long runs were dealt with where they ended; one that touches the unit's end: below)
          }
          // a run that touches the unit's end ends there
          const int lp = static_cast<int>(s.len2 & 0xFFFFu), ln = static_cast<int>(s.len2 >> 16);
          p.acc = s.acc_p + (mode == 0 && lp >= 8 ? unit_long_run<(LEN > 128)>(u + LEN - lp, lp) : s.seq_p);
          n.acc = s.acc_n + (mode == 0 && ln >= 8 ? unit_long_run<(LEN > 128)>(u + LEN - ln, ln) : s.seq_n);
          p.cnt = walked - static_cast<int>(s.unsel2 & 0xFFFFu);
          n.cnt = walked - static_cast<int>(s.unsel2 >> 16);
          long_runs |= !by_compares && (s.long2 & 0x00F800F8u) != 0;      // (listing / list walks: some run reached 8)
        }
        if (long_runs) {
          LanePair s{0.f, 0.f, 0.f, 0.f, 0u, 0u, 0u};
          walk_unit_long<kTuples>(s, xr, guess, u);
          const int lp = static_cast<int>(s.len2 & 0xFFFFu), ln = static_cast<int>(s.len2 >> 16);
          p.acc = s.acc_p + (lp >= 8 ? unit_long_run<(LEN > 128)>(u + LEN - lp, lp) : s.seq_p);
          n.acc = s.acc_n + (ln >= 8 ? unit_long_run<(LEN > 128)>(u + LEN - ln, ln) : s.seq_n);
          p.cnt = LEN - static_cast<int>(s.unsel2 & 0xFFFFu);
          n.cnt = LEN - static_cast<int>(s.unsel2 >> 16);
        }
        if (by_compares) {
          const float hi = guess, lo = -guess;
          ExactWalk w{LaneMask{0.f, 0.f, 0}, LaneMask{0.f, 0.f, 0}, 0, 0};
          exact_unit<kTuples>(w, xr, hi, lo, u);
          p = w.p;
          n = w.n;
          p.acc = p.acc + (w.lp >= 8 ? unit_long_run<(LEN > 128)>(u + LEN - w.lp, w.lp) : p.seq);
          n.acc = n.acc + (w.ln >= 8 ? unit_long_run<(LEN > 128)>(u + LEN - w.ln, w.ln) : n.seq);
        }
      }
      const OctavStep st = octav_step(guess, p.acc, n.acc, p.cnt, n.cnt, LEN, a.s, a.count_is_f64);
      a.hist[static_cast<long long>(it) * a.units + unit] = st.next;
      still_moving = !st.close;
      may_list = !special && guess > 0.f && st.next >= guess && st.next < __builtin_inff() && p.cnt + n.cnt <= kCand;
      if (reached_fixed_point(guess, st.next)) {
        repeat_iterate(a, it, unit, st.next);
        done = true;
      }
      guess = st.next;
    }
    if (__ballot(still_moving) != 0) moved |= 1ull << it;      // (outside the divergent part: every lane keeps the wave's mask)
    if (__ballot(!done) == 0) break;
    if (mode == 1) {
      // the lists are complete: empty slots behind every lane's own, the longest list bounds the walks
      int longest = cand_count;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) longest = max(longest, __shfl_xor(longest, off));
      cand_slots = __builtin_amdgcn_readfirstlane(longest);
      for (int k = cand_count; k < cand_slots; ++k) {
        cand.v[k * kWave + lane] = 0.f;
        cand.pos[k * kWave + lane] = kCandNoPos;
      }
      mode = 2;
    } else if (mode == 0 && __ballot(!done && !may_list) == 0) {
      mode = 1;
    }
  }
  if (lane == 0) publish_moving(a.moving, moved);
}

size_t octav_groups_smem() {
  const size_t row_floats = static_cast<size_t>((kGroupLen + (kGroupLen >> 4) + 4) & ~3);
  const size_t cap = static_cast<size_t>(((kGroupLen / 2 + 2 + 63) & ~63) + 64);
  return 512 + kGroupThreads * sizeof(float) + row_floats * sizeof(float) + cap * 2 * sizeof(float) +
         2 * kGroupThreads * sizeof(unsigned short) + 4 * kGroupThreads * sizeof(int) + kGroupThreads * sizeof(float) + 16;
}

// MI355Q_OCTAV_UNIT_LANES=0: blockwise units take octav_groups_kernel again (A / B, tests compare the two)
bool octav_unit_lanes_on() {
  const char* e = getenv("MI355Q_OCTAV_UNIT_LANES");
  return e == nullptr || atoi(e) != 0;
}

// candidates per mask a row may hand to its tail (an eighth of the row, a multiple of 64)
int octav_tail_cap(int len) {
  static const int div = [] { const char* e = getenv("MI355Q_OCTAV_TAIL_DIV"); const int v = e ? atoi(e) : 0; return v >= 2 && v <= 256 ? v : 8; }();
  const int c = ((len / div) + 63) & ~63;
  return c < 64 ? 64 : c;
}

int octav_rows_threads(int len) {
  const int npieces = (len + kPiece - 1) / kPiece;
  if (npieces > 256 && !getenv("MI355Q_OCTAV_NARROW_ROWS")) return npieces <= 512 ? 512 : 1024;
  return npieces <= 64 ? 64 : (npieces <= 128 ? 128 : 256);
}

size_t octav_rows_smem(int len) {
  const int kRowsThreads = octav_rows_threads(len);
  const int npieces = (len + kPiece - 1) / kPiece;
  const size_t row_floats = static_cast<size_t>((len + (len >> 4) + 4) & ~3);
  const size_t cap = static_cast<size_t>(((len / 2 + 2 + 63) & ~63) + 64);
  return 2 * rows_xchg_floats(kRowsThreads) * sizeof(float) + kRowsThreads * sizeof(float) + row_floats * sizeof(float) + cap * 2 * sizeof(float) +
         static_cast<size_t>((npieces + 1) & ~1) * 2 * sizeof(unsigned short);
}

// ---- general [outer, channels, inner] view: unit c = the `outer` segments x[o, c, :] -----
// NumPy hands each contiguous segment to the inner loop separately and keeps one running
// total per channel: acc = acc + pairwise(run) for every run of every segment, in order
// (tests/test_numpy_sum_model.py::test_mixed_reductions). One wave per channel; segments are
// read from global memory (L2-resident after the first Newton iteration).
__global__ void octav_seg_kernel(OctavArgs a, long long channels, long long outer) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const long long c = static_cast<long long>(blockIdx.x) * (blockDim.x / kWave) + wave;
  if (c >= channels) return;
  const int inner = a.len;
  const float qnan = __builtin_nanf("");
  float guess = 1.0f;
  unsigned long long moved = 0;  // iterations in which this unit's guess still moved
  for (int it = 0; it < a.max_iter; ++it) {
    RunSum pos, neg;
    long long npos = 0, nneg = 0;
    const float hi = guess, lo = -guess;
    for (long long o = 0; o < outer; ++o) {
      const float* u = a.x + (o * channels + c) * inner;
      for (int base = 0; base < inner; base += kWave) {
        const int i = base + lane;
        const float v = i < inner ? u[i] : qnan;
        pos.feed(__ballot(v >= hi), base, v, u, lane);
        neg.feed(__ballot(v <= lo), base, v, u, lane);
      }
      pos.flush(u, lane);
      neg.flush(u, lane);
      npos += pos.count;
      nneg += neg.count;
      pos.count = neg.count = 0;
    }
    const OctavStep st = octav_step(guess, pos.acc, neg.acc, npos, nneg, outer * inner, a.s, 1);
    if (lane == 0) a.hist[static_cast<long long>(it) * a.units + c] = st.next;
    if (!st.close) moved |= 1ull << it;
    if (reached_fixed_point(guess, st.next)) {
      if (lane == 0) repeat_iterate(a, it, c, st.next);
      break;
    }
    guess = st.next;
  }
  if (lane == 0) publish_moving(a.moving, moved);
}

__global__ void mse_scale_seg_kernel(const float* __restrict__ x, long long outer, long long channels,
                                     int inner, float multiplier, float* __restrict__ scale) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const long long c = static_cast<long long>(blockIdx.x) * (blockDim.x / kWave) + wave;
  if (c >= channels) return;
  float acc = 0.f;
  bool first = true;
  for (long long o = 0; o < outer; ++o) {
    const float* u = x + (o * channels + c) * inner;
    for (int k = 0; k < inner; k += kChunk) {
      const int n = inner - k < kChunk ? inner - k : kChunk;
      const float part = pairwise_sum<true, 8>(u + k, n, lane);
      acc = first ? part : acc + part;
      first = false;
    }
  }
  const float mean = acc / static_cast<float>(outer * inner);
  if (lane == 0) scale[c] = multiplier * __builtin_sqrtf(mean);
}

// ---- channel-last units: x viewed as [outer, channels], one unit per channel -------------
// NumPy reduces the leading axes of a C-contiguous array row by row (`out[c] += x[o, c]`,
// masked elements skipped), i.e. each channel is a plain left-to-right float32 sum over o
// (tests/numpy_sum_model.py states and checks this). One lane per channel keeps that order and
// makes every load a coalesced row segment.
constexpr int kColsUnroll = 8;

__global__ __launch_bounds__(256) void octav_cols_kernel(OctavArgs a, long long channels) {
  const long long c_raw = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  const bool live = c_raw < channels;           // dead lanes shadow the last channel, write nothing
  const long long c = live ? c_raw : channels - 1;
  const long long outer = a.len;
  float guess = 1.0f;
  bool settled = false;
  unsigned long long moved = 0;  // iterations in which this unit's guess still moved
  for (int it = 0; it < a.max_iter; ++it) {
    const float hi = guess, lo = -guess;
    float pos = 0.f, neg = 0.f;
    long long npos = 0, nneg = 0;
    const float* p = a.x + c;
    long long o = 0;
    for (; o + kColsUnroll <= outer; o += kColsUnroll) {
      float v[kColsUnroll];
#pragma unroll
      for (int k = 0; k < kColsUnroll; ++k) v[k] = p[static_cast<long long>(k) * channels];
#pragma unroll
      for (int k = 0; k < kColsUnroll; ++k) {
        if (v[k] >= hi) { pos = pos + v[k]; ++npos; }
        if (v[k] <= lo) { neg = neg + v[k]; ++nneg; }
      }
      p += kColsUnroll * channels;
    }
    for (; o < outer; ++o, p += channels) {
      const float v = *p;
      if (v >= hi) { pos = pos + v; ++npos; }
      if (v <= lo) { neg = neg + v; ++nneg; }
    }
    const OctavStep st = octav_step(guess, pos, neg, npos, nneg, outer, a.s, a.count_is_f64);
    if (live && !settled) a.hist[static_cast<long long>(it) * a.units + c] = st.next;
    if (live && !settled && !st.close) moved |= 1ull << it;
    if (!settled && reached_fixed_point(guess, st.next)) {
      settled = true;
      if (live) repeat_iterate(a, it, c, st.next);
    }
    guess = st.next;
    if (__all(settled)) break;
  }
  // one lane per 64 channels publishes the union of the wave
  for (int off = kWave / 2; off > 0; off >>= 1) moved |= __shfl_xor(moved, off, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0) publish_moving(a.moving, moved);
}

__global__ __launch_bounds__(256) void mse_scale_cols_kernel(const float* __restrict__ x, long long outer,
                                                            long long channels, float multiplier,
                                                            float* __restrict__ scale) {
  const long long c = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (c >= channels) return;
  const float* p = x + c;
  float acc = 0.f;
  long long o = 0;
  for (; o + kColsUnroll <= outer; o += kColsUnroll) {
    float v[kColsUnroll];
#pragma unroll
    for (int k = 0; k < kColsUnroll; ++k) v[k] = p[static_cast<long long>(k) * channels];
#pragma unroll
    for (int k = 0; k < kColsUnroll; ++k) acc = acc + v[k] * v[k];
    p += kColsUnroll * channels;
  }
  for (; o < outer; ++o, p += channels) acc = acc + (*p) * (*p);
  scale[c] = multiplier * __builtin_sqrtf(acc / static_cast<float>(outer));
}

// The reference stops at the first iteration where *every* unit is close
// (octav.py:109); every unit ran all iterations, so just pick that iterate.
__global__ __launch_bounds__(256) void octav_pick_kernel(const float* __restrict__ hist,
                                                        const unsigned long long* __restrict__ moving,
                                                        long long units, int max_iter, int early_stop,
                                                        float* __restrict__ clip, int* iters_out) {
  int k = max_iter - 1;
  if (early_stop)
    for (int it = 0; it < max_iter; ++it)
      if (((*moving >> it) & 1ull) == 0) { k = it; break; }
  const long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (i < units) clip[i] = hist[static_cast<long long>(k) * units + i];
  if (i == 0 && iters_out != nullptr) *iters_out = k + 1;
}

struct MseArgs {
  const float* x;
  long long units;
  int len;
  int lds_stride;
  float multiplier;
  float* scale;
  // mi355q_mse_requant_f32: the wave that made a unit's scale also quantizes the unit (it has just read it: the second
  // pass comes out of the L2), four int8 per lane and store; null: scales only
  unsigned* q = nullptr;     // [units][len / 4]
  float lo = 0.f, hi = 0.f;
};

// Leaves of NumPy's pairwise recursion over [lo, lo + n), in order (n <= 8192: at most 128).
__device__ void list_leaves(int lo, int n, int* leaf_lo, int* leaf_n, int* count) {
  if (n <= 128) {
    leaf_lo[*count] = lo;
    leaf_n[*count] = n;
    ++*count;
    return;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  list_leaves(lo, n2, leaf_lo, leaf_n, count);
  list_leaves(lo + n2, n - n2, leaf_lo, leaf_n, count);
}

// The recursion again with the leaves' sums known (consumed in order).
__device__ float fold_leaves(int n, const float* leaf_sum, int* next) {
  if (n <= 128) return leaf_sum[(*next)++];
  int n2 = n / 2;
  n2 -= n2 % 8;
  const float left = fold_leaves(n2, leaf_sum, next);
  return left + fold_leaves(n - n2, leaf_sum, next);
}

// sum(x*x) of contiguous units in NumPy's order, all lanes busy: a unit is a sequence of
// 8192-element chunks added in order; a chunk is a tree of leaves of <= 128 elements; a leaf is
// eight strided accumulators. One wave per unit; eight lanes share a leaf, one per accumulator
// (a wave step reads eight 32-byte runs straight from global memory, every byte of every cache
// line exactly once), the accumulators fold by an xor butterfly -- ((r0+r1)+(r2+r3))+((r4+r5)+
// (r6+r7)), IEEE addition commutes so both partners agree -- and lane 0 folds the leaf sums in
// the recursion's order. The two leaf tables a unit needs (full chunk, last chunk) are listed
// once per block. This is the general form (any length); mse_scale_balanced_kernel below takes
// the lengths whose tree is complete.
__global__ __launch_bounds__(256) void mse_scale_leaves_kernel(MseArgs a) {
  __shared__ int leaf_lo[2][128], leaf_n[2][128], leaves[2];
  __shared__ float leaf_sum[4][128];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int k = lane & 7, slot = lane >> 3;
  const int last_len = a.len - (a.len - 1) / kChunk * kChunk;  // 1 .. 8192
  if (threadIdx.x == 0 || threadIdx.x == kWave) {
    const int t = threadIdx.x == 0 ? 0 : 1;
    int c = 0;
    list_leaves(0, t == 0 ? (a.len < kChunk ? a.len : kChunk) : last_len, leaf_lo[t], leaf_n[t], &c);
    leaves[t] = c;
  }
  __syncthreads();
  const long long unit = static_cast<long long>(blockIdx.x) * 4 + wave;
  const bool live_unit = unit < a.units;
  const float* u = a.x + (live_unit ? unit : 0) * a.len;
  float acc = 0.f;
  for (int c0 = 0; c0 < a.len; c0 += kChunk) {
    const int n = a.len - c0 < kChunk ? a.len - c0 : kChunk;
    const int t = (c0 + kChunk >= a.len) ? 1 : 0;
    const int count = leaves[t];
    for (int base = 0; base < count; base += 8) {
      const int leaf = base + slot;
      const bool live = leaf < count;
      const float* p = u + c0 + (live ? leaf_lo[t][leaf] : 0);
      const int ln = live ? leaf_n[t][leaf] : 0;
      float res = 0.f;
      if (ln < 8) {  // short leaf (only a chunk shorter than 8 has one): left to right
        for (int i = 0; i < ln; ++i) res = res + p[i] * p[i];
      } else {
        float v[16];  // all loads first, then the ordered adds
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = (8 * (q + 1) <= ln) ? p[8 * q + k] : 0.f;
        float r = v[0] * v[0];
#pragma unroll
        for (int q = 1; q < 16; ++q)
          if (8 * (q + 1) <= ln) r = r + v[q] * v[q];
        r = r + __shfl_xor(r, 1, kWave);
        r = r + __shfl_xor(r, 2, kWave);
        r = r + __shfl_xor(r, 4, kWave);
        for (int i = ln & ~7; i < ln; ++i) r = r + p[i] * p[i];
        res = r;
      }
      if (live && k == 0) leaf_sum[wave][leaf] = res;
    }
    __syncthreads();
    if (lane == 0) {
      int next = 0;
      const float part = fold_leaves(n, leaf_sum[wave], &next);
      acc = c0 == 0 ? part : acc + part;
    }
    __syncthreads();
  }
  if (lane == 0 && live_unit) {
    const float mean = acc / static_cast<float>(a.len);
    a.scale[unit] = a.multiplier * __builtin_sqrtf(mean);
  }
}

// The same sum for units whose chunks are balanced (common.h; every full chunk is: 8192 =
// 128 << 6): no
// tables, no LDS, no serial fold. Eight lanes share a leaf as above; leaf index = 8 * step +
// (lane >> 3), so the three lowest tree levels are lane butterflies (xor 8, 16, 32), and the
// levels above pair whole steps, folded as they arrive like a binary counter (the step loop is
// unrolled, so that is straight-line code).
__global__ __launch_bounds__(256) void mse_scale_balanced_kernel(MseArgs a, int last_leaf, int last_depth) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int k = lane & 7, slot = lane >> 3;
  const long long unit = static_cast<long long>(blockIdx.x) * 4 + wave;
  if (unit >= a.units) return;
  const float* u = a.x + unit * a.len;
  float acc = 0.f;
  for (int c0 = 0; c0 < a.len; c0 += kChunk) {
    const bool last = c0 + kChunk >= a.len;
    const int leaf_len = last ? last_leaf : 128, depth = last ? last_depth : 6;
    const int count = 1 << depth;
    const int steps = count > 8 ? count >> 3 : 1;
    float lvl0 = 0.f, lvl1 = 0.f, lvl2 = 0.f, x = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < steps) {
        const int leaf = s * 8 + slot;
        const float* p = u + c0 + static_cast<long long>(leaf < count ? leaf : 0) * leaf_len + k;
        float v[16];  // all loads first, then the ordered adds
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = (8 * (q + 1) <= leaf_len) ? p[8 * q] : 0.f;
        float r = v[0] * v[0];
#pragma unroll
        for (int q = 1; q < 16; ++q)
          if (8 * (q + 1) <= leaf_len) r = r + v[q] * v[q];
        r = r + __shfl_xor(r, 1, kWave);  // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)); addition commutes
        r = r + __shfl_xor(r, 2, kWave);
        r = r + __shfl_xor(r, 4, kWave);
        if (count > 1) r = r + __shfl_xor(r, 8, kWave);   // leaves 2i, 2i+1
        if (count > 2) r = r + __shfl_xor(r, 16, kWave);
        if (count > 4) r = r + __shfl_xor(r, 32, kWave);
        x = r;  // the sum of this step's (up to) eight leaves
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
    acc = c0 == 0 ? x : acc + x;
  }
  acc = lane_bcast(acc, 0);      // (lane 0's total is the unit's: short last chunks leave other slots with copies of leaf 0)
  const float mean = acc / static_cast<float>(a.len);
  const float scale = a.multiplier * __builtin_sqrtf(mean);
  if (lane == 0) a.scale[unit] = scale;
  if (a.q != nullptr) {
    // uniform_quantize with a zero zero point: rint(x / scale) clipped (ref uniform_quantize_tensor.py:357-360; the reference's
    // `+ zero_point` in float64 changes no value here: it only turns -0.0 into +0.0, and both round to the integer 0)
    const float4* u4 = reinterpret_cast<const float4*>(u);
    unsigned* q = a.q + unit * (a.len / 4);
    for (int i = lane; i < a.len / 4; i += kWave) {
      const float4 v = u4[i];
      const unsigned w = (static_cast<unsigned>(round_clip(v.x / scale, a.lo, a.hi)) & 0xFFu) |
                         ((static_cast<unsigned>(round_clip(v.y / scale, a.lo, a.hi)) & 0xFFu) << 8) |
                         ((static_cast<unsigned>(round_clip(v.z / scale, a.lo, a.hi)) & 0xFFu) << 16) |
                         ((static_cast<unsigned>(round_clip(v.w / scale, a.lo, a.hi)) & 0xFFu) << 24);
      __builtin_nontemporal_store(w, q + i);
    }
  }
}

struct UnitPlan {
  bool use_lds;
  int waves;       // waves (units) per block
  int lds_stride;  // floats
  size_t smem;
};

UnitPlan plan_units(int len) {
  UnitPlan p{};
  const int stride = (len + 3) / 4 * 4;
  const size_t per_wave = static_cast<size_t>(stride) * sizeof(float);
  constexpr size_t kBudget = 64 * 1024;  // per block: keeps >= 2 blocks per CU
  int waves = static_cast<int>(kBudget / (per_wave ? per_wave : 1));
  if (waves > 4) waves = 4;
  // (one-wave blocks would fit ten 16 KB slices per CU instead of eight; measured 12 % slower:
  // the loops are bound by scalar-instruction issue, not by waves in flight)
  if (waves >= 1) {
    p.use_lds = true;
    p.waves = waves;
    p.lds_stride = stride;
    p.smem = per_wave * waves;
  } else {
    p.use_lds = false;
    p.waves = 4;
    p.lds_stride = 0;
    p.smem = 0;
  }
  return p;
}


// ---------------------------------------------------------------- OCTAV, one read (opt-in, tolerance class T2) ---
// ref: algorithms/uniform_quantize/octav.py:30-112. The kernels above reproduce NumPy's float32 summation ORDER (pairwise
// over runs of selected elements) so that the clipping constants are the reference's bit for bit; the price is 0.045 of
// one read of the tensor. The reference pins neither NumPy nor its summation order (SURVEY 7 classes OCTAV as T2), so this
// kernel offers the other trade: a unit (row / block) stays in REGISTERS as |x| for all iterations -- one pass over HBM --
// and a masked sum is per-lane float32 partials (<= 64 elements each), a float32 tree over a wave's lanes and a float64
// sum over a unit's waves: within 1e-6 of the reference's scale, not bit-identical to it.
//   c <- sum_{|x| >= c} |x| / (f32(n_sel) (1 - s) + f64(s) N):  for c > 0 the reference's two masks (x >= c, x <= -c) are
// disjoint and their sums' difference is the sum of |x| over their union; for c == 0 (every weight tensor's second iterate:
// a first guess of 1 selects nothing) each zero sits in both masks, counted twice in n_sel as the reference counts it
// (octav.py:87-100). NaNs fail both tests there and `|x| >= c` here.
// The reference's early stop is global (octav.py:109): every unit runs all iterations, records its iterates and the
// iterations in which it still moved, and octav_pick_kernel takes the first iterate at which no unit moved.
struct OctavFastArgs {
  const float* x;
  long long units;
  int len;          // elements per unit, a multiple of 4
  int max_iter;
  float s;
  float* hist;      // [max_iter][units]
  unsigned long long* moving;
};

// Sum over the LANES (<= 64) lanes that own a unit, result on every one of them, without LDS traffic: butterflies inside a
// row of 16 lanes are DPP operands of the add itself (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror); rows
// are combined through four v_readlane (the wave's four row totals in scalar registers). A ds_bpermute per step (what
// __shfl_xor compiles to) made the twelve dependent exchanges of one iteration cost more than its 64 compare-select-adds.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }

template <int LANES>
__device__ __forceinline__ float unit_sum_f32(float v) {
  if constexpr (LANES >= 2) v += dpp_f32<0xB1>(v);
  if constexpr (LANES >= 4) v += dpp_f32<0x4E>(v);
  if constexpr (LANES >= 8) v += dpp_f32<0x141>(v);
  if constexpr (LANES >= 16) v += dpp_f32<0x140>(v);
  if constexpr (LANES >= 32) {
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    if constexpr (LANES >= 64) v = (r0 + r1) + (r2 + r3);
    else v = (threadIdx.x & 32) ? r2 + r3 : r0 + r1;
  }
  return v;
}
template <int LANES>
__device__ __forceinline__ int unit_sum_i32(int v) {
  if constexpr (LANES >= 2) v += dpp_i32<0xB1>(v);
  if constexpr (LANES >= 4) v += dpp_i32<0x4E>(v);
  if constexpr (LANES >= 8) v += dpp_i32<0x141>(v);
  if constexpr (LANES >= 16) v += dpp_i32<0x140>(v);
  if constexpr (LANES >= 32) {
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    if constexpr (LANES >= 64) v = (r0 + r1) + (r2 + r3);
    else v = (threadIdx.x & 32) ? r2 + r3 : r0 + r1;
  }
  return v;
}

// LANES lanes own a unit (1 .. 64: part of a wave or a wave; 256 / 1024: the workgroup), V float4 per lane.
template <int LANES, int V>
__global__ __launch_bounds__(LANES > 256 ? LANES : 256) void octav_fast_kernel(OctavFastArgs a) {
  constexpr int THREADS = LANES > 256 ? LANES : 256;
  constexpr int UPB = THREADS / LANES;                     // units per workgroup
  constexpr int WAVES = LANES / kWave;                    // waves per unit (0: several units per wave)
  const int l = threadIdx.x % LANES;
  const long long u = static_cast<long long>(blockIdx.x) * UPB + threadIdx.x / LANES;
  const bool live = u < a.units;
  const int len4 = a.len / 4;
  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(a.x) + (live ? u : 0) * len4;
  const float qnan = __builtin_nanf("");
  // every load is issued before the first value is looked at (V 16-byte loads in flight per lane): slots past the
  // unit's end re-read its last piece and are blanked afterwards instead of branching around the load
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f raw[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int c = j * LANES + l;
    raw[j] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(&x4[c < len4 ? c : len4 - 1]));     // read once
  }
  float v[V][4];
  int zeros = 0;                                          // per lane, or -- units that own whole waves -- per wave
  float top = 0.f;                                        // this lane's largest |x| (fmaxf passes NaNs and blanks by)
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const bool mine = live && j * LANES + l < len4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[j][e] = mine ? fabsf(raw[j][e]) : qnan;           // a blank slot, like a NaN, is never selected and never a zero
      top = fmaxf(top, v[j][e]);
      if constexpr (LANES >= kWave) zeros += __builtin_popcountll(__builtin_amdgcn_ballot_w64(v[j][e] == 0.f));
      else zeros += v[j][e] == 0.f;
    }
  }
  __shared__ double red_sum[2][WAVES > 0 ? WAVES : 1];
  __shared__ int red_cnt[2][WAVES > 0 ? WAVES : 1];
  float unit_top = 0.f;                                  // the unit's largest |x| (units that own whole waves)
  if constexpr (LANES >= kWave) {
    unit_top = top;
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) unit_top = fmaxf(unit_top, __shfl_xor(unit_top, off, kWave));
    if constexpr (WAVES > 1) {
      __shared__ float red_top[WAVES > 0 ? WAVES : 1];
      if ((threadIdx.x & (kWave - 1)) == 0) red_top[threadIdx.x / kWave] = unit_top;
      __syncthreads();
#pragma unroll
      for (int k = 0; k < WAVES; ++k) unit_top = fmaxf(unit_top, red_top[k]);
    }
  }
  float guess = 1.0f;
  unsigned long long moved = 0;
  int it = 0;
  for (; it < a.max_iter; ++it) {
    float part = 0.f;
    int cnt = 0;
    if constexpr (LANES >= kWave) {
      // The guess is the same on every lane of the wave: a guess above the unit's largest |x| (the first guess, 1, of
      // weights below 1) selects nothing, and the iteration costs nothing.
      if (!(guess > unit_top)) {
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
        for (int j = 0; j < V; ++j) {
          // compare -> select -> lane count, element by element on VCC: left to the scheduler, the 64 compares are issued
          // first and their lane masks (128 scalar registers) spill through v_writelane / v_readlane
          // four elements per step, the four compares first: each lane mask (an SGPR pair of its own) is three
          // instructions old when its select and its s_bcnt1 read it, so neither waits for the compare to retire
          float t0, t1, t2, t3;
          int n0, n1, n2, n3;
          unsigned long long m0, m1, m2, m3;
          asm volatile(
              "v_cmp_ge_f32_e64 %[m0], %[x0], %[g]\n\tv_cmp_ge_f32_e64 %[m1], %[x1], %[g]\n\t"
              "v_cmp_ge_f32_e64 %[m2], %[x2], %[g]\n\tv_cmp_ge_f32_e64 %[m3], %[x3], %[g]\n\t"
              "v_cndmask_b32_e64 %[t0], 0, %[x0], %[m0]\n\tv_cndmask_b32_e64 %[t1], 0, %[x1], %[m1]\n\t"
              "v_cndmask_b32_e64 %[t2], 0, %[x2], %[m2]\n\tv_cndmask_b32_e64 %[t3], 0, %[x3], %[m3]\n\t"
              "s_bcnt1_i32_b64 %[n0], %[m0]\n\ts_bcnt1_i32_b64 %[n1], %[m1]\n\t"
              "s_bcnt1_i32_b64 %[n2], %[m2]\n\ts_bcnt1_i32_b64 %[n3], %[m3]\n\t"
              "v_add_f32 %[p0], %[p0], %[t0]\n\tv_add_f32 %[p1], %[p1], %[t1]\n\t"
              "v_add_f32 %[p2], %[p2], %[t2]\n\tv_add_f32 %[p3], %[p3], %[t3]\n\t"
              "s_add_i32 %[n0], %[n0], %[n1]\n\ts_add_i32 %[n2], %[n2], %[n3]\n\t"
              "s_add_i32 %[c], %[c], %[n0]\n\ts_add_i32 %[c], %[c], %[n2]"
              : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [c] "+s"(cnt),
                [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),
                [n0] "=&s"(n0), [n1] "=&s"(n1), [n2] "=&s"(n2), [n3] "=&s"(n3),
                [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)
              : [x0] "v"(v[j][0]), [x1] "v"(v[j][1]), [x2] "v"(v[j][2]), [x3] "v"(v[j][3]), [g] "v"(guess)
              : "scc");
        }
        part = (p0 + p1) + (p2 + p3);
        if (guess == 0.f) cnt += zeros;                  // every zero sits in both of the reference's masks
      }
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool sel = v[j][e] >= guess;
          part += sel ? v[j][e] : 0.f;
          cnt += sel;
        }
      cnt = unit_sum_i32<LANES>(cnt + (guess == 0.f ? zeros : 0));
    }
    // per-lane float32 partials (<= 64 elements), float32 tree over the wave's lanes, float64 across the unit's waves
    double sum = static_cast<double>(unit_sum_f32<(LANES < kWave ? LANES : kWave)>(part));
    if constexpr (WAVES > 1) {
      const int w = threadIdx.x / kWave, b = it & 1;
      if ((threadIdx.x & (kWave - 1)) == 0) { red_sum[b][w] = sum; red_cnt[b][w] = cnt; }
      __syncthreads();
      sum = 0.0; cnt = 0;
#pragma unroll
      for (int k = 0; k < WAVES; ++k) { sum += red_sum[b][k]; cnt += red_cnt[b][k]; }
    }
    const OctavStep st = octav_step(guess, static_cast<float>(sum), 0.f, cnt, 0, a.len, a.s, 1);
    if (live && l == 0) a.hist[static_cast<long long>(it) * a.units + u] = st.next;
    if (!st.close) moved |= 1ull << it;
    const bool fixed = reached_fixed_point(guess, st.next);
    guess = st.next;
    // every lane of a unit holds the same iterate: the unit leaves together; lanes of other units in the wave go on
    if constexpr (LANES >= kWave) {
      if (fixed) { ++it; break; }
    }
  }
  if constexpr (LANES >= kWave) {
    if (live && l == 0)
      for (int k = it; k < a.max_iter; ++k) a.hist[static_cast<long long>(k) * a.units + u] = guess;
  }
  if (!live) moved = 0;
  // one flag update per wave at most
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    const unsigned lo = __shfl_xor(static_cast<unsigned>(moved), off, kWave);
    const unsigned hi = __shfl_xor(static_cast<unsigned>(moved >> 32), off, kWave);
    moved |= (static_cast<unsigned long long>(hi) << 32) | lo;
  }
  if ((threadIdx.x & (kWave - 1)) == 0) publish_moving(a.moving, moved);
}

template <int LANES, int V>
void launch_octav_fast(const OctavFastArgs& a, hipStream_t st) {
  constexpr int THREADS = LANES > 256 ? LANES : 256;
  constexpr int UPB = THREADS / LANES;
  hipLaunchKernelGGL((octav_fast_kernel<LANES, V>), dim3(static_cast<unsigned>((a.units + UPB - 1) / UPB)), dim3(THREADS), 0, st, a);
}

}  // namespace
}  // namespace mi355q

using namespace mi355q;

extern "C" size_t mi355q_octav_workspace_bytes(int64_t units, int32_t max_iter) {
  if (units <= 0 || max_iter <= 0) return 0;
  return static_cast<size_t>(units) * max_iter * sizeof(float) + 64 * sizeof(int);
}

extern "C" size_t mi355q_octav_rows_workspace_bytes(int64_t units, int64_t unit_len, int32_t max_iter) {
  const size_t base = mi355q_octav_workspace_bytes(units, max_iter);
  if (base == 0 || unit_len < kRowsMinLen || unit_len > kRowsMaxLen) return base;
  const size_t cap = static_cast<size_t>(octav_tail_cap(static_cast<int>(unit_len)));
  return ((base + 63) & ~static_cast<size_t>(63)) + static_cast<size_t>(units) * (sizeof(TailState) + 2 * cap * (sizeof(float) + sizeof(unsigned short)));
}

extern "C" int32_t mi355q_octav_clip_f32(const float* x, int64_t units, int64_t unit_len,
                                         int32_t bits, int32_t max_iter, float exponent_divisor,
                                         int32_t early_stop, int32_t count_is_f64, float* clip_out,
                                         int32_t* iters_out, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  clear_error();
  if (units < 0 || unit_len < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  if (max_iter < 1 || max_iter > 64) return fail(MI355Q_BAD_ARG, "max_iter must be in [1, 64]");
  if (bits < 1 || bits > 16) return fail(MI355Q_BAD_ARG, "bits must be in [1, 16]");
  if (units == 0) return MI355Q_OK;
  if (unit_len == 0) return fail(MI355Q_BAD_SHAPE, "empty reduction unit");
  if (unit_len > 0x7FFFFFFFLL - 64) return fail(MI355Q_UNSUPPORTED, "unit_len too large");
  if (!x || !clip_out) return fail(MI355Q_BAD_ARG, "null pointer");
  const size_t need = mi355q_octav_workspace_bytes(units, max_iter);
  if (!workspace || workspace_bytes < need)
    return fail(MI355Q_BAD_ARG, "workspace too small: need %zu bytes", need);
  hipStream_t st = as_stream(stream);
  float* hist = static_cast<float*>(workspace);
  unsigned long long* not_close = reinterpret_cast<unsigned long long*>(hist + (units * max_iter + 1) / 2 * 2);
  if (hipMemsetAsync(not_close, 0, sizeof(unsigned long long), st) != hipSuccess)
    return fail(MI355Q_HIP_ERROR, "hipMemsetAsync failed");
  // scale = np.asarray(4.0 ** (-bits) / exponent_divisor, dtype=np.float32)  (octav.py:64)
  double p4 = 1.0;
  for (int i = 0; i < bits; ++i) p4 *= 0.25;
  const float s = static_cast<float>(p4 / static_cast<double>(exponent_divisor));
  if ((unit_len == 32 || unit_len == 64 || unit_len == 128 || unit_len == 256) && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
      (units + kWave - 1) / kWave <= 0x7FFFFFFFLL && octav_unit_lanes_on() && !getenv("MI355Q_OCTAV_WAVE_KERNEL")) {
    // blockwise units: a lane per unit, the unit in registers (octav_unit_lanes_kernel)
    OctavArgs a{x, units, static_cast<int>(unit_len), 0, max_iter, count_is_f64, s, hist, not_close};
    const dim3 grid(static_cast<unsigned>((units + kWave - 1) / kWave));
    if (unit_len == 32) hipLaunchKernelGGL(octav_unit_lanes_kernel<32>, grid, dim3(kWave), 0, st, a);
    else if (unit_len == 64) hipLaunchKernelGGL(octav_unit_lanes_kernel<64>, grid, dim3(kWave), 0, st, a);
    else if (unit_len == 128) hipLaunchKernelGGL(octav_unit_lanes_kernel<128>, grid, dim3(kWave), 0, st, a);
    else hipLaunchKernelGGL(octav_unit_lanes_kernel<256>, grid, dim3(kWave), 0, st, a);
    MI355Q_CHECK_LAUNCH("octav unit lanes launch");
    hipLaunchKernelGGL(octav_pick_kernel, dim3(static_cast<unsigned>((units + 255) / 256)), dim3(256), 0, st,
                       hist, not_close, units, max_iter, early_stop, clip_out, iters_out);
    MI355Q_CHECK_LAUNCH("octav pick launch");
    return MI355Q_OK;
  }
  if (unit_len >= kRowsMinLen && unit_len <= kRowsMaxLen && !getenv("MI355Q_OCTAV_WAVE_KERNEL")) {
    // rows of a weight matrix: lanes own 64-element batches of the LDS-resident row
    // late iterations on a one-wave tail kernel (octav_tail_kernel) when the caller's workspace has room for
    // the hand-over (mi355q_octav_rows_workspace_bytes); MI355Q_OCTAV_TAIL=0 keeps every iteration in the rows kernel
    static const bool tail_on = [] { const char* e = getenv("MI355Q_OCTAV_TAIL"); return e == nullptr || atoi(e) != 0; }();
    const int threads = octav_rows_threads(static_cast<int>(unit_len));
    OctavArgs a{x, units, static_cast<int>(unit_len), 0, max_iter, count_is_f64, s, hist, not_close, nullptr, nullptr, nullptr, 0};
    a.tail_cap = octav_tail_cap(a.len);
    const int slots_ = ((a.len + kPiece - 1) / kPiece + threads - 1) / threads;
    if (tail_on && slots_ == 1 && workspace_bytes >= mi355q_octav_rows_workspace_bytes(units, unit_len, max_iter)) {
      unsigned char* tb = reinterpret_cast<unsigned char*>(workspace) + ((need + 63) & ~static_cast<size_t>(63));
      a.tail = reinterpret_cast<TailState*>(tb);
      a.tail_values = reinterpret_cast<float*>(tb + static_cast<size_t>(units) * sizeof(TailState));
      a.tail_pos = reinterpret_cast<unsigned short*>(a.tail_values + static_cast<size_t>(units) * 2 * a.tail_cap);
    }
    const size_t smem = octav_rows_smem(a.len);
    const int slots = ((a.len + kPiece - 1) / kPiece + threads - 1) / threads;   // 1 .. 4 (> 1 only with 256 threads)
    const int variant = threads == 1024 ? 6 : threads == 512 ? 5 : slots;   // (slots 2 .. 4: MI355Q_OCTAV_NARROW_ROWS)
    const void* fn = variant == 6 ? reinterpret_cast<const void*>(octav_rows_kernel<1, 1024>)
                   : variant == 5 ? reinterpret_cast<const void*>(octav_rows_kernel<1, 512>)
                   : slots == 4 ? reinterpret_cast<const void*>(octav_rows_kernel<4, 256>)
                   : slots == 3 ? reinterpret_cast<const void*>(octav_rows_kernel<3, 256>)
                   : slots == 2 ? reinterpret_cast<const void*>(octav_rows_kernel<2, 256>)
                   : threads == 256 ? reinterpret_cast<const void*>(octav_rows_kernel<1, 256>)
                   : threads == 128 ? reinterpret_cast<const void*>(octav_rows_kernel<1, 128>)
                                    : reinterpret_cast<const void*>(octav_rows_kernel<1, 64>);
    (void)variant;
    if (smem > 64 * 1024) {   // > 64 KB of dynamic LDS has to be asked for (per device: set on every call)
      const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return fail(MI355Q_HIP_ERROR, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    if (units > 0x7FFFFFFFLL) return fail(MI355Q_UNSUPPORTED, "too many units");
    void* kargs[] = {&a};
    if (hipLaunchKernel(fn, dim3(static_cast<unsigned>(units)), dim3(threads), kargs, smem, st) != hipSuccess)
      return fail(MI355Q_HIP_ERROR, "octav rows launch failed");
    MI355Q_CHECK_LAUNCH("octav rows launch");
    if (a.tail != nullptr) {
      hipLaunchKernelGGL(octav_tail_kernel, dim3(static_cast<unsigned>(units)), dim3(kWave), static_cast<size_t>(a.tail_cap) * 12, st, a);
      MI355Q_CHECK_LAUNCH("octav tail launch");
    }
    hipLaunchKernelGGL(octav_pick_kernel, dim3(static_cast<unsigned>((units + 255) / 256)), dim3(256), 0, st,
                       hist, not_close, units, max_iter, early_stop, clip_out, iters_out);
    MI355Q_CHECK_LAUNCH("octav pick launch");
    return MI355Q_OK;
  }
  if (unit_len >= 2 * kPiece && unit_len <= 512 && (unit_len & (unit_len - 1)) == 0 &&
      (units * unit_len) % kGroupLen == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
      units * unit_len / kGroupLen <= 0x7FFFFFFFLL && !getenv("MI355Q_OCTAV_WAVE_KERNEL")) {
    // blockwise units: 4096 contiguous elements (8 .. 128 whole units) per workgroup
    OctavArgs a{x, units, static_cast<int>(unit_len), 0, max_iter, count_is_f64, s, hist, not_close};
    const size_t smem = octav_groups_smem();
    hipLaunchKernelGGL(octav_groups_kernel, dim3(static_cast<unsigned>(units * unit_len / kGroupLen)), dim3(kGroupThreads), smem,
                       st, a, static_cast<int>(unit_len));
    MI355Q_CHECK_LAUNCH("octav groups launch");
    hipLaunchKernelGGL(octav_pick_kernel, dim3(static_cast<unsigned>((units + 255) / 256)), dim3(256), 0, st,
                       hist, not_close, units, max_iter, early_stop, clip_out, iters_out);
    MI355Q_CHECK_LAUNCH("octav pick launch");
    return MI355Q_OK;
  }
  const UnitPlan p = plan_units(static_cast<int>(unit_len));
  OctavArgs a{x, units, static_cast<int>(unit_len), p.lds_stride, max_iter, count_is_f64, s, hist, not_close};
  const dim3 blk(p.waves * kWave);
  const dim3 grid(static_cast<unsigned>((units + p.waves - 1) / p.waves));
  if (p.use_lds)
    hipLaunchKernelGGL(octav_kernel<true>, grid, blk, p.smem, st, a);
  else
    hipLaunchKernelGGL(octav_kernel<false>, grid, blk, 0, st, a);
  MI355Q_CHECK_LAUNCH("octav launch");
  hipLaunchKernelGGL(octav_pick_kernel, dim3(static_cast<unsigned>((units + 255) / 256)), dim3(256), 0, st,
                     hist, not_close, units, max_iter, early_stop, clip_out, iters_out);
  MI355Q_CHECK_LAUNCH("octav pick launch");
  return MI355Q_OK;
}

extern "C" int32_t mi355q_octav_clip_fast_f32(const float* x, int64_t units, int64_t unit_len, int32_t bits,
                                              int32_t max_iter, float exponent_divisor, int32_t early_stop,
                                              float* clip_out, int32_t* iters_out, void* workspace,
                                              size_t workspace_bytes, void* stream) {
  clear_error();
  if (units < 0 || unit_len < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  if (max_iter < 1 || max_iter > 64) return fail(MI355Q_BAD_ARG, "max_iter must be in [1, 64]");
  if (bits < 1 || bits > 16) return fail(MI355Q_BAD_ARG, "bits must be in [1, 16]");
  if (units == 0) return MI355Q_OK;
  if (unit_len == 0) return fail(MI355Q_BAD_SHAPE, "empty reduction unit");
  if (unit_len % 4 != 0 || unit_len > 65536)
    return fail(MI355Q_UNSUPPORTED, "the one-read OCTAV kernel takes units of 4 .. 65536 elements, a multiple of 4 (got %lld):"
                                    " use mi355q_octav_clip_f32", static_cast<long long>(unit_len));
  if (units > 0x7FFFFFFFLL) return fail(MI355Q_UNSUPPORTED, "too many units");
  if (!x || !clip_out) return fail(MI355Q_BAD_ARG, "null pointer");
  if (reinterpret_cast<uintptr_t>(x) & 15u) return fail(MI355Q_BAD_ARG, "x must be 16-byte aligned");
  const size_t need = mi355q_octav_workspace_bytes(units, max_iter);
  if (!workspace || workspace_bytes < need)
    return fail(MI355Q_BAD_ARG, "workspace too small: need %zu bytes", need);
  hipStream_t st = as_stream(stream);
  float* hist = static_cast<float*>(workspace);
  unsigned long long* moving = reinterpret_cast<unsigned long long*>(hist + (units * max_iter + 1) / 2 * 2);
  if (hipMemsetAsync(moving, 0, sizeof(unsigned long long), st) != hipSuccess)
    return fail(MI355Q_HIP_ERROR, "hipMemsetAsync failed");
  double p4 = 1.0;
  for (int i = 0; i < bits; ++i) p4 *= 0.25;
  const float s = static_cast<float>(p4 / static_cast<double>(exponent_divisor));
  const OctavFastArgs a{x, units, static_cast<int>(unit_len), max_iter, s, hist, moving};
  const int len4 = static_cast<int>(unit_len / 4);
  if (len4 <= 4) launch_octav_fast<1, 4>(a, st);
  else if (len4 <= 8) launch_octav_fast<2, 4>(a, st);
  else if (len4 <= 16) launch_octav_fast<4, 4>(a, st);
  else if (len4 <= 32) launch_octav_fast<8, 4>(a, st);
  else if (len4 <= 64) launch_octav_fast<16, 4>(a, st);
  else if (len4 <= 128) launch_octav_fast<32, 4>(a, st);
  else if (len4 <= 256) launch_octav_fast<64, 4>(a, st);
  else if (len4 <= 512) launch_octav_fast<64, 8>(a, st);
  else if (len4 <= 1024) launch_octav_fast<64, 16>(a, st);
  else if (len4 <= 2048) launch_octav_fast<256, 8>(a, st);
  else if (len4 <= 4096) launch_octav_fast<256, 16>(a, st);
  else if (len4 <= 8192) launch_octav_fast<1024, 8>(a, st);
  else launch_octav_fast<1024, 16>(a, st);
  MI355Q_CHECK_LAUNCH("octav one-read launch");
  hipLaunchKernelGGL(octav_pick_kernel, dim3(static_cast<unsigned>((units + 255) / 256)), dim3(256), 0, st,
                     hist, moving, static_cast<long long>(units), max_iter, early_stop, clip_out, iters_out);
  MI355Q_CHECK_LAUNCH("octav pick launch");
  return MI355Q_OK;
}

extern "C" int32_t mi355q_octav_clip_nd_f32(const float* x, int64_t outer, int64_t channels,
                                            int64_t inner, int32_t bits, int32_t max_iter,
                                            float exponent_divisor, int32_t early_stop, float* clip_out,
                                            int32_t* iters_out, void* workspace,
                                            size_t workspace_bytes, void* stream) {
  if (outer == 1)  // contiguous units: the LDS-staged kernel
    return mi355q_octav_clip_f32(x, channels, inner, bits, max_iter, exponent_divisor, early_stop, 1,
                                 clip_out, iters_out, workspace, workspace_bytes, stream);
  clear_error();
  if (outer < 0 || channels < 0 || inner < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  if (max_iter < 1 || max_iter > 64) return fail(MI355Q_BAD_ARG, "max_iter must be in [1, 64]");
  if (bits < 1 || bits > 16) return fail(MI355Q_BAD_ARG, "bits must be in [1, 16]");
  if (channels == 0) return MI355Q_OK;
  if (outer == 0 || inner == 0) return fail(MI355Q_BAD_SHAPE, "empty reduction unit");
  if (outer > 0x7FFFFFFFLL || inner > 0x7FFFFFFFLL - 64 || outer * inner > 0x7FFFFFFFLL)
    return fail(MI355Q_UNSUPPORTED, "reduction unit too large");
  if (!x || !clip_out) return fail(MI355Q_BAD_ARG, "null pointer");
  const size_t need = mi355q_octav_workspace_bytes(channels, max_iter);
  if (!workspace || workspace_bytes < need)
    return fail(MI355Q_BAD_ARG, "workspace too small: need %zu bytes", need);
  hipStream_t st = as_stream(stream);
  float* hist = static_cast<float*>(workspace);
  unsigned long long* not_close = reinterpret_cast<unsigned long long*>(hist + (channels * max_iter + 1) / 2 * 2);
  if (hipMemsetAsync(not_close, 0, sizeof(unsigned long long), st) != hipSuccess)
    return fail(MI355Q_HIP_ERROR, "hipMemsetAsync failed");
  double p4 = 1.0;
  for (int i = 0; i < bits; ++i) p4 *= 0.25;
  const float s = static_cast<float>(p4 / static_cast<double>(exponent_divisor));
  // an axis is always given in this form, so s * N is evaluated in float64
  if (inner == 1) {
    OctavArgs a{x, channels, static_cast<int>(outer), 0, max_iter, 1, s, hist, not_close};
    hipLaunchKernelGGL(octav_cols_kernel, dim3(static_cast<unsigned>((channels + 255) / 256)), dim3(256), 0,
                       st, a, static_cast<long long>(channels));
  } else {
    OctavArgs a{x, channels, static_cast<int>(inner), 0, max_iter, 1, s, hist, not_close};
    hipLaunchKernelGGL(octav_seg_kernel, dim3(static_cast<unsigned>((channels + 3) / 4)), dim3(4 * kWave), 0,
                       st, a, static_cast<long long>(channels), static_cast<long long>(outer));
  }
  MI355Q_CHECK_LAUNCH("octav nd launch");
  hipLaunchKernelGGL(octav_pick_kernel, dim3(static_cast<unsigned>((channels + 255) / 256)), dim3(256), 0, st,
                     hist, not_close, static_cast<long long>(channels), max_iter, early_stop, clip_out,
                     iters_out);
  MI355Q_CHECK_LAUNCH("octav pick launch");
  return MI355Q_OK;
}

extern "C" int32_t mi355q_mse_scale_nd_f32(const float* x, int64_t outer, int64_t channels,
                                           int64_t inner, float multiplier, float* scale_out,
                                           void* stream) {
  if (outer == 1) return mi355q_mse_scale_f32(x, channels, inner, multiplier, scale_out, stream);
  clear_error();
  if (outer < 0 || channels < 0 || inner < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  if (channels == 0) return MI355Q_OK;
  if (outer == 0 || inner == 0) return fail(MI355Q_BAD_SHAPE, "empty reduction unit");
  if (inner > 0x7FFFFFFFLL - 64) return fail(MI355Q_UNSUPPORTED, "inner too large");
  if (!x || !scale_out) return fail(MI355Q_BAD_ARG, "null pointer");
  hipStream_t st = as_stream(stream);
  if (inner == 1)
    hipLaunchKernelGGL(mse_scale_cols_kernel, dim3(static_cast<unsigned>((channels + 255) / 256)), dim3(256),
                       0, st, x, static_cast<long long>(outer), static_cast<long long>(channels),
                       multiplier, scale_out);
  else
    hipLaunchKernelGGL(mse_scale_seg_kernel, dim3(static_cast<unsigned>((channels + 3) / 4)), dim3(4 * kWave),
                       0, st, x, static_cast<long long>(outer), static_cast<long long>(channels),
                       static_cast<int>(inner), multiplier, scale_out);
  MI355Q_CHECK_LAUNCH("mse nd launch");
  return MI355Q_OK;
}

extern "C" int32_t mi355q_mse_scale_f32(const float* x, int64_t units, int64_t unit_len,
                                        float multiplier, float* scale_out, void* stream) {
  clear_error();
  if (units < 0 || unit_len < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  if (units == 0) return MI355Q_OK;
  if (unit_len == 0) return fail(MI355Q_BAD_SHAPE, "empty reduction unit");
  if (unit_len > 0x7FFFFFFFLL - 64) return fail(MI355Q_UNSUPPORTED, "unit_len too large");
  if (!x || !scale_out) return fail(MI355Q_BAD_ARG, "null pointer");
  MseArgs a{x, units, static_cast<int>(unit_len), 0, multiplier, scale_out};
  hipStream_t st = as_stream(stream);
  const dim3 grid(static_cast<unsigned>((units + 3) / 4)), blk(4 * kWave);
  int leaf = 0, depth = 0;
  if (balanced_chunk(a.len - (a.len - 1) / kChunk * kChunk, &leaf, &depth))
    hipLaunchKernelGGL(mse_scale_balanced_kernel, grid, blk, 0, st, a, leaf, depth);
  else
    hipLaunchKernelGGL(mse_scale_leaves_kernel, grid, blk, 0, st, a);
  MI355Q_CHECK_LAUNCH("mse_scale launch");
  return MI355Q_OK;
}

extern "C" int32_t mi355q_mse_requant_f32(const float* x, int64_t units, int64_t unit_len, float multiplier, int32_t bits,
                                          int32_t narrow, float* scale_out, int8_t* q_out, void* stream) {
  clear_error();
  if (units < 0 || unit_len < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  if (bits < 2 || bits > 8) return fail(MI355Q_BAD_ARG, "bits must be in [2, 8]");
  if (units == 0) return MI355Q_OK;
  if (unit_len == 0) return fail(MI355Q_BAD_SHAPE, "empty reduction unit");
  if (unit_len > 0x7FFFFFFFLL - 64) return fail(MI355Q_UNSUPPORTED, "unit_len too large");
  if (!x || !scale_out || !q_out) return fail(MI355Q_BAD_ARG, "null pointer");
  int leaf = 0, depth = 0;
  const int len = static_cast<int>(unit_len);
  const bool one_kernel = unit_len % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(q_out) & 3) == 0 &&
                          balanced_chunk(len - (len - 1) / kChunk * kChunk, &leaf, &depth) && !getenv("MI355Q_MSE_TWO_KERNELS");
  if (!one_kernel) {      // the scale kernel of any length, then the quantizer every other algorithm uses
    const int32_t st = mi355q_mse_scale_f32(x, units, unit_len, multiplier, scale_out, stream);
    if (st != MI355Q_OK) return st;
    return mi355q_quantize_f32(x, 1, units, unit_len, scale_out, 0, nullptr, 1, bits, narrow, 8, q_out, stream);
  }
  MseArgs a{x, units, len, 0, multiplier, scale_out};
  a.q = reinterpret_cast<unsigned*>(q_out);
  a.hi = static_cast<float>((1 << (bits - 1)) - 1);
  a.lo = -static_cast<float>(1 << (bits - 1)) + (narrow ? 1.f : 0.f);
  hipLaunchKernelGGL(mse_scale_balanced_kernel, dim3(static_cast<unsigned>((units + 3) / 4)), dim3(4 * kWave), 0, as_stream(stream), a,
                     leaf, depth);
  MI355Q_CHECK_LAUNCH("mse requant launch");
  return MI355Q_OK;
}
