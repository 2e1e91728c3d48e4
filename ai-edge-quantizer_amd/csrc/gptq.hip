// GPTQ on gfx950: Hessian (K8), damped inverse through a blocked FP64 Cholesky
// (K9) and the column-serial OBS weight update (K10).
//
//   ref: algorithms/uniform_quantize/gptq.py:100-107  H = (2/num_samples) X^T X
//   ref: algorithms/uniform_quantize/gptq.py:111-128  _prepare_hessian_inverse
//   ref: algorithms/uniform_quantize/gptq.py:131-216  _apply_gptq
//   ref: utils/qsv_utils.py:71-88                     _gptq_merge_hessian
//
// Dense contractions go through the MFMA GEMMs of gemm.hip; everything that is
// O(d^2) or column-serial is plain VALU code. The reference computes X^T X with
// sgemm (FP32), scales it into FLOAT64 (the `2.0 / np.array(n)` factor promotes),
// factors in FLOAT64 and inverts through FP32 LAPACK; here the whole inverse is
// done in FP64 and cast to FP32 at the end (tolerance class T2).
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "gemm.h"

namespace mi355q {
namespace {

constexpr int NB = 64;  // Cholesky / TRTRI block size and the reference's GPTQ blocksize

// ------------------------------------------------------------ Hessian ----
// H = alpha * P for a symmetric P of which only the lower triangle (j <= i) was computed:
// 32 x 32 tiles, the upper ones read their mirror tile through LDS (coalesced both ways).
__global__ __launch_bounds__(256) void mirror_scale_to_f64_kernel(const float* __restrict__ p, int d,
                                                                 double alpha, double* __restrict__ h) {
  __shared__ float tile[32][33];
  const int bi = blockIdx.y, bj = blockIdx.x;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  const bool upper = bj > bi;
  const int si = upper ? bj : bi, sj = upper ? bi : bj;         // source tile (lower triangle)
  for (int r = ty; r < 32; r += 8) {
    const int i = si * 32 + r, j = sj * 32 + tx;
    float v = 0.0f;
    if (i < d && j < d) v = p[static_cast<long long>(i >= j ? i : j) * d + (i >= j ? j : i)];
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int i = bi * 32 + r, j = bj * 32 + tx;
    if (i < d && j < d)
      h[static_cast<long long>(i) * d + j] = alpha * static_cast<double>(upper ? tile[tx][r] : tile[r][tx]);
  }
}

// (h0*n0 + h1*n1) / (n0+n1) in FP64, as NumPy evaluates it.
__global__ __launch_bounds__(256) void hessian_merge_kernel(const double* __restrict__ h0, double n0,
                                                           const double* __restrict__ h1, double n1,
                                                           long long n, double* __restrict__ out) {
  const double total = n0 + n1;
  const long long stride = static_cast<long long>(gridDim.x) * 256;
  for (long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride)
    out[i] = (h0[i] * n0 + h1[i] * n1) / total;
}

// ---------------------------------------------------- damping + copy ----
// sum of where(diag, diag, 1.0) -> out[0]
// (S = double: the Hessian itself, alpha = 1; S = float: the float32 product X^T X it is alpha times of --
// alpha * double(p) is exactly what mi355q_gptq_xtx_finish_f64 would have stored)
template <typename S>
__global__ __launch_bounds__(256) void diag_sum_kernel(const S* __restrict__ h, double alpha, int d, double* out) {
  __shared__ double part[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < d; i += 256) {
    const double v = alpha * static_cast<double>(h[static_cast<long long>(i) * d + i]);
    s += (v != 0.0) ? v : 1.0;
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = part[0];
}

// a = lower(h) with the damped diagonal; the band of kZeroBand elements above the diagonal zeroed.
// Nothing reads `a` further up: every kernel touches the upper triangle only inside tiles (<= 128
// wide, at offsets that are multiples of 64) that straddle the diagonal; a third of the copy's
// traffic was zeros nobody looked at (tests/test_gpu_gptq.py poisons the workspace to prove it).
constexpr int kZeroBand = 256;
template <typename S>
__global__ __launch_bounds__(256) void copy_damped_lower_kernel(const S* __restrict__ h, double alpha, int d,
                                                               const double* __restrict__ diag_sum,
                                                               double damp, double* __restrict__ a) {
  const double add = damp * (diag_sum[0] / static_cast<double>(d));
  const long long n = static_cast<long long>(d) * d;
  const long long stride = static_cast<long long>(gridDim.x) * 256;
  for (long long e = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; e < n; e += stride) {
    const int i = static_cast<int>(e / d), j = static_cast<int>(e % d);
    double v = 0.0;
    if (j < i) v = alpha * static_cast<double>(h[e]);
    if (j == i) { v = alpha * static_cast<double>(h[e]); v = ((v != 0.0) ? v : 1.0) + add; }
    if (j - i < kZeroBand) a[e] = v;
  }
}

// The same for columns [c0, c1) only (grid: x over the columns, y over groups of 8 rows): with a
// look-ahead factorization the first outer block's columns are copied on the caller's stream and
// everything behind them on the side stream, underneath the first block's chain of small kernels.
template <typename S>
__global__ __launch_bounds__(256) void copy_damped_lower_cols_kernel(const S* __restrict__ h, double alpha, int d,
                                                                    const double* __restrict__ diag_sum, double damp,
                                                                    double* __restrict__ a, int c0, int c1) {
  const double add = damp * (diag_sum[0] / static_cast<double>(d));
  const int j = c0 + static_cast<int>(blockIdx.x) * 256 + static_cast<int>(threadIdx.x);
  if (j >= c1) return;
  const int i_end = min(d, (static_cast<int>(blockIdx.y) + 1) * 8);
  for (int i = static_cast<int>(blockIdx.y) * 8; i < i_end; ++i) {
    const long long e = static_cast<long long>(i) * d + j;
    double v = 0.0;
    if (j < i) v = alpha * static_cast<double>(h[e]);
    if (j == i) { v = alpha * static_cast<double>(h[e]); v = ((v != 0.0) ? v : 1.0) + add; }
    if (j - i < kZeroBand) a[e] = v;
  }
}

// ---------------------------------------------------------- Cholesky ----
// The 64-column step of the blocked factorization is a serial chain of small kernels
// (diagonal block -> panel below it -> trailing update), so each link is latency-tuned and the
// chain holds nothing that can run elsewhere:
//   potf2_kernel        factors the nb x nb diagonal block in registers (one workgroup);
//   trsm_panel_kernel   solves L21 L11^T = A21 in place, one wave per 64 rows, by forward
//                       substitution against L11 held in LDS (no inverse of L11 is needed);
//   diag_inverse_kernel inverts all diagonal blocks of the finished factor in one launch, off
//                       the chain: the base level of the triangular inverse.
__device__ __forceinline__ int tri(int r, int c) { return r * (r + 1) / 2 + c; }  // c <= r

__device__ __forceinline__ float lane_bcast32(float v, int src_lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

__device__ __forceinline__ double lane_bcast64(double v, int src_lane) {
  const long long bits = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane(static_cast<int>(bits), src_lane);
  const int hi = __builtin_amdgcn_readlane(static_cast<int>(bits >> 32), src_lane);
  return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}

#if defined(MI355Q_POTF2_PROF)   // tools/kbench/potf2_bench.hip: phase stamps
#define MI355Q_STAMP() do { long long now__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now__) : : "memory"); \
    if (prof && threadIdx.x == 0) prof[stamp] = now__; ++stamp; } while (0)
#define MI355Q_PROF_ARG , static_cast<long long*>(nullptr)
#else
#define MI355Q_STAMP() do { } while (0)
#define MI355Q_PROF_ARG
#endif

// Unblocked lower Cholesky of the nb x nb diagonal block at (k, k), in place. info != 0 if a
// pivot is not positive (LAPACK's convention: 1 + its index); `lt` (NB x NB) receives the
// transposed factor, reciprocals on its diagonal, for the panel solve.
//
// Lane = row, wave w owns columns 16w..16w+15 of every row (a short block is padded with the
// identity, which factors to itself: straight-line code). A 16-column panel is factored inside
// its wave and applied to the waves on its right as a rank-16 update (4 barriers in all). Inside
// the panel the factorization is root-free: column j stays unscaled (u), its multipliers are
// m = u / p_j (hardware reciprocal + two Newton steps), and v_i -= u m_i. What this kernel's
// time is made of is the chain from one pivot to the next and the cross-lane traffic around it
// (a v_readlane pair and the use of its result cost ~30 cycles), so only the pivot and the next
// column's multiplier travel by v_readlane; the other multipliers go through LDS (one write,
// wave-uniform reads) and are used a column later. The 1/sqrt(p_j) scaling that turns u into L
// is done for all 64 columns at once at the end.
#ifndef MI355Q_POTF2_PW     // columns per wave (tuning hook: 16 -> 4 waves, 8 -> 8 waves)
#define MI355Q_POTF2_PW 8
#endif
constexpr int kPW = MI355Q_POTF2_PW, kPotf2Waves = NB / kPW, kPotf2Threads = 64 * kPotf2Waves;

struct Potf2Shared {
  double PU[2][NB][kPW + 1];                              // the last two panels, unscaled: [row][column]
  __attribute__((aligned(16))) double PM[2][kPW][NB];     // their multipliers: [column][row]
  __attribute__((aligned(16))) double piv[NB], ys[NB];
  int bad[kPotf2Waves];
};

// The factorization proper, by kPotf2Threads threads: v = this thread's kPW values of the
// (identity padded) block on entry, the unscaled columns u on exit; sh.ys[c] = 1 / sqrt(p_c), so
// L[r][c] = v * ys[c]; returns the index of the first non-positive pivot, NB if there is none (to
// every thread).
__device__ __forceinline__ int potf2_core(double (&v)[kPW], Potf2Shared& sh, int t) {
  const int r = t & 63, cq = t >> 6;
#pragma unroll
  for (int jq = 0; jq < kPotf2Waves; ++jq) {
    double (*pu)[kPW + 1] = sh.PU[jq & 1];
    double (*pm)[NB] = sh.PM[jq & 1];
    if (cq == jq) {  // wave-uniform
      // (every instruction of this loop costs the chain ~8 cycles x 64 columns: the pivot test and
      // the pivots themselves are taken from the finished panel below, not tracked per column)
#pragma unroll
      for (int ji = 0; ji < kPW; ++ji) {
        const int j = jq * kPW + ji;
        const double u = v[ji];
        const double p = lane_bcast64(u, j);
        double rp = __builtin_amdgcn_rcp(p);                      // ~2^-40 relative (measured): one Newton step
        rp = __builtin_fma(rp, __builtin_fma(-p, rp, 1.0), rp);
        const double m = u * rp;
        if (ji + 1 < kPW) v[ji + 1] = __builtin_fma(-u, lane_bcast64(m, j + 1), v[ji + 1]);
        pm[ji][r] = m;
        __builtin_amdgcn_wave_barrier();   // same wave: the LDS unit keeps the order, the compiler must too
#pragma unroll
        for (int i = ji + 2; i < kPW; ++i) v[i] = __builtin_fma(-u, pm[ji][jq * kPW + i], v[i]);
      }
#pragma unroll
      for (int i = 0; i < kPW; ++i) pu[r][i] = v[i];
      __builtin_amdgcn_wave_barrier();
      // the panel's pivots sit on its diagonal: u_j of row j is p_j
      const bool mine = r / kPW == jq;
      const double pv = pu[r][r % kPW];
      if (mine) sh.piv[r] = pv;
      const unsigned long long nonpos = __ballot(mine && !(pv > 0.0));
      if (r == 0) sh.bad[jq] = nonpos ? __builtin_ctzll(nonpos) : NB;
    }
    __syncthreads();
    if (cq > jq) {
      double mine[kPW];
#pragma unroll
      for (int jj = 0; jj < kPW; ++jj) mine[jj] = pu[r][jj];
#pragma unroll
      for (int i = 0; i < kPW; ++i) {
        const int c = cq * kPW + i;
#pragma unroll
        for (int jj = 0; jj < kPW; ++jj) v[i] = __builtin_fma(-mine[jj], pm[jj][c], v[i]);
      }
    }
  }
  // L = u / sqrt(p): 1/sqrt(p) by hardware estimate + two Newton steps, one pivot per thread
  if (t < NB) {
    const double p = sh.piv[t];
    double y = __builtin_amdgcn_rsq(p);
    y = y * (1.5 - 0.5 * p * y * y);
    y = y * (1.5 - 0.5 * p * y * y);   // (off the chain: kept at two steps)
    sh.ys[t] = y;
  }
  __syncthreads();
  int j = NB;
#pragma unroll
  for (int q = kPotf2Waves - 1; q >= 0; --q) j = sh.bad[q] < NB ? sh.bad[q] : j;
  return j;
}

__global__ __launch_bounds__(kPotf2Threads) void potf2_kernel(double* __restrict__ a, int d, int k, int nb, int* info,
                                                   double* __restrict__ lt
#if defined(MI355Q_POTF2_PROF)
                                                   , long long* prof
#endif
                                                   ) {
  __shared__ Potf2Shared sh;
  const int t = threadIdx.x, r = t & 63, cq = t >> 6;
#if defined(MI355Q_POTF2_PROF)
  int stamp = 0;
#endif
  MI355Q_STAMP();
  double v[kPW];
  const double* row = a + static_cast<long long>(k + (r < nb ? r : 0)) * d + k;
#pragma unroll
  for (int i = 0; i < kPW; ++i) {
    const int c = cq * kPW + i;
    const double g = row[c < nb ? c : 0];           // clamped, not predicated: no branch per load
    v[i] = (r < nb && c < nb) ? (c <= r ? g : 0.0) : (c == r ? 1.0 : 0.0);
  }
  MI355Q_STAMP();
  const int first_bad = potf2_core(v, sh, t);
  MI355Q_STAMP();
  double* out = a + static_cast<long long>(k + (r < nb ? r : 0)) * d + k;
#pragma unroll
  for (int i = 0; i < kPW; ++i) {
    const int c = cq * kPW + i;
    const double y = sh.ys[c];
    const double l = v[i] * y;                  // on the diagonal u = p: sqrt(p)
    if (r < nb && c <= r) out[c] = l;
    lt[c * NB + r] = c == r ? y : l;            // 1 / L_jj = 1 / sqrt(p_j)
  }
  if (t == 0 && first_bad < nb) atomicCAS(info, 0, k + first_bad + 1);
  MI355Q_STAMP();
}

// L21 := A21 inv(L11)^T in place for the m rows below the diagonal block: one wave per 64 rows,
// lane = row, the row's 64 values in registers (128 of the 256 VGPRs an ALU operand can name).
// potf2_kernel left L11 transposed (1 / L_jj on the diagonal) in `lt`, so that step j
// (x_j /= L_jj, then x_i -= x_j L_ij for i > j) reads one contiguous LDS column with
// wave-uniform (broadcast) addresses. Straight-line code over (column, 16-row chunk) items,
// software pipelined by hand: the chunks of the next two items are in flight while this item's
// FMAs issue. One wave issues one instruction every ~4.5 cycles whatever its kind, so the
// solve costs (2016 FMAs + 1056 LDS reads + waits) x 4.5 cycles: ~17 k cycles per tile.
typedef double Pair __attribute__((ext_vector_type(2)));

// Item N of the solve: column J against rows 16 Q .. 16 Q + 15 (template recursion instead of
// loops: every register index has to be a compile-time constant). The empty asm statements pin
// the schedule: LDS reads cannot cross the first (its pointer operand makes the never-escaping
// LDS array "memory"), and the FMAs of this item cannot sink below the second, which would keep
// every chunk alive until its rows' own columns come up.
// (Tried: the wave-uniform L values as scalar operands through s_load. The 16 KB of L11 do not
// stay in the scalar cache next to 30 other waves' copies, s_waitcnt lgkmcnt(0) is the only wait
// there is for scalar loads, and 32 SGPRs per item leave no room to request further ahead:
// 31.6 k cycles per tile against 17 k through LDS.)
template <int J, int Q, int N>
__device__ __forceinline__ void trsm_item(double (&x)[NB], Pair (&buf)[3][8], const Pair* Lt) {
  constexpr int NJ = (Q + 1 == 4) ? J + 1 : J;               // the next item
  constexpr int NQ = (Q + 1 == 4) ? NJ / 16 : Q + 1;
  constexpr int PJ = (NQ + 1 == 4) ? NJ + 1 : NJ;            // the one after it: its chunk is requested now
  constexpr int PQ = (NQ + 1 == 4) ? PJ / 16 : NQ + 1;
  if constexpr (NJ < NB && PJ < NB) {
#pragma unroll
    for (int u = 0; u < 8; ++u) buf[(N + 2) % 3][u] = Lt[(PJ * NB + PQ * 16) / 2 + u];
  }
  asm volatile("" : "+v"(x[J]) : "v"(Lt) : "memory");
  const Pair* c = buf[N % 3];
  if constexpr (Q == J / 16) x[J] *= (J & 1) ? c[(J % 16) / 2].y : c[(J % 16) / 2].x;
#pragma unroll
  for (int i = Q * 16; i < Q * 16 + 16; ++i) {
    if (i > J) x[i] = __builtin_fma(-x[J], (i & 1) ? c[(i % 16) / 2].y : c[(i % 16) / 2].x, x[i]);
  }
  constexpr int B = Q * 16;
  asm volatile("" : "+v"(x[B]), "+v"(x[B + 1]), "+v"(x[B + 2]), "+v"(x[B + 3]), "+v"(x[B + 4]), "+v"(x[B + 5]),
               "+v"(x[B + 6]), "+v"(x[B + 7]), "+v"(x[B + 8]), "+v"(x[B + 9]), "+v"(x[B + 10]), "+v"(x[B + 11]),
               "+v"(x[B + 12]), "+v"(x[B + 13]), "+v"(x[B + 14]), "+v"(x[B + 15]));
  if constexpr (NJ < NB) trsm_item<NJ, NQ, N + 1>(x, buf, Lt);
}

__global__ __launch_bounds__(64) void trsm_panel_kernel(double* __restrict__ a, int d, int k, int nb, int m,
                                                       const double* __restrict__ lt
#if defined(MI355Q_POTF2_PROF)
                                                       , long long* prof
#endif
                                                       ) {
#if defined(MI355Q_POTF2_PROF)
  int stamp = 8;
  if (blockIdx.x != 0) prof = nullptr;
#endif
  MI355Q_STAMP();
  __shared__ Pair Lt[NB * NB / 2];     // Lt[(j * NB + i) / 2] = (L11[i][j], L11[i + 1][j]), identity padded
  __shared__ double X[NB * (NB + 1)];  // the 64 x 64 tile of A21 on its way between row-per-lane and coalesced
  const int t = threadIdx.x;
  const long long r0 = static_cast<long long>(blockIdx.x) * 64;      // first row of this tile (below the block)
  const int rows = m - r0 < 64 ? static_cast<int>(m - r0) : 64;
  double* tile = a + (static_cast<long long>(k + nb) + r0) * d + k;
  // global -> LDS with lane = column (512 contiguous bytes per row), then LDS -> registers with lane = row
  // (clamped addresses instead of predicated loads: a branch per load would serialize 64 round trips)
  const int tc = t < nb ? t : nb - 1;
  {
    double g[NB];
#pragma unroll
    for (int rr = 0; rr < NB; ++rr) g[rr] = tile[static_cast<long long>(rr < rows ? rr : rows - 1) * d + tc];
    const Pair* src = reinterpret_cast<const Pair*>(lt);
#pragma unroll 8
    for (int e = t; e < NB * NB / 2; e += 64) Lt[e] = src[e];
#pragma unroll
    for (int rr = 0; rr < NB; ++rr) asm volatile("" : "+v"(g[rr]));   // "used" whatever the row count: 64 loads in flight, no branches
#pragma unroll
    for (int rr = 0; rr < NB; ++rr) X[rr * (NB + 1) + t] = (rr < rows && t < nb) ? g[rr] : 0.0;
  }
  __syncthreads();
  MI355Q_STAMP();
  double x[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) x[c] = X[t * (NB + 1) + c];
  Pair buf[3][8];
#pragma unroll
  for (int u = 0; u < 8; ++u) { buf[0][u] = Lt[u]; buf[1][u] = Lt[8 + u]; }   // items (0, 0) and (0, 1)
  trsm_item<0, 0, 0>(x, buf, Lt);
  MI355Q_STAMP();
#pragma unroll
  for (int c = 0; c < NB; ++c) X[t * (NB + 1) + c] = x[c];
  __syncthreads();
#pragma unroll 1
  for (int r16 = 0; r16 < NB; r16 += 16) {
    double g[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) g[i] = X[(r16 + i) * (NB + 1) + t];
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (r16 + i < rows && t < nb) tile[static_cast<long long>(r16 + i) * d + t] = g[i];
  }
  MI355Q_STAMP();
}

// A whole 64-column step -- diagonal block, panel solve, trailing update -- in ONE launch (whole
// tiles only; small d). A step's three kernels each cost 7-8 us before they do anything (launch,
// first loads, last stores), which is most of what a step costs at d = 2048. Here the workgroup
// that owns trailing tile (i, j) factors the diagonal block itself (all four waves, potf2_core),
// solves the two row tiles it needs itself -- wave 0 rows i, wave 1 rows j, against L11^T in LDS,
// with the same code as trsm_panel_kernel -- while its C tile is already on the way, then
// multiplies them out of LDS (FP64 MFMA, 16 per wave) and subtracts. The diagonal block is
// factored by every workgroup and row tile i solved by every workgroup of tile row i (up to 7
// times): the chip is idle at this point of the chain. Only workgroup (0, 0) writes L11 back, only
// the workgroups of tile column 0 their L21 tile.
// The panel solve of the step kernel: FOUR LANES PER ROW (a DPP quad, lane c of the quad owns
// columns 4k + c), 16 rows per wave, so that the eight waves of the workgroup solve its two row
// tiles together: a lane's 2016 FMAs become ~560, and x_j crosses the quad as two DPP moves.
// LtQ[(j * 4 + c) * 16 + k] = L11[4k + c][j] (1 / L_jj in the diagonal's slot): a lane's values of
// column j are contiguous. One column ahead in flight, schedule pinned as in trsm_item.
template <int CO>
__device__ __forceinline__ double quad_bcast64(double v) {
  const long long bits = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, static_cast<int>(bits), CO * 0x55, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, static_cast<int>(bits >> 32), CO * 0x55, 0xF, 0xF, false);
  return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}

template <int J>
__device__ __forceinline__ void trsm_quad_item(double (&x)[16], Pair (&buf)[2][8], const Pair* LtQ, int c4) {
  constexpr int KO = J >> 2, CO = J & 3;
  if constexpr (J + 1 < NB) {
    constexpr int NKO = (J + 1) >> 2;
#pragma unroll
    for (int u = NKO / 2; u < 8; ++u) buf[(J + 1) & 1][u] = LtQ[((J + 1) * 4 + c4) * 8 + u];
  }
  asm volatile("" : "+v"(x[KO]) : "v"(LtQ) : "memory");
  const Pair* c = buf[J & 1];
  const double lko = (KO & 1) ? c[KO / 2].y : c[KO / 2].x;   // the owner's: 1 / L_JJ; lanes right of it: L[4 KO + c4][J]
  const double own = x[KO] * lko;
  const double xj = quad_bcast64<CO>(own);
  x[KO] = c4 == CO ? own : (c4 > CO ? __builtin_fma(-xj, lko, x[KO]) : x[KO]);
#pragma unroll
  for (int k = KO + 1; k < 16; ++k) x[k] = __builtin_fma(-xj, (k & 1) ? c[k / 2].y : c[k / 2].x, x[k]);
  asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
               "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
  if constexpr (J + 1 < NB) trsm_quad_item<J + 1>(x, buf, LtQ, c4);
}

// rows 16 * part .. + 15 of the tile X ([row][column], row stride NB + 1), by one wave
__device__ __forceinline__ void trsm_quad_in_lds(double* __restrict__ X, const Pair* LtQ, int part, int lane) {
  const int c4 = lane & 3, row = part * 16 + (lane >> 2);
  double x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = X[row * (NB + 1) + 4 * k + c4];
  Pair buf[2][8];
#pragma unroll
  for (int u = 0; u < 8; ++u) buf[0][u] = LtQ[c4 * 8 + u];
  trsm_quad_item<0>(x, buf, LtQ, c4);
#pragma unroll
  for (int k = 0; k < 16; ++k) X[row * (NB + 1) + 4 * k + c4] = x[k];
}

// s[i][0:64] = a[k + i][k : k + 64], i < rows: a diagonal block and the panel below it, compact.
__global__ __launch_bounds__(256) void copy_panel_kernel(const double* __restrict__ a, int d, int k, int rows,
                                                        double* __restrict__ s) {
  const long long n = static_cast<long long>(rows) * NB;
  const long long stride = static_cast<long long>(gridDim.x) * 256;
  for (long long e = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; e < n; e += stride)
    s[e] = a[(k + e / NB) * d + k + e % NB];
}

__device__ __forceinline__ void trsm_tile_in_lds(double* __restrict__ X, const Pair* Lt, int lane) {
  double x[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) x[c] = X[lane * (NB + 1) + c];
  Pair buf[3][8];
#pragma unroll
  for (int u = 0; u < 8; ++u) { buf[0][u] = Lt[u]; buf[1][u] = Lt[8 + u]; }   // items (0, 0) and (0, 1)
  trsm_item<0, 0, 0>(x, buf, Lt);
#pragma unroll
  for (int c = 0; c < NB; ++c) X[lane * (NB + 1) + c] = x[c];
}

//
// What a step reads is rewritten (factored, solved) by one workgroup while others still read it,
// so it cannot be read from where L goes: the unsolved block column comes from `s_cur`
// ((64 + m) x 64, compact; tile 0 = the diagonal block), which the PREVIOUS step's tile-column-0
// workgroups filled with the columns they had just updated (`s_next` here; the first step of an
// outer block copies it from `a`: copy_panel_kernel).
// update == 0: the last step of an outer block (nothing left to update inside it): grid (m / 64, 1),
// diagonal block and panel solve only.
__global__ __launch_bounds__(kPotf2Threads) void chol_step_kernel(double* __restrict__ a, int d, int k, int* info,
                                                                 const double* __restrict__ s_cur, double* __restrict__ s_next,
                                                                 int update) {
  constexpr int T = kPotf2Threads, W = kPotf2Waves;
  constexpr int SC = NB / (W / 2), MB = SC / 16;     // a wave's part of the C tile: 32 rows x SC columns, 2 x MB MFMA tiles
  const int ti = blockIdx.x, tj = blockIdx.y;        // trailing tile (ti, tj): rows / columns k + 64 + 64 t ..
  if (tj > ti) return;
  __shared__ Potf2Shared sh;
  __shared__ Pair Lt[NB * NB / 2];                   // Lt[(j * NB + i) / 2] = (L11[i][j], L11[i + 1][j]), 1 / L_jj on the diagonal
  __shared__ double Xi[NB * (NB + 1)], Xj[NB * (NB + 1)];   // A21 row tiles i and j, [row][column of the panel]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // everything this workgroup reads is requested up front: the two row tiles, its C tile, the diagonal block
  double gi[NB * NB / T], gj[NB * NB / T];
  {
    const double* src_i = s_cur + static_cast<long long>(ti + 1) * NB * NB;
    const double* src_j = s_cur + static_cast<long long>(tj + 1) * NB * NB;
#pragma unroll
    for (int it = 0; it < NB * NB / T; ++it) {
      gi[it] = src_i[tid + T * it];
      gj[it] = src_j[tid + T * it];
    }
  }
  const int wy = wave / (W / 2), wx = wave % (W / 2);
  double* ctile = a + static_cast<long long>(k + NB + ti * NB) * d + (k + NB + tj * NB);
  double cold[2][MB][4] = {};                        // this thread's elements of it, MFMA accumulator layout
  if (update) {
#pragma unroll
    for (int ma = 0; ma < 2; ++ma)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          cold[ma][mb][r] = ctile[static_cast<long long>(wy * 32 + ma * 16 + (lane >> 4) + 4 * r) * d + wx * SC + mb * 16 + (lane & 15)];
  }
  {
    // the diagonal block: lane = row, wave = kPW-column panel
    const int r = lane, cq = wave;
    double v[kPW];
#pragma unroll
    for (int i = 0; i < kPW; ++i) {
      const int c = cq * kPW + i;
      const double g = s_cur[r * NB + c];
      v[i] = c <= r ? g : 0.0;
    }
    const int first_bad = potf2_core(v, sh, tid);
    double* ltd = reinterpret_cast<double*>(Lt);
    double* out = a + static_cast<long long>(k + r) * d + k;
#pragma unroll
    for (int i = 0; i < kPW; ++i) {
      const int c = cq * kPW + i;
      const double y = sh.ys[c];
      const double l = v[i] * y;                  // on the diagonal u = p: sqrt(p)
      if constexpr (W == 8) ltd[(c * 4 + (r & 3)) * 16 + (r >> 2)] = c == r ? y : l;   // trsm_quad_item's layout
      else ltd[c * NB + r] = c == r ? y : l;
      if (ti == 0 && tj == 0 && c <= r) out[c] = l;
    }
    if (ti == 0 && tj == 0 && tid == 0 && first_bad < NB) atomicCAS(info, 0, k + first_bad + 1);
  }
#pragma unroll
  for (int it = 0; it < NB * NB / T; ++it) {
    const int e = tid + T * it, rr = e >> 6, c = e & 63;
    Xi[rr * (NB + 1) + c] = gi[it];
    Xj[rr * (NB + 1) + c] = gj[it];
  }
  __syncthreads();
  if constexpr (W == 8) {
    if (wave < 4 || (tj != ti && update)) trsm_quad_in_lds(wave < 4 ? Xi : Xj, Lt, wave & 3, lane);
  } else {
    if (wave == 0 || (wave == 1 && tj != ti && update)) trsm_tile_in_lds(wave == 0 ? Xi : Xj, Lt, lane);
  }
  __syncthreads();
  if (update) {
  const double* XJ = tj == ti ? Xi : Xj;
  __attribute__((ext_vector_type(4))) double acc[2][MB];
#pragma unroll
  for (int ma = 0; ma < 2; ++ma)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[ma][mb] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k0 = 0; k0 < NB; k0 += 4) {
    double af[2], bf[MB];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) af[t2] = Xi[(wy * 32 + t2 * 16 + (lane & 15)) * (NB + 1) + k0 + (lane >> 4)];
#pragma unroll
    for (int t2 = 0; t2 < MB; ++t2) bf[t2] = XJ[(wx * SC + t2 * 16 + (lane & 15)) * (NB + 1) + k0 + (lane >> 4)];
#pragma unroll
    for (int ma = 0; ma < 2; ++ma)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
        acc[ma][mb] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ma], bf[mb], acc[ma][mb], 0, 0, 0);
  }
#pragma unroll
  for (int ma = 0; ma < 2; ++ma)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wy * 32 + ma * 16 + (lane >> 4) + 4 * r, col = wx * SC + mb * 16 + (lane & 15);
        const double cnew = cold[ma][mb][r] - acc[ma][mb][r];
        if (tj != ti || col <= row) ctile[static_cast<long long>(row) * d + col] = cnew;
        // tile column 0 is the next step's block column (its tile 0 the next diagonal block)
        if (tj == 0 && s_next != nullptr) s_next[static_cast<long long>(ti) * NB * NB + row * NB + col] = cnew;
      }
  }
  if (tj == 0) {   // L21 tile i goes back in place (coalesced: a row of the tile is 512 contiguous bytes)
    double* rows_i = a + static_cast<long long>(k + NB + ti * NB) * d + k;
#pragma unroll
    for (int it = 0; it < NB * NB / T; ++it) {
      const int e = tid + T * it, rr = e >> 6, c = e & 63;
      rows_i[static_cast<long long>(rr) * d + c] = Xi[rr * (NB + 1) + c];
    }
  }
}

// One level of the in-LDS inverse: every pair of adjacent inverted S-blocks becomes one
// inverted 2S-block. Fixed trip counts (terms outside the triangles are masked, the LDS index
// clamped) so the S loads of a dot product are all in flight at once.
template <int S>
__device__ __forceinline__ void merge_level(const double* __restrict__ Lp, double* __restrict__ Xp,
                                            double* __restrict__ T, int t) {
  constexpr int OUTS = (NB / 2) * S;  // pairs * S * S
  for (int o = t; o < OUTS; o += 256) {   // T = L21 X11
    const int bb = o % S, aa = (o / S) % S, p = (o / (S * S)) * 2 * S;
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < S; ++m) {
      const int mm = m < bb ? bb : m;
      const double term = Lp[tri(p + S + aa, p + mm)] * Xp[tri(p + mm, p + bb)];
      acc += m < bb ? 0.0 : term;
    }
    T[o] = acc;
  }
  __syncthreads();
  for (int o = t; o < OUTS; o += 256) {   // X21 = -X22 T
    const int bb = o % S, aa = (o / S) % S, pr = o / (S * S), p = pr * 2 * S;
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < S; ++m) {
      const int mm = m > aa ? aa : m;
      const double term = Xp[tri(p + S + aa, p + S + mm)] * T[(pr * S + mm) * S + bb];
      acc += m > aa ? 0.0 : term;
    }
    Xp[tri(p + S + aa, p + bb)] = -acc;
  }
  __syncthreads();
}

// Every diagonal NB-block of the finished factor := its inverse, in place (one workgroup per
// block): pairwise merging in LDS, levels s = 1, 2, ..., 32 with X21 = -X22 (L21 X11).
__global__ __launch_bounds__(256) void diag_inverse_kernel(double* __restrict__ a, int d, long long matrix_stride = 0) {
  a += static_cast<long long>(blockIdx.z) * matrix_stride;
  __shared__ double Lp[NB * (NB + 1) / 2];   // packed lower triangles
  __shared__ double Xp[NB * (NB + 1) / 2];
  __shared__ double T[NB * NB / 4];          // per level: all pairs' s x s products (32 * s values)
  const int t = threadIdx.x, k = blockIdx.x * NB;
  const int nb = d - k < NB ? d - k : NB;
  for (int e = t; e < NB * NB; e += 256) {
    const int r = e / NB, c = e % NB;
    if (c <= r) Lp[tri(r, c)] = (r < nb) ? a[static_cast<long long>(k + r) * d + k + c] : (c == r ? 1.0 : 0.0);
  }
  __syncthreads();
  if (t < NB) Xp[tri(t, t)] = 1.0 / Lp[tri(t, t)];
  __syncthreads();
  merge_level<1>(Lp, Xp, T, t);
  merge_level<2>(Lp, Xp, T, t);
  merge_level<4>(Lp, Xp, T, t);
  merge_level<8>(Lp, Xp, T, t);
  merge_level<16>(Lp, Xp, T, t);
  merge_level<32>(Lp, Xp, T, t);
  for (int e = t; e < NB * NB; e += 256) {
    const int r = e / NB, c = e % NB;
    if (r < nb && c <= r) a[static_cast<long long>(k + r) * d + k + c] = Xp[tri(r, c)];
  }
}
#undef MI355Q_STAMP

// h[i][j] = h[j][i] for j > i: the upper triangle of a float32 matrix from its lower one, 32 x 32
// tiles through LDS (coalesced both ways); one workgroup per tile strictly above the diagonal
// plus the diagonal tiles.
__global__ __launch_bounds__(256) void mirror_lower_f32_kernel(float* __restrict__ h, int d) {
  __shared__ float tile[32][33];
  const int bi = blockIdx.y, bj = blockIdx.x;          // destination tile (bi, bj), bj >= bi
  if (bj < bi) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int i = bj * 32 + r, j = bi * 32 + tx;       // source: the mirror tile in the lower triangle
    tile[r][tx] = (i < d && j < d) ? h[static_cast<long long>(i) * d + j] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int i = bi * 32 + r, j = bj * 32 + tx;
    if (i < d && j < d && j > i) h[static_cast<long long>(i) * d + j] = tile[tx][r];
  }
}

// ---- several equally sized inverses in lock step (d < 4096, a multiple of 64) ---------------------------------
// A d = 2048 inverse is a chain of 32 dependent 64-column steps; alone, each step is one launch whose every workgroup
// factors the diagonal block again (chol_step_kernel: free on an idle chip, 117 KB of LDS, one workgroup per CU).
// Eight such chains on eight streams (round 3) therefore share the chip badly: 0.67 ms per inverse where 0.11 would be
// the FP64 peak's. Here G matrices advance through the SAME step together, blockIdx.z = matrix, three launches per
// step whatever G is:
//   potf2_batched_kernel   one workgroup per matrix factors its diagonal block ONCE (potf2_core) and leaves L11^T in the
//                          quad layout of the panel solve in the matrix's `lt` slot;
//   solve_batched_kernel   one workgroup per 64-row tile of the panel below: the DPP-quad forward substitution of the
//                          step kernel (trsm_quad_in_lds), once per tile instead of once per trailing tile's workgroup;
//   update_batched_kernel  the rank-64 update of the outer block's remaining columns on FP64 MFMA, the step kernel's
//                          own product loop; 66 KB of LDS: two workgroups per CU.
// Every element sees the operations of the single call in the same order (the same device functions on the same
// values): bit-identical inverses (tests/test_gpu_gptq.py). The GEMMs between outer blocks, the triangular-inverse
// levels and L^-T L^-1 go out once for all matrices (GemmArgs::outer), the damped copy and the final mirror take the
// callers' G pointers as a kernel-argument table.
constexpr int kHinvGroup = 32;
struct HinvTable {
  const double* h[kHinvGroup];     // the Hessians
  float* out[kHinvGroup];          // their inverses
};

__global__ __launch_bounds__(256) void diag_sum_batched_kernel(HinvTable t, int d, double* __restrict__ scal, long long matrix_stride) {
  __shared__ double part[256];
  const double* h = t.h[blockIdx.z];
  double s = 0.0;
  for (int i = threadIdx.x; i < d; i += 256) {
    const double v = h[static_cast<long long>(i) * d + i];
    s += (v != 0.0) ? v : 1.0;
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) scal[static_cast<long long>(blockIdx.z) * matrix_stride] = part[0];
}

__global__ __launch_bounds__(256) void copy_damped_lower_batched_kernel(HinvTable t, int d, const double* __restrict__ scal, double damp,
                                                                       double* __restrict__ a, long long matrix_stride) {
  const double* h = t.h[blockIdx.z];
  a += static_cast<long long>(blockIdx.z) * matrix_stride;
  const double add = damp * (scal[static_cast<long long>(blockIdx.z) * matrix_stride] / static_cast<double>(d));
  const long long n = static_cast<long long>(d) * d;
  const long long stride = static_cast<long long>(gridDim.x) * 256;
  for (long long e = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; e < n; e += stride) {
    const int i = static_cast<int>(e / d), j = static_cast<int>(e % d);
    double v = 0.0;
    if (j < i) v = h[e];
    if (j == i) { v = h[e]; v = ((v != 0.0) ? v : 1.0) + add; }
    if (j - i < kZeroBand) a[e] = v;
  }
}

// the diagonal block at (k, k) of every matrix: L11 in place, L11^T (1 / L_jj on the diagonal) in trsm_quad_item's layout in `ltq`
__global__ __launch_bounds__(kPotf2Threads) void potf2_batched_kernel(double* __restrict__ a, long long matrix_stride, int d, int k,
                                                                     int* __restrict__ info, double* __restrict__ ltq) {
  static_assert(kPotf2Waves == 8, "the quad layout of the panel solve");
  __shared__ Potf2Shared sh;
  a += static_cast<long long>(blockIdx.z) * matrix_stride;
  ltq += static_cast<long long>(blockIdx.z) * matrix_stride;
  const int tid = threadIdx.x, r = tid & 63, cq = tid >> 6;
  double v[kPW];
  const double* row = a + static_cast<long long>(k + r) * d + k;
#pragma unroll
  for (int i = 0; i < kPW; ++i) {
    const int c = cq * kPW + i;
    const double g = row[c];
    v[i] = c <= r ? g : 0.0;
  }
  const int first_bad = potf2_core(v, sh, tid);
  double* out = a + static_cast<long long>(k + r) * d + k;
#pragma unroll
  for (int i = 0; i < kPW; ++i) {
    const int c = cq * kPW + i;
    const double y = sh.ys[c];
    const double l = v[i] * y;                  // on the diagonal u = p: sqrt(p)
    ltq[(c * 4 + (r & 3)) * 16 + (r >> 2)] = c == r ? y : l;
    if (c <= r) out[c] = l;
  }
  if (tid == 0 && first_bad < NB) atomicCAS(info + blockIdx.z, 0, k + first_bad + 1);
}

// L21 tile blockIdx.x of every matrix := A21 inv(L11)^T, in place
__global__ __launch_bounds__(256) void solve_batched_kernel(double* __restrict__ a, long long matrix_stride, int d, int k,
                                                           const double* __restrict__ ltq) {
  __shared__ Pair Lt[NB * NB / 2];
  __shared__ double X[NB * (NB + 1)];
  a += static_cast<long long>(blockIdx.z) * matrix_stride;
  ltq += static_cast<long long>(blockIdx.z) * matrix_stride;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* tile = a + static_cast<long long>(k + NB + blockIdx.x * NB) * d + k;
  double g[NB * NB / 256];
#pragma unroll
  for (int it = 0; it < NB * NB / 256; ++it) {
    const int e = tid + 256 * it;
    g[it] = tile[static_cast<long long>(e >> 6) * d + (e & 63)];
  }
  {
    const Pair* src = reinterpret_cast<const Pair*>(ltq);
#pragma unroll
    for (int it = 0; it < NB * NB / 2 / 256; ++it) Lt[tid + 256 * it] = src[tid + 256 * it];
  }
#pragma unroll
  for (int it = 0; it < NB * NB / 256; ++it) {
    const int e = tid + 256 * it;
    X[(e >> 6) * (NB + 1) + (e & 63)] = g[it];
  }
  __syncthreads();
  trsm_quad_in_lds(X, Lt, wave, lane);
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NB * NB / 256; ++it) {
    const int e = tid + 256 * it;
    tile[static_cast<long long>(e >> 6) * d + (e & 63)] = X[(e >> 6) * (NB + 1) + (e & 63)];
  }
}

// trailing tile (blockIdx.x, blockIdx.y) of every matrix -= L21 tile i . L21 tile j ^T  (tiles above the diagonal: nothing)
__global__ __launch_bounds__(kPotf2Threads) void update_batched_kernel(double* __restrict__ a, long long matrix_stride, int d, int k) {
  constexpr int T = kPotf2Threads, W = kPotf2Waves;
  constexpr int SC = NB / (W / 2), MB = SC / 16;
  const int ti = blockIdx.x, tj = blockIdx.y;
  if (tj > ti) return;
  __shared__ double Xi[NB * (NB + 1)], Xj[NB * (NB + 1)];
  a += static_cast<long long>(blockIdx.z) * matrix_stride;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double gi[NB * NB / T], gj[NB * NB / T];
  {
    const double* src_i = a + static_cast<long long>(k + NB + ti * NB) * d + k;
    const double* src_j = a + static_cast<long long>(k + NB + tj * NB) * d + k;
#pragma unroll
    for (int it = 0; it < NB * NB / T; ++it) {
      const int e = tid + T * it;
      gi[it] = src_i[static_cast<long long>(e >> 6) * d + (e & 63)];
      gj[it] = src_j[static_cast<long long>(e >> 6) * d + (e & 63)];
    }
  }
  const int wy = wave / (W / 2), wx = wave % (W / 2);
  double* ctile = a + static_cast<long long>(k + NB + ti * NB) * d + (k + NB + tj * NB);
  double cold[2][MB][4];
#pragma unroll
  for (int ma = 0; ma < 2; ++ma)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        cold[ma][mb][r] = ctile[static_cast<long long>(wy * 32 + ma * 16 + (lane >> 4) + 4 * r) * d + wx * SC + mb * 16 + (lane & 15)];
#pragma unroll
  for (int it = 0; it < NB * NB / T; ++it) {
    const int e = tid + T * it, rr = e >> 6, c = e & 63;
    Xi[rr * (NB + 1) + c] = gi[it];
    Xj[rr * (NB + 1) + c] = gj[it];
  }
  __syncthreads();
  const double* XJ = tj == ti ? Xi : Xj;
  __attribute__((ext_vector_type(4))) double acc[2][MB];
#pragma unroll
  for (int ma = 0; ma < 2; ++ma)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[ma][mb] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k0 = 0; k0 < NB; k0 += 4) {
    double af[2], bf[MB];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) af[t2] = Xi[(wy * 32 + t2 * 16 + (lane & 15)) * (NB + 1) + k0 + (lane >> 4)];
#pragma unroll
    for (int t2 = 0; t2 < MB; ++t2) bf[t2] = XJ[(wx * SC + t2 * 16 + (lane & 15)) * (NB + 1) + k0 + (lane >> 4)];
#pragma unroll
    for (int ma = 0; ma < 2; ++ma)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
        acc[ma][mb] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ma], bf[mb], acc[ma][mb], 0, 0, 0);
  }
#pragma unroll
  for (int ma = 0; ma < 2; ++ma)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wy * 32 + ma * 16 + (lane >> 4) + 4 * r, col = wx * SC + mb * 16 + (lane & 15);
        const double cnew = cold[ma][mb][r] - acc[ma][mb][r];
        if (tj != ti || col <= row) ctile[static_cast<long long>(row) * d + col] = cnew;
      }
}

// out[z] = the symmetric float32 matrix whose lower triangle is src + z * matrix_stride (32 x 32 tiles through LDS)
__global__ __launch_bounds__(256) void mirror_out_batched_kernel(HinvTable t, const float* __restrict__ src, long long matrix_stride, int d) {
  __shared__ float tile[32][33];
  const int bi = blockIdx.y, bj = blockIdx.x;          // tiles (bj, bi) [lower] and (bi, bj) [upper], bj >= bi
  if (bj < bi) return;
  src += static_cast<long long>(blockIdx.z) * matrix_stride;
  float* out = t.out[blockIdx.z];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int i = bj * 32 + r, j = bi * 32 + tx;
    const float v = (i < d && j < d) ? src[static_cast<long long>(i) * d + j] : 0.f;
    tile[r][tx] = v;
    if (i < d && j < d && j <= i) out[static_cast<long long>(i) * d + j] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int i = bi * 32 + r, j = bj * 32 + tx;
    if (i < d && j < d && j > i) out[static_cast<long long>(i) * d + j] = tile[tx][r];
  }
}

// -------------------------------------------------------- OBS update ----
struct ApplyArgs {
  float* w;             // [rows, d] working copy (updated in place)
  int rows, d, c0, nb;  // current column block [c0, c0+nb)
  const float* hinv;    // [d, d]
  const void* scale;    // float or double
  const int32_t* zp;    // null = zeros
  int scale_mode;       // 0: one scale; 1: per row; 2: per (row, col / block_size)
  int block_size, nblk;
  float lo, hi;
  int zp_via_f64;       // int32/int64 zero points are added in FP64
  int diff_bits;        // width of (q - zp) in dequantize (8: wraps like int8 - int8)
  int container32;      // targets of 17..32 bits: an int32 container -- (q - zp) * float32 scale is a float64 product
                        // (NumPy's int32 x float32), a NaN or a quotient of 2^31 and beyond casts to INT32_MIN (x86)
  float* err;           // [rows, kErrLd]: this block's 64 columns start at err_col
  int err_col;
  int8_t* q;            // [rows, d]
  int32_t* q32;         // [rows, d] instead of q for targets wider than 8 bits (mi355q_gptq_apply_wide_f32), else null
};

// One 64-column block of the OBS sweep (ref gptq.py:180-212): columns are visited left to
// right; each is quantized and its error pushed into the columns to its right - a dependent
// chain of 64 quantize -> divide -> update steps per row. A row is spread over kRowLanes lanes
// (NB / kRowLanes consecutive columns each): with 32 lanes the update is 2 multiply-subtracts per
// lane, the column's value is fetched with 2 readlanes, and 2048 rows make 256 workgroups (one
// thread per row made 8, 16 lanes per row 128); the chain is quantize-latency bound.
constexpr int kLazyBlocks = 4;          // blocks whose far update is applied together
constexpr int kErrLd = kLazyBlocks * NB;
// lanes per row: 32 while the rows alone cannot fill the part (the chain is latency-bound and
// shorter with 2 columns per lane), 16 once they can (every lane of a row repeats the
// quantization arithmetic, so fewer lanes per row is less total issue)
inline int row_lanes_for(long long rows) { return rows >= 8192 ? 16 : 32; }

// kPlain: a full 64-column block, no zero points, scales that change at most every 32 columns --
// what symmetric weight recipes produce; the uniform conditions (ragged last block, per-column
// scale lookup, zero-point sums, int8 wrap of q - zp) then drop out of the 64 unrolled column
// steps at compile time (they were ~15 branches and half of the ~170 instructions per step).
template <typename ST, int kRowLanes, bool kPlain>
// (the 16-lane form runs where rows fill the part: capped at 128 VGPRs for 4 waves per SIMD;
// the 32-lane form is latency-bound and keeps the registers its hoisted LDS loads want)
__global__ __launch_bounds__(256, kRowLanes == 16 ? 4 : 1) void gptq_block_kernel(ApplyArgs a) {
  constexpr int kColsPerLane = NB / kRowLanes;
  __shared__ __attribute__((aligned(16))) float h[NB][NB];
  __shared__ float hd[NB];
  __shared__ float es[256 / kRowLanes][NB];
  const int l = threadIdx.x % kRowLanes;              // which columns
  const int r = blockIdx.x * (256 / kRowLanes) + threadIdx.x / kRowLanes;
  const bool live = r < a.rows;
  const int rr = live ? r : a.rows - 1;               // idle lanes shadow the last row, write nothing
  float* wrow = a.w + static_cast<long long>(rr) * a.d + a.c0;
  float w[kColsPerLane];
#pragma unroll
  for (int k = 0; k < kColsPerLane; ++k) {
    const int c = l * kColsPerLane + k;
    w[k] = c < a.nb ? wrow[c] : 0.f;
  }
  // Errors of the earlier blocks of this group have not reached these columns yet (the host
  // applies them to the columns beyond the group once per group): W[:, block] -= E_b @
  // Hinv[block b, this block] for every earlier block b, each product summed over its 64 columns
  // and then subtracted, as the reference's per-block matmul does (gptq.py:213-214). 64 x 64
  // tiles through LDS; ~1.3 us per earlier block instead of a GEMM launch per block.
  for (int pb = 0; pb < a.err_col; pb += NB) {
    __syncthreads();
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
      const int k = e / NB, c = e % NB;
      h[k][c] = c < a.nb ? a.hinv[static_cast<long long>(a.c0 - a.err_col + pb + k) * a.d + a.c0 + c] : 0.f;
    }
    for (int e = threadIdx.x; e < (256 / kRowLanes) * NB; e += 256) {
      const int rw = e / NB, k = e % NB;
      const long long row = static_cast<long long>(blockIdx.x) * (256 / kRowLanes) + rw;
      es[rw][k] = row < a.rows ? a.err[row * kErrLd + pb + k] : 0.f;
    }
    __syncthreads();
    float sum[kColsPerLane];
#pragma unroll
    for (int k = 0; k < kColsPerLane; ++k) sum[k] = 0.f;
#pragma unroll 8
    for (int k = 0; k < NB; ++k) {
      const float ek = es[threadIdx.x / kRowLanes][k];
      const float* hp = &h[k][l * kColsPerLane];
#pragma unroll
      for (int j = 0; j < kColsPerLane; ++j) sum[j] = sum[j] + ek * hp[j];
    }
#pragma unroll
    for (int j = 0; j < kColsPerLane; ++j) w[j] = w[j] - sum[j];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < NB * NB; e += 256) {
    const int r2 = e / NB, c = e % NB;
    const float v = (r2 < a.nb && c < a.nb) ? a.hinv[static_cast<long long>(a.c0 + r2) * a.d + a.c0 + c] : 0.f;
    h[r2][c] = v;
    if (r2 == c) hd[r2] = v;
  }
  __syncthreads();
  const ST* sc = static_cast<const ST*>(a.scale);
  const int group = (threadIdx.x & 63) / kRowLanes;    // which of the wave's rows
  // Scale / zero point change at most every 32 columns when the block size is a multiple of 32
  // (all BLOCKWISE_* granularities; the block starts at a multiple of 64): two loads per kernel,
  // none inside the dependent chain.
  auto scale_index = [&](int col) -> long long {
    if (a.scale_mode == 1) return rr;
    if (a.scale_mode == 2) return static_cast<long long>(rr) * a.nblk + col / a.block_size;
    return 0;
  };
  const bool per_step = !kPlain && a.scale_mode == 2 && a.block_size % 32 != 0;
  const long long si0 = scale_index(a.c0), si1 = scale_index(a.c0 + (a.nb > 32 ? 32 : 0));
  const ST s_lo = sc[si0], s_hi = sc[si1];
  const bool with_zp = !kPlain && a.zp != nullptr;
  const int z_lo = with_zp ? a.zp[si0] : 0, z_hi = with_zp ? a.zp[si1] : 0;
  // every lane keeps the results of its own columns and stores them once, after the chain
  int q_own[kColsPerLane];
  float e_own[kColsPerLane];
#pragma unroll
  for (int k = 0; k < kColsPerLane; ++k) { q_own[k] = 0; e_own[k] = 0.f; }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    if (kPlain || i < a.nb) {  // uniform
      // column i of each of the wave's rows lives in lane kRowLanes*g + i/kColsPerLane: uniform
      // readlanes and selects instead of a ds_bpermute round trip
      const float mine = w[i % kColsPerLane];
      float wi = lane_bcast32(mine, i / kColsPerLane);
#pragma unroll
      for (int g = 1; g < kWave / kRowLanes; ++g) {
        const float wg = lane_bcast32(mine, g * kRowLanes + i / kColsPerLane);
        wi = group == g ? wg : wi;
      }
      ST s = i < 32 ? s_lo : s_hi;
      int z = i < 32 ? z_lo : z_hi;
      if (per_step) {  // uniform; odd block sizes (the reference's tests use them): look it up
        const long long si = scale_index(a.c0 + i);
        s = sc[si];
        z = a.zp ? a.zp[si] : 0;
      }
      // quantize (ref gptq.py:191-195 -> uniform_quantize); every lane of the row computes it
      int qi;
      float e;
      if constexpr (sizeof(ST) == 8) {
        double v = static_cast<double>(wi) / s;
        if (with_zp) v = v + static_cast<double>(z);  // uniform; a zero zero-point changes nothing
        double q = __builtin_rint(v);
        q = fmin(fmax(q, static_cast<double>(a.lo)), static_cast<double>(a.hi));
        qi = (v != v) ? (a.container32 ? INT32_MIN : 0) : (q >= 2147483648.0 ? INT32_MIN : static_cast<int>(q));
        int dd = static_cast<int>(static_cast<unsigned>(qi) - static_cast<unsigned>(z));
        if (with_zp && a.diff_bits == 8) dd = static_cast<int8_t>(dd);  // (q itself fits int8)
        if (with_zp && a.diff_bits == 16) dd = static_cast<int16_t>(dd);
        const double dq = static_cast<double>(dd) * s;
        e = static_cast<float>(static_cast<double>(wi) - dq);  // np.subtract(f32, f64, out=f32)
      } else {
        float v = wi / s;
        if (with_zp)  // uniform; without zero points (symmetric weights) the sum is v itself
          v = a.zp_via_f64 ? static_cast<float>(static_cast<double>(v) + static_cast<double>(z))
                           : v + static_cast<float>(z);
        float q = __builtin_rintf(v);
        q = fminf(fmaxf(q, a.lo), a.hi);
        const bool c32 = !kPlain && a.container32 != 0;   // uniform
        qi = (v != v) ? (c32 ? INT32_MIN : 0) : (q >= 2147483648.f ? INT32_MIN : static_cast<int>(q));
        int dd = static_cast<int>(static_cast<unsigned>(qi) - static_cast<unsigned>(z));
        if (with_zp && a.diff_bits == 8) dd = static_cast<int8_t>(dd);
        if (with_zp && a.diff_bits == 16) dd = static_cast<int16_t>(dd);
        if (c32) {   // int32 * float32 is a float64 product in NumPy, and np.subtract(f32, f64, out=f32) rounds once
          e = static_cast<float>(static_cast<double>(wi) - static_cast<double>(dd) * static_cast<double>(s));
        } else {
          const float dq = static_cast<float>(dd) * s;
          e = wi - dq;
        }
      }
      e = e / hd[i];
      const bool mine_col = l == i / kColsPerLane;
      q_own[i % kColsPerLane] = mine_col ? qi : q_own[i % kColsPerLane];
      e_own[i % kColsPerLane] = mine_col ? e : e_own[i % kColsPerLane];
      // intra-block rank-1 update: w[:, j] -= outer(err, hinv[c, j]) (product rounded, then subtracted)
      const float* hv = &h[i][l * kColsPerLane];
#pragma unroll
      for (int k = 0; k < kColsPerLane; ++k) {
        const float p = e * hv[k];
        const float updated = w[k] - p;
        w[k] = (l * kColsPerLane + k > i) ? updated : w[k];   // select, not a branch
      }
    }
  }
  if (live) {
    int8_t* qrow = a.q + static_cast<long long>(r) * a.d + a.c0 + l * kColsPerLane;
    float* erow = a.err + static_cast<long long>(r) * kErrLd + a.err_col + l * kColsPerLane;
    if (a.q32 != nullptr) {                     // wide targets: the integers as they are (uniform branch)
      int32_t* q32row = a.q32 + static_cast<long long>(r) * a.d + a.c0 + l * kColsPerLane;
#pragma unroll
      for (int k = 0; k < kColsPerLane; ++k) {
        const int c = l * kColsPerLane + k;
        if (c < a.nb) q32row[k] = q_own[k];
        erow[k] = c < a.nb ? e_own[k] : 0.f;
      }
    } else if (kPlain && (a.d % kColsPerLane) == 0) {  // one packed store each (c0 is a multiple of 64)
      unsigned packed = 0;
#pragma unroll
      for (int k = 0; k < kColsPerLane; ++k) packed |= (static_cast<unsigned>(q_own[k]) & 0xFFu) << (8 * k);
      if constexpr (kColsPerLane == 4) {
        *reinterpret_cast<unsigned*>(qrow) = packed;
        *reinterpret_cast<float4*>(erow) = make_float4(e_own[0], e_own[1], e_own[2], e_own[3]);
      } else {
        *reinterpret_cast<unsigned short*>(qrow) = static_cast<unsigned short>(packed);
        *reinterpret_cast<float2*>(erow) = make_float2(e_own[0], e_own[1]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < kColsPerLane; ++k) {
        const int c = l * kColsPerLane + k;
        if (c < a.nb) qrow[k] = static_cast<int8_t>(q_own[k]);
        erow[k] = c < a.nb ? e_own[k] : 0.f;
      }
    }
  }
}

// ---- the same block with the column chain inside a DPP QUAD (symmetric recipes, float scales) ----
// With a row spread over 16 / 32 lanes every column step broadcasts the column's value through
// v_readlane, every lane of the row repeats the quantization, and the step costs ~1000 cycles of
// exposed latency (1.34 ms for 2048 x 2048). The chain itself is only quantize -> error -> divide ->
// one multiply-subtract for the next column. Round 2 first moved a whole row into one lane (64
// statically indexed registers, nothing crosses lanes: 1.19 ms), then to the quad layout below.
// The arithmetic per element is exactly the spread kernel's (same operations in the same order:
// identical integers). The errors of the group's earlier blocks are applied by the same kernel
// before the chain starts (the spread kernel's first phase; all four waves of the workgroup).
// Four lanes per row (a DPP quad), 16 rows per wave: lane c4 of the quad owns columns 4k + c4 of
// the block, k = 0..15, in registers. One wave issues an instruction every ~4.5 cycles whatever
// its kind, and with a whole row per lane the 2 x (63 - i) multiply-subtracts of step i were 60 %
// of the 7200 instructions of a block (18.9 us at 2048 rows, on 32 of 256 CUs). Spread over a
// quad they are a quarter of that, four times as many waves work, and what crosses lanes -- the
// step's error, once -- is one quad_perm DPP move. The arithmetic per element is unchanged.
constexpr int kRowsPerWave = kWave / 4;

template <int kLaneInQuad>
__device__ __forceinline__ float quad_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), kLaneInQuad * 0x55, 0xF, 0xF, false));
}

template <int I>
__device__ __forceinline__ void gptq_quad_step(const ApplyArgs& a, const float (*hp)[4][16], int c4, float s_lo, float s_hi,
                                               float (&w)[16], float& mye, unsigned& myq, unsigned* qrow, float* erow,
                                               float* es_row) {
  constexpr int KO = I >> 2, CO = I & 3;
  // this lane's Hinv values of row I (columns 4k + c4, k >= KO) and the row's diagonal entry
  float hv[16];
#pragma unroll
  for (int k = (KO & ~3); k < 16; k += 4) {
    const float4 v = *reinterpret_cast<const float4*>(&hp[I][c4][k]);
    hv[k] = v.x; hv[k + 1] = v.y; hv[k + 2] = v.z; hv[k + 3] = v.w;
  }
  const float hii = hp[I][CO][KO];
  // the chain: every lane runs it on its own column KO, the owner's result (c4 == CO) is the step's
  const float s = I < 32 ? s_lo : s_hi;
  const float wi = w[KO];
  const float v = wi / s;
  float q = __builtin_rintf(v);
  q = fminf(fmaxf(q, a.lo), a.hi);
  const int qi = (v != v) ? 0 : static_cast<int>(q);
  const float dq = static_cast<float>(qi) * s;
  float e_own = wi - dq;
  e_own = e_own / hii;
  const float e = quad_bcast<CO>(e_own);
  myq = c4 == CO ? (static_cast<unsigned>(qi) & 0xFFu) << (8 * CO) : myq;
  mye = c4 == CO ? e_own : mye;
  if constexpr (CO < 3) {
    const float p = e * hv[KO];          // product rounded, then subtracted (np.outer, then -=)
    w[KO] = c4 > CO ? w[KO] - p : w[KO];
  }
  // (pairs: v_pk_mul_f32 + v_pk_add_f32 do two columns per instruction, with the same two
  // roundings per column; a wave issues an instruction every ~4.5 cycles whatever it does)
  typedef float Pair32 __attribute__((ext_vector_type(2)));
  constexpr int K1 = KO + 1, KP = K1 + (K1 & 1);
  if constexpr ((K1 & 1) != 0 && K1 < 16) {
    const float p = e * hv[K1];
    w[K1] = w[K1] - p;
  }
#pragma unroll
  for (int k = KP; k < 16; k += 2) {
    const Pair32 hh = {hv[k], hv[k + 1]};
    Pair32 ww = {w[k], w[k + 1]};
    const Pair32 pp = hh * e;
    ww = ww - pp;
    w[k] = ww.x; w[k + 1] = ww.y;
  }
  if constexpr (CO == 3) {
    // Group KO is complete: its four bytes are gathered over the quad, every lane stores its own
    // error. Unconditional (idle rows shadow the last row and store its values again): a branch
    // here would cut the 64 steps into basic blocks, and the optimizer then sinks the updates of
    // the far columns block by block towards their use.
    unsigned word = myq;
    word |= static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(word), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    word |= static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(word), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    qrow[KO] = word;
    erow[4 * KO + c4] = mye;       // for the update of the columns behind the group (one GEMM per group)
    es_row[4 * KO + c4] = mye;     // for the group's later blocks, which this workgroup does next
  }
  if constexpr (I + 1 < NB) gptq_quad_step<I + 1>(a, hp, c4, s_lo, s_hi, w, mye, myq, qrow, erow, es_row);
}

// One launch per GROUP of kLazyBlocks 64-column blocks (a.c0 = the group's first column, a.nb = its
// number of blocks): rows are independent, so the workgroup that owns 16 rows takes them through
// the group's blocks one after the other -- catch-up with the earlier blocks' errors (kept in
// LDS), then the chain -- and nothing but the update of the columns behind the group needs a
// kernel boundary (~5 us each on this stack; they were a quarter of the 2048 x 2048 time).
__global__ __launch_bounds__(256) void gptq_rows_kernel(ApplyArgs a) {
  // one buffer, two uses -- catch-up: h[k][c] = Hinv[g0 + pb + k][c0 + c]; chain: hp[i][c][k] = Hinv[c0 + i][c0 + 4 k + c]
  __shared__ __attribute__((aligned(16))) float hbuf[NB * NB];
  __shared__ __attribute__((aligned(16))) float es[kRowsPerWave][kErrLd];   // the group's errors so far
  __shared__ __attribute__((aligned(16))) float wl[kRowsPerWave][NB + 4];   // the rows' block on its way into the quad layout
  const int tid = threadIdx.x;
  const int rw = tid >> 4, l = tid & 15;               // catch-up layout: 16 lanes per row, 4 columns per lane
  const long long row = static_cast<long long>(blockIdx.x) * kRowsPerWave + rw;
  const long long rrow = row < a.rows ? row : a.rows - 1;
  const int lane = tid & 63, c4 = lane & 3, rl = lane >> 2;   // chain layout (wave 0): 4 lanes per row
  const int r = blockIdx.x * kRowsPerWave + rl;
  const int rr = r < a.rows ? r : a.rows - 1;          // idle lanes shadow the last row and store its values again
  const float* sc = static_cast<const float*>(a.scale);
  for (int blk = 0; blk < a.nb; ++blk) {
    const int c0 = a.c0 + blk * NB, err_col = blk * NB;
    {
      // Catch-up, by the whole workgroup: the errors of the group's earlier blocks reach this
      // block's columns now,
      //   W[rows, c0:c0+64] -= err[rows, 0:err_col] @ Hinv[g0:g0+err_col, c0:c0+64],
      // one 64-deep sum per earlier block, subtracted in block order.
      const float4 w4 = *reinterpret_cast<const float4*>(a.w + rrow * a.d + c0 + 4 * l);
      float w[4] = {w4.x, w4.y, w4.z, w4.w};
      for (int pb = 0; pb < err_col; pb += NB) {
        __syncthreads();
        const float* hsrc = a.hinv + static_cast<long long>(a.c0 + pb) * a.d + c0;
#pragma unroll
        for (int k = 0; k < NB * NB / 4 / 256; ++k) {
          const int e4 = k * 256 + tid, hr = e4 / (NB / 4), hc = e4 % (NB / 4);
          reinterpret_cast<float4*>(hbuf)[e4] = *reinterpret_cast<const float4*>(hsrc + static_cast<long long>(hr) * a.d + 4 * hc);
        }
        __syncthreads();
        typedef float Pair32 __attribute__((ext_vector_type(2)));   // packed FP32: two columns per instruction
        Pair32 s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
#pragma unroll 8
        for (int k = 0; k < NB; ++k) {
          const float ek = es[rw][pb + k];
          const float4 h4 = *reinterpret_cast<const float4*>(&hbuf[k * NB + 4 * l]);
          const Pair32 h01 = {h4.x, h4.y}, h23 = {h4.z, h4.w};
          const Pair32 p01 = h01 * ek, p23 = h23 * ek;
          s01 = s01 + p01;
          s23 = s23 + p23;
        }
        w[0] = w[0] - s01.x; w[1] = w[1] - s01.y; w[2] = w[2] - s23.x; w[3] = w[3] - s23.y;
      }
      __syncthreads();                 // everyone is done with hbuf (and, for blk > 0, wave 0 with wl)
      *reinterpret_cast<float4*>(&wl[rw][4 * l]) = make_float4(w[0], w[1], w[2], w[3]);
      const float* hblock = a.hinv + static_cast<long long>(c0) * a.d + c0;
      float (*hpw)[4][16] = reinterpret_cast<float (*)[4][16]>(hbuf);
#pragma unroll
      for (int k = 0; k < NB * NB / 4 / 256; ++k) {
        const int e4 = k * 256 + tid, hr = e4 / (NB / 4), kk = e4 % (NB / 4);
        const float4 v = *reinterpret_cast<const float4*>(hblock + static_cast<long long>(hr) * a.d + 4 * kk);
        hpw[hr][0][kk] = v.x; hpw[hr][1][kk] = v.y; hpw[hr][2][kk] = v.z; hpw[hr][3][kk] = v.w;
      }
      __syncthreads();
    }
    if (tid < kWave) {                 // the chain is one wave's
      const float (*hp)[4][16] = reinterpret_cast<const float (*)[4][16]>(hbuf);
      const long long si0 = a.scale_mode == 1 ? rr : (a.scale_mode == 2 ? static_cast<long long>(rr) * a.nblk + c0 / a.block_size : 0);
      const long long si1 = a.scale_mode == 2 ? static_cast<long long>(rr) * a.nblk + (c0 + 32) / a.block_size : si0;
      const float s_lo = sc[si0], s_hi = sc[si1];
      unsigned* qrow = reinterpret_cast<unsigned*>(a.q + static_cast<long long>(rr) * a.d + c0);
      float* erow = a.err + static_cast<long long>(rr) * kErrLd + err_col;
      float w[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) w[k] = wl[rl][4 * k + c4];
      float mye = 0.f;
      unsigned myq = 0;
      gptq_quad_step<0>(a, hp, c4, s_lo, s_hi, w, mye, myq, qrow, erow, &es[rl][err_col]);
    }
  }
}

inline unsigned grid1d(long long n) {
  long long b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > 256 * 16) b = 256 * 16;
  return static_cast<unsigned>(b);
}

}  // namespace
}  // namespace mi355q

using namespace mi355q;

namespace mi355q {
// xtx_bf16x3.hip: the product on the bf16 matrix cores (three-way split of every float32)
bool xtx_bf16x3_usable(int64_t n, int64_t d);
size_t xtx_bf16x3_workspace_bytes(int64_t n, int64_t d);
int32_t xtx_bf16x3(const float* x, int64_t n, int64_t d, float* p, void* workspace, hipStream_t st, bool accumulate_first);
// xtx_f16x2.hip: the same product from a two-way float16 split (half the MFMA work; the default)
bool xtx_f16x2_usable(int64_t n, int64_t d);
size_t xtx_f16x2_workspace_bytes(int64_t n, int64_t d);
int32_t xtx_f16x2(const float* x, int64_t n, int64_t d, float* p, void* workspace, hipStream_t st, bool accumulate_first);
// ... and the same split for the update behind a group of columns of the OBS apply
bool upd_bf16x3_usable(int64_t rows, int64_t d);
size_t upd_bf16x3_workspace_bytes(int64_t rows, int64_t d, int64_t kk_max);
int32_t upd_bf16x3_prepare(const float* hinv, int64_t d, void* workspace, hipStream_t st);
int32_t upd_bf16x3(const float* err, int64_t ld, int64_t rows, int64_t d, int64_t g0, int64_t kk, int64_t g1, float* w,
                   void* workspace, hipStream_t st);
// ... and for the single-precision steps of the Hessian inverse (triangular inverse, L^-T L^-1)
size_t hinv_split_scratch_bytes(int64_t d);
int32_t trtri_level_bf16x3(double* a, int64_t d, int64_t s, int64_t items, void* scratch, hipStream_t st);
int32_t ltl_bf16x3(const double* linv, int64_t d, float* hinv, void* scratch, hipStream_t st);
}  // namespace mi355q

namespace {
size_t xtx_scratch_bytes(int64_t n, int64_t d) {
  const size_t fp32 = gemm_splitk_workspace_bytes<float>(static_cast<int>(d), static_cast<int>(d), static_cast<int>(n < 0 ? 0 : n), true);
  size_t split = xtx_bf16x3_usable(n, d) ? xtx_bf16x3_workspace_bytes(n, d) : 0;
  if (xtx_f16x2_usable(n, d) && xtx_f16x2_workspace_bytes(n, d) > split) split = xtx_f16x2_workspace_bytes(n, d);
  return split > fp32 ? split : fp32;
}

// product (+)= X^T X on the lower-triangular tiles
int32_t xtx_product(const float* x, int64_t n, int64_t d, float* p, bool accumulate, void* scratch, size_t scratch_bytes,
                    hipStream_t st) {
  // P = X^T X : A(i,k) = X[k][i], B(k,j) = X[k][j]; long K is split over gridDim.z. P is
  // symmetric and P[i][j], P[j][i] are the same k-ordered sum of the same (commuting) products,
  // so only the lower triangle is computed (triangular launch grid: half the flops) and mirrored.
  if (xtx_f16x2_usable(n, d)) return xtx_f16x2(x, n, d, p, scratch, st, accumulate);
  if (xtx_bf16x3_usable(n, d)) return xtx_bf16x3(x, n, d, p, scratch, st, accumulate);
  GemmArgs<float> g{x, 1, d, x, d, 1, p, d, 1, static_cast<int>(d), static_cast<int>(d),
                    static_cast<int>(n), 1.0f, accumulate ? 1.0f : 0.0f, 1, 0};
  return launch_gemm<float>(g, st, scratch, scratch_bytes);
}
}  // namespace

extern "C" size_t mi355q_gptq_xtx_workspace_bytes(int64_t n, int64_t d) {
  if (d <= 0 || d > 0x7FFFFFFF || n > 0x7FFFFFFF) return 0;
  return static_cast<size_t>(d) * d * sizeof(float) + xtx_scratch_bytes(n, d);
}

extern "C" int32_t mi355q_gptq_xtx_f32(const float* x, int64_t n, int64_t d, double alpha,
                                       double* hessian_out, void* workspace, size_t workspace_bytes,
                                       void* stream) {
  clear_error();
  if (n < 0 || d < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  if (d == 0) return MI355Q_OK;
  if (d > 0x7FFFFFFF || n > 0x7FFFFFFF) return fail(MI355Q_UNSUPPORTED, "dimension too large");
  if (!x || !hessian_out) return fail(MI355Q_BAD_ARG, "null pointer");
  const size_t need = mi355q_gptq_xtx_workspace_bytes(n, d);
  if (!workspace || workspace_bytes < need)
    return fail(MI355Q_BAD_ARG, "workspace too small: need %zu bytes", need);
  hipStream_t st = as_stream(stream);
  float* p = static_cast<float*>(workspace);
  if (int32_t s = xtx_product(x, n, d, p, false, p + d * d, need - static_cast<size_t>(d) * d * sizeof(float), st)) return s;
  const unsigned t32 = static_cast<unsigned>((d + 31) / 32);
  hipLaunchKernelGGL(mirror_scale_to_f64_kernel, dim3(t32, t32), dim3(256), 0, st, p, static_cast<int>(d),
                     alpha, hessian_out);
  MI355Q_CHECK_LAUNCH("hessian scale launch");
  return MI355Q_OK;
}

extern "C" size_t mi355q_gptq_xtx_accum_workspace_bytes(int64_t n, int64_t d) {
  if (d <= 0 || d > 0x7FFFFFFF || n > 0x7FFFFFFF) return 0;
  return xtx_scratch_bytes(n, d);
}

extern "C" int32_t mi355q_gptq_xtx_accum_f32(const float* x, int64_t n, int64_t d, float* product,
                                             int32_t accumulate, void* workspace, size_t workspace_bytes,
                                             void* stream) {
  clear_error();
  if (n < 0 || d < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  if (d == 0) return MI355Q_OK;
  if (d > 0x7FFFFFFF || n > 0x7FFFFFFF) return fail(MI355Q_UNSUPPORTED, "dimension too large");
  if (!x || !product) return fail(MI355Q_BAD_ARG, "null pointer");
  const size_t need = mi355q_gptq_xtx_accum_workspace_bytes(n, d);
  if (need && (!workspace || workspace_bytes < need))
    return fail(MI355Q_BAD_ARG, "workspace too small: need %zu bytes", need);
  return xtx_product(x, n, d, product, accumulate != 0, workspace, workspace_bytes, as_stream(stream));
}

extern "C" int32_t mi355q_gptq_xtx_finish_f64(const float* product, int64_t d, double alpha,
                                              double* hessian_out, void* stream) {
  clear_error();
  if (d < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  if (d == 0) return MI355Q_OK;
  if (d > 0x7FFFFFFF) return fail(MI355Q_UNSUPPORTED, "dimension too large");
  if (!product || !hessian_out) return fail(MI355Q_BAD_ARG, "null pointer");
  const unsigned t32 = static_cast<unsigned>((d + 31) / 32);
  hipLaunchKernelGGL(mirror_scale_to_f64_kernel, dim3(t32, t32), dim3(256), 0, as_stream(stream), product,
                     static_cast<int>(d), alpha, hessian_out);
  MI355Q_CHECK_LAUNCH("hessian scale launch");
  return MI355Q_OK;
}

extern "C" int32_t mi355q_gptq_hessian_merge_f64(const double* h_cur, double n_cur, const double* h_new,
                                                 double n_new, int64_t d, double* h_out, void* stream) {
  clear_error();
  if (d < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  if (d == 0) return MI355Q_OK;
  if (!h_cur || !h_new || !h_out) return fail(MI355Q_BAD_ARG, "null pointer");
  if (n_cur + n_new == 0.0) return fail(MI355Q_BAD_ARG, "total sample count is zero");
  hipLaunchKernelGGL(hessian_merge_kernel, dim3(grid1d(d * d)), dim3(256), 0, as_stream(stream), h_cur,
                     n_cur, h_new, n_new, static_cast<long long>(d) * d, h_out);
  MI355Q_CHECK_LAUNCH("hessian merge launch");
  return MI355Q_OK;
}

namespace {
// Look-ahead for the blocked Cholesky: the rank-512 update of the matrix behind the next outer
// block runs on a side stream while the caller's stream already factors that next block (serial
// 64 x 64 diagonal work and small GEMMs that would otherwise leave the chip idle).
// One side stream + two events per device, created on first use, released by mi355q_shutdown().
struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t panel_done = nullptr, update_done = nullptr;
  bool tried = false;
  long long debug_delay_ticks = 0;   // MI355Q_DEBUG_SIDE_DELAY_US: see side_delay_kernel
};

// Test hook (tests/test_gpu_gptq.py): holds the side stream back for a while in front of every
// look-ahead update, so that an ordering bug between the two streams shows up as a wrong inverse
// instead of depending on how short the side GEMM happens to be. wall_clock64 ticks at 100 MHz.
__global__ void side_delay_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
// Two slots per device: a large inverse owns one for the duration of its call, so two of them -- the pair
// mi355q_gptq_hinv_*_batched keeps in flight at d >= 4096 -- each have a look-ahead stream of their own (sharing one,
// the second matrix's rank-512 updates would queue behind ALL of the first's). Single calls use slot 0.
constexpr int kSideSlots = 2;
SideStream g_side[64][kSideSlots];
// One call at a time enqueues look-ahead work per slot: a slot's stream and its two events are shared by every
// caller of a device, and hipStreamWaitEvent captures the event's state at the time of the call, so record/wait
// pairs of two host threads must not interleave. The lock is held only while a call enqueues (the entry point never
// synchronizes).
std::mutex g_side_mutex[kSideSlots];

SideStream* side_stream(int slot = 0) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || slot < 0 || slot >= kSideSlots) return nullptr;
  SideStream& s = g_side[dev][slot];
  if (!s.tried) {
    s.tried = true;
    if (getenv("MI355Q_NO_LOOKAHEAD")) return nullptr;
    // non-blocking: the caller's stream is usually the legacy default stream, which serializes
    // with every blocking stream (hipExtStreamCreateWithCUMask only makes blocking ones)
    // lowest priority: when both queues have workgroups to place, the caller's small kernels go first
    hipStream_t st = nullptr;
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
    if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, least) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    hipEvent_t a = nullptr, b = nullptr;
    if (hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipStreamDestroy(st);
      return nullptr;
    }
    s.stream = st;
    s.panel_done = a;
    s.update_done = b;
    if (const char* us = getenv("MI355Q_DEBUG_SIDE_DELAY_US")) s.debug_delay_ticks = atoll(us) * 100;
  }
  return s.stream ? &s : nullptr;
}
}  // namespace

namespace { void release_hinv_pools(); }
namespace mi355q { void release_file_io(); }   // file_io.hip

namespace { void prepare_hinv_pool(); }

// Device memory of the caller's own (host code: ctypes releases the interpreter lock around these, the framework's
// allocator does not around its hipMalloc): see include/mi355q.h.
extern "C" int32_t mi355q_device_alloc(size_t nbytes, void** out) {
  clear_error();
  if (!out) return fail(MI355Q_BAD_ARG, "null pointer");
  *out = nullptr;
  if (nbytes == 0) return MI355Q_OK;
  if (hipError_t e = hipMalloc(out, nbytes)) return fail(MI355Q_HIP_ERROR, "hipMalloc(%zu): %s", nbytes, hipGetErrorString(e));
  return MI355Q_OK;
}

extern "C" int32_t mi355q_device_free(void* p) {
  clear_error();
  if (!p) return MI355Q_OK;
  if (hipError_t e = hipFree(p)) return fail(MI355Q_HIP_ERROR, "hipFree: %s", hipGetErrorString(e));
  return MI355Q_OK;
}

extern "C" int32_t mi355q_prepare_device(void) {
  clear_error();
  for (int slot = 0; slot < kSideSlots; ++slot) {
    std::lock_guard<std::mutex> lock(g_side_mutex[slot]);
    (void)side_stream(slot);    // (look-ahead switched off or no device: nothing to prepare)
  }
  // ... and the lanes of the batched inverse: a hardware queue costs ~4 ms to create now and 3 - 4 times that
  // once the application has made its own streams (16 lanes made at the first batched call of an 18-layer
  // run: 220 ms). Eight lanes: 35 ms here, 0.67 ms per d = 2048 inverse (sixteen: 70 ms and 0.59).
  prepare_hinv_pool();
  return MI355Q_OK;
}

extern "C" int32_t mi355q_shutdown(void) {
  clear_error();
  release_hinv_pools();
  release_file_io();
  for (int slot = 0; slot < kSideSlots; ++slot) {
    std::lock_guard<std::mutex> lock(g_side_mutex[slot]);
    for (auto& dev : g_side) {
      SideStream& s = dev[slot];
      if (s.stream) {
        (void)hipStreamSynchronize(s.stream);
        (void)hipEventDestroy(s.panel_done);
        (void)hipEventDestroy(s.update_done);
        (void)hipStreamDestroy(s.stream);
      }
      s = SideStream();
    }
  }
  return MI355Q_OK;
}

extern "C" size_t mi355q_gptq_hinv_workspace_bytes(int64_t d) {
  // two d x d FP64 matrices + one transposed NB x NB diagonal block + two d x NB panels + scalars
  if (d <= 0) return 0;
  return (static_cast<size_t>(d) * d * 2 + NB * NB + static_cast<size_t>(d) * NB * 2 + 8) * sizeof(double);
}

namespace {
// hessian (FLOAT64 [d, d]) or, when it is null, alpha * product (product FLOAT32 [d, d], lower triangle valid)
int32_t hinv_impl(const double* hessian, const float* product, double alpha, int64_t d64, double damp_factor,
                  float* hinv_out, int32_t* info_out, void* workspace, size_t workspace_bytes, void* stream,
                  int side_slot = 0) {
  clear_error();
  if (d64 < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  if (d64 == 0) return MI355Q_OK;
  if (d64 > 46000) return fail(MI355Q_UNSUPPORTED, "d too large");
  if ((!hessian && !product) || !hinv_out || !info_out) return fail(MI355Q_BAD_ARG, "null pointer");
  const size_t need = mi355q_gptq_hinv_workspace_bytes(d64);
  if (!workspace || workspace_bytes < need)
    return fail(MI355Q_BAD_ARG, "workspace too small: need %zu bytes", need);
  const int d = static_cast<int>(d64);
  hipStream_t st = as_stream(stream);
  const int nblocks = (d + NB - 1) / NB;
  double* a = static_cast<double*>(workspace);           // L, then L^-1 (lower, zeros above)
  double* out = a + static_cast<size_t>(d) * d;          // lower(H^-1) in FP64
  double* lt = out + static_cast<size_t>(d) * d;         // NB x NB: the current diagonal block, transposed
  double* spanel = lt + NB * NB;                         // 2 x (d x NB): the unsolved panels of the fused steps
  double* scal = spanel + static_cast<size_t>(d) * NB * 2;
  if (hipMemsetAsync(info_out, 0, sizeof(int32_t), st) != hipSuccess)
    return fail(MI355Q_HIP_ERROR, "hipMemsetAsync failed");
  if (hessian) hipLaunchKernelGGL((diag_sum_kernel<double>), dim3(1), dim3(256), 0, st, hessian, 1.0, d, scal);
  else hipLaunchKernelGGL((diag_sum_kernel<float>), dim3(1), dim3(256), 0, st, product, alpha, d, scal);
  MI355Q_CHECK_LAUNCH("gptq damp launch");
  // ---- blocked right-looking Cholesky (lower), FP64, two levels. Per 64-column step:
  //   diagonal block: factor (one workgroup)
  //   panel:    L21 L11^T = A21, in place                   (forward substitution, one wave per 64 rows)
  //   trailing: A22 -= L21 * L21^T, lower triangle only     (MFMA GEMM)
  // A rank-64 update of the whole trailing matrix is memory-bound (it reads and writes
  // (d-k)^2/2 doubles for 64 flops each), so the 64-column steps only update the rest of their
  // own 512-column outer block; the matrix behind it gets one rank-512 update per outer block.
  static const int OB = [] { const char* e = getenv("MI355Q_CHOL_OB"); const int v = e ? atoi(e) : 0; return v >= 64 && v % 64 == 0 ? v : 8 * NB; }();
  std::unique_lock<std::mutex> side_lock(g_side_mutex[side_slot], std::defer_lock);
  if (d >= 4096) side_lock.lock();
  SideStream* side = d >= 4096 ? side_stream(side_slot) : nullptr;
  bool side_busy = false;
  // Whatever way this call leaves the loop, the caller's stream waits for the side stream's last
  // update: the workspace it writes belongs to the caller, who may free it right after an error.
  struct JoinSide {
    SideStream*& side; bool& busy; hipStream_t st;
    ~JoinSide() { if (busy && side) (void)hipStreamWaitEvent(st, side->update_done, 0); }
  } join_side{side, side_busy, st};
  if (side != nullptr && d > 2 * OB && getenv("MI355Q_NO_SPLIT_COPY") == nullptr) {
    // the first outer block's columns now; the other 2 GB (d = 16384: 0.75 ms) on the side stream,
    // underneath the first block's chain. update_done doubles as "copied": the first update behind
    // outer block 0 waits for it like for any earlier side-stream update.
    if (hipEventRecord(side->panel_done, st) != hipSuccess || hipStreamWaitEvent(side->stream, side->panel_done, 0) != hipSuccess)
      return fail(MI355Q_HIP_ERROR, "look-ahead: copy hand-over failed");
    if (hessian) {
      hipLaunchKernelGGL((copy_damped_lower_cols_kernel<double>), dim3((OB + 255) / 256, (d + 7) / 8), dim3(256), 0, st, hessian, 1.0,
                         d, scal, damp_factor, a, 0, OB);
      hipLaunchKernelGGL((copy_damped_lower_cols_kernel<double>), dim3((d - OB + 255) / 256, (d + 7) / 8), dim3(256), 0, side->stream,
                         hessian, 1.0, d, scal, damp_factor, a, OB, d);
    } else {
      hipLaunchKernelGGL((copy_damped_lower_cols_kernel<float>), dim3((OB + 255) / 256, (d + 7) / 8), dim3(256), 0, st, product, alpha,
                         d, scal, damp_factor, a, 0, OB);
      hipLaunchKernelGGL((copy_damped_lower_cols_kernel<float>), dim3((d - OB + 255) / 256, (d + 7) / 8), dim3(256), 0, side->stream,
                         product, alpha, d, scal, damp_factor, a, OB, d);
    }
    if (hipEventRecord(side->update_done, side->stream) != hipSuccess)
      return fail(MI355Q_HIP_ERROR, "look-ahead: record failed");
    side_busy = true;
  } else {
    if (hessian)
      hipLaunchKernelGGL((copy_damped_lower_kernel<double>), dim3(grid1d(static_cast<long long>(d) * d)), dim3(256), 0, st,
                         hessian, 1.0, d, scal, damp_factor, a);
    else
      hipLaunchKernelGGL((copy_damped_lower_kernel<float>), dim3(grid1d(static_cast<long long>(d) * d)), dim3(256), 0, st,
                         product, alpha, d, scal, damp_factor, a);
  }
  MI355Q_CHECK_LAUNCH("gptq damp launch");
  double* step_panel[2] = {spanel, spanel + static_cast<size_t>(d) * NB};   // see chol_step_kernel
  int step_parity = 0;
  for (int k0 = 0; k0 < d; k0 += OB) {
    const int ob = d - k0 < OB ? d - k0 : OB;
    bool step_panel_ready = false;   // an outer block's first step reads its panel from `a`
    for (int k = k0; k < k0 + ob; k += NB) {
      const int nb = k0 + ob - k < NB ? k0 + ob - k : NB;
      const int m = d - k - nb;            // rows below the diagonal block
      const int w = k0 + ob - k - nb;      // columns left in this outer block
      // small d (no look-ahead, the chip idle around the chain): the whole step in one launch
      static const bool fused_step = getenv("MI355Q_NO_FUSED_STEP") == nullptr;
      // (with look-ahead, i.e. d >= 4096, only for the last 2048 rows: above that the workgroups' redundant
      // factor + solve work competes with the side stream's update for CUs -- 92 vs 87 ms at d = 16384 everywhere)
      static const int fused_max_m = [] { const char* e = getenv("MI355Q_FUSED_MAX_M"); return e ? atoi(e) : 2048; }();
      if (m > 0 && nb == NB && m % NB == 0 && w % NB == 0 && fused_step && (side == nullptr || m <= fused_max_m) &&
          (w > 0 || step_panel_ready)) {
        if (!step_panel_ready)
          hipLaunchKernelGGL(copy_panel_kernel, dim3(grid1d(static_cast<long long>(m + NB) * NB)), dim3(256), 0, st, a, d, k,
                             m + NB, step_panel[step_parity]);
        hipLaunchKernelGGL(chol_step_kernel, dim3(m / NB, w > 0 ? w / NB : 1), dim3(kPotf2Threads), 0, st, a, d, k, info_out,
                           step_panel[step_parity], step_panel[step_parity ^ 1], w > 0 ? 1 : 0);
        step_parity ^= 1;
        step_panel_ready = true;     // the next step's block column (if it is fused too) is in step_panel[step_parity]
        continue;
      }
      // whole tiles: the three non-redundant step kernels of the lock-step batch (one matrix): the panel solve on DPP
      // quads (four waves per 64-row tile instead of one lane per row) and the in-block update on the step kernel's MFMA
      // loop, no workgroup repeating another's work -- beside the look-ahead GEMM, where the chain is what the wall clock
      // follows for two thirds of the outer blocks of a d = 16384 factorization
      static const bool asu = [] { const char* e = getenv("MI355Q_CHOL_ASU"); return e == nullptr || atoi(e) != 0; }();
      if (asu && nb == NB && m % NB == 0 && w % NB == 0) {
        hipLaunchKernelGGL(potf2_batched_kernel, dim3(1, 1, 1), dim3(kPotf2Threads), 0, st, a, 0LL, d, k, info_out, lt);
        if (m > 0) {
          hipLaunchKernelGGL(solve_batched_kernel, dim3(static_cast<unsigned>(m / NB), 1, 1), dim3(256), 0, st, a, 0LL, d, k, lt);
          if (w > 0)
            hipLaunchKernelGGL(update_batched_kernel, dim3(static_cast<unsigned>(m / NB), static_cast<unsigned>(w / NB), 1),
                               dim3(kPotf2Threads), 0, st, a, 0LL, d, k);
        }
        step_panel_ready = false;
        continue;
      }
      hipLaunchKernelGGL(potf2_kernel, dim3(1), dim3(kPotf2Threads), 0, st, a, d, k, nb, info_out, lt MI355Q_PROF_ARG);
      if (m > 0) {
        hipLaunchKernelGGL(trsm_panel_kernel, dim3((m + 63) / 64), dim3(64), 0, st, a, d, k, nb, m, lt MI355Q_PROF_ARG);
        if (w > 0) {
          const double* l21 = a + static_cast<long long>(k + nb) * d + k;
          double* a22 = a + static_cast<long long>(k + nb) * d + k + nb;
          GemmArgs<double> gt{l21, d, 1, l21, 1, d, a22, d, 1, m, w, nb, -1.0, 1.0, 1, 0};
          if (int32_t e = launch_gemm<double>(gt, st)) return e;
        }
      }
    }
    const int m2 = d - k0 - ob;
    if (m2 > 0) {
      const double* l = a + static_cast<long long>(k0 + ob) * d + k0;   // m2 x ob, finished columns
      double* c = a + static_cast<long long>(k0 + ob) * d + k0 + ob;
      const int next = m2 < OB ? m2 : OB;       // width of the next outer block
      if (side == nullptr || m2 - next < 2048) {
        // the previous outer block's `rest` update may still be read-modify-writing this very
        // trailing region on the side stream: it has to land before this update starts
        if (side_busy) {
          if (hipStreamWaitEvent(st, side->update_done, 0) != hipSuccess)
            return fail(MI355Q_HIP_ERROR, "look-ahead: wait failed");
          side_busy = false;
        }
        GemmArgs<double> gu{l, d, 1, l, 1, d, c, d, 1, m2, m2, ob, -1.0, 1.0, 1, 0};
        if (int32_t e = launch_gemm<double>(gu, st)) return e;
      } else {
        // the strip the next outer block lives in: here, now (it must also wait for the side
        // stream's previous update, which wrote the same strip)
        if (side_busy && hipStreamWaitEvent(st, side->update_done, 0) != hipSuccess)
          return fail(MI355Q_HIP_ERROR, "look-ahead: wait failed");
        if (hipEventRecord(side->panel_done, st) != hipSuccess)
          return fail(MI355Q_HIP_ERROR, "look-ahead: record failed");
        GemmArgs<double> strip{l, d, 1, l, 1, d, c, d, 1, m2, next, ob, -1.0, 1.0, 1, 0};
        if (int32_t e = launch_gemm<double>(strip, st)) return e;
        // everything behind that strip: on the side stream, overlapping the next block's
        // factorization (disjoint columns)
        if (hipStreamWaitEvent(side->stream, side->panel_done, 0) != hipSuccess)
          return fail(MI355Q_HIP_ERROR, "look-ahead: wait failed");
        if (side->debug_delay_ticks > 0)
          hipLaunchKernelGGL(side_delay_kernel, dim3(1), dim3(1), 0, side->stream, side->debug_delay_ticks);
        const double* l2 = l + static_cast<long long>(next) * d;
        double* c2 = c + static_cast<long long>(next) * d + next;
        GemmArgs<double> rest{l2, d, 1, l2, 1, d, c2, d, 1, m2 - next, m2 - next, ob, -1.0, 1.0, 1, 0};
        if (int32_t e = launch_gemm<double>(rest, side->stream)) return e;
        if (hipEventRecord(side->update_done, side->stream) != hipSuccess)
          return fail(MI355Q_HIP_ERROR, "look-ahead: record failed");
        side_busy = true;
      }
    }
  }
  if (side_busy) {
    if (hipStreamWaitEvent(st, side->update_done, 0) != hipSuccess)
      return fail(MI355Q_HIP_ERROR, "look-ahead: wait failed");
    side_busy = false;
  }
  if (side_lock.owns_lock()) side_lock.unlock();
  MI355Q_CHECK_LAUNCH("gptq cholesky launch");
  // ---- in-place inverse of the lower-triangular factor by pairwise merging: the diagonal
  // NB-blocks are already inverted; at level s every pair of adjacent inverted blocks
  //   [ L11^-1        0     ]
  //   [ L21        L22^-1   ]      becomes one inverted block with  L21 <- -L22^-1 (L21 L11^-1).
  // All flops are in large triangular-operand GEMMs (two per pair); `out` is free until the
  // final product and holds the intermediate L21 L11^-1.
  hipLaunchKernelGGL(diag_inverse_kernel, dim3(nblocks), dim3(256), 0, st, a, d);
  // The reference runs everything from here on in single precision (scipy's strtri on the float32
  // cast of the factor, a float32 einsum for the product, ref gptq.py:121-128). For d >= 4096 the
  // large merge levels and the product therefore run on the bf16 matrix cores with float32-class
  // accuracy (exact three-way split of every operand rounded to float32, csrc/xtx_bf16x3.hip);
  // the factorization above stays FP64 like the reference's np.linalg.cholesky of the float64
  // Hessian. MI355Q_HINV_FP64=1 keeps FP64 MFMA throughout. `out` (free until the product) is the scratch.
  static const bool split_ok = getenv("MI355Q_HINV_FP64") == nullptr;
  static const long long split_min_s = [] { const char* e = getenv("MI355Q_HINV_SPLIT_MIN_S"); const long long v = e ? atoll(e) : 0; return v >= 128 ? v : 1024LL; }();
  const bool split = split_ok && d >= 4096 && d % 128 == 0 && d <= 16384 &&
                     hinv_split_scratch_bytes(d) + 4096 <= static_cast<size_t>(d) * d * sizeof(double);
  for (long long s = NB; s < d; s *= 2) {
    // the pairs of one level are independent and (but for a ragged last one) equally shaped:
    // the low levels, hundreds of one-tile GEMMs, go out as two batched launches per level
    long long first = 0;
    const long long full = (d - s) / (2 * s) + ((d - s) % (2 * s) >= s ? 1 : 0);   // pairs with n2 == s
    if (split && s >= split_min_s && full >= 1) {
      if (int32_t e = trtri_level_bf16x3(a, d, s, full, out, st)) return e;
      first = full * 2 * s;
    } else if (full >= 2 && s <= 2048) {
      const int n = static_cast<int>(s);
      const long long hop = 2 * s * (static_cast<long long>(d) + 1);
      GemmArgs<double> g1{a + s * d, d, 1, a, d, 1, out, n, 1, n, n, n, 1.0, 0.0, 0, 3,
                          static_cast<int>(full), hop, hop, s * s};
      if (int32_t e = launch_gemm<double>(g1, st)) return e;
      GemmArgs<double> g2{a + s * d + s, d, 1, out, n, 1, a + s * d, d, 1, n, n, n, -1.0, 0.0, 0, 1,
                          static_cast<int>(full), hop, s * s, hop};
      if (int32_t e = launch_gemm<double>(g2, st)) return e;
      first = full * 2 * s;
    }
    for (long long p = first; p + s < d; p += 2 * s) {
      const int n2 = static_cast<int>(d - (p + s) < s ? d - (p + s) : s);
      const int n1 = static_cast<int>(s);
      double* l11 = a + p * d + p;                 // n1 x n1, inverted, lower
      double* l21 = a + (p + s) * d + p;           // n2 x n1
      double* l22 = a + (p + s) * d + (p + s);     // n2 x n2, inverted, lower
      double* t = out;                             // n2 x n1 scratch, row stride n1
      GemmArgs<double> g1{l21, d, 1, l11, d, 1, t, n1, 1, n2, n1, n1, 1.0, 0.0, 0, 3};
      if (int32_t e = launch_gemm<double>(g1, st)) return e;
      GemmArgs<double> g2{l22, d, 1, t, n1, 1, l21, d, 1, n2, n1, n2, -1.0, 0.0, 0, 1};
      if (int32_t e = launch_gemm<double>(g2, st)) return e;
    }
  }
  MI355Q_CHECK_LAUNCH("gptq trtri launch");
  // ---- H^-1 = L^-T L^-1 : out(i,j) = sum_k Linv[k][i] * Linv[k][j], k >= max(i,j); lower half
  // (stored as float32 straight from the accumulators; the upper triangle is mirrored afterwards)
  if (split) {
    if (int32_t s = ltl_bf16x3(a, d, hinv_out, out, st)) return s;
  } else {
    GemmArgs<double> gp{a, 1, d, a, d, 1, out, d, 1, d, d, d, 1.0, 0.0, 1, 2};
    gp.c32 = hinv_out;
    if (int32_t s = launch_gemm<double>(gp, st)) return s;
  }
  {
    const unsigned nt = static_cast<unsigned>((d + 31) / 32);
    hipLaunchKernelGGL(mirror_lower_f32_kernel, dim3(nt, nt), dim3(256), 0, st, hinv_out, d);
  }
  MI355Q_CHECK_LAUNCH("gptq mirror launch");
  return MI355Q_OK;
}
}  // namespace

extern "C" int32_t mi355q_gptq_hinv_f64(const double* hessian, int64_t d, double damp_factor, float* hinv_out,
                                        int32_t* info_out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!hessian) { clear_error(); return d == 0 ? MI355Q_OK : fail(d < 0 ? MI355Q_BAD_ARG : MI355Q_BAD_ARG, d < 0 ? "negative shape" : "null pointer"); }
  return hinv_impl(hessian, nullptr, 1.0, d, damp_factor, hinv_out, info_out, workspace, workspace_bytes, stream);
}

extern "C" int32_t mi355q_gptq_hinv_from_product_f32(const float* product, int64_t d, double alpha, double damp_factor,
                                                     float* hinv_out, int32_t* info_out, void* workspace,
                                                     size_t workspace_bytes, void* stream) {
  if (!product) { clear_error(); return d == 0 ? MI355Q_OK : fail(MI355Q_BAD_ARG, d < 0 ? "negative shape" : "null pointer"); }
  return hinv_impl(nullptr, product, alpha, d, damp_factor, hinv_out, info_out, workspace, workspace_bytes, stream);
}

// ---- several independent inverses at once ---------------------------------------------------
// A d = 2048 inverse is a chain of ~32 dependent step kernels of ~28 us, each a few hundred
// workgroups at most: 250 CUs idle through it, and a Gemma-2B has 54 such Hessians. The chains of
// different Hessians share nothing, so they go out on a small pool of streams (per device, made on
// first use, released by mi355q_shutdown) and interleave on the chip; each lane of the pool has its
// own slice of the workspace, instances on one lane follow each other. Every instance runs the
// launches of mi355q_gptq_hinv_f64 unchanged: bit-identical results.
namespace {
constexpr int kHinvLanes = 8;
struct HinvPool {
  hipStream_t lane[kHinvLanes] = {};
  hipEvent_t fork = nullptr, done[kHinvLanes] = {};
  bool tried = false, ok = false;
};
HinvPool g_hinv_pool[64];
std::mutex g_hinv_pool_mutex;

HinvPool* hinv_pool() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  HinvPool& p = g_hinv_pool[dev];
  if (!p.tried) {
    p.tried = true;
    bool ok = getenv("MI355Q_HINV_NO_LANES") == nullptr && hipEventCreateWithFlags(&p.fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < kHinvLanes; ++i)
      ok = hipStreamCreateWithFlags(&p.lane[i], hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&p.done[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    p.ok = ok;
  }
  return p.ok ? &p : nullptr;
}

size_t hinv_lane_bytes(int64_t d) { return (mi355q_gptq_hinv_workspace_bytes(d) + 255) & ~static_cast<size_t>(255); }
// d >= 4096: one matrix at a time by default. MI355Q_HINV_PAIRS=1 keeps TWO in flight (a lane each, each with the
// look-ahead stream of its slot): built and measured in round 5 (tools/hinv_pairs_bench.py, profiles/r05_hinv_pairs.txt) --
// bit-identical and SLOWER: 55.6 against 53.2 ms per d = 16384 inverse, 12.7 against 10.5 at 8192, 5.2 against 4.0 at 4608.
// A lone call already keeps the machine busy (its own rank-512 update beside its own chain); what it loses, it loses to the
// two interfering (the GEMM at 42-50 of its 64 TFLOP/s, the chain's kernels waiting for slots), and a second
// matrix adds a second pair of the same to the same CUs instead of filling idle ones.
int hinv_lanes_for(int32_t count, int64_t d) {
  if (count < 2) return 1;
  if (d >= 4096) {
    static const bool pairs = [] { const char* e = getenv("MI355Q_HINV_PAIRS"); return e != nullptr && atoi(e) != 0; }();
    return pairs ? kSideSlots : 1;
  }
  return count < kHinvLanes ? count : kHinvLanes;
}
// matrices that advance in lock step (hinv_lockstep): whole 64-column steps only, at most kHinvGroup, at most 4 GiB of workspace
bool hinv_lockstep_ok(int32_t count, int64_t d) {
  static const bool on = getenv("MI355Q_HINV_LANES") == nullptr;
  return on && count >= 2 && d >= 2 * NB && d < 4096 && d % NB == 0;
}
int hinv_group_for(int32_t count, int64_t d) {
  const size_t fit = (static_cast<size_t>(4) << 30) / hinv_lane_bytes(d);
  int g = count < kHinvGroup ? count : kHinvGroup;
  if (static_cast<size_t>(g) > fit) g = static_cast<int>(fit);
  return g < 1 ? 1 : g;
}

// G <= kHinvGroup damped inverses of order d, all steps in lock step (the kernels above). ws: G slices of `per` bytes,
// each laid out as mi355q_gptq_hinv_f64's workspace.
int32_t hinv_lockstep(const double* const* hessians, int G, int d, double damp_factor, float* const* outs, int32_t* info,
                      unsigned char* ws, size_t per, hipStream_t st) {
  HinvTable tab{};
  for (int z = 0; z < G; ++z) {
    if (!hessians[z] || !outs[z]) return fail(MI355Q_BAD_ARG, "null pointer");
    tab.h[z] = hessians[z];
    tab.out[z] = outs[z];
  }
  const long long ms = static_cast<long long>(per / sizeof(double));      // matrix stride in doubles
  double* a = reinterpret_cast<double*>(ws);
  double* out = a + static_cast<size_t>(d) * d;
  double* lt = out + static_cast<size_t>(d) * d;
  double* scal = lt + NB * NB + static_cast<size_t>(d) * NB * 2;
  const int nblocks = d / NB;
  const unsigned uG = static_cast<unsigned>(G);
  if (hipMemsetAsync(info, 0, sizeof(int32_t) * G, st) != hipSuccess) return fail(MI355Q_HIP_ERROR, "hipMemsetAsync failed");
  hipLaunchKernelGGL(diag_sum_batched_kernel, dim3(1, 1, uG), dim3(256), 0, st, tab, d, scal, ms);
  hipLaunchKernelGGL(copy_damped_lower_batched_kernel, dim3(grid1d(static_cast<long long>(d) * d), 1, uG), dim3(256), 0, st, tab, d,
                     scal, damp_factor, a, ms);
  MI355Q_CHECK_LAUNCH("gptq damp launch");
  static const int OB = [] { const char* e = getenv("MI355Q_CHOL_OB"); const int v = e ? atoi(e) : 0; return v >= 64 && v % 64 == 0 ? v : 8 * NB; }();
  for (int k0 = 0; k0 < d; k0 += OB) {
    const int ob = d - k0 < OB ? d - k0 : OB;
    for (int k = k0; k < k0 + ob; k += NB) {
      const int m = d - k - NB;             // rows below the diagonal block
      const int w = k0 + ob - k - NB;       // columns left in this outer block
      hipLaunchKernelGGL(potf2_batched_kernel, dim3(1, 1, uG), dim3(kPotf2Threads), 0, st, a, ms, d, k, info, lt);
      if (m > 0) {
        hipLaunchKernelGGL(solve_batched_kernel, dim3(static_cast<unsigned>(m / NB), 1, uG), dim3(256), 0, st, a, ms, d, k, lt);
        if (w > 0)
          hipLaunchKernelGGL(update_batched_kernel, dim3(static_cast<unsigned>(m / NB), static_cast<unsigned>(w / NB), uG),
                             dim3(kPotf2Threads), 0, st, a, ms, d, k);
      }
    }
    const int m2 = d - k0 - ob;
    if (m2 > 0) {
      const double* l = a + static_cast<long long>(k0 + ob) * d + k0;
      double* c = a + static_cast<long long>(k0 + ob) * d + k0 + ob;
      GemmArgs<double> gu{l, d, 1, l, 1, d, c, d, 1, m2, m2, ob, -1.0, 1.0, 1, 0};
      gu.outer = G; gu.oa = gu.ob = gu.oc = ms;
      if (int32_t e = launch_gemm<double>(gu, st)) return e;
    }
  }
  MI355Q_CHECK_LAUNCH("gptq cholesky launch");
  hipLaunchKernelGGL(diag_inverse_kernel, dim3(static_cast<unsigned>(nblocks), 1, uG), dim3(256), 0, st, a, d, ms);
  for (long long s = NB; s < d; s *= 2) {
    long long first = 0;
    const long long full = (d - s) / (2 * s) + ((d - s) % (2 * s) >= s ? 1 : 0);
    if (full >= 2 && s <= 2048) {
      const int n = static_cast<int>(s);
      const long long hop = 2 * s * (static_cast<long long>(d) + 1);
      GemmArgs<double> g1{a + s * d, d, 1, a, d, 1, out, n, 1, n, n, n, 1.0, 0.0, 0, 3, static_cast<int>(full), hop, hop, s * s};
      g1.outer = G; g1.oa = g1.ob = g1.oc = ms;
      if (int32_t e = launch_gemm<double>(g1, st)) return e;
      GemmArgs<double> g2{a + s * d + s, d, 1, out, n, 1, a + s * d, d, 1, n, n, n, -1.0, 0.0, 0, 1, static_cast<int>(full), hop, s * s, hop};
      g2.outer = G; g2.oa = g2.ob = g2.oc = ms;
      if (int32_t e = launch_gemm<double>(g2, st)) return e;
      first = full * 2 * s;
    }
    for (long long p = first; p + s < d; p += 2 * s) {
      const int n2 = static_cast<int>(d - (p + s) < s ? d - (p + s) : s);
      const int n1 = static_cast<int>(s);
      double* l11 = a + p * d + p;
      double* l21 = a + (p + s) * d + p;
      double* l22 = a + (p + s) * d + (p + s);
      GemmArgs<double> g1{l21, d, 1, l11, d, 1, out, n1, 1, n2, n1, n1, 1.0, 0.0, 0, 3};
      g1.outer = G; g1.oa = g1.ob = g1.oc = ms;
      if (int32_t e = launch_gemm<double>(g1, st)) return e;
      GemmArgs<double> g2{l22, d, 1, out, n1, 1, l21, d, 1, n2, n1, n2, -1.0, 0.0, 0, 1};
      g2.outer = G; g2.oa = g2.ob = g2.oc = ms;
      if (int32_t e = launch_gemm<double>(g2, st)) return e;
    }
  }
  MI355Q_CHECK_LAUNCH("gptq trtri launch");
  // H^-1 = L^-T L^-1, lower half, as float32 into the slice's `out` region; then out[z] <- its mirror image
  GemmArgs<double> gp{a, 1, d, a, d, 1, out, d, 1, d, d, d, 1.0, 0.0, 1, 2};
  gp.c32 = reinterpret_cast<float*>(out);
  gp.outer = G; gp.oa = gp.ob = gp.oc = ms; gp.oc32 = 2 * ms;
  if (int32_t e = launch_gemm<double>(gp, st)) return e;
  const unsigned nt = static_cast<unsigned>((d + 31) / 32);
  hipLaunchKernelGGL(mirror_out_batched_kernel, dim3(nt, nt, uG), dim3(256), 0, st, tab, reinterpret_cast<const float*>(out), 2 * ms, d);
  MI355Q_CHECK_LAUNCH("gptq mirror launch");
  return MI355Q_OK;
}
}  // namespace

namespace {
void prepare_hinv_pool() {
  std::lock_guard<std::mutex> lock(g_hinv_pool_mutex);
  (void)hinv_pool();
}

void release_hinv_pools() {
  std::lock_guard<std::mutex> lock(g_hinv_pool_mutex);
  for (HinvPool& p : g_hinv_pool) {
    if (p.ok) {
      for (int i = 0; i < kHinvLanes; ++i) {
        (void)hipStreamSynchronize(p.lane[i]);
        (void)hipStreamDestroy(p.lane[i]);
        (void)hipEventDestroy(p.done[i]);
      }
      (void)hipEventDestroy(p.fork);
    }
    p = HinvPool();
  }
}
}  // namespace

extern "C" size_t mi355q_gptq_hinv_batched_workspace_bytes(int32_t count, int64_t d) {
  if (count <= 0 || d <= 0) return 0;
  if (hinv_lockstep_ok(count, d)) return hinv_lane_bytes(d) * static_cast<size_t>(hinv_group_for(count, d));
  return hinv_lane_bytes(d) * static_cast<size_t>(hinv_lanes_for(count, d));
}
// (Hessians handed over as float32 products are never taken in lock step: one slice per lane)
extern "C" size_t mi355q_gptq_hinv_from_product_batched_workspace_bytes(int32_t count, int64_t d) {
  if (count <= 0 || d <= 0) return 0;
  return hinv_lane_bytes(d) * static_cast<size_t>(hinv_lanes_for(count, d));
}

namespace {
// hessians_host[i] (FLOAT64) or, when that table is null, alphas_host[i] * products_host[i] (FLOAT32 products, lower triangle)
int32_t hinv_batched_impl(const double* const* hessians_host, const float* const* products_host, const double* alphas_host,
                          int32_t count, int64_t d, double damp_factor, float* const* hinv_out_host, int32_t* info_out,
                          void* workspace, size_t workspace_bytes, void* stream) {
  clear_error();
  if (count < 0 || d < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  if (count == 0 || d == 0) return MI355Q_OK;
  if ((!hessians_host && (!products_host || !alphas_host)) || !hinv_out_host || !info_out) return fail(MI355Q_BAD_ARG, "null pointer");
  for (int32_t i = 0; i < count; ++i)
    if (!(hessians_host ? static_cast<const void*>(hessians_host[i]) : static_cast<const void*>(products_host[i])) || !hinv_out_host[i])
      return fail(MI355Q_BAD_ARG, "null pointer");
  const bool lockstep = hessians_host != nullptr && hinv_lockstep_ok(count, d);
  const size_t per = hinv_lane_bytes(d);
  const size_t need = per * static_cast<size_t>(lockstep ? hinv_group_for(count, d) : hinv_lanes_for(count, d));
  if (!workspace || workspace_bytes < need) return fail(MI355Q_BAD_ARG, "workspace too small: need %zu bytes", need);
  auto one = [&](int32_t i, void* ws, void* st, int slot) {
    return hinv_impl(hessians_host ? hessians_host[i] : nullptr, hessians_host ? nullptr : products_host[i],
                     hessians_host ? 1.0 : alphas_host[i], d, damp_factor, hinv_out_host[i], info_out + i, ws, per, st, slot);
  };
  if (lockstep) {
    // equally sized small Hessians advance through every step together: three launches per 64-column step for all of them
    const int group = hinv_group_for(count, d);
    for (int32_t i = 0; i < count; i += group) {
      const int g = count - i < group ? count - i : group;
      if (g == 1) {
        if (int32_t e = one(i, workspace, stream, 0)) return e;
      } else if (int32_t e = hinv_lockstep(hessians_host + i, g, static_cast<int>(d), damp_factor, hinv_out_host + i, info_out + i,
                                           static_cast<unsigned char*>(workspace), per, as_stream(stream))) {
        return e;
      }
    }
    return MI355Q_OK;
  }
  int lanes = hinv_lanes_for(count, d);
  std::unique_lock<std::mutex> lock(g_hinv_pool_mutex, std::defer_lock);
  HinvPool* pool = nullptr;
  if (lanes > 1) {
    lock.lock();
    pool = hinv_pool();
    if (!pool) lanes = 1;
  }
  hipStream_t st = as_stream(stream);
  if (lanes == 1) {
    for (int32_t i = 0; i < count; ++i)
      if (int32_t e = one(i, workspace, stream, 0)) return e;
    return MI355Q_OK;
  }
  if (hipEventRecord(pool->fork, st) != hipSuccess) return fail(MI355Q_HIP_ERROR, "hinv lanes: record failed");
  for (int l = 0; l < lanes; ++l)
    if (hipStreamWaitEvent(pool->lane[l], pool->fork, 0) != hipSuccess) return fail(MI355Q_HIP_ERROR, "hinv lanes: wait failed");
  // One inverse is ~130 launches (d = 2048; a thousand at d = 16384) of a few microseconds of host time each: a single
  // thread enqueuing 54 chains is the bottleneck long before the chip is (39 ms of enqueue for 54 x 0.73 ms), and two
  // large matrices enqueued one after the other would not run side by side. Every lane therefore gets a thread of its
  // own for the duration of the call. A lane of large matrices also has the look-ahead stream of its slot.
  int32_t status = MI355Q_OK;
  {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::vector<int32_t> lane_status(static_cast<size_t>(lanes), MI355Q_OK);
    std::vector<std::string> lane_error(static_cast<size_t>(lanes));
    std::vector<std::thread> workers;
    for (int l = 0; l < lanes; ++l)
      workers.emplace_back([&, l] {
        (void)hipSetDevice(dev);
        for (int32_t i = l; i < count; i += lanes) {
          const int32_t e = one(i, static_cast<unsigned char*>(workspace) + static_cast<size_t>(l) * per, pool->lane[l],
                                d >= 4096 ? l % kSideSlots : 0);
          if (e != MI355Q_OK) {
            lane_status[static_cast<size_t>(l)] = e;
            lane_error[static_cast<size_t>(l)] = mi355q_last_error();
            break;
          }
        }
      });
    for (std::thread& w : workers) w.join();
    for (int l = 0; l < lanes; ++l)
      if (lane_status[static_cast<size_t>(l)] != MI355Q_OK && status == MI355Q_OK)
        status = fail(static_cast<mi355q_status>(lane_status[static_cast<size_t>(l)]), "%s", lane_error[static_cast<size_t>(l)].c_str());
  }
  // whatever happened, the caller's stream waits for every lane: the workspace is the caller's
  for (int l = 0; l < lanes; ++l)
    if (hipEventRecord(pool->done[l], pool->lane[l]) != hipSuccess || hipStreamWaitEvent(st, pool->done[l], 0) != hipSuccess)
      return fail(MI355Q_HIP_ERROR, "hinv lanes: join failed");
  return status;
}
}  // namespace

extern "C" int32_t mi355q_gptq_hinv_f64_batched(const double* const* hessians_host, int32_t count, int64_t d,
                                                double damp_factor, float* const* hinv_out_host, int32_t* info_out,
                                                void* workspace, size_t workspace_bytes, void* stream) {
  if (count > 0 && d > 0 && !hessians_host) { clear_error(); return fail(MI355Q_BAD_ARG, "null pointer"); }
  return hinv_batched_impl(hessians_host, nullptr, nullptr, count, d, damp_factor, hinv_out_host, info_out, workspace,
                           workspace_bytes, stream);
}

extern "C" int32_t mi355q_gptq_hinv_from_product_f32_batched(const float* const* products_host, const double* alphas_host,
                                                             int32_t count, int64_t d, double damp_factor,
                                                             float* const* hinv_out_host, int32_t* info_out, void* workspace,
                                                             size_t workspace_bytes, void* stream) {
  if (count > 0 && d > 0 && (!products_host || !alphas_host)) { clear_error(); return fail(MI355Q_BAD_ARG, "null pointer"); }
  return hinv_batched_impl(nullptr, products_host, alphas_host, count, d, damp_factor, hinv_out_host, info_out, workspace,
                           workspace_bytes, stream);
}

extern "C" size_t mi355q_gptq_apply_workspace_bytes(int64_t rows, int64_t d) {
  if (rows <= 0 || d <= 0) return 0;
  return (static_cast<size_t>(rows) * d + static_cast<size_t>(rows) * kErrLd) * sizeof(float) +
         (upd_bf16x3_usable(rows, d) ? upd_bf16x3_workspace_bytes(rows, d, kErrLd) : 0);
}

namespace {
int32_t gptq_apply_impl(const float* w, int64_t rows, int64_t d, const float* hinv,
                        const void* scale, int32_t scale_is_f64, const int32_t* zero_point,
                        int32_t scale_mode, int32_t block_size, int32_t bits, int32_t narrow,
                        int32_t zp_via_f64, int32_t diff_bits, int8_t* q_out, int32_t* q_out32,
                        void* workspace, size_t workspace_bytes, void* stream) {
  const bool wide = q_out32 != nullptr;
  if (rows < 0 || d < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  if (rows == 0 || d == 0) return MI355Q_OK;
  if (rows > 0x7FFFFFFF || d > 0x7FFFFFFF) return fail(MI355Q_UNSUPPORTED, "dimension too large");
  if (!wide && (bits < 2 || bits > 8)) return fail(MI355Q_UNSUPPORTED, "gptq apply supports 2..8 bits (wider targets: mi355q_gptq_apply_wide_f32)");
  if (wide && (bits < 9 || bits > 32)) return fail(MI355Q_UNSUPPORTED, "gptq apply (wide) supports 9..32 bits");
  if (scale_mode < 0 || scale_mode > 2) return fail(MI355Q_BAD_ARG, "bad scale_mode");
  if (scale_mode == 2 && (block_size <= 0 || d % block_size != 0))
    return fail(MI355Q_BAD_SHAPE, "Quantized dimension %lld is not divisible by block size %d.",
                static_cast<long long>(d), block_size);

  if (!w || !hinv || !scale || (!q_out && !q_out32)) return fail(MI355Q_BAD_ARG, "null pointer");
  const size_t need = mi355q_gptq_apply_workspace_bytes(rows, d);
  if (!workspace || workspace_bytes < need)
    return fail(MI355Q_BAD_ARG, "workspace too small: need %zu bytes", need);
  hipStream_t st = as_stream(stream);
  float* wc = static_cast<float*>(workspace);
  float* err = wc + rows * d;
  if (hipMemcpyAsync(wc, w, static_cast<size_t>(rows) * d * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
    return fail(MI355Q_HIP_ERROR, "hipMemcpyAsync failed");
  // wide layers: the update behind a group runs on the bf16 matrix cores (exact three-way split of
  // the errors and of Hinv, xtx_bf16x3.hip); Hinv's planes are made once here
  const bool split_upd = upd_bf16x3_usable(rows, d);
  void* upd_ws = err + rows * kErrLd;
  if (split_upd)
    if (int32_t s = upd_bf16x3_prepare(hinv, d, upd_ws, st)) return s;
  // (the bounds become float32 the way np.clip's Python floats do: 2^(bits-1) - 1 rounds up to 2^(bits-1) from 26 bits on)
  const double qmax = static_cast<double>((1LL << (bits - 1)) - 1), qmin = -static_cast<double>(1LL << (bits - 1));
  ApplyArgs a{};
  a.w = wc; a.rows = static_cast<int>(rows); a.d = static_cast<int>(d); a.hinv = hinv; a.scale = scale;
  a.zp = zero_point; a.scale_mode = scale_mode; a.block_size = block_size > 0 ? block_size : 1;
  a.nblk = scale_mode == 2 ? static_cast<int>(d / block_size) : 1;
  a.lo = static_cast<float>(narrow ? qmin + 1 : qmin); a.hi = static_cast<float>(qmax);
  a.zp_via_f64 = zp_via_f64; a.diff_bits = diff_bits; a.err = err; a.q = q_out; a.q32 = q_out32;
  a.container32 = bits > 16 ? 1 : 0;
  // (Tried: look-ahead -- only the next group's 256 columns updated on this stream, the rest on the
  // library's side stream underneath the next chain, two alternating error buffers. The chain's
  // workgroups then wait for CUs the update holds: 0.77 -> 0.91 ms at 2048 x 2048, 11.0 -> 11.1 ms
  // at [2048, 16384]. Dropped.)
  // Lazy batch updates: a block's error reaches the later blocks of its own group of kLazyBlocks
  // inside their block kernels (they are quantized next), the columns beyond the group once per
  // group, as one K = 256 product instead of four K = 64 ones -- a quarter of the passes over the
  // trailing matrix, which is what the K = 64 updates were bound by (43 % of HBM bandwidth).
  // Same mathematics as ref gptq.py:213-214 applied block by block; the far columns see the four
  // products summed inside one GEMM before the subtraction instead of four subtractions.
  for (int g0 = 0; g0 < a.d; g0 += kErrLd) {
    const int g1 = a.d - g0 < kErrLd ? a.d : g0 + kErrLd;
    // symmetric recipes with float scales: one launch takes every workgroup's rows through all of
    // the group's blocks (see gptq_rows_kernel)
    const bool plain_group = zero_point == nullptr && !(scale_mode == 2 && block_size % 32 != 0) && !scale_is_f64 &&
                             d % NB == 0 && !wide && !getenv("MI355Q_GPTQ_SPREAD");   // (gptq_rows_kernel packs bytes)
    if (plain_group) {
      a.c0 = g0;
      a.nb = (g1 - g0) / NB;
      a.err_col = 0;
      hipLaunchKernelGGL(gptq_rows_kernel, dim3(static_cast<unsigned>((rows + kRowsPerWave - 1) / kRowsPerWave)), dim3(256), 0, st, a);
    }
    for (int c0 = g0; c0 < g1 && !plain_group; c0 += NB) {
      a.c0 = c0;
      a.nb = g1 - c0 < NB ? g1 - c0 : NB;
      a.err_col = c0 - g0;
      const int rl = row_lanes_for(rows);
      const dim3 grid(static_cast<unsigned>((rows + (256 / rl) - 1) / (256 / rl)));
      const bool plain = a.nb == NB && zero_point == nullptr && !(scale_mode == 2 && block_size % 32 != 0) && !a.container32;
#define MI355Q_BLOCK(ST, RL, PL) hipLaunchKernelGGL((gptq_block_kernel<ST, RL, PL>), grid, dim3(256), 0, st, a)
      if (scale_is_f64) {
        if (rl == 32) { if (plain) MI355Q_BLOCK(double, 32, true); else MI355Q_BLOCK(double, 32, false); }
        else          { if (plain) MI355Q_BLOCK(double, 16, true); else MI355Q_BLOCK(double, 16, false); }
      } else {
        if (rl == 32) { if (plain) MI355Q_BLOCK(float, 32, true); else MI355Q_BLOCK(float, 32, false); }
        else          { if (plain) MI355Q_BLOCK(float, 16, true); else MI355Q_BLOCK(float, 16, false); }
      }
#undef MI355Q_BLOCK
    }
    if (g1 < a.d) {
      // W[:, g1:] -= err[:, group] @ Hinv[g0:g1, g1:]. One K = 256 product rounds the far columns once
      // where the reference's per-block `W[:, rest] -= err_blk @ Hinv[blk, rest]` (gptq.py:213-214) rounds
      // them four times. At 4 bits that never moved an integer (0 of 393 216 against the oracle); on an
      // 8-bit grid, 18 x finer, it did, so 8-bit targets take the reference's sequence: one product and one
      // subtraction per 64-column block (a quarter of the speed-up of the lazy update). What remains at 8 bits
      // is the order of the float32 additions INSIDE a block's product: with the same inverse 63 of 262 144
      // integers at [16384, 2048] differ from the as-stated oracle (one K = 64 sgemm pass) and NONE from the
      // oracle with that product formed as two K = 32 halves -- the oracle's own re-ordering floor is the
      // same 63 (profiles/r04_parity_rates.txt: three rates per floor-based comparison).
      static const int far_env = [] { const char* e = getenv("MI355Q_GPTQ_FAR_PER_BLOCK"); return e ? atoi(e) : -1; }();
      const bool per_block = far_env >= 0 ? far_env != 0 : bits >= 8;
      const int step = per_block ? NB : g1 - g0;
      for (int b0 = g0; b0 < g1; b0 += step) {
        const int kk = g1 - b0 < step ? g1 - b0 : step;
        const float* eb = err + (b0 - g0);
        if (split_upd && kk % 16 == 0 && g1 % 128 == 0) {
          if (int32_t s = upd_bf16x3(eb, kErrLd, rows, d, b0, kk, g1, wc, upd_ws, st)) return s;
        } else {
          GemmArgs<float> g{eb, kErrLd, 1, hinv + static_cast<long long>(b0) * d + g1, d, 1, wc + g1, d, 1,
                            a.rows, a.d - g1, kk, -1.0f, 1.0f, 0, 0};
          if (int32_t s = launch_gemm<float>(g, st)) return s;
        }
      }
    }
  }
  MI355Q_CHECK_LAUNCH("gptq apply launch");
  return MI355Q_OK;
}
}  // namespace

extern "C" int32_t mi355q_gptq_apply_f32(const float* w, int64_t rows, int64_t d, const float* hinv,
                                         const void* scale, int32_t scale_is_f64, const int32_t* zero_point,
                                         int32_t scale_mode, int32_t block_size, int32_t bits, int32_t narrow,
                                         int32_t zp_via_f64, int32_t diff_bits, int8_t* q_out,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  clear_error();
  if (rows > 0 && d > 0 && !q_out) return fail(MI355Q_BAD_ARG, "null pointer");
  return gptq_apply_impl(w, rows, d, hinv, scale, scale_is_f64, zero_point, scale_mode, block_size, bits, narrow, zp_via_f64,
                         diff_bits, q_out, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int32_t mi355q_gptq_apply_wide_f32(const float* w, int64_t rows, int64_t d, const float* hinv,
                                              const void* scale, int32_t scale_is_f64, const int32_t* zero_point,
                                              int32_t scale_mode, int32_t block_size, int32_t bits, int32_t narrow,
                                              int32_t zp_via_f64, int32_t diff_bits, int32_t* q_out,
                                              void* workspace, size_t workspace_bytes, void* stream) {
  clear_error();
  if (rows > 0 && d > 0 && !q_out) return fail(MI355Q_BAD_ARG, "null pointer");
  return gptq_apply_impl(w, rows, d, hinv, scale, scale_is_f64, zero_point, scale_mode, block_size, bits, narrow, zp_via_f64,
                         diff_bits, nullptr, q_out, workspace, workspace_bytes, stream);
}
