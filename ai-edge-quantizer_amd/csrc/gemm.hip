// Strided MFMA GEMMs for the GPTQ dense contractions (gfx950).
//
//   C[i,j] = beta * C[i,j] + alpha * sum_k A(i,k) * B(k,j)
//
// with A(i,k) at A + i*a_i + k*a_k (any strides, so transposes and sub-blocks are
// free), same for B and C. Two instantiations:
//   float  : v_mfma_f32_32x32x2_f32   -- Hessian X^T X (ref: gptq.py:100-107) and the
//            inter-block OBS update err @ Hinv (ref: gptq.py:213-214)
//   double : v_mfma_f64_16x16x4_f64   -- blocked Cholesky trailing updates, triangular
//            inverse and L^-T L^-1 (ref: gptq.py:111-128)
// The f32-input MFMA is an exact k-ordered fmaf chain (no TF32 on gfx950), i.e.
// sgemm-class numerics; see MI355X_MICROARCH.md.
//
// Structure: 256 threads = 4 waves in a 2x2 grid; each wave owns a 2x2 arrangement
// of MFMA tiles (32x32 for float -> block tile 128x128, 16x16 for double -> 64x64).
// K is consumed BK = 16 at a time through LDS tiles stored k-major ([k][i]) so that an MFMA operand
// fragment (lane -> (i = lane % T, k = lane / T)) is one conflict-free ds_read per
// lane. The LDS tiles are double buffered: the 16-byte global loads of tile t+1 are
// issued before the MFMAs of tile t and land in the other buffer afterwards, one
// barrier per K step. Long-K / few-tile shapes (the Hessian at d = 2048: K = 65536,
// 256 tiles) are split along K over gridDim.z into a workspace and summed in a fixed
// order by a second kernel (deterministic, unlike atomics).
#include <type_traits>

#include "gemm.h"

#ifndef MI355Q_BIG_MIN_K
#define MI355Q_BIG_MIN_K 256   // below this the 128x128 tile's prologue / epilogue dominate
#endif

namespace mi355q {
namespace {

// 16-byte operand pieces as NATIVE vectors: with HIP's float4 / double2 structs a staged piece was a memcpy into a stack
// object the compiler did not promote (global -> scratch -> LDS in every K step of the m-contiguous operand modes).
typedef float F32x4 __attribute__((ext_vector_type(4)));
typedef double F64x2 __attribute__((ext_vector_type(2)));

template <typename T>
struct Tile;
template <>
struct Tile<float> {
  static constexpr int MF = 32;   // MFMA tile edge
  static constexpr int KF = 2;    // k per MFMA
  static constexpr int TM = 2;    // MFMA tiles per wave along each of i, j
  static constexpr int BM = 128;  // block tile edge = 2 waves x TM x MF
  static constexpr int BK = 16;   // k per LDS stage
  static constexpr int VEC = 4;   // elements per 16-byte load
  static constexpr bool DBUF = true;  // two LDS buffers (one barrier per K step)
  using Elem = float;
  using Acc = __attribute__((ext_vector_type(16))) float;
  using Vec = F32x4;
};
template <>
struct Tile<double> {
  static constexpr int MF = 16;
  static constexpr int KF = 4;
  static constexpr int TM = 2;
  static constexpr int BM = 64;   // 2 waves x 2 tiles x 16
#ifndef MI355Q_F64_BK        // tuning hooks (tools/gemm_bench.py)
#define MI355Q_F64_BK 16
#endif
#ifndef MI355Q_F64_DBUF
#define MI355Q_F64_DBUF 1
#endif
  static constexpr int BK = MI355Q_F64_BK;
  static constexpr int VEC = 2;
  static constexpr bool DBUF = MI355Q_F64_DBUF != 0;
  using Elem = double;
  using Acc = __attribute__((ext_vector_type(4))) double;
  using Vec = F64x2;
};
// The merge products of a small triangular inverse (d = 2048: two batched launches per level, at
// most one 64 x 64 tile per CU) are chains of K / BK steps that each wait ~1.2 us for their
// operands -- the 16 MFMAs of a BK = 16 step take 0.43 us -- so for those the step is twice as
// deep (a K = 1024 product 99 -> 71 us). Not for the product L^-T L^-1, whose 528 tiles share CUs
// and lose more occupancy than they gain (0.41 -> 0.44 ms); BK = 64 and a four-stage register
// ring measured the same or worse on both.
// The same idea in FP32 for GPTQ's update of the columns behind a group: [rows, <= d] += err[rows, 256] x
// Hinv[256, <= d] is at most 224 tiles of 128 x 128 with K = 256, i.e. sixteen BK = 16 steps of ~3 us
// (operand latency; their 32 MFMAs per wave take 0.85) = 52 us whatever the width. Four steps of 64.
struct TileF32K64 {
  static constexpr int MF = 32;
  static constexpr int KF = 2;
  static constexpr int TM = 2;
  static constexpr int BM = 128;
  static constexpr int BK = 64;
  static constexpr int VEC = 4;
  static constexpr bool DBUF = false;   // 2 x 33 KB of LDS
  using Elem = float;
  using Acc = __attribute__((ext_vector_type(16))) float;
  using Vec = F32x4;
};
// ... and when even those 128 x 128 tiles are fewer than the CUs (every group but the first few of a
// d = 2048 layer), a quarter of the tile: its 512 MFMAs per wave (13.7 us at K = 256) become 128.
struct TileF32Small {
  static constexpr int MF = 32;
  static constexpr int KF = 2;
  static constexpr int TM = 1;
  static constexpr int BM = 64;
  static constexpr int BK = 64;
  static constexpr int VEC = 4;
  static constexpr bool DBUF = false;
  using Elem = float;
  using Acc = __attribute__((ext_vector_type(16))) float;
  using Vec = F32x4;
};
struct TileF64K32 {
  static constexpr int MF = 16;
  static constexpr int KF = 4;
  static constexpr int TM = 2;
  static constexpr int BM = 64;
  static constexpr int BK = 32;
  static constexpr int VEC = 2;
  static constexpr bool DBUF = false;
  using Elem = double;
  using Acc = __attribute__((ext_vector_type(4))) double;
  using Vec = F64x2;
};
// Large FP64 shapes: at 64x64 the kernel needs 16 B / cycle / CU from L2 (9.8 TB/s chip-wide)
// and saturates near 40 TFLOP/s; a 128x128 block halves that. Each wave owns 4x4 MFMA tiles
// (128 accumulator registers, kept in AccVGPRs); one LDS buffer (33 KB) is refilled from
// registers behind a second barrier - a K step is 64 MFMAs = 4096 cycles per wave, so the two
// barriers cost ~2 %, and fragments are double-buffered in registers by hand so the single
// resident wave per SIMD never waits on ds_read.
// (Round 2 re-measured the alternatives on the GPU -- two LDS buffers / one barrier per K step,
// BK = 32, both: 8192^3 63.7 -> 58.2 / 56.3 / 48.7 TFLOP/s -- and an XCD-aware tile order that
// walks 4 x 8 tile patches per XCD: the HBM fetch of the d = 16384 product fell from 83.5 to
// 61.7 GB with no change in its 26.5 ms, and the rank-512 updates got slower (13.4 -> 16.7 ms).
// The kernel is not bandwidth-bound; profiles/r02_hinv_phases.txt.)
#ifndef MI355Q_BIG_BK      // tuning hooks (tools/gemm_bench.py)
#define MI355Q_BIG_BK 16
#endif
#ifndef MI355Q_BIG_DBUF
#define MI355Q_BIG_DBUF 0
#endif
struct TileF64Big {
  static constexpr int MF = 16;
  static constexpr int KF = 4;
  static constexpr int TM = 4;
  static constexpr int BM = 128;
  static constexpr int BK = MI355Q_BIG_BK;
  static constexpr int VEC = 2;
  static constexpr bool DBUF = MI355Q_BIG_DBUF != 0;
  using Elem = double;
  using Acc = __attribute__((ext_vector_type(4))) double;
  using Vec = F64x2;
};

__device__ __forceinline__ Tile<float>::Acc mfma(float a, float b, Tile<float>::Acc c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ Tile<double>::Acc mfma(double a, double b, Tile<double>::Acc c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// row of accumulator register `reg` for this lane (C/D layouts, cdna_hip_programming.md section 3)
__device__ __forceinline__ int acc_row(float, int reg, int lane) {
  return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ int acc_row(double, int reg, int lane) { return (lane >> 4) + 4 * reg; }

__device__ __forceinline__ float comp(const F32x4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }
__device__ __forceinline__ double comp(const F64x2& v, int c) { return c == 0 ? v.x : v.y; }

// How one operand tile (BM x BK, "m" = the non-k index) is fetched.
enum LoadMode { kGeneric = 0, kMFast = 1, kKFast = 2 };

// Per-thread staging registers of one operand tile: NL x 16 bytes.
template <typename TL>
struct Stage {
  static constexpr int NL = TL::BM * TL::BK / TL::VEC / 256;
  typename TL::Vec v[NL];
};

// Global -> registers. p(m, k) = base + m*s_m + k*s_k; the tile origin is (m0, k0).
template <typename TL>
__device__ __forceinline__ void load_tile(Stage<TL>& st, const typename TL::Elem* __restrict__ base, long long s_m,
                                          long long s_k, int m0, int k0, int M, int K, int mode, int tid) {
  using T = typename TL::Elem;
  constexpr int BM = TL::BM, VEC = TL::VEC, BK = TL::BK, NL = Stage<TL>::NL;
  using Vec = typename TL::Vec;
  const bool inside = m0 + BM <= M && k0 + BK <= K;
  if (mode == kMFast && inside) {
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const int e = tid + 256 * l, mv = e % (BM / VEC), k = e / (BM / VEC);
      st.v[l] = *reinterpret_cast<const Vec*>(base + (m0 + mv * VEC) * s_m + static_cast<long long>(k0 + k) * s_k);
    }
  } else if (mode == kKFast && inside) {
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const int e = tid + 256 * l, kv = e % (BK / VEC), m = e / (BK / VEC);
      st.v[l] = *reinterpret_cast<const Vec*>(base + static_cast<long long>(m0 + m) * s_m + (k0 + kv * VEC) * s_k);
    }
  } else {  // generic strides, ragged edges: scalar loads in the kMFast register layout
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const int e = tid + 256 * l, mv = e % (BM / VEC), k = e / (BM / VEC);
      T tmp[VEC];
#pragma unroll
      for (int c = 0; c < VEC; ++c) {
        const long long m = m0 + mv * VEC + c, kk = k0 + k;
        tmp[c] = (m < M && kk < K) ? base[m * s_m + kk * s_k] : T(0);
      }
      if constexpr (VEC == 4) st.v[l] = Vec{tmp[0], tmp[1], tmp[2], tmp[3]};
      else st.v[l] = Vec{tmp[0], tmp[1]};
    }
  }
}

// Registers -> LDS tile [BK][LD] (k-major).
template <typename TL, int LD>
__device__ __forceinline__ void store_tile(const Stage<TL>& st, typename TL::Elem (*lds)[LD], int m0, int k0,
                                           int M, int K, int mode, int tid) {
  constexpr int BM = TL::BM, VEC = TL::VEC, BK = TL::BK, NL = Stage<TL>::NL;
  using Vec = typename TL::Vec;
  const bool inside = m0 + BM <= M && k0 + BK <= K;
  if (mode == kKFast && inside) {
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const int e = tid + 256 * l, kv = e % (BK / VEC), m = e / (BK / VEC);
#pragma unroll
      for (int c = 0; c < VEC; ++c) lds[kv * VEC + c][m] = comp(st.v[l], c);
    }
  } else {
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const int e = tid + 256 * l, mv = e % (BM / VEC), k = e / (BM / VEC);
      *reinterpret_cast<Vec*>(&lds[k][mv * VEC]) = st.v[l];
    }
  }
}

template <typename T>
__device__ __forceinline__ void batch_offset(GemmArgs<T>& g) {
  long long b = blockIdx.z;
  if (g.outer > 1) {
    const long long inner = g.batch > 1 ? g.batch : 1;
    const long long o = b / inner;
    b -= o * inner;
    g.A += o * g.oa;
    g.B += o * g.ob;
    g.C += o * g.oc;
    if (g.c32 != nullptr) g.c32 += o * g.oc32;
  }
  if (g.batch > 1) {
    g.A += b * g.sa;
    g.B += b * g.sb;
    g.C += b * g.sc;
  }
}

template <typename TL>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs<typename TL::Elem> g, int a_mode, int b_mode,
                                                   int k_chunk, typename TL::Elem* __restrict__ partial) {
  using T = typename TL::Elem;
  constexpr int BM = TL::BM, MF = TL::MF, KF = TL::KF, TM = TL::TM, BK = TL::BK;
  constexpr int LD = BM + TL::VEC;  // keeps every row 16-byte aligned, breaks the power-of-2 stride
  constexpr int NREG = sizeof(typename TL::Acc) / sizeof(T);
  constexpr int NBUF = TL::DBUF ? 2 : 1;
  static_assert(BM * BK / TL::VEC % 256 == 0, "whole 16-byte loads per thread per operand tile");
  batch_offset(g);
  __shared__ __attribute__((aligned(16))) T As[NBUF][BK][LD];
  __shared__ __attribute__((aligned(16))) T Bs[NBUF][BK][LD];

  const int bi = blockIdx.y, bj = blockIdx.x;
  if (g.lower_only && bj > bi) return;  // only tiles touching the lower triangle
  const int i0 = bi * BM, j0 = bj * BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = (wave >> 1) * (TM * MF), wj = (wave & 1) * (TM * MF);
  const int fi = lane % MF, fk = lane / MF;

  typename TL::Acc acc[TM][TM];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < NREG; ++r) acc[a][b][r] = T(0);

  int k_begin = 0, k_end = g.K;
  if (g.k_mode == 1) k_end = min(g.K, i0 + BM);
  if (g.k_mode == 2) k_begin = (max(i0, j0) / BK) * BK;  // i0, j0 are multiples of BM >= BK
  if (g.k_mode == 3) k_begin = (j0 / BK) * BK;
  if (k_chunk > 0) {  // split-K slice of this z
    k_begin = max(k_begin, static_cast<int>(blockIdx.z) * k_chunk);
    k_end = min(k_end, (static_cast<int>(blockIdx.z) + 1) * k_chunk);
  }

  Stage<TL> sa, sb;
  if (k_begin < k_end) {
    load_tile<TL>(sa, g.A, g.a_i, g.a_k, i0, k_begin, g.M, g.K, a_mode, tid);
    load_tile<TL>(sb, g.B, g.b_j, g.b_k, j0, k_begin, g.N, g.K, b_mode, tid);
    store_tile<TL, LD>(sa, As[0], i0, k_begin, g.M, g.K, a_mode, tid);
    store_tile<TL, LD>(sb, Bs[0], j0, k_begin, g.N, g.K, b_mode, tid);
  }
  __syncthreads();
  int buf = 0;
  for (int k0 = k_begin; k0 < k_end; k0 += BK) {
    const int kn = k0 + BK;
    const bool more = kn < k_end;
    if (more) {  // next tile's global loads fly while this tile's MFMAs run
      load_tile<TL>(sa, g.A, g.a_i, g.a_k, i0, kn, g.M, g.K, a_mode, tid);
      load_tile<TL>(sb, g.B, g.b_j, g.b_k, j0, kn, g.N, g.K, b_mode, tid);
    }
    // operand fragments are double-buffered in registers: the ds_reads of k-substep s+1 are
    // issued before the MFMAs of substep s
    T af[2][TM], bf[2][TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      af[0][t] = As[buf][fk][wi + t * MF + fi];
      bf[0][t] = Bs[buf][fk][wj + t * MF + fi];
    }
#pragma unroll
    for (int s = 0; s < BK / KF; ++s) {
      const int cur = s & 1;
      if (s + 1 < BK / KF) {
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          af[cur ^ 1][t] = As[buf][(s + 1) * KF + fk][wi + t * MF + fi];
          bf[cur ^ 1][t] = Bs[buf][(s + 1) * KF + fk][wj + t * MF + fi];
        }
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = mfma(af[cur][a], bf[cur][b], acc[a][b]);
    }
    if constexpr (TL::DBUF) {
      if (more) {
        store_tile<TL, LD>(sa, As[buf ^ 1], i0, kn, g.M, g.K, a_mode, tid);
        store_tile<TL, LD>(sb, Bs[buf ^ 1], j0, kn, g.N, g.K, b_mode, tid);
      }
      __syncthreads();
      buf ^= 1;
    } else {
      __syncthreads();  // everyone is done reading the tile
      if (more) {
        store_tile<TL, LD>(sa, As[0], i0, kn, g.M, g.K, a_mode, tid);
        store_tile<TL, LD>(sb, Bs[0], j0, kn, g.N, g.K, b_mode, tid);
      }
      __syncthreads();
    }
  }
  // ---- epilogue
  if (partial == nullptr && g.c32 == nullptr && g.beta != T(0)) {
    // read-modify-write: a row of MFMA tiles asks for its old values before it needs the first (see gemm_fast_body)
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      T old[TM][NREG];
#pragma unroll
      for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
          const long long i = i0 + wi + a * MF + acc_row(T(0), r, lane);
          const long long j = j0 + wj + b * MF + fi;
          old[b][r] = (i < g.M && j < g.N && (!g.lower_only || j <= i)) ? g.C[i * g.c_i + j * g.c_j] : T(0);
        }
#pragma unroll
      for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
          const long long i = i0 + wi + a * MF + acc_row(T(0), r, lane);
          const long long j = j0 + wj + b * MF + fi;
          const T p = g.alpha * acc[a][b][r];  // P is rounded first, as sgemm-then-update does
          if (i < g.M && j < g.N && (!g.lower_only || j <= i)) g.C[i * g.c_i + j * g.c_j] = g.beta * old[b][r] + p;
        }
    }
    return;
  }
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < NREG; ++r) {
        const long long i = i0 + wi + a * MF + acc_row(T(0), r, lane);
        const long long j = j0 + wj + b * MF + fi;
        if (i < g.M && j < g.N && (!g.lower_only || j <= i)) {
          if (partial != nullptr) {  // raw partial sums; alpha / beta applied by the reducer
            partial[(static_cast<long long>(blockIdx.z) * g.M + i) * g.N + j] = acc[a][b][r];
          } else {
            const T p = g.alpha * acc[a][b][r];  // P is rounded first, as sgemm-then-update does
            if (g.c32 != nullptr) {
              g.c32[i * g.c_i + j * g.c_j] = static_cast<float>(p);
            } else {
              T* c = g.C + i * g.c_i + j * g.c_j;
              *c = (g.beta == T(0)) ? p : g.beta * (*c) + p;
            }
          }
        }
      }
}

// ---- lean main loop for the common case -------------------------------------------------
// Whole tiles only (M, N multiples of BM; the K range a multiple of BK) and both operands in a
// 16-byte-loadable layout known at compile time: no per-step mode or bounds decisions, one
// 64-bit pointer per operand advanced by a constant. With one wave per SIMD (the 128x128 FP64
// tile) every issue slot between MFMAs counts - the generic kernel above spends ~6 extra
// instructions per MFMA on that bookkeeping and reaches 50 % MFMA-busy where this loop is
// MFMA-bound.
template <typename TL, int MODE>
struct FastOperand {
  using T = typename TL::Elem;
  using Vec = typename TL::Vec;
  static constexpr int BM = TL::BM, BK = TL::BK, VEC = TL::VEC, NL = Stage<TL>::NL;
  const T* p;          // this thread's first element of the current K step
  long long step;      // elements per K step
  long long lstride;   // elements between this thread's consecutive loads
  int lds_a, lds_b;    // LDS coordinates of load 0 (see store)

  __device__ __forceinline__ void init(const T* base, long long s_m, long long s_k, int m0, int k0, int tid) {
    if constexpr (MODE == kMFast) {       // m contiguous: a load is VEC consecutive m at one k
      constexpr int PER_K = BM / VEC;     // loads per k row
      const int mv = tid % PER_K, k = tid / PER_K;
      p = base + (m0 + mv * VEC) + static_cast<long long>(k0 + k) * s_k;
      lstride = static_cast<long long>(256 / PER_K) * s_k;
      lds_a = k;
      lds_b = mv * VEC;
    } else {                              // k contiguous: a load is VEC consecutive k at one m
      constexpr int PER_M = BK / VEC;
      const int kv = tid % PER_M, m = tid / PER_M;
      p = base + static_cast<long long>(m0 + m) * s_m + (k0 + kv * VEC);
      lstride = static_cast<long long>(256 / PER_M) * s_m;
      lds_a = kv * VEC;
      lds_b = m;
    }
    step = static_cast<long long>(BK) * s_k;
  }
  // Unconditional on purpose: a load guarded by `if (more)` makes the compiler merge the two
  // register states right behind the branch, i.e. wait for the loads before the MFMAs. On the
  // last K step the current tile is simply fetched again (and never stored).
  __device__ __forceinline__ void load(Stage<TL>& st, bool more) {
    const T* q = more ? p : p - step;
#pragma unroll
    for (int l = 0; l < NL; ++l) st.v[l] = *reinterpret_cast<const Vec*>(q + l * lstride);
    p += step;
  }
  template <int LD>
  __device__ __forceinline__ void store(const Stage<TL>& st, T (*lds)[LD]) const {
    if constexpr (MODE == kMFast) {
      constexpr int KSTEP = 256 / (BM / VEC);
#pragma unroll
      for (int l = 0; l < NL; ++l) *reinterpret_cast<Vec*>(&lds[lds_a + l * KSTEP][lds_b]) = st.v[l];
    } else {
      constexpr int MSTEP = 256 / (BK / VEC);
#pragma unroll
      for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int c = 0; c < VEC; ++c) lds[lds_a + c][lds_b + l * MSTEP] = comp(st.v[l], c);
    }
  }
};

template <typename TL, int AMODE, int BMODE>
__device__ __forceinline__ void gemm_fast_body(const GemmArgs<typename TL::Elem>& g, int k_chunk,
                                               typename TL::Elem* __restrict__ partial) {
  using T = typename TL::Elem;
  constexpr int BM = TL::BM, MF = TL::MF, KF = TL::KF, TM = TL::TM, BK = TL::BK;
  constexpr int LD = BM + TL::VEC;
  constexpr int NREG = sizeof(typename TL::Acc) / sizeof(T);
  constexpr int NBUF = TL::DBUF ? 2 : 1;
  __shared__ __attribute__((aligned(16))) T As[NBUF][BK][LD];
  __shared__ __attribute__((aligned(16))) T Bs[NBUF][BK][LD];

  int bi = blockIdx.y, bj = blockIdx.x;
  if (g.k_mode == 3 && g.lower_only == 0) {
    // K starts at the tile COLUMN's diagonal: walk the tiles column by column (grid.x = tile rows),
    // so that what runs together shares its K range and moves through one B panel in step, the
    // longest columns first -- the mirror image of k_mode 1, whose row-major walk does the same
    // for A. (Row-major, the 8192-wide first product of the top merge level ran at 55 % MFMA
    // utilisation against 83 % for its k_mode 1 twin and fetched 2.7 x the bytes.)
    bi = blockIdx.x;
    bj = blockIdx.y;
  }
  if (g.lower_only == 2) {
    // triangular grid: block r of nt (nt + 1) / 2, the longest tile rows (most K work under the
    // k_modes) first; no empty blocks above the diagonal
    // (k_mode 2: K starts at the tile row's diagonal, the first rows are the long ones)
    const int r = g.k_mode == 2 ? static_cast<int>(blockIdx.x) : static_cast<int>(gridDim.x) - 1 - static_cast<int>(blockIdx.x);
    bi = static_cast<int>((__builtin_sqrt(8.0 * r + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= r) ++bi;
    while (bi * (bi + 1) / 2 > r) --bi;
    bj = r - bi * (bi + 1) / 2;
  } else if (g.lower_only && bj > bi) {
    return;
  } else if (g.k_mode == 1) {
    bi = static_cast<int>(gridDim.y) - 1 - bi;   // K grows with the tile row: longest rows first
  }
  const int i0 = bi * BM, j0 = bj * BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = (wave >> 1) * (TM * MF), wj = (wave & 1) * (TM * MF);
  const int fi = lane % MF, fk = lane / MF;

  typename TL::Acc acc[TM][TM];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < NREG; ++r) acc[a][b][r] = T(0);

  int k_begin = 0, k_end = g.K;
  if (g.k_mode == 1) k_end = min(g.K, i0 + BM);
  if (g.k_mode == 2) k_begin = max(i0, j0);
  if (g.k_mode == 3) k_begin = j0;
  if (k_chunk > 0) {
    k_begin = max(k_begin, static_cast<int>(blockIdx.z) * k_chunk);
    k_end = min(k_end, (static_cast<int>(blockIdx.z) + 1) * k_chunk);
  }
  const int steps = (k_end - k_begin) / BK;

  FastOperand<TL, AMODE> fa;
  FastOperand<TL, BMODE> fb;
  fa.init(g.A, g.a_i, g.a_k, i0, k_begin, tid);
  fb.init(g.B, g.b_j, g.b_k, j0, k_begin, tid);
  Stage<TL> sa, sb;
  if (steps > 0) {
    fa.load(sa, true);
    fb.load(sb, true);
    fa.template store<LD>(sa, As[0]);
    fb.template store<LD>(sb, Bs[0]);
  }
  __syncthreads();
  int buf = 0;
  for (int it = 0; it < steps; ++it) {
    const bool more = it + 1 < steps;
    fa.load(sa, more);
    fb.load(sb, more);
    T af[2][TM], bf[2][TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      af[0][t] = As[buf][fk][wi + t * MF + fi];
      bf[0][t] = Bs[buf][fk][wj + t * MF + fi];
    }
#pragma unroll
    for (int s = 0; s < BK / KF; ++s) {
      const int cur = s & 1;
      if (s + 1 < BK / KF) {
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          af[cur ^ 1][t] = As[buf][(s + 1) * KF + fk][wi + t * MF + fi];
          bf[cur ^ 1][t] = Bs[buf][(s + 1) * KF + fk][wj + t * MF + fi];
        }
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = mfma(af[cur][a], bf[cur][b], acc[a][b]);
    }
    if constexpr (TL::DBUF) {
      if (more) {
        fa.template store<LD>(sa, As[buf ^ 1]);
        fb.template store<LD>(sb, Bs[buf ^ 1]);
      }
      __syncthreads();
      buf ^= 1;
    } else {
      __syncthreads();
      if (more) {
        fa.template store<LD>(sa, As[0]);
        fb.template store<LD>(sb, Bs[0]);
      }
      __syncthreads();
    }
  }
  const bool diag = g.lower_only && bi == bj;
#if !defined(MI355Q_GEMM_SERIAL_EPILOGUE)   // (tuning hook, tools/gemm_bench.py: defined = the epilogue of rounds 1-4 alone)
  // A read-modify-write tile asks for its old values a row of MFMA tiles at a time, before it needs the first of them:
  // written as `*c = beta * *c + p` per element (below), the stores and loads may alias as far as the compiler knows, so
  // every element waits for its own round trip to HBM -- TM * TM * NREG = 64 of them in a row for the 128 x 128 FP64 tile.
  // Whole tiles only here: elements above a diagonal tile's diagonal are inside C, read and not written.
  if (partial == nullptr && g.c32 == nullptr && g.beta != T(0)) {
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      T old[TM][NREG];
#pragma unroll
      for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int r = 0; r < NREG; ++r)
          old[b][r] = g.C[(i0 + wi + a * MF + acc_row(T(0), r, lane)) * g.c_i + (j0 + wj + b * MF + fi) * g.c_j];
#pragma unroll
      for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
          const long long i = i0 + wi + a * MF + acc_row(T(0), r, lane);
          const long long j = j0 + wj + b * MF + fi;
          const T p = g.alpha * acc[a][b][r];
          if (!diag || j <= i) g.C[i * g.c_i + j * g.c_j] = g.beta * old[b][r] + p;
        }
    }
    return;
  }
#endif
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < NREG; ++r) {
        const long long i = i0 + wi + a * MF + acc_row(T(0), r, lane);
        const long long j = j0 + wj + b * MF + fi;
        if (!diag || j <= i) {
          if (partial != nullptr) {
            partial[(static_cast<long long>(blockIdx.z) * g.M + i) * g.N + j] = acc[a][b][r];
          } else {
            const T p = g.alpha * acc[a][b][r];
            if (g.c32 != nullptr) {
              g.c32[i * g.c_i + j * g.c_j] = static_cast<float>(p);
            } else {
              T* c = g.C + i * g.c_i + j * g.c_j;
#if defined(MI355Q_GEMM_SERIAL_EPILOGUE)
              *c = (g.beta == T(0)) ? p : g.beta * (*c) + p;
#else
              *c = p;                    // (beta == 0 here: a read-modify-write tile has left above)
#endif
            }
          }
        }
      }
}

template <typename TL, int AMODE, int BMODE>
__global__ __launch_bounds__(256) void gemm_fast_kernel(GemmArgs<typename TL::Elem> g, int k_chunk,
                                                        typename TL::Elem* __restrict__ partial) {
  batch_offset(g);
  gemm_fast_body<TL, AMODE, BMODE>(g, k_chunk, partial);
}

// The 128x128 FP64 tile needs > 256 registers per lane (128 accumulators in AccVGPRs + staging
// + fragments): tell the compiler it owns the whole 512-entry file (one wave per SIMD), or it
// budgets for two waves, spills the staging registers to scratch and shuffles accumulators
// between AccVGPRs and VGPRs every K step.
template <int AMODE, int BMODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
void gemm_fast_big_kernel(GemmArgs<double> g, int k_chunk, double* __restrict__ partial) {
  batch_offset(g);
  gemm_fast_body<TileF64Big, AMODE, BMODE>(g, k_chunk, partial);
}

// C = beta*C + alpha * (partial[0] + partial[1] + ... ), slices added in order.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs<T> g, const T* __restrict__ partial,
                                                            int slices) {
  const long long n = static_cast<long long>(g.M) * g.N;
  const long long stride = static_cast<long long>(gridDim.x) * 256;
  for (long long e = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; e < n; e += stride) {
    const long long i = e / g.N, j = e % g.N;
    if (g.lower_only && j > i) continue;
    T s = partial[e];
    for (int z = 1; z < slices; ++z) s = s + partial[z * n + e];
    T* c = g.C + i * g.c_i + j * g.c_j;
    const T p = g.alpha * s;
    *c = (g.beta == T(0)) ? p : g.beta * (*c) + p;
  }
}

template <typename T>
int pick_mode(const T* base, long long s_m, long long s_k) {
  constexpr int VEC = Tile<T>::VEC;
  const bool aligned = (reinterpret_cast<uintptr_t>(base) & 15u) == 0;
  if (aligned && s_m == 1 && s_k % VEC == 0) return kMFast;
  if (aligned && s_k == 1 && s_m % VEC == 0) return kKFast;
  return kGeneric;
}

}  // namespace

template <typename T>
size_t gemm_splitk_workspace_bytes(int M, int N, int K, bool lower_only) {
  const int s = gemm_pick_splitk<T>(M, N, K, lower_only);
  return s > 1 ? static_cast<size_t>(s) * M * N * sizeof(T) : 0;
}

template <typename T>
int gemm_pick_splitk(int M, int N, int K, bool lower_only) {
  constexpr int BM = Tile<T>::BM;
  long long tiles = static_cast<long long>((M + BM - 1) / BM) * ((N + BM - 1) / BM);
  if (lower_only && M == N) {           // only the tiles on and below the diagonal are launched
    const long long nt = (M + BM - 1) / BM;
    tiles = nt * (nt + 1) / 2;
  }
  if (tiles >= 1024 || K < 4096) return 1;
  // 1024 workgroups run at a time (4 per CU: 97-113 VGPRs, 33 KB of LDS), so tiles x slices should
  // fill whole rounds of 1024: 136 tiles x 8 slices = 1088 was one full round and one at 6 %
  // (the d = 2048 Hessian at 72 of ~130 TFLOP/s). The fewest slices that fill their rounds to
  // within 5 % of the best filling win (every slice is another partial matrix to write and add).
  int max_s = K / 1024;        // >= 1024 of K per slice
  if (max_s > 16) max_s = 16;
  if (max_s < 1) max_s = 1;
  auto fill = [&](int s) {
    const long long wg = tiles * s, rounds = (wg + 1023) / 1024;
    return static_cast<double>(wg) / static_cast<double>(rounds * 1024);
  };
  double best_fill = 0.0;
  for (int s = 1; s <= max_s; ++s) best_fill = fill(s) > best_fill ? fill(s) : best_fill;
  for (int s = 1; s <= max_s; ++s)
    if (fill(s) >= best_fill - 0.05) return s;
  return 1;
}

template <typename TL>
int32_t launch_with(const GemmArgs<typename TL::Elem>& g, hipStream_t st, void* splitk_ws, size_t splitk_ws_bytes,
                    int a_mode, int b_mode) {
  using T = typename TL::Elem;
  constexpr int BM = TL::BM;
  int slices = 1;
  if (g.batch > 1 || g.outer > 1) {
    splitk_ws = nullptr;
  }
  if (splitk_ws != nullptr && g.k_mode == 0) {
    slices = gemm_pick_splitk<T>(g.M, g.N, g.K, g.lower_only != 0);
    if (static_cast<size_t>(slices) * g.M * g.N * sizeof(T) > splitk_ws_bytes) slices = 1;
  }
  const dim3 grid(static_cast<unsigned>((g.N + BM - 1) / BM), static_cast<unsigned>((g.M + BM - 1) / BM),
                  static_cast<unsigned>((g.batch > 1 ? g.batch : slices) * (g.outer > 1 ? g.outer : 1)));
  // whole tiles + 16-byte loadable operands: the lean kernel
  const bool whole = g.M % BM == 0 && g.N % BM == 0 && g.K % TL::BK == 0 && a_mode != kGeneric &&
                     b_mode != kGeneric;
  int chunk = 0;
  if (slices > 1) {
    constexpr int BK = TL::BK;
    chunk = (g.K + slices - 1) / slices;
    chunk = (chunk + BK - 1) / BK * BK;
  }
#if defined(MI355Q_GEMM_NOFAST)
  if (false) {
#else
  if (whole) {
#endif
    T* part = slices > 1 ? static_cast<T*>(splitk_ws) : nullptr;
    GemmArgs<T> h = g;
    dim3 fgrid = grid;
    if (g.k_mode == 3 && g.lower_only == 0) fgrid = dim3(grid.y, grid.x, grid.z);   // column-major tile walk
    if (g.lower_only == 1 && g.M == g.N) {
      const unsigned nt = grid.y;
      h.lower_only = 2;
      fgrid = dim3(nt * (nt + 1) / 2, 1, grid.z);
    }
#define MI355Q_FAST(AM, BMO)                                                                        \
  do {                                                                                               \
    if constexpr (std::is_same_v<TL, TileF64Big>)                                                    \
      hipLaunchKernelGGL((gemm_fast_big_kernel<AM, BMO>), fgrid, dim3(256), 0, st, h, chunk, part);  \
    else                                                                                             \
      hipLaunchKernelGGL((gemm_fast_kernel<TL, AM, BMO>), fgrid, dim3(256), 0, st, h, chunk, part);  \
  } while (0)
    if (a_mode == kMFast && b_mode == kMFast) MI355Q_FAST(kMFast, kMFast);
    else if (a_mode == kMFast) MI355Q_FAST(kMFast, kKFast);
    else if (b_mode == kMFast) MI355Q_FAST(kKFast, kMFast);
    else MI355Q_FAST(kKFast, kKFast);
#undef MI355Q_FAST
    MI355Q_CHECK_LAUNCH("gemm launch");
    if (slices > 1) {
      long long n = static_cast<long long>(g.M) * g.N;
      unsigned blocks = static_cast<unsigned>((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
      hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, st, g,
                         static_cast<const T*>(splitk_ws), slices);
      MI355Q_CHECK_LAUNCH("gemm split-k reduce launch");
    }
    return MI355Q_OK;
  }
  if (slices > 1) {
    hipLaunchKernelGGL((gemm_kernel<TL>), grid, dim3(256), 0, st, g, a_mode, b_mode, chunk,
                       static_cast<T*>(splitk_ws));
    MI355Q_CHECK_LAUNCH("gemm launch");
    long long n = static_cast<long long>(g.M) * g.N;
    unsigned blocks = static_cast<unsigned>((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, st, g,
                       static_cast<const T*>(splitk_ws), slices);
    MI355Q_CHECK_LAUNCH("gemm split-k reduce launch");
  } else {
    hipLaunchKernelGGL((gemm_kernel<TL>), grid, dim3(256), 0, st, g, a_mode, b_mode, 0, static_cast<T*>(nullptr));
    MI355Q_CHECK_LAUNCH("gemm launch");
  }
  return MI355Q_OK;
}

template <typename T>
int32_t launch_gemm(const GemmArgs<T>& g, hipStream_t st, void* splitk_ws, size_t splitk_ws_bytes) {
  if (g.M <= 0 || g.N <= 0) return MI355Q_OK;
  const int a_mode = pick_mode<T>(g.A, g.a_i, g.a_k);
  const int b_mode = pick_mode<T>(g.B, g.b_j, g.b_k);
  if constexpr (sizeof(T) == 8) {
    // the 128x128 tile pays when there are enough of them to fill the chip; k_mode offsets are
    // multiples of the block edge either way
    const long long tiles128 = static_cast<long long>((g.M + 127) / 128) * ((g.N + 127) / 128);
    const bool splitk = splitk_ws != nullptr && g.k_mode == 0 && gemm_pick_splitk<T>(g.M, g.N, g.K, g.lower_only != 0) > 1;
    const bool whole128 = g.M % 128 == 0 && g.N % 128 == 0 && g.K % 16 == 0 && a_mode != kGeneric &&
                          b_mode != kGeneric;
#ifndef MI355Q_BIG_MIN_TRI_TILES
#define MI355Q_BIG_MIN_TRI_TILES 1024
#endif
#if !defined(MI355Q_GEMM_NOBIG)   // tuning hook (tools/gemm_bench.py)
    // triangular outputs (lower_only) leave the chip half empty at the tail with tiles this big
    if (tiles128 >= 256 && !splitk && whole128 && g.K >= MI355Q_BIG_MIN_K &&
        (!g.lower_only || tiles128 >= MI355Q_BIG_MIN_TRI_TILES))
      return launch_with<TileF64Big>(g, st, nullptr, 0, a_mode, b_mode);
#endif
  }
  if constexpr (sizeof(T) == 8) {
    static const long long kK32MaxTiles = [] { const char* e = getenv("MI355Q_K32_MAX_TILES"); return e ? atoll(e) : 512LL; }();
    const long long tiles64 = static_cast<long long>((g.M + 63) / 64) * ((g.N + 63) / 64) * (g.batch > 1 ? g.batch : 1);
    if ((g.k_mode == 1 || g.k_mode == 3 || (g.k_mode == 0 && g.lower_only != 0)) && tiles64 <= kK32MaxTiles && g.M % 64 == 0 && g.N % 64 == 0 && g.K % 32 == 0 &&
        g.K >= 128 && a_mode != kGeneric && b_mode != kGeneric)
      return launch_with<TileF64K32>(g, st, nullptr, 0, a_mode, b_mode);
  }
  if constexpr (sizeof(T) == 4) {
    const long long tiles128 = static_cast<long long>((g.M + 127) / 128) * ((g.N + 127) / 128) * (g.batch > 1 ? g.batch : 1);
    static const bool f32_k64 = getenv("MI355Q_NO_F32_K64") == nullptr;
    const bool short_k = f32_k64 && g.k_mode == 0 && !g.lower_only && g.K % 64 == 0 && g.K >= 128 && g.K <= 1024 &&
                         a_mode != kGeneric && b_mode != kGeneric;
    if (short_k && tiles128 <= 256 && g.M % 64 == 0 && g.N % 64 == 0)
      return launch_with<TileF32Small>(g, st, nullptr, 0, a_mode, b_mode);
    if (short_k && tiles128 <= 512 && g.M % 128 == 0 && g.N % 128 == 0)
      return launch_with<TileF32K64>(g, st, nullptr, 0, a_mode, b_mode);
  }
  return launch_with<Tile<T>>(g, st, splitk_ws, splitk_ws_bytes, a_mode, b_mode);
}

template int32_t launch_gemm<float>(const GemmArgs<float>&, hipStream_t, void*, size_t);
template int32_t launch_gemm<double>(const GemmArgs<double>&, hipStream_t, void*, size_t);
template int gemm_pick_splitk<float>(int, int, int, bool);
template int gemm_pick_splitk<double>(int, int, int, bool);
template size_t gemm_splitk_workspace_bytes<float>(int, int, int, bool);
template size_t gemm_splitk_workspace_bytes<double>(int, int, int, bool);

}  // namespace mi355q

using namespace mi355q;

namespace {
template <typename T>
int32_t gemm_entry(const void* A, int64_t a_i, int64_t a_k, const void* B, int64_t b_k, int64_t b_j,
                   void* C, int64_t c_i, int64_t c_j, int64_t M, int64_t N, int64_t K, double alpha,
                   double beta, int32_t lower_only, void* stream) {
  clear_error();
  if (M < 0 || N < 0 || K < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  if (M > 0x7FFFFFFF || N > 0x7FFFFFFF || K > 0x7FFFFFFF) return fail(MI355Q_UNSUPPORTED, "dimension too large");
  if (M == 0 || N == 0) return MI355Q_OK;
  if (!A || !B || !C) return fail(MI355Q_BAD_ARG, "null pointer");
  GemmArgs<T> g{static_cast<const T*>(A), a_i, a_k, static_cast<const T*>(B), b_k, b_j,
                static_cast<T*>(C), c_i, c_j, static_cast<int>(M), static_cast<int>(N),
                static_cast<int>(K), static_cast<T>(alpha), static_cast<T>(beta), lower_only, 0};
  return launch_gemm<T>(g, as_stream(stream), nullptr, 0);
}
}  // namespace

extern "C" int32_t mi355q_gemm_f32(const float* A, int64_t a_i, int64_t a_k, const float* B, int64_t b_k,
                                   int64_t b_j, float* C, int64_t c_i, int64_t c_j, int64_t M, int64_t N,
                                   int64_t K, float alpha, float beta, int32_t lower_only, void* stream) {
  return gemm_entry<float>(A, a_i, a_k, B, b_k, b_j, C, c_i, c_j, M, N, K, alpha, beta, lower_only, stream);
}

extern "C" int32_t mi355q_gemm_f64(const double* A, int64_t a_i, int64_t a_k, const double* B, int64_t b_k,
                                   int64_t b_j, double* C, int64_t c_i, int64_t c_j, int64_t M, int64_t N,
                                   int64_t K, double alpha, double beta, int32_t lower_only, void* stream) {
  return gemm_entry<double>(A, a_i, a_k, B, b_k, b_j, C, c_i, c_j, M, N, K, alpha, beta, lower_only, stream);
}
