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
// of MFMA tiles; K is consumed BK = 16 at a time through LDS tiles stored k-major
// ([k][i]) so that an MFMA operand fragment (lane -> (i = lane % T, k = lane / T))
// is one conflict-free ds_read per lane.
#include "gemm.h"

namespace mi355q {
namespace {

constexpr int BK = 16;

template <typename T>
struct Tile;
template <>
struct Tile<float> {
  static constexpr int MF = 32;        // MFMA tile edge
  static constexpr int KF = 2;         // k per MFMA
  static constexpr int BM = 128;       // block tile edge (2 waves x 2 MFMA tiles x 32)
  using Acc = __attribute__((ext_vector_type(16))) float;
};
template <>
struct Tile<double> {
  static constexpr int MF = 16;
  static constexpr int KF = 4;
  static constexpr int BM = 64;        // 2 waves x 2 MFMA tiles x 16
  using Acc = __attribute__((ext_vector_type(4))) double;
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

template <typename T>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs<T> g) {
  using TL = Tile<T>;
  constexpr int BM = TL::BM, MF = TL::MF, KF = TL::KF;
  constexpr int PAD = 4;
  constexpr int NREG = sizeof(typename TL::Acc) / sizeof(T);
  __shared__ T As[BK][BM + PAD];
  __shared__ T Bs[BK][BM + PAD];

  const int bi = blockIdx.y, bj = blockIdx.x;
  if (g.lower_only && bj > bi) return;  // only tiles touching the lower triangle
  const int i0 = bi * BM, j0 = bj * BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = (wave >> 1) * (2 * MF), wj = (wave & 1) * (2 * MF);
  const int fi = lane % MF, fk = lane / MF;

  typename TL::Acc acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < NREG; ++r) acc[a][b][r] = T(0);

  const bool a_k_fast = g.a_k == 1 && g.a_i != 1;  // which index is contiguous in memory
  const bool b_k_fast = g.b_k == 1 && g.b_j != 1;

  int k_begin = 0, k_end = g.K;
  if (g.k_mode == 1) k_end = min(g.K, i0 + BM);
  if (g.k_mode == 2) k_begin = (max(i0, j0) / BK) * BK;
  for (int k0 = k_begin; k0 < k_end; k0 += BK) {
    // ---- stage A(i0:i0+BM, k0:k0+BK) and B(k0:k0+BK, j0:j0+BM) into LDS
#pragma unroll
    for (int e = tid; e < BM * BK; e += 256) {
      int i, k;
      if (a_k_fast) { k = e % BK; i = e / BK; } else { i = e % BM; k = e / BM; }
      const long long gi = i0 + i, gk = k0 + k;
      As[k][i] = (gi < g.M && gk < g.K) ? g.A[gi * g.a_i + gk * g.a_k] : T(0);
    }
#pragma unroll
    for (int e = tid; e < BM * BK; e += 256) {
      int j, k;
      if (b_k_fast) { k = e % BK; j = e / BK; } else { j = e % BM; k = e / BM; }
      const long long gj = j0 + j, gk = k0 + k;
      Bs[k][j] = (gj < g.N && gk < g.K) ? g.B[gk * g.b_k + gj * g.b_j] : T(0);
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; kk += KF) {
      const T a0 = As[kk + fk][wi + fi], a1 = As[kk + fk][wi + MF + fi];
      const T b0 = Bs[kk + fk][wj + fi], b1 = Bs[kk + fk][wj + MF + fi];
      acc[0][0] = mfma(a0, b0, acc[0][0]);
      acc[0][1] = mfma(a0, b1, acc[0][1]);
      acc[1][0] = mfma(a1, b0, acc[1][0]);
      acc[1][1] = mfma(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }
  // ---- epilogue: C = beta*C + alpha*P  (P rounded first, as sgemm-then-update does)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < NREG; ++r) {
        const long long i = i0 + wi + a * MF + acc_row(T(0), r, lane);
        const long long j = j0 + wj + b * MF + fi;
        if (i < g.M && j < g.N && (!g.lower_only || j <= i)) {
          T* c = g.C + i * g.c_i + j * g.c_j;
          const T p = g.alpha * acc[a][b][r];
          *c = (g.beta == T(0)) ? p : g.beta * (*c) + p;
        }
      }
}

}  // namespace

template <typename T>
int32_t launch_gemm(const GemmArgs<T>& g, hipStream_t st) {
  if (g.M <= 0 || g.N <= 0) return MI355Q_OK;
  constexpr int BM = Tile<T>::BM;
  const dim3 grid(static_cast<unsigned>((g.N + BM - 1) / BM), static_cast<unsigned>((g.M + BM - 1) / BM));
  hipLaunchKernelGGL((gemm_kernel<T>), grid, dim3(256), 0, st, g);
  MI355Q_CHECK_LAUNCH("gemm launch");
  return MI355Q_OK;
}

template int32_t launch_gemm<float>(const GemmArgs<float>&, hipStream_t);
template int32_t launch_gemm<double>(const GemmArgs<double>&, hipStream_t);

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
  return launch_gemm<T>(g, as_stream(stream));
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
