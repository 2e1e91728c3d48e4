// Internal API of gemm.hip (strided MFMA GEMMs) for the other translation units.
#pragma once

#include "common.h"

namespace mi355q {

template <typename T>
struct GemmArgs {
  const T* A;
  long long a_i, a_k;  // element strides of A(i, k)
  const T* B;
  long long b_k, b_j;  // element strides of B(k, j)
  T* C;
  long long c_i, c_j;
  int M, N, K;
  T alpha, beta;       // C = beta*C + alpha*(A.B); beta == 0 never reads C
  int lower_only;      // write only j <= i (square C), skip tiles above the diagonal
  int k_mode;          // 0: all k; 1: A(i,k) == 0 for k > i (lower-triangular A): k < i0+BM;
                       // 2: A(i,k) == 0 for k < i and B(k,j) == 0 for k < j: k >= max(i0, j0)
};

template <typename T>
int32_t launch_gemm(const GemmArgs<T>& g, hipStream_t st);

}  // namespace mi355q
