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
  int lower_only;      // write only j <= i (square C), skip tiles above the diagonal; 1: triangular
                       // launch grid (no empty workgroups), 3: square grid with early exit (keeps
                       // blockIdx.x = tile column, i.e. one set of B panels per XCD's L2, for long K)
  int k_mode;          // 0: all k; 1: A(i,k) == 0 for k > i (lower-triangular A): k < i0+BM;
                       // 2: A(i,k) == 0 for k < i and B(k,j) == 0 for k < j: k >= max(i0, j0)
                       // 3: B(k,j) == 0 for k < j (lower-triangular B): k >= j0
  int batch;           // > 1: grid.z independent problems of this shape; operand b starts
  long long sa, sb, sc;  //      batch strides (elements) after problem b - 1 (no split-K then)
  float* c32;          // not null (beta == 0, no split-K): alpha * (A.B) is stored here as float32, same strides, C untouched
  // An OUTER batch on top (gptq.hip: the same GEMM of several equally sized Hessian inverses in one launch): grid.z =
  // max(batch, 1) * outer, outer problem o starts oa / ob / oc elements (oc32 floats for c32) after problem o - 1. It
  // takes no part in the choice of the tile kernel, which stays the single problem's: the same bits as `outer` launches.
  int outer;
  long long oa, ob, oc, oc32;
};

// Number of K slices launch_gemm would use for this shape (1 = no split) and the
// workspace that needs; pass a workspace to launch_gemm to allow the split.
template <typename T>
int gemm_pick_splitk(int M, int N, int K, bool lower_only = false);
template <typename T>
size_t gemm_splitk_workspace_bytes(int M, int N, int K, bool lower_only = false);

template <typename T>
int32_t launch_gemm(const GemmArgs<T>& g, hipStream_t st, void* splitk_ws = nullptr,
                    size_t splitk_ws_bytes = 0);

}  // namespace mi355q
