// Block-diagonal Hadamard rotation (K6): out = reshape(x, (-1, h)) @ (H_h / sqrt(h)).
//
//   ref: algorithms/uniform_quantize/hadamard_rotation.py:48-90 (Sylvester H / sqrt(h), FP32)
//   ref: algorithms/uniform_quantize/hadamard_rotation.py:93-134 (reshape(-1, h) @ H)
//
// The reference multiplies by the dense matrix with sgemm (O(h) work per output);
// here every length-h vector is transformed in LDS with a fast Walsh-Hadamard
// butterfly (log2 h stages), which is the same linear map because the Sylvester
// (Kronecker) order is the natural-order WHT and H is symmetric. Each input is
// first multiplied by fl(1/fl(sqrt(h))) -- the value of every |H entry| in the
// reference -- so only the order of the FP32 additions differs from sgemm
// (tolerance class T2).
#include "common.h"

namespace mi355q {
namespace {

// One block transforms `vecs` vectors of length h (vecs * h floats in LDS).
__global__ __launch_bounds__(256) void fwht_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                  long long n_vec, int h, int log2h, int vecs,
                                                  float r) {
  extern __shared__ __attribute__((aligned(16))) float buf[];
  const int tile = vecs * h;
  const long long first = static_cast<long long>(blockIdx.x) * vecs;
  const long long remain = (n_vec - first) * h;
  const int valid = remain < tile ? static_cast<int>(remain) : tile;
  const float* src = x + first * h;
  float* dst = out + first * h;
  if ((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (valid & 3) == 0) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* b4 = reinterpret_cast<float4*>(buf);
    for (int i = threadIdx.x; i < valid / 4; i += 256) {
      float4 v = s4[i];
      b4[i] = make_float4(v.x * r, v.y * r, v.z * r, v.w * r);
    }
  } else {
    for (int i = threadIdx.x; i < valid; i += 256) buf[i] = src[i] * r;
  }
  __syncthreads();
  const int pairs = valid / 2;
  for (int s = 0; s < log2h; ++s) {
    const int half = 1 << s;
    for (int i = threadIdx.x; i < pairs; i += 256) {
      const int lo = ((i >> s) << (s + 1)) | (i & (half - 1));
      const float u = buf[lo], v = buf[lo + half];
      buf[lo] = u + v;
      buf[lo + half] = u - v;
    }
    __syncthreads();
  }
  if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (valid & 3) == 0) {
    const float4* b4 = reinterpret_cast<const float4*>(buf);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int i = threadIdx.x; i < valid / 4; i += 256) d4[i] = b4[i];
  } else {
    for (int i = threadIdx.x; i < valid; i += 256) dst[i] = buf[i];
  }
}

}  // namespace
}  // namespace mi355q

using namespace mi355q;

extern "C" int32_t mi355q_hadamard_rotate_f32(const float* x, int64_t n_vec, int32_t h, float* out,
                                              void* stream) {
  clear_error();
  if (n_vec < 0) return fail(MI355Q_BAD_ARG, "negative vector count");
  if (h <= 0 || (h & (h - 1)) != 0)
    return fail(MI355Q_BAD_ARG, "Hadamard matrix size must be a power of 2. ");
  if (h > 16384) return fail(MI355Q_UNSUPPORTED, "hadamard size > 16384 does not fit one LDS tile");
  if (n_vec == 0) return MI355Q_OK;
  if (!x || !out) return fail(MI355Q_BAD_ARG, "null pointer");
  int log2h = 0;
  while ((1 << log2h) < h) ++log2h;
  // |H entry| of the reference: int8(1) / np.sqrt(h, dtype=float32)
  const float r = 1.0f / __builtin_sqrtf(static_cast<float>(h));
  int vecs = h >= 2048 ? 1 : 2048 / h;
  if (vecs > n_vec) vecs = static_cast<int>(n_vec);
  const size_t smem = static_cast<size_t>(vecs) * h * sizeof(float);
  const long long blocks = (n_vec + vecs - 1) / vecs;
  if (blocks > 0x7FFFFFFFLL) return fail(MI355Q_UNSUPPORTED, "too many vectors");
  hipLaunchKernelGGL(fwht_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), smem,
                     as_stream(stream), x, out, static_cast<long long>(n_vec), h, log2h, vecs, r);
  MI355Q_CHECK_LAUNCH("hadamard launch");
  return MI355Q_OK;
}
