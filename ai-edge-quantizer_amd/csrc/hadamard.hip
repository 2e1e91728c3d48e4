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

// ---- h >= 256: radix-16 passes over a tile of 4096 / 8192 / 16384 elements --------------------
// The radix-2 kernel above sends every element through LDS twice per stage (12 stages at
// h = 4096: 24 LDS accesses per element for 2 HBM accesses) and is LDS-bound at ~2.6 TB/s. Here a
// thread keeps 16 elements in registers and does up to four butterfly stages at once over a tile
// of T = 2^LOG2T elements (T / 16 threads):
//   pass 1  index bits {0, 1, LOG2T-2, LOG2T-1}: straight from the coalesced float4 global loads
//   pass 2  index bits {2..5}      (LDS read + write)
//   pass 3  index bits {6..9}      (LDS read; for T = 4096 the results go to HBM as 256-byte rows)
//   pass 4  index bits {10..LOG2T-3} (T = 8192: bit 10, T = 16384: bits 10 and 11; then HBM)
// i.e. 4 (T = 4096) or 6 LDS accesses per element. Stages above log2(h) are skipped (those bits
// select the vector inside the tile). The LDS index is XOR-swizzled (bits 2-4 ^= bits 6-8) so
// that pass 2, whose lanes differ in bits {0,1,6..}, still spreads over all banks.
constexpr int kTile = 4096;

__device__ __forceinline__ int swz(int i) { return i ^ (((i >> 6) & 7) << 2); }

template <int NBITS>   // butterflies over the low NBITS bits of the register index, radix 2
__device__ __forceinline__ void butterflies(float (&v)[16], int first_bit, int log2h) {
#pragma unroll
  for (int b = 0; b < NBITS; ++b) {
    if (first_bit + b < log2h) {  // uniform
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if ((k >> b) & 1) continue;
        const float u = v[k], w = v[k | (1 << b)];
        v[k] = u + w;
        v[k | (1 << b)] = u - w;
      }
    }
  }
}

template <int LOG2T>
__global__ __launch_bounds__((1 << LOG2T) / 16) void fwht_tile_kernel(const float* __restrict__ x,
                                                                      float* __restrict__ out,
                                                                      long long total, int log2h, float r) {
  constexpr int T = 1 << LOG2T, NT = T / 16;
  __shared__ __attribute__((aligned(16))) float buf[T];
  const int t = threadIdx.x;
  const long long base = static_cast<long long>(blockIdx.x) * T;
  const long long left = total - base;
  const int valid = left < T ? static_cast<int>(left) : T;   // a multiple of h (and of 4)
  const float4* src = reinterpret_cast<const float4*>(x + base);
  float v[16];
  // pass 1: register index k = c | (j << 2)  <->  tile index bits (0,1) and (LOG2T-2, LOG2T-1)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int f4 = t + NT * j;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f4 * 4 < valid) q = src[f4];
    v[j * 4 + 0] = q.x * r; v[j * 4 + 1] = q.y * r; v[j * 4 + 2] = q.z * r; v[j * 4 + 3] = q.w * r;
  }
  butterflies<2>(v, 0, log2h);            // bits 0, 1 (always below log2h: h >= 256)
  {                                       // the two top bits = register bits 2, 3
    float hi[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) hi[k] = v[((k & 3) << 2) | (k >> 2)];   // transpose c <-> j
    butterflies<2>(hi, LOG2T - 2, log2h);
#pragma unroll
    for (int k = 0; k < 16; ++k) v[((k & 3) << 2) | (k >> 2)] = hi[k];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = 4 * (t + NT * j);
    *reinterpret_cast<float4*>(&buf[swz(i)]) = make_float4(v[j * 4], v[j * 4 + 1], v[j * 4 + 2], v[j * 4 + 3]);
  }
  __syncthreads();
  // pass 2: bits 2..5 in registers; thread bits -> index bits {0,1} and {6..LOG2T-1}
  {
    const int fixed = (t & 3) | ((t >> 2) << 6);
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = buf[swz(fixed | (k << 2))];
    butterflies<4>(v, 2, log2h);
#pragma unroll
    for (int k = 0; k < 16; ++k) buf[swz(fixed | (k << 2))] = v[k];
  }
  __syncthreads();
  // pass 3: bits 6..9 in registers; thread bits -> index bits {0..5} and {10..LOG2T-1}
  {
    const int fixed = (t & 63) | ((t >> 6) << 10);
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = buf[swz(fixed | (k << 6))];
    butterflies<4>(v, 6, log2h);
    if constexpr (LOG2T == 12) {
      float* dst = out + base;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int i = fixed | (k << 6);
        if (i < valid) dst[i] = v[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) buf[swz(fixed | (k << 6))] = v[k];
    }
  }
  if constexpr (LOG2T > 12) {
    __syncthreads();
    // pass 4: the remaining bits 10 .. LOG2T-3. Registers <-> index bits {LOG2T-4 .. LOG2T-1} (for
    // T = 8192 that is {9, 10, 11, 12}: bit 9 only picks a second group), thread bits <-> the rest.
    constexpr int kLow = LOG2T - 4;               // the threads cover index bits 0 .. kLow-1
    const int fixed = t;                          // NT = 2^kLow threads
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = buf[swz(fixed | (k << kLow))];
    {
      // register bit (10 - kLow) is index bit 10
      constexpr int shift = 10 - kLow;            // 1 for T = 8192, 0 for T = 16384
      float w[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) w[k] = v[((k << shift) | (k >> (4 - shift))) & 15];   // rotate index bit 10 down to bit 0
      butterflies<LOG2T - 12>(w, 10, log2h);
#pragma unroll
      for (int k = 0; k < 16; ++k) v[((k << shift) | (k >> (4 - shift))) & 15] = w[k];
    }
    float* dst = out + base;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = fixed | (k << kLow);
      if (i < valid) dst[i] = v[k];
    }
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
  const long long total = n_vec * static_cast<long long>(h);
  if (h >= 256 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    const int tile = h <= kTile ? kTile : h;   // 4096, 8192 or 16384 elements per workgroup
    const long long tiles = (total + tile - 1) / tile;
    if (tiles > 0x7FFFFFFFLL) return fail(MI355Q_UNSUPPORTED, "too many vectors");
    const dim3 grid(static_cast<unsigned>(tiles));
    if (tile == 4096)
      hipLaunchKernelGGL(fwht_tile_kernel<12>, grid, dim3(256), 0, as_stream(stream), x, out, total, log2h, r);
    else if (tile == 8192)
      hipLaunchKernelGGL(fwht_tile_kernel<13>, grid, dim3(512), 0, as_stream(stream), x, out, total, log2h, r);
    else
      hipLaunchKernelGGL(fwht_tile_kernel<14>, grid, dim3(1024), 0, as_stream(stream), x, out, total, log2h, r);
    MI355Q_CHECK_LAUNCH("hadamard launch");
    return MI355Q_OK;
  }
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
