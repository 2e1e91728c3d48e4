// Unfused building blocks: weight min/max (K1), uniform quantize with given
// parameters (K3, "T1" path), dequantize, bit packing (K4) and activation
// statistics (K7). All are streaming, HBM-bound kernels for gfx950.
#include <climits>

#include <cstdlib>

#include "common.h"

namespace mi355q {
namespace {

struct MinMax {
  float mn, mx;
  bool nan;
};

__device__ __forceinline__ MinMax mm_identity() {
  return {__builtin_huge_valf(), -__builtin_huge_valf(), false};
}
__device__ __forceinline__ void mm_add(MinMax& a, float v) {
  a.mn = v < a.mn ? v : a.mn;
  a.mx = v > a.mx ? v : a.mx;
  a.nan |= (v != v);
}
__device__ __forceinline__ void mm_merge(MinMax& a, float mn, float mx, bool nan) {
  a.mn = mn < a.mn ? mn : a.mn;
  a.mx = mx > a.mx ? mx : a.mx;
  a.nan |= nan;
}
__device__ __forceinline__ MinMax wave_reduce(MinMax a) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    const float mn = __shfl_xor(a.mn, off, kWave);
    const float mx = __shfl_xor(a.mx, off, kWave);
    const int nn = __shfl_xor(static_cast<int>(a.nan), off, kWave);
    mm_merge(a, mn, mx, nn != 0);
  }
  return a;
}
__device__ __forceinline__ void mm_store(const MinMax& a, float* mn, float* mx, int64_t i) {
  const float qnan = __builtin_nanf("");
  mn[i] = a.nan ? qnan : a.mn;
  mx[i] = a.nan ? qnan : a.mx;
}

// ------------------------------------------------------------------ K1 ---
// One wave per (channel, split) unit. Elements of a channel: `outer` runs of
// `inner` contiguous floats at (o*channels + c)*inner.
__global__ __launch_bounds__(256) void minmax_runs_kernel(
    const float* __restrict__ x, int64_t outer, int64_t channels, int64_t inner,
    int64_t splits, int64_t chunk, float* mn_out, float* mx_out) {
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * (256 / kWave) + threadIdx.x / kWave;
  if (unit >= channels * splits) return;
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t c = unit / splits, s = unit % splits;
  const int64_t len = outer * inner;
  const int64_t beg = s * chunk;
  const int64_t end = beg + chunk < len ? beg + chunk : len;
  MinMax a = mm_identity();
  if (outer == 1 && ((reinterpret_cast<uintptr_t>(x + c * inner + beg) & 15) == 0)) {
    // 16-byte loads, four in flight per lane (two left the per-channel reduction of a 4096 x 4096 weight at 0.44 of the
    // HBM peak: 8192 waves with 2 KiB in flight each), then two, then one
    const float* p = x + c * inner;
    const float4* p4 = reinterpret_cast<const float4*>(p + beg);
    const int64_t n4 = (end - beg) / 4;
    int64_t i = lane;
    for (; i + 3 * kWave < n4; i += 4 * kWave) {
      const float4 v = p4[i], w = p4[i + kWave], y = p4[i + 2 * kWave], z = p4[i + 3 * kWave];
      mm_add(a, v.x); mm_add(a, v.y); mm_add(a, v.z); mm_add(a, v.w);
      mm_add(a, w.x); mm_add(a, w.y); mm_add(a, w.z); mm_add(a, w.w);
      mm_add(a, y.x); mm_add(a, y.y); mm_add(a, y.z); mm_add(a, y.w);
      mm_add(a, z.x); mm_add(a, z.y); mm_add(a, z.z); mm_add(a, z.w);
    }
    for (; i + kWave < n4; i += 2 * kWave) {
      const float4 v = p4[i], w = p4[i + kWave];
      mm_add(a, v.x); mm_add(a, v.y); mm_add(a, v.z); mm_add(a, v.w);
      mm_add(a, w.x); mm_add(a, w.y); mm_add(a, w.z); mm_add(a, w.w);
    }
    for (; i < n4; i += kWave) {
      const float4 v = p4[i];
      mm_add(a, v.x); mm_add(a, v.y); mm_add(a, v.z); mm_add(a, v.w);
    }
    for (int64_t e = beg + n4 * 4 + lane; e < end; e += kWave) mm_add(a, p[e]);
  } else if (outer == 1) {
    const float* p = x + c * inner;
    int64_t e = beg + lane;
    for (; e + 3 * kWave < end; e += 4 * kWave) {
      const float v0 = p[e], v1 = p[e + kWave], v2 = p[e + 2 * kWave], v3 = p[e + 3 * kWave];
      mm_add(a, v0); mm_add(a, v1); mm_add(a, v2); mm_add(a, v3);
    }
    for (; e < end; e += kWave) mm_add(a, p[e]);
  } else {
    for (int64_t e = beg + lane; e < end; e += kWave) {
      const int64_t o = e / inner, i = e - o * inner;
      mm_add(a, x[(o * channels + c) * inner + i]);
    }
  }
  a = wave_reduce(a);
  if (lane == 0) mm_store(a, mn_out, mx_out, s * channels + c);
}

// Channel-last layout (inner == 1): x[outer][channels]; lanes run along channels.
__global__ __launch_bounds__(256) void minmax_lastdim_kernel(
    const float* __restrict__ x, int64_t outer, int64_t channels, int64_t splits,
    float* mn_out, float* mx_out) {
  __shared__ float smn[4][kWave], smx[4][kWave];
  __shared__ int snan[4][kWave];
  const int lane = threadIdx.x & (kWave - 1), slice = threadIdx.x / kWave;
  const int64_t c = static_cast<int64_t>(blockIdx.x) * kWave + lane;
  const int64_t s = blockIdx.y;
  const int64_t rows_per = (outer + splits - 1) / splits;
  const int64_t r0 = s * rows_per, r1 = r0 + rows_per < outer ? r0 + rows_per : outer;
  MinMax a = mm_identity();
  if (c < channels)
    for (int64_t r = r0 + slice; r < r1; r += 4) mm_add(a, x[r * channels + c]);
  smn[slice][lane] = a.mn; smx[slice][lane] = a.mx; snan[slice][lane] = a.nan;
  __syncthreads();
  if (slice == 0 && c < channels) {
#pragma unroll
    for (int k = 1; k < 4; ++k) mm_merge(a, smn[k][lane], smx[k][lane], snan[k][lane] != 0);
    mm_store(a, mn_out, mx_out, s * channels + c);
  }
}

// partial[s][c] -> out[c]
__global__ __launch_bounds__(256) void minmax_finalize_kernel(
    const float* __restrict__ pmn, const float* __restrict__ pmx, int64_t channels,
    int64_t splits, float* mn_out, float* mx_out) {
  const int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (c >= channels) return;
  MinMax a = mm_identity();
  for (int64_t s = 0; s < splits; ++s) {
    const float mn = pmn[s * channels + c], mx = pmx[s * channels + c];
    mm_merge(a, mn, mx, mn != mn);
  }
  mm_store(a, mn_out, mx_out, c);
}

// The same for FEW channels with MANY partials (TENSORWISE: one channel, thousands of splits -- a lane per channel walked
// them one after the other, 0.46 ms for a 4096 x 4096 tensor whose reduction itself takes 15 us): a workgroup per
// channel, threads stride over the splits four pairs at a time, butterfly + one LDS exchange at the end. min / max are
// order-free, so the result is the same.
__global__ __launch_bounds__(256) void minmax_finalize_wave_kernel(
    const float* __restrict__ pmn, const float* __restrict__ pmx, int64_t channels,
    int64_t splits, float* mn_out, float* mx_out) {
  __shared__ float smn[4], smx[4];
  __shared__ int snan[4];
  const int64_t c = blockIdx.x;
  if (c >= channels) return;
  MinMax a = mm_identity();
  int64_t s = threadIdx.x;
  for (; s + 3 * 256 < splits; s += 4 * 256) {       // four pairs of partials in flight per thread
    float mn[4], mx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mn[k] = pmn[(s + 256 * k) * channels + c];
      mx[k] = pmx[(s + 256 * k) * channels + c];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) mm_merge(a, mn[k], mx[k], mn[k] != mn[k]);
  }
  for (; s < splits; s += 256) {
    const float mn = pmn[s * channels + c], mx = pmx[s * channels + c];
    mm_merge(a, mn, mx, mn != mn);
  }
  a = wave_reduce(a);
  const int wave = threadIdx.x / kWave;
  if ((threadIdx.x & (kWave - 1)) == 0) {
    smn[wave] = a.mn; smx[wave] = a.mx; snan[wave] = a.nan;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 1; k < 4; ++k) mm_merge(a, smn[k], smx[k], snan[k] != 0);
    mm_store(a, mn_out, mx_out, c);
  }
}

struct MinMaxPlan {
  bool lastdim;
  int64_t splits, chunk;
};

MinMaxPlan plan_minmax(int64_t outer, int64_t channels, int64_t inner) {
  MinMaxPlan p{};
  if (inner == 1 && outer > 1) {
    p.lastdim = true;
    p.splits = outer / 256;  // >= 256 rows per split
    if (p.splits < 1) p.splits = 1;
    if (p.splits > 128) p.splits = 128;
    p.chunk = 0;
    return p;
  }
  const int64_t len = outer * inner;
  // Aim for ~8192 wave-units on the chip, >= 4096 elements per unit.
  int64_t want = 8192 / (channels > 0 ? channels : 1);
  if (want < 1) want = 1;
  int64_t maxs = (len + 4095) / 4096;
  if (maxs < 1) maxs = 1;
  p.splits = want < maxs ? want : maxs;
  p.chunk = (len + p.splits - 1) / p.splits;
  // keep chunks a multiple of 64 so waves stay aligned
  p.chunk = (p.chunk + kWave - 1) / kWave * kWave;
  p.splits = p.chunk > 0 ? (len + p.chunk - 1) / p.chunk : 1;
  if (p.splits < 1) p.splits = 1;
  return p;
}

// ------------------------------------------------------------------ K3 ---
template <typename OutT>
__device__ __forceinline__ OutT sat_cast(float r, bool isnan_) {
  if (isnan_) return 0;
  if constexpr (sizeof(OutT) == 4) {
    // x86 cvttss2si "integer indefinite" for out-of-range values
    if (!(r < 2147483648.0f) || r < -2147483648.0f) return INT_MIN;
  }
  return static_cast<OutT>(r);
}
template <typename OutT>
__device__ __forceinline__ OutT sat_cast(double r, bool isnan_) {
  if (isnan_) return 0;
  if constexpr (sizeof(OutT) == 4) {
    if (!(r < 2147483648.0) || r < -2147483648.0) return INT_MIN;
  }
  return static_cast<OutT>(r);
}

template <typename ScaleT, typename OutT>
__global__ __launch_bounds__(256) void quantize_kernel(
    const float* __restrict__ x, int64_t n, int64_t channels, int64_t inner,
    const ScaleT* __restrict__ scale, const int32_t* __restrict__ zp, int zp_via_f64,
    double lo64, double hi64, OutT* __restrict__ q) {
  const float lo = static_cast<float>(lo64), hi = static_cast<float>(hi64);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < n; e += stride) {
    const int64_t c = channels == 1 ? 0 : (e / inner) % channels;
    const int z = zp ? zp[c] : 0;
    if constexpr (sizeof(ScaleT) == 8) {
      // float64 scale: NumPy promotes the whole chain to float64 (bounds exact up to 32 bits)
      double v = static_cast<double>(x[e]) / scale[c] + static_cast<double>(z);
      double r = __builtin_rint(v);
      r = fmin(fmax(r, lo64), hi64);
      q[e] = sat_cast<OutT>(r, v != v);
    } else {
      float v = x[e] / scale[c];
      if (zp_via_f64)
        v = static_cast<float>(static_cast<double>(v) + static_cast<double>(z));
      else
        v = v + static_cast<float>(z);
      float r = __builtin_rintf(v);
      r = fminf(fmaxf(r, lo), hi);
      q[e] = sat_cast<OutT>(r, v != v);
    }
  }
}

// Fast path of the above: float32 scale, int8 containers, inner % 4 == 0 and 16-byte
// aligned buffers -- each lane owns whole float4 pieces (one channel lookup per piece),
// four pieces in flight, dword stores.
__global__ __launch_bounds__(256) void quantize_vec4_kernel(
    const float4* __restrict__ x, int64_t n4, int64_t channels, int64_t inner4,
    const float* __restrict__ scale, const int32_t* __restrict__ zp, int zp_via_f64,
    float lo, float hi, uint32_t* __restrict__ q) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; base < n4; base += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = base + u * stride;
      v[u] = i < n4 ? x[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = base + u * stride;
      if (i >= n4) continue;
      const int64_t c = channels == 1 ? 0 : (i / inner4) % channels;
      const float s = scale[c];
      const int z = zp ? zp[c] : 0;
      const float in[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
      uint32_t w = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float t = in[k] / s;
        t = zp_via_f64 ? static_cast<float>(static_cast<double>(t) + static_cast<double>(z))
                       : t + static_cast<float>(z);
        const int qi = round_clip(t, lo, hi);
        w |= static_cast<uint32_t>(qi & 0xFF) << (8 * k);
      }
      q[i] = w;
    }
  }
}

// The same arithmetic with the channel looked up ONCE per workgroup: the grid runs over (run, piece) where a run is one
// (outer index, channel) stretch of `inner` contiguous elements and a piece is up to kRowPiece float4s of it, so scale
// and zero point are wave-uniform and no 64-bit division sits between the loads (the grid-stride kernel above spends
// more on `(i / inner4) % channels` per float4 than on the IEEE quotients: 0.47 of HBM peak at 4096 x 4096). Rows of
// weights (inner >= 256) take this one.
constexpr int kRowPiece = 1024;   // float4s per workgroup: 4 per thread in flight
typedef float vec4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void quantize_rows_vec4_kernel(
    const float4* __restrict__ x, int64_t channels, int64_t inner4, int32_t pieces,
    const float* __restrict__ scale, const int32_t* __restrict__ zp, int zp_via_f64,
    float lo, float hi, uint32_t* __restrict__ q) {
  const int64_t run = blockIdx.x / pieces;
  const int32_t piece = static_cast<int32_t>(blockIdx.x - run * pieces);
  const int64_t c = channels == 1 ? 0 : run % channels;
  const float s = scale[c];
  const int z = zp ? zp[c] : 0;
  const float zf = static_cast<float>(z);
  const double zd = static_cast<double>(z);
  const int64_t first = static_cast<int64_t>(piece) * kRowPiece;
  const int64_t here = inner4 - first < kRowPiece ? inner4 - first : kRowPiece;
  const float4* xr = x + run * inner4 + first;
  uint32_t* qr = q + run * inner4 + first;
  float4 v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = threadIdx.x + u * 256;
    if (i < here) {
      const vec4f t = __builtin_nontemporal_load(reinterpret_cast<const vec4f*>(xr + i));
      v[u] = make_float4(t[0], t[1], t[2], t[3]);
    } else {
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = threadIdx.x + u * 256;
    if (i >= here) continue;
    const float in[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float t = in[k] / s;
      t = zp_via_f64 ? static_cast<float>(static_cast<double>(t) + zd) : t + zf;
      const int qi = round_clip(t, lo, hi);
      w |= static_cast<uint32_t>(qi & 0xFF) << (8 * k);
    }
    __builtin_nontemporal_store(w, qr + i);
  }
}

// int8 -> float32, four elements per lane (dword in, float4 out), channel per workgroup as above.
__global__ __launch_bounds__(256) void dequantize_rows_vec4_kernel(
    const uint32_t* __restrict__ q, int64_t channels, int64_t inner4, int32_t pieces,
    const float* __restrict__ scale, const int32_t* __restrict__ zp, int diff_bits,
    float4* __restrict__ out) {
  const int64_t run = blockIdx.x / pieces;
  const int32_t piece = static_cast<int32_t>(blockIdx.x - run * pieces);
  const int64_t c = channels == 1 ? 0 : run % channels;
  const float s = scale[c];
  const int z = zp ? zp[c] : 0;
  const int64_t first = static_cast<int64_t>(piece) * kRowPiece;
  const int64_t here = inner4 - first < kRowPiece ? inner4 - first : kRowPiece;
  const uint32_t* qr = q + run * inner4 + first;
  float4* outr = out + run * inner4 + first;
  uint32_t v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = threadIdx.x + u * 256;
    v[u] = i < here ? __builtin_nontemporal_load(qr + i) : 0u;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = threadIdx.x + u * 256;
    if (i >= here) continue;
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int d = static_cast<int>(static_cast<int8_t>((v[u] >> (8 * k)) & 0xFF)) - z;
      if (diff_bits == 8) d = static_cast<int8_t>(d);
      else if (diff_bits == 16) d = static_cast<int16_t>(d);
      r[k] = static_cast<float>(d) * s;
    }
    const vec4f t = {r[0], r[1], r[2], r[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<vec4f*>(outr + i));
  }
}

template <typename InT, typename OutT>
__global__ __launch_bounds__(256) void dequantize_kernel(
    const InT* __restrict__ q, int64_t n, int64_t channels, int64_t inner,
    const float* __restrict__ scale, const int32_t* __restrict__ zp, int diff_bits,
    OutT* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < n; e += stride) {
    const int64_t c = channels == 1 ? 0 : (e / inner) % channels;
    const int z = zp ? zp[c] : 0;
    // NumPy subtracts in the promoted integer type of (q, zero_point) and wraps:
    // int8 - int8 stays int8 (ref: uniform_quantize_tensor.py:407-409).
    int d = static_cast<int>(q[e]) - z;
    if (diff_bits == 8) d = static_cast<int8_t>(d);
    else if (diff_bits == 16) d = static_cast<int16_t>(d);
    out[e] = static_cast<OutT>(d) * static_cast<OutT>(scale[c]);
  }
}

// Inverse of K4: one packed byte per loop step -> PER sign-extended int8 values (the int8
// containers UniformQuantParams.quantized_data holds). Every thread expands 4 packed bytes.
template <int BITS>
__global__ __launch_bounds__(256) void unpack_kernel(const uint8_t* __restrict__ packed, int64_t n,
                                                    int8_t* __restrict__ q) {
  constexpr int PER = 8 / BITS;
  const int64_t n_in = (n + PER - 1) / PER;
  const int64_t words = (n_in + 3) / 4;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  const bool al = ((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(q)) & 3) == 0;
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; w < words; w += stride) {
    uint32_t v = 0;
    if (al && w * 4 + 4 <= n_in) v = reinterpret_cast<const uint32_t*>(packed)[w];
    else for (int b = 0; b < 4 && w * 4 + b < n_in; ++b) v |= static_cast<uint32_t>(packed[w * 4 + b]) << (8 * b);
    const int64_t out0 = w * 4 * PER;
    uint32_t o[PER];   // 4 * PER output bytes as PER dwords
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      uint32_t acc = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int idx = k * 4 + b;   // output element index within the group
        const int field = static_cast<int>((v >> ((idx / PER) * 8 + (idx % PER) * BITS)) & ((1 << BITS) - 1));
        const int val = (field ^ (1 << (BITS - 1))) - (1 << (BITS - 1));   // sign extend
        acc |= static_cast<uint32_t>(val & 0xFF) << (8 * b);
      }
      o[k] = acc;
    }
    if (al && out0 + 4 * PER <= n) {
#pragma unroll
      for (int k = 0; k < PER; ++k) reinterpret_cast<uint32_t*>(q + out0)[k] = o[k];
    } else {
      for (int idx = 0; idx < 4 * PER && out0 + idx < n; ++idx)
        q[out0 + idx] = static_cast<int8_t>((o[idx / 4] >> (8 * (idx % 4))) & 0xFF);
    }
  }
}

// ------------------------------------------------------------------ K4 ---
// Each thread builds 4 output bytes from 8 (int4) or 16 (int2) input bytes.
template <int BITS>
__global__ __launch_bounds__(256) void pack_kernel(const int8_t* __restrict__ q, int64_t n,
                                                  uint8_t* __restrict__ out, int64_t n_out) {
  constexpr int PER = 8 / BITS;
  constexpr int MASK = (1 << BITS) - 1;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  const int64_t words = (n_out + 3) / 4;
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; w < words; w += stride) {
    const int64_t in0 = w * 4 * PER;
    uint32_t acc = 0;
    if (in0 + 4 * PER <= n && (reinterpret_cast<uintptr_t>(q) & 3) == 0) {
#pragma unroll
      for (int k = 0; k < PER; ++k) {  // one dword = 4 input values
        const uint32_t v = reinterpret_cast<const uint32_t*>(q + in0)[k];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int idx = k * 4 + b;  // input element index within the word group
          acc |= ((v >> (8 * b)) & MASK) << ((idx / PER) * 8 + (idx % PER) * BITS);
        }
      }
    } else {
      for (int idx = 0; idx < 4 * PER; ++idx) {
        const int64_t i = in0 + idx;
        const uint32_t v = i < n ? static_cast<uint8_t>(q[i]) : 0u;
        acc |= (v & MASK) << ((idx / PER) * 8 + (idx % PER) * BITS);
      }
    }
    if (w * 4 + 4 <= n_out && (reinterpret_cast<uintptr_t>(out) & 3) == 0) {
      reinterpret_cast<uint32_t*>(out)[w] = acc;
    } else {
      for (int b = 0; b < 4 && w * 4 + b < n_out; ++b) out[w * 4 + b] = (acc >> (8 * b)) & 0xFF;
    }
  }
}

__global__ __launch_bounds__(256) void copy_bytes_kernel(const int8_t* __restrict__ q, int64_t n,
                                                        uint8_t* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride)
    out[i] = static_cast<uint8_t>(q[i]);
}

// ------------------------------------------------------------------ K7 ---
struct ActAcc {
  float mn_m, mx_m;  // masked (x > lo / x < hi)
  float mn_a, mx_a;  // plain
  bool nan;
};
__device__ __forceinline__ void act_add(ActAcc& a, float v, float lo, float hi, bool ranged) {
  const bool in_lo = !ranged || v > lo;  // NaN fails both masks, as in NumPy
  const bool in_hi = !ranged || v < hi;
  a.mn_m = (in_lo && v < a.mn_m) ? v : a.mn_m;
  a.mx_m = (in_hi && v > a.mx_m) ? v : a.mx_m;
  a.mn_a = v < a.mn_a ? v : a.mn_a;
  a.mx_a = v > a.mx_a ? v : a.mx_a;
  a.nan |= (v != v);
}

constexpr int kActBlocks = 64;  // blocks per tensor (<= 64: one wave finalizes)

// grid (kActBlocks, count). partial layout: [count][kActBlocks][5]
template <int UNROLL>
__global__ __launch_bounds__(256) void act_minmax_kernel(
    const float* const* __restrict__ xs, const int64_t* __restrict__ numel, float lo, float hi,
    int ranged, float* __restrict__ partial) {
  const int t = blockIdx.y;
  const float* __restrict__ x = xs[t];
  const int64_t n = numel[t];
  const float inf = __builtin_huge_valf();
  ActAcc a{inf, -inf, inf, -inf, false};
  const bool r = ranged != 0;
  const int64_t tid = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t nthreads = static_cast<int64_t>(gridDim.x) * 256;
  const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  // Fast pass: plain min / max (v_min / v_max skip NaN) and the largest |x| bit pattern (any NaN
  // sorts above +inf). If every value of this lane's share lies strictly inside (lo, hi) and
  // none is NaN, the masked extrema ARE the plain ones; otherwise the share is scanned again
  // with the full per-element masks (sentinels are rare: the second scan hits L2).
  bool fast_ok = false;
  if (aligned) {
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const int64_t n4 = n / 4;
    float mn = inf, mx = -inf;
    uint32_t top = 0;
    auto eat = [&](const float4& v) {
      mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w)));
      mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
      const uint32_t t01 = abs_bits(v.x) > abs_bits(v.y) ? abs_bits(v.x) : abs_bits(v.y);
      const uint32_t t23 = abs_bits(v.z) > abs_bits(v.w) ? abs_bits(v.z) : abs_bits(v.w);
      const uint32_t t4 = t01 > t23 ? t01 : t23;
      top = t4 > top ? t4 : top;
    };
    int64_t i = tid;
    for (; i + (UNROLL - 1) * nthreads < n4; i += UNROLL * nthreads) {  // UNROLL 16-byte loads in flight per lane
      float4 v[UNROLL];
#pragma unroll
      // read once: non-temporal loads keep the stream out of L2 / MALL (+5.6 %, tools/path_bench.py)
      for (int u = 0; u < UNROLL; ++u) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f q = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(&x4[i + u * nthreads]));
        v[u] = make_float4(q.x, q.y, q.z, q.w);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) eat(v[u]);
    }
    for (; i < n4; i += nthreads) eat(x4[i]);
    for (int64_t e = n4 * 4 + tid; e < n; e += nthreads) {
      const float v = x[e];
      mn = fminf(mn, v);
      mx = fmaxf(mx, v);
      top = abs_bits(v) > top ? abs_bits(v) : top;
    }
    const bool no_nan = top <= 0x7F800000u;
    fast_ok = no_nan && (!r || (mn > lo && mx < hi));
    if (fast_ok) {
      a.mn_m = a.mn_a = mn;
      a.mx_m = a.mx_a = mx;
    }
  }
  if (!fast_ok) {
    if (aligned) {
      const float4* x4 = reinterpret_cast<const float4*>(x);
      const int64_t n4 = n / 4;
      for (int64_t i = tid; i < n4; i += nthreads) {
        const float4 v = x4[i];
        act_add(a, v.x, lo, hi, r); act_add(a, v.y, lo, hi, r);
        act_add(a, v.z, lo, hi, r); act_add(a, v.w, lo, hi, r);
      }
      for (int64_t e = n4 * 4 + tid; e < n; e += nthreads) act_add(a, x[e], lo, hi, r);
    } else {
      for (int64_t e = tid; e < n; e += nthreads) act_add(a, x[e], lo, hi, r);
    }
  }
  // block reduce
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    const float b0 = __shfl_xor(a.mn_m, off, kWave), b1 = __shfl_xor(a.mx_m, off, kWave);
    const float b2 = __shfl_xor(a.mn_a, off, kWave), b3 = __shfl_xor(a.mx_a, off, kWave);
    const int nn = __shfl_xor(static_cast<int>(a.nan), off, kWave);
    a.mn_m = b0 < a.mn_m ? b0 : a.mn_m; a.mx_m = b1 > a.mx_m ? b1 : a.mx_m;
    a.mn_a = b2 < a.mn_a ? b2 : a.mn_a; a.mx_a = b3 > a.mx_a ? b3 : a.mx_a;
    a.nan |= nn != 0;
  }
  __shared__ float sh[4][5];
  const int wave = threadIdx.x / kWave;
  if ((threadIdx.x & (kWave - 1)) == 0) {
    sh[wave][0] = a.mn_m; sh[wave][1] = a.mx_m; sh[wave][2] = a.mn_a; sh[wave][3] = a.mx_a;
    sh[wave][4] = a.nan ? 1.f : 0.f;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* p = partial + (static_cast<int64_t>(t) * kActBlocks + blockIdx.x) * 5;
    p[0] = fminf(fminf(sh[0][0], sh[1][0]), fminf(sh[2][0], sh[3][0]));
    p[1] = fmaxf(fmaxf(sh[0][1], sh[1][1]), fmaxf(sh[2][1], sh[3][1]));
    p[2] = fminf(fminf(sh[0][2], sh[1][2]), fminf(sh[2][2], sh[3][2]));
    p[3] = fmaxf(fmaxf(sh[0][3], sh[1][3]), fmaxf(sh[2][3], sh[3][3]));
    p[4] = sh[0][4] + sh[1][4] + sh[2][4] + sh[3][4];
  }
}

// One wave per tensor combines the kActBlocks partials and applies the fallback
// (ref: common_quantize.py:1393-1394, 1405-1406).
__global__ __launch_bounds__(64) void act_minmax_finalize_kernel(const float* __restrict__ partial,
                                                                int count, int blocks,
                                                                float* __restrict__ out) {
  const int t = blockIdx.x;
  if (t >= count) return;
  const int lane = threadIdx.x;
  const float inf = __builtin_huge_valf();
  float v0 = inf, v1 = -inf, v2 = inf, v3 = -inf, v4 = 0.f;
  if (lane < blocks) {
    const float* p = partial + (static_cast<int64_t>(t) * kActBlocks + lane) * 5;
    v0 = p[0]; v1 = p[1]; v2 = p[2]; v3 = p[3]; v4 = p[4];
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    v0 = fminf(v0, __shfl_xor(v0, off, kWave)); v1 = fmaxf(v1, __shfl_xor(v1, off, kWave));
    v2 = fminf(v2, __shfl_xor(v2, off, kWave)); v3 = fmaxf(v3, __shfl_xor(v3, off, kWave));
    v4 += __shfl_xor(v4, off, kWave);
  }
  if (lane == 0) {
    const float qnan = __builtin_nanf("");
    const float plain_min = v4 > 0.f ? qnan : v2;  // np.min propagates NaN
    const float plain_max = v4 > 0.f ? qnan : v3;
    out[2 * t + 0] = (v0 == inf) ? plain_min : v0;
    out[2 * t + 1] = (v1 == -inf) ? plain_max : v1;
  }
}

inline unsigned grid_for(int64_t n, int per_thread = 1) {
  int64_t b = (n + 256LL * per_thread - 1) / (256LL * per_thread);
  if (b < 1) b = 1;
  if (b > 256 * 16) b = 256 * 16;  // 16 blocks per CU, grid-stride beyond
  return static_cast<unsigned>(b);
}

}  // namespace
}  // namespace mi355q

using namespace mi355q;

extern "C" size_t mi355q_minmax_workspace_bytes(int64_t outer, int64_t channels, int64_t inner) {
  if (outer <= 0 || channels <= 0 || inner <= 0) return 0;
  const MinMaxPlan p = plan_minmax(outer, channels, inner);
  return p.splits > 1 ? static_cast<size_t>(p.splits) * channels * 2 * sizeof(float) : 0;
}

extern "C" int32_t mi355q_minmax_f32(const float* x, int64_t outer, int64_t channels,
                                     int64_t inner, float* min_out, float* max_out,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  clear_error();
  if (outer < 0 || channels < 0 || inner < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  if (channels == 0) return MI355Q_OK;
  if (outer == 0 || inner == 0)
    return fail(MI355Q_BAD_SHAPE, "zero-size array to reduction operation minimum which has no identity");
  if (!x || !min_out || !max_out) return fail(MI355Q_BAD_ARG, "null pointer");
  const MinMaxPlan p = plan_minmax(outer, channels, inner);
  const size_t need = mi355q_minmax_workspace_bytes(outer, channels, inner);
  if (need > 0 && (workspace == nullptr || workspace_bytes < need))
    return fail(MI355Q_BAD_ARG, "workspace too small: need %zu bytes", need);
  hipStream_t st = as_stream(stream);
  float* pmn = p.splits > 1 ? static_cast<float*>(workspace) : min_out;
  float* pmx = p.splits > 1 ? pmn + p.splits * channels : max_out;
  if (p.lastdim) {
    const dim3 grid(static_cast<unsigned>((channels + kWave - 1) / kWave),
                    static_cast<unsigned>(p.splits));
    hipLaunchKernelGGL(minmax_lastdim_kernel, grid, dim3(256), 0, st, x, outer, channels,
                       p.splits, pmn, pmx);
  } else {
    const int64_t units = channels * p.splits;
    const dim3 grid(static_cast<unsigned>((units + 3) / 4));
    hipLaunchKernelGGL(minmax_runs_kernel, grid, dim3(256), 0, st, x, outer, channels, inner,
                       p.splits, p.chunk, pmn, pmx);
  }
  MI355Q_CHECK_LAUNCH("minmax launch");
  if (p.splits >= 16 && channels <= 4096) {
    hipLaunchKernelGGL(minmax_finalize_wave_kernel, dim3(static_cast<unsigned>(channels)), dim3(256), 0, st, pmn,
                       pmx, channels, p.splits, min_out, max_out);
    MI355Q_CHECK_LAUNCH("minmax finalize launch");
  } else if (p.splits > 1) {
    hipLaunchKernelGGL(minmax_finalize_kernel, dim3(static_cast<unsigned>((channels + 255) / 256)),
                       dim3(256), 0, st, pmn, pmx, channels, p.splits, min_out, max_out);
    MI355Q_CHECK_LAUNCH("minmax finalize launch");
  }
  return MI355Q_OK;
}

namespace {
template <typename ScaleT>
int32_t launch_quantize(const float* x, int64_t n, int64_t channels, int64_t inner,
                        const void* scale, const int32_t* zp, int zp_via_f64, double lo, double hi,
                        int out_bits, void* q, hipStream_t st) {
  const dim3 grid(grid_for(n)), blk(256);
  const ScaleT* s = static_cast<const ScaleT*>(scale);
  if constexpr (sizeof(ScaleT) == 4) {
    const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(q)) & 15u) == 0;
    if (out_bits == 8 && inner % 4 == 0 && aligned && inner >= 1024 && (n / inner) * ((inner / 4 + kRowPiece - 1) / kRowPiece) < 0x7FFFFFFFLL) {
      const int64_t inner4 = inner / 4, pieces = (inner4 + kRowPiece - 1) / kRowPiece;
      hipLaunchKernelGGL(quantize_rows_vec4_kernel, dim3(static_cast<unsigned>((n / inner) * pieces)), blk, 0, st,
                         reinterpret_cast<const float4*>(x), channels, inner4, static_cast<int32_t>(pieces), s, zp,
                         zp_via_f64, static_cast<float>(lo), static_cast<float>(hi), static_cast<uint32_t*>(q));
      MI355Q_CHECK_LAUNCH("quantize launch");
      return MI355Q_OK;
    }
    if (out_bits == 8 && inner % 4 == 0 && aligned) {
      hipLaunchKernelGGL(quantize_vec4_kernel, dim3(grid_for(n / 4, 4)), blk, 0, st,
                         reinterpret_cast<const float4*>(x), n / 4, channels, inner / 4, s, zp, zp_via_f64,
                         static_cast<float>(lo), static_cast<float>(hi), static_cast<uint32_t*>(q));
      MI355Q_CHECK_LAUNCH("quantize launch");
      return MI355Q_OK;
    }
  }
  switch (out_bits) {
    case 8: hipLaunchKernelGGL((quantize_kernel<ScaleT, int8_t>), grid, blk, 0, st, x, n, channels, inner, s, zp, zp_via_f64, lo, hi, static_cast<int8_t*>(q)); break;
    case 16: hipLaunchKernelGGL((quantize_kernel<ScaleT, int16_t>), grid, blk, 0, st, x, n, channels, inner, s, zp, zp_via_f64, lo, hi, static_cast<int16_t*>(q)); break;
    case 32: hipLaunchKernelGGL((quantize_kernel<ScaleT, int32_t>), grid, blk, 0, st, x, n, channels, inner, s, zp, zp_via_f64, lo, hi, static_cast<int32_t*>(q)); break;
    default: return fail(MI355Q_UNSUPPORTED, "out_bits must be 8, 16 or 32");
  }
  MI355Q_CHECK_LAUNCH("quantize launch");
  return MI355Q_OK;
}
}  // namespace

extern "C" int32_t mi355q_quantize_f32(const float* x, int64_t outer, int64_t channels,
                                       int64_t inner, const void* scale, int32_t scale_is_f64,
                                       const int32_t* zero_point, int32_t zp_via_f64,
                                       int32_t bits, int32_t narrow, int32_t out_bits,
                                       void* q_out, void* stream) {
  clear_error();
  if (outer < 0 || channels < 0 || inner < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  if (bits < 2 || bits > 32 || bits > out_bits)
    return fail(MI355Q_BAD_ARG, "bits must be in [2, 32] and fit out_bits");
  const int64_t n = outer * channels * inner;
  if (n == 0) return MI355Q_OK;
  if (!x || !scale || !q_out) return fail(MI355Q_BAD_ARG, "null pointer");
  // Bounds as NumPy sees them: Python floats cast to the working float type.
  const double qmax = static_cast<double>((1LL << (bits - 1)) - 1);
  const double qmin = -static_cast<double>(1LL << (bits - 1));
  // float32 scale: the chain stays float32 and NumPy casts the bounds to float32 (2^31 - 1
  // rounds to 2^31); float64 scale: everything, bounds included, is float64
  const double lo = narrow ? qmin + 1.0 : qmin;
  const double hi = qmax;
  hipStream_t st = as_stream(stream);
  return scale_is_f64
             ? launch_quantize<double>(x, n, channels, inner, scale, zero_point, zp_via_f64, lo, hi, out_bits, q_out, st)
             : launch_quantize<float>(x, n, channels, inner, scale, zero_point, zp_via_f64, lo, hi, out_bits, q_out, st);
}

extern "C" int32_t mi355q_dequantize_f32(const void* q, int32_t in_bits, int64_t outer,
                                         int64_t channels, int64_t inner, const float* scale,
                                         const int32_t* zero_point, int32_t diff_bits,
                                         int32_t out_is_f64, void* out, void* stream) {
  clear_error();
  if (outer < 0 || channels < 0 || inner < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  const int64_t n = outer * channels * inner;
  if (n == 0) return MI355Q_OK;
  if (!q || !scale || !out) return fail(MI355Q_BAD_ARG, "null pointer");
  if (diff_bits != 8 && diff_bits != 16 && diff_bits != 32)
    return fail(MI355Q_BAD_ARG, "diff_bits must be 8, 16 or 32");
  hipStream_t st = as_stream(stream);
  const dim3 grid(grid_for(n)), blk(256);
  if (in_bits == 8 && !out_is_f64 && inner % 4 == 0 && inner >= 1024 &&
      ((reinterpret_cast<uintptr_t>(q) & 3u) | (reinterpret_cast<uintptr_t>(out) & 15u)) == 0 &&
      (n / inner) * ((inner / 4 + kRowPiece - 1) / kRowPiece) < 0x7FFFFFFFLL) {
    const int64_t inner4 = inner / 4, pieces = (inner4 + kRowPiece - 1) / kRowPiece;
    hipLaunchKernelGGL(dequantize_rows_vec4_kernel, dim3(static_cast<unsigned>((n / inner) * pieces)), blk, 0, st,
                       static_cast<const uint32_t*>(q), channels, inner4, static_cast<int32_t>(pieces), scale, zero_point,
                       diff_bits, static_cast<float4*>(out));
    MI355Q_CHECK_LAUNCH("dequantize launch");
    return MI355Q_OK;
  }
#define MI355Q_DQ(IN, OUT)                                                                 \
  hipLaunchKernelGGL((dequantize_kernel<IN, OUT>), grid, blk, 0, st, static_cast<const IN*>(q), n, \
                     channels, inner, scale, zero_point, diff_bits, static_cast<OUT*>(out))
  switch (in_bits) {
    case 8: if (out_is_f64) MI355Q_DQ(int8_t, double); else MI355Q_DQ(int8_t, float); break;
    case 16: if (out_is_f64) MI355Q_DQ(int16_t, double); else MI355Q_DQ(int16_t, float); break;
    case 32: if (out_is_f64) MI355Q_DQ(int32_t, double); else MI355Q_DQ(int32_t, float); break;
    default: return fail(MI355Q_UNSUPPORTED, "in_bits must be 8, 16 or 32");
  }
#undef MI355Q_DQ
  MI355Q_CHECK_LAUNCH("dequantize launch");
  return MI355Q_OK;
}

extern "C" int32_t mi355q_pack_bits(const int8_t* q, int64_t n, int32_t bits, uint8_t* out,
                                    void* stream) {
  clear_error();
  if (n < 0) return fail(MI355Q_BAD_ARG, "negative size");
  if (n == 0) return MI355Q_OK;
  if (!q || !out) return fail(MI355Q_BAD_ARG, "null pointer");
  hipStream_t st = as_stream(stream);
  if (bits == 8) {
    hipLaunchKernelGGL(copy_bytes_kernel, dim3(grid_for(n)), dim3(256), 0, st, q, n, out);
  } else if (bits == 4) {
    const int64_t n_out = (n + 1) / 2;
    hipLaunchKernelGGL((pack_kernel<4>), dim3(grid_for((n_out + 3) / 4)), dim3(256), 0, st, q, n, out, n_out);
  } else if (bits == 2) {
    const int64_t n_out = (n + 3) / 4;
    hipLaunchKernelGGL((pack_kernel<2>), dim3(grid_for((n_out + 3) / 4)), dim3(256), 0, st, q, n, out, n_out);
  } else {
    return fail(MI355Q_UNSUPPORTED, "pack_bits supports 2, 4 and 8 bits");
  }
  MI355Q_CHECK_LAUNCH("pack launch");
  return MI355Q_OK;
}

extern "C" int32_t mi355q_unpack_bits(const uint8_t* packed, int64_t n, int32_t bits, int8_t* q_out,
                                      void* stream) {
  clear_error();
  if (n < 0) return fail(MI355Q_BAD_ARG, "negative size");
  if (n == 0) return MI355Q_OK;
  if (!packed || !q_out) return fail(MI355Q_BAD_ARG, "null pointer");
  hipStream_t st = as_stream(stream);
  if (bits == 8) {
    hipLaunchKernelGGL(copy_bytes_kernel, dim3(grid_for(n)), dim3(256), 0, st,
                       reinterpret_cast<const int8_t*>(packed), n, reinterpret_cast<uint8_t*>(q_out));
  } else if (bits == 4) {
    hipLaunchKernelGGL((unpack_kernel<4>), dim3(grid_for((n + 7) / 8)), dim3(256), 0, st, packed, n, q_out);
  } else if (bits == 2) {
    hipLaunchKernelGGL((unpack_kernel<2>), dim3(grid_for((n + 15) / 16)), dim3(256), 0, st, packed, n, q_out);
  } else {
    return fail(MI355Q_UNSUPPORTED, "unpack_bits supports 2, 4 and 8 bits");
  }
  MI355Q_CHECK_LAUNCH("unpack launch");
  return MI355Q_OK;
}

// float32 -> float16, round to nearest even, overflow to inf, subnormal halves kept: what
// ndarray.astype(np.float16) does (ref: algorithms/nonlinear_quantize/float_casting.py:157-160).
__global__ __launch_bounds__(256) void cast_f16_kernel(const float* __restrict__ x, int64_t n,
                                                       _Float16* __restrict__ out) {
  const int64_t quad = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t e = quad * 4;
  if (e + 4 <= n && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(out) & 7) == 0) {
    const float4 v = *reinterpret_cast<const float4*>(x + e);
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 h;
    h.x = static_cast<_Float16>(v.x);
    h.y = static_cast<_Float16>(v.y);
    h.z = static_cast<_Float16>(v.z);
    h.w = static_cast<_Float16>(v.w);
    *reinterpret_cast<h4*>(out + e) = h;
  } else {
    for (int64_t i = e; i < n && i < e + 4; ++i) out[i] = static_cast<_Float16>(x[i]);
  }
}

extern "C" int32_t mi355q_cast_f32_to_f16(const float* x, int64_t n, uint16_t* out, void* stream) {
  clear_error();
  if (n < 0) return fail(MI355Q_BAD_ARG, "negative size");
  if (n == 0) return MI355Q_OK;
  if (!x || !out) return fail(MI355Q_BAD_ARG, "null pointer");
  hipLaunchKernelGGL(cast_f16_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, as_stream(stream), x,
                     n, reinterpret_cast<_Float16*>(out));
  MI355Q_CHECK_LAUNCH("cast_f16 launch");
  return MI355Q_OK;
}

extern "C" size_t mi355q_act_minmax_workspace_bytes(int32_t count) {
  return count > 0 ? static_cast<size_t>(count) * kActBlocks * 5 * sizeof(float) : 0;
}

extern "C" int32_t mi355q_act_minmax_f32(const float* const* x_ptrs, const int64_t* numel,
                                         int32_t count, float lo, float hi, int32_t use_range,
                                         float* minmax_out, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  clear_error();
  if (count < 0 || count > 65535) return fail(MI355Q_BAD_ARG, "count must be in [0, 65535]");
  if (count == 0) return MI355Q_OK;
  if (!x_ptrs || !numel || !minmax_out) return fail(MI355Q_BAD_ARG, "null pointer");
  const size_t need = mi355q_act_minmax_workspace_bytes(count);
  if (!workspace || workspace_bytes < need)
    return fail(MI355Q_BAD_ARG, "workspace too small: need %zu bytes", need);
  hipStream_t st = as_stream(stream);
  // 64 blocks per tensor, eight 16-byte loads in flight per lane (measured best of
  // {16, 32, 64} x {2, 4, 8, 16}: 5.9 TB/s at 128 x 4 MiB tensors)
  const int blocks = kActBlocks;
  hipLaunchKernelGGL(act_minmax_kernel<8>, dim3(blocks, static_cast<unsigned>(count)), dim3(256), 0,
                     st, x_ptrs, numel, lo, hi, use_range, static_cast<float*>(workspace));
  MI355Q_CHECK_LAUNCH("act_minmax launch");
  hipLaunchKernelGGL(act_minmax_finalize_kernel, dim3(static_cast<unsigned>(count)), dim3(64), 0, st,
                     static_cast<const float*>(workspace), count, blocks, minmax_out);
  MI355Q_CHECK_LAUNCH("act_minmax finalize launch");
  return MI355Q_OK;
}

// ---- shader clock seen from inside (measurement aid, not on the product path) -------------------------------------
// One wave that watches the constant 100 MHz counter (wall_clock64) for `ticks_100mhz` ticks and reports how many shader
// clocks (clock64) went by: launched on a stream of its own beside a kernel under test, it gives the clock the part
// sustains WHILE that kernel runs -- the hwmon / SMI files on these boxes report a constant 2407 MHz.
namespace mi355q {
namespace {
__global__ __launch_bounds__(64) void clock_probe_kernel(long long ticks_100mhz, long long* out) {
  if (threadIdx.x != 0) return;
  const long long w0 = wall_clock64();
  const long long c0 = clock64();
  long long w = w0;
  while (w - w0 < ticks_100mhz) {
    __builtin_amdgcn_s_sleep(32);
    w = wall_clock64();
  }
  out[0] = clock64() - c0;
  out[1] = w - w0;
}
}  // namespace
}  // namespace mi355q

extern "C" int32_t mi355q_clock_probe(double seconds, int64_t* clocks_and_ticks_out, void* stream) {
  mi355q::clear_error();
  if (!clocks_and_ticks_out) return mi355q::fail(MI355Q_BAD_ARG, "null pointer");
  if (!(seconds > 0.0 && seconds <= 1.0)) return mi355q::fail(MI355Q_BAD_ARG, "probe length must be in (0, 1] s");
  hipLaunchKernelGGL(mi355q::clock_probe_kernel, dim3(1), dim3(64), 0, mi355q::as_stream(stream),
                     static_cast<long long>(seconds * 1e8), reinterpret_cast<long long*>(clocks_and_ticks_out));
  MI355Q_CHECK_LAUNCH("clock probe launch");
  return MI355Q_OK;
}
