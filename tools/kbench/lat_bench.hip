// Single-wave instruction latency / issue-rate probes for gfx950 (s_memtime deltas; the values a
// probe works on are operands of the timer statements, so its work cannot leave the timed region).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kbench/lat_bench.hip -o tools/kbench/lat_bench
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 256
typedef double Pair __attribute__((ext_vector_type(2)));

#define NOW8(t, T, v)                                                                                          \
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)"                    \
               : "=s"(t), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) \
               : : "memory")

__global__ __launch_bounds__(64) void probe(double* out, long long* cyc, double seed) {
  __shared__ __attribute__((aligned(16))) double lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = seed + i;
  __syncthreads();
  long long t0, t1;
  int slot = 0;
  double x[8];
  float f[8];
  for (int k = 0; k < 8; ++k) { x[k] = seed + threadIdx.x + k; f[k] = static_cast<float>(x[k]); }
  double b = seed * 0.5, c = 1.0;
  float fb = 0.5f, fc = 1.0f;
  asm volatile("" : "+v"(b), "+v"(c), "+v"(fb), "+v"(fc));
  // 1. dependent f64 FMA chain
  NOW8(t0, double, x);
#pragma unroll
  for (int i = 0; i < REP; ++i) x[0] = __builtin_fma(x[0], b, c);
  NOW8(t1, double, x);
  if (threadIdx.x == 0) cyc[slot] = t1 - t0; ++slot;
  // 2. 8 independent f64 FMA chains
  NOW8(t0, double, x);
#pragma unroll
  for (int i = 0; i < REP / 8; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = __builtin_fma(x[k], b, c);
  }
  NOW8(t1, double, x);
  if (threadIdx.x == 0) cyc[slot] = t1 - t0; ++slot;
  // 3. dependent f32 chain
  NOW8(t0, float, f);
#pragma unroll
  for (int i = 0; i < REP; ++i) f[0] = __builtin_fmaf(f[0], fb, fc);
  NOW8(t1, float, f);
  if (threadIdx.x == 0) cyc[slot] = t1 - t0; ++slot;
  // 4. 8 independent f32 chains
  NOW8(t0, float, f);
#pragma unroll
  for (int i = 0; i < REP / 8; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = __builtin_fmaf(f[k], fb, fc);
  }
  NOW8(t1, float, f);
  if (threadIdx.x == 0) cyc[slot] = t1 - t0; ++slot;
  // 5. (2 readlanes -> f64 fma) dependent chain
  NOW8(t0, double, x);
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const long long bits = __double_as_longlong(x[0]);
    const int lo = __builtin_amdgcn_readlane(static_cast<int>(bits), (i * 7) & 63);
    const int hi = __builtin_amdgcn_readlane(static_cast<int>(bits >> 32), (i * 7) & 63);
    const double s = __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
    x[0] = __builtin_fma(x[0], b, s);
  }
  NOW8(t1, double, x);
  if (threadIdx.x == 0) cyc[slot] = t1 - t0; ++slot;
  // 6. 32 ds_read_b128 with a wave-uniform address, all in flight, one wait
  const Pair* l2 = reinterpret_cast<const Pair*>(lds);
  {
    Pair v[32];
    NOW8(t0, double, x);
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = l2[k];
#pragma unroll
    for (int k = 0; k < 32; k += 4) asm volatile("" : "+v"(v[k]), "+v"(v[k + 1]), "+v"(v[k + 2]), "+v"(v[k + 3]));
    NOW8(t1, double, x);
#pragma unroll
    for (int k = 0; k < 32; ++k) x[k & 7] += v[k].x + v[k].y;
  }
  if (threadIdx.x == 0) cyc[slot] = t1 - t0; ++slot;
  // 7. 32 ds_read_b64 with a wave-uniform address, all in flight, one wait
  {
    double v[32];
    NOW8(t0, double, x);
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = lds[2 * k + 1];
#pragma unroll
    for (int k = 0; k < 32; k += 4) asm volatile("" : "+v"(v[k]), "+v"(v[k + 1]), "+v"(v[k + 2]), "+v"(v[k + 3]));
    NOW8(t1, double, x);
#pragma unroll
    for (int k = 0; k < 32; ++k) x[k & 7] += v[k];
  }
  if (threadIdx.x == 0) cyc[slot] = t1 - t0; ++slot;
  // 8. dependent v_rcp_f64
  NOW8(t0, double, x);
#pragma unroll
  for (int i = 0; i < 64; ++i) x[0] = __builtin_amdgcn_rcp(x[0]);
  NOW8(t1, double, x);
  if (threadIdx.x == 0) cyc[slot] = t1 - t0; ++slot;
  // 9. one LDS round trip, dependent (address from the value read)
  int idx = threadIdx.x & 1;
  NOW8(t0, double, x);
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const double v = lds[idx];
    idx = (__double_as_longlong(v) >> 60) & 1;   // 0: the values are ~1.0 .. 1024
  }
  x[1] += idx;
  NOW8(t1, double, x);
  if (threadIdx.x == 0) cyc[slot] = t1 - t0; ++slot;
  // 10. fixed wall time reference: s_sleep 100 x 64 clocks
  NOW8(t0, double, x);
  for (int i = 0; i < 100; ++i) __builtin_amdgcn_s_sleep(1);
  NOW8(t1, double, x);
  if (threadIdx.x == 0) cyc[slot] = t1 - t0; ++slot;
  double s = 0;
  for (int k = 0; k < 8; ++k) s += x[k] + f[k];
  out[threadIdx.x] = s;
}

int main() {
  double* out;
  long long* cyc;
  hipMalloc(&out, 64 * 8);
  hipMalloc(&cyc, 16 * 8);
  long long h[16];
  for (int it = 0; it < 3; ++it) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, 1.000001);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"f64 fma dependent", "f64 fma 8 chains", "f32 fma dependent", "f32 fma 8 chains",
                         "2 readlane + f64 fma dependent", "ds_read_b128 uniform, 32 in flight", "ds_read_b64 uniform, 32 in flight",
                         "rcp f64 dependent", "LDS round trip (b64)", "s_sleep 1 (64 clocks each)"};
  const int counts[] = {256, 256, 256, 256, 64, 32, 32, 64, 64, 100};
  for (int i = 0; i < 10; ++i) printf("%-36s %4d ops %7lld ticks  %.2f per op\n", names[i], counts[i], h[i], double(h[i]) / counts[i]);
  return 0;
}
