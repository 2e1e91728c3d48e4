// kbench -- times variants of the fused requant kernels on one MI355X and checks
// every variant bit-for-bit against the plain-IEEE-division baseline.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude \
//         -Iai-edge-quantizer_amd/csrc tools/kbench/kbench.hip -o tools/kbench/kbench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "requant_kernels.h"

namespace mi355q {  // the kernels reference these; no host error plumbing needed here
void set_error(const char*, ...) {}
int32_t fail(mi355q_status st, const char*, ...) { return st; }
void clear_error() {}
}  // namespace mi355q

using namespace mi355q;
using namespace mi355q::requant;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } \
  } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// N(0, sigma) via Box-Muller on a counter hash; mode 1 plants values near half-integer quotients.
__global__ void fill_kernel(float* x, size_t n, uint32_t seed, float sigma, int mode) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t a = hash32((uint32_t)i * 2u + seed), b = hash32((uint32_t)i * 2u + 1u + seed * 31u);
    float u1 = (a >> 8) * (1.0f / 16777216.0f) + 1e-7f, u2 = (b >> 8) * (1.0f / 16777216.0f);
    float v = sigma * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
    if (mode == 1 && (i & 3) == 1) {
      // k + 0.5 (+- a few ulp) in units of a plausible scale; the row max is pinned elsewhere
      int k = (int)(a % 255u) - 127;
      float base = ((float)k + 0.5f) * (4.0f / 127.0f);
      int ulps = (int)(b % 9u) - 4;
      v = __uint_as_float(__float_as_uint(base) + ulps);
    }
    x[i] = v;
  }
}
__global__ void pin_rowmax_kernel(float* x, int rows, int cols, float value) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) x[(size_t)r * cols] = value;
}

// traffic-equivalent lower bound: read float4, write 1 byte per element, no math
__global__ __launch_bounds__(256) void stream_bound_kernel(const float4* __restrict__ x, uint32_t* __restrict__ q, size_t n4) {
  size_t i = blockIdx.x * (size_t)1024 + threadIdx.x;
  float4 v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) v[u] = i + u * 256 < n4 ? x[i + u * 256] : make_float4(0, 0, 0, 0);
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (i + u * 256 < n4) q[i + u * 256] = __float_as_uint(v[u].x) ^ __float_as_uint(v[u].y) ^ __float_as_uint(v[u].z) ^ __float_as_uint(v[u].w);
}

struct Buf { float* x; int8_t* q; uint8_t* p; float* s; uint16_t* s16; };

struct Case {
  const char* name;
  int64_t rows, cols; int block, bits, pool; bool packed_only;
  std::vector<Buf> bufs;
  void** tx; void** tq; void** tp; void** ts; void** t16;
  size_t alg_bytes;
};

static void alloc_case(Case& c, float sigma, int mode) {
  size_t n = (size_t)c.rows * c.cols;
  size_t nscale = c.block ? n / c.block : c.rows;
  std::vector<void*> hx, hq, hp, hs, h16;
  for (int i = 0; i < c.pool; ++i) {
    Buf b{};
    CK(hipMalloc(&b.x, n * 4)); CK(hipMalloc(&b.q, n)); CK(hipMalloc(&b.p, n)); CK(hipMalloc(&b.s, nscale * 4));
    CK(hipMalloc(&b.s16, nscale * 2));
    fill_kernel<<<4096, 256>>>(b.x, n, 1234u + i, sigma, mode);
    if (mode == 1) pin_rowmax_kernel<<<(c.rows + 255) / 256, 256>>>(b.x, (int)c.rows, (int)c.cols, 4.0f);
    c.bufs.push_back(b);
    hx.push_back(b.x); hq.push_back(b.q); hp.push_back(b.p); hs.push_back(b.s); h16.push_back(b.s16);
  }
  auto up = [&](std::vector<void*>& h, void**& d) { CK(hipMalloc(&d, h.size() * sizeof(void*))); CK(hipMemcpy(d, h.data(), h.size() * sizeof(void*), hipMemcpyHostToDevice)); };
  up(hx, c.tx); up(hq, c.tq); up(hp, c.tp); up(hs, c.ts); up(h16, c.t16);
  CK(hipDeviceSynchronize());
}

template <typename F>
static float time_ms(F launch, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; ++i) launch();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}

static std::vector<uint8_t> snapshot(const Case& c, bool packed) {
  size_t n = (size_t)c.rows * c.cols;
  size_t nb = packed ? n * c.bits / 8 : n;
  size_t nscale = c.block ? n / c.block : c.rows;
  std::vector<uint8_t> h(nb + nscale * 4);
  CK(hipMemcpy(h.data(), packed ? (void*)c.bufs[1].p : (void*)c.bufs[1].q, nb, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h.data() + nb, c.bufs[1].s, nscale * 4, hipMemcpyDeviceToHost));
  return h;
}

static std::vector<uint8_t> g_ref;

struct Variant {
  std::string name;
  std::function<void()> launch;
  bool packed;
  std::vector<float> ms;
  std::string verdict;
};

static void run_interleaved(const Case& c, std::vector<Variant>& vs, int rounds, int iters) {
  for (size_t k = 0; k < vs.size(); ++k) {  // correctness first
    vs[k].launch();
    CK(hipDeviceSynchronize());
    auto snap = snapshot(c, vs[k].packed);
    if (k == 0) { g_ref = snap; vs[k].verdict = "ref"; }
    else vs[k].verdict = (snap == g_ref) ? "bit-exact" : "MISMATCH";
  }
  for (int r = 0; r < rounds; ++r)
    for (auto& v : vs) v.ms.push_back(time_ms(v.launch, iters));
  double bytes = (double)c.alg_bytes * c.pool;
  for (auto& v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    float med = v.ms[v.ms.size() / 2], mn = v.ms.front();
    printf("%-28s %-30s med %8.2f us  min %8.2f us  %7.1f GB/s alg (med)  %5.1f%% of 8TB/s  [%s]\n", c.name, v.name.c_str(),
           med * 1e3, mn * 1e3, bytes / med / 1e6, bytes / med / 1e6 / 80.0, v.verdict.c_str());
  }
  fflush(stdout);
}

template <int BITS, int TPR, int R, bool FAST, bool NT = false>
static Variant rows_variant(Case& c, const char* tag) {
  RequantArgs a{c.tx, c.tq, nullptr, c.ts, nullptr, nullptr, c.rows, c.cols, 0};
  if (c.packed_only) { a.q = nullptr; a.packed = c.tp; }
  dim3 grid((unsigned)((c.rows + (256 / TPR) - 1) / (256 / TPR)), c.pool);
  return Variant{tag, [=] { hipLaunchKernelGGL((requant_rows_kernel<BITS, TPR, R, FAST, true, NT>), grid, dim3(256), 0, 0, a); }, c.packed_only, {}, ""};
}

template <int BITS, int G4, int U, int CL, bool FAST, bool NT = false>
static Variant groups_variant(Case& c, const char* tag) {
  RequantArgs a{c.tx, nullptr, c.tp, c.ts, c.t16, nullptr, c.rows, c.cols, c.block};
  if (!c.packed_only) { a.q = c.tq; a.packed = nullptr; }
  int64_t n4 = c.rows * c.cols / 4;
  dim3 grid((unsigned)((n4 + 256 * U * CL - 1) / (256 * U * CL)), c.pool);
  return Variant{tag, [=] { hipLaunchKernelGGL((requant_groups_kernel<BITS, G4, U, CL, FAST, true, NT>), grid, dim3(256), 0, 0, a); }, c.packed_only, {}, ""};
}

int main(int argc, char** argv) {
  int mode = argc > 1 ? atoi(argv[1]) : 0;  // 1 = adversarial half-integer data
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("# device %s, %d CUs; data mode %d\n", prop.gcnArchName, prop.multiProcessorCount, mode);

  const int rounds = 7;
  {  // ---- C2: 4096x4096 per-channel int8, pool of 16 (1 GiB in)
    Case c{"C2 4096x4096 cw int8", 4096, 4096, 0, 8, 16, false};
    c.alg_bytes = 4096ull * 4096 * 5 + 4096 * 4 + 4096;
    alloc_case(c, 1.0f, mode);
    std::vector<Variant> vs;
    vs.push_back(rows_variant<8, 256, 4, false>(c, "rows TPR=256 R=4 ieee"));
    vs.push_back(rows_variant<8, 256, 4, false, true>(c, "rows TPR=256 R=4 ieee nt"));
    vs.push_back(rows_variant<8, 256, 8, false>(c, "rows TPR=256 R=8(pred) ieee"));
    run_interleaved(c, vs, rounds, 30);
    {
      RequantArgs a{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 4096, 4096, 0};
      int k = 0;
      float ms1 = time_ms([&] {
        const Buf& b = c.bufs[k++ % c.pool];
        a.x = b.x; a.q = b.q; a.scale = b.s;
        hipLaunchKernelGGL((requant_rows_kernel<8, 256, 4, false, false>), dim3(4096), dim3(256), 0, 0, a);
      }, 640);
      printf("%-28s %-30s %9.2f us/launch %8.1f GB/s alg  %5.1f%% of 8TB/s  [1 buffer per launch, same stream]\n", c.name,
             "rows single ieee", ms1 * 1e3, (double)c.alg_bytes / ms1 / 1e6, (double)c.alg_bytes / ms1 / 1e6 / 80.0);
      hipStream_t st[4];
      for (auto& s_ : st) CK(hipStreamCreate(&s_));
      k = 0;
      float ms4 = time_ms([&] {
        for (int j = 0; j < 4; ++j) {
          const Buf& b = c.bufs[k++ % c.pool];
          a.x = b.x; a.q = b.q; a.scale = b.s;
          hipLaunchKernelGGL((requant_rows_kernel<8, 256, 4, false, false>), dim3(4096), dim3(256), 0, st[j], a);
        }
      }, 160) / 4;
      CK(hipDeviceSynchronize());
      printf("%-28s %-30s %9.2f us/launch %8.1f GB/s alg  %5.1f%% of 8TB/s  [1 buffer per launch, 4 streams round robin]\n", c.name,
             "rows single ieee", ms4 * 1e3, (double)c.alg_bytes / ms4 / 1e6, (double)c.alg_bytes / ms4 / 1e6 / 80.0);
    }
    Case c4{"C2 4096x4096 cw int4 packed", 4096, 4096, 0, 4, 16, true};
    c4.alg_bytes = 4096ull * 4096 * 4 + 4096ull * 4096 / 2 + 4096 * 4;
    c4.bufs = c.bufs; c4.tx = c.tx; c4.tq = c.tq; c4.tp = c.tp; c4.ts = c.ts; c4.t16 = c.t16;
    std::vector<Variant> v4;
    v4.push_back(rows_variant<4, 256, 4, false>(c4, "rows TPR=256 R=4 ieee"));
    v4.push_back(rows_variant<4, 256, 4, true>(c4, "rows TPR=256 R=4 fast"));
    run_interleaved(c4, v4, rounds, 30);
    for (auto& b : c.bufs) { hipFree(b.x); hipFree(b.q); hipFree(b.p); hipFree(b.s); hipFree(b.s16); }
  }
  {  // ---- C3: 4096x11008 blockwise-128 int4, fused pack, pool of 6 (1 GiB in)
    Case c{"C3 4096x11008 bw128 int4", 4096, 11008, 128, 4, 6, true};
    c.alg_bytes = 4096ull * 11008 * 4 + 4096ull * 11008 / 2 + (4096ull * 11008 / 128) * 2;
    alloc_case(c, 0.02f, mode);
    std::vector<Variant> vs;
    vs.push_back(groups_variant<4, 32, 4, 1, false>(c, "groups U=4 CL=1 ieee"));
    vs.push_back(groups_variant<4, 32, 1, 2, false>(c, "groups U=1 CL=2 ieee"));
    vs.push_back(groups_variant<4, 32, 1, 2, false, true>(c, "groups U=1 CL=2 ieee nt"));
    vs.push_back(groups_variant<4, 32, 2, 2, false, true>(c, "groups U=2 CL=2 ieee nt"));
    vs.push_back(groups_variant<4, 32, 2, 2, false>(c, "groups U=2 CL=2 ieee"));
    vs.push_back(groups_variant<4, 32, 3, 2, false>(c, "groups U=3 CL=2 ieee"));
    vs.push_back(groups_variant<4, 32, 4, 2, false>(c, "groups U=4 CL=2 ieee"));
    vs.push_back(groups_variant<4, 32, 1, 2, true>(c, "groups U=1 CL=2 fast"));
    vs.push_back(groups_variant<4, 32, 2, 2, true>(c, "groups U=2 CL=2 fast"));
    vs.push_back(groups_variant<4, 32, 2, 4, false>(c, "groups U=2 CL=4 ieee"));
    run_interleaved(c, vs, rounds, 20);
    Case c8{"C3 4096x11008 bw128 int8", 4096, 11008, 128, 8, 6, false};
    c8.alg_bytes = 4096ull * 11008 * 5 + (4096ull * 11008 / 128) * 4;
    c8.bufs = c.bufs; c8.tx = c.tx; c8.tq = c.tq; c8.tp = c.tp; c8.ts = c.ts; c8.t16 = c.t16;
    std::vector<Variant> v8;
    v8.push_back(groups_variant<8, 32, 4, 1, false>(c8, "groups U=4 CL=1 ieee"));
    v8.push_back(groups_variant<8, 32, 2, 2, false>(c8, "groups U=2 CL=2 ieee"));
    v8.push_back(groups_variant<8, 32, 2, 1, false>(c8, "groups U=2 CL=1 ieee"));
    v8.push_back(groups_variant<8, 32, 8, 1, false>(c8, "groups U=8 CL=1 ieee"));
    run_interleaved(c8, v8, rounds, 20);
  }
  return 0;
}
