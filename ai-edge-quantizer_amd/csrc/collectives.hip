// RCCL behind the C ABI: the exchange steps of the calibration path.
//
//   X1  activation ranges: all-gather of per-sample (min, max) pairs, replayed in dataset order
//       by the host (the reference's moving average, ref utils/qsv_utils.py:43-68, is order
//       dependent); all-reduce(min) / (max) when the update rule is min_max_update (ref :105-122).
//   X2  GPTQ Hessians: the sample-weighted mean of ref utils/qsv_utils.py:71-102 over all ranks'
//       samples = sum over ranks of (n_rank / N) * H_rank: one in-place all-reduce(sum) of d x d
//       FP64 per distinct Hessian (16 MiB at d = 2048, 2 GiB at d = 16384).
//
// One process per GPU; the communicator is an opaque handle created from a 128-byte unique id the
// host broadcasts over whatever rendezvous it already has. RCCL is bound at run time (dlopen of
// librccl.so.1 -- the copy already in the process when PyTorch-ROCm is loaded, so both share one
// runtime): the library itself has no link-time dependency on it, and a missing RCCL surfaces
// as MI355Q_RCCL_ERROR from these entry points only.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "common.h"

namespace mi355q {
namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
char g_rccl_why[256] = "unknown";    // why binding failed, captured when it happened (dlerror() is one-shot)

void keep_why(const char* what) {
  const char* e = dlerror();
  snprintf(g_rccl_why, sizeof(g_rccl_why), "%s: %s", what, e ? e : "unknown");
}

template <typename F>
bool bind(void* h, const char* name, F* out) {
  *out = reinterpret_cast<F>(dlsym(h, name));
  return *out != nullptr;
}

const Rccl* rccl() {
  std::call_once(g_rccl_once, [] {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);   // the copy PyTorch-ROCm brought
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) { keep_why("dlopen(librccl.so.1)"); return; }
    Rccl& r = g_rccl;
    r.handle = h;
    r.ok = bind(h, "ncclGetUniqueId", &r.GetUniqueId) && bind(h, "ncclCommInitRank", &r.CommInitRank) &&
           bind(h, "ncclCommDestroy", &r.CommDestroy) && bind(h, "ncclCommCount", &r.CommCount) &&
           bind(h, "ncclCommUserRank", &r.CommUserRank) && bind(h, "ncclAllReduce", &r.AllReduce) &&
           bind(h, "ncclAllGather", &r.AllGather) && bind(h, "ncclReduce", &r.Reduce) && bind(h, "ncclGroupStart", &r.GroupStart) &&
           bind(h, "ncclGroupEnd", &r.GroupEnd) && bind(h, "ncclGetErrorString", &r.GetErrorString);
    if (!r.ok) keep_why("dlsym");
  });
  return g_rccl.ok ? &g_rccl : nullptr;
}

int32_t no_rccl() { return fail(MI355Q_RCCL_ERROR, "librccl.so.1 could not be bound (%s)", g_rccl_why); }

#define MI355Q_RCCL(call, what)                                                             \
  do {                                                                                      \
    ncclResult_t r__ = (call);                                                              \
    if (r__ != ncclSuccess)                                                                 \
      return fail(MI355Q_RCCL_ERROR, "%s: %s", what, R->GetErrorString(r__));               \
  } while (0)

__global__ __launch_bounds__(256) void scale_f64_kernel(double* __restrict__ x, long long n, double a) {
  const long long stride = static_cast<long long>(gridDim.x) * 256;
  for (long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) x[i] *= a;
}

// The Hessian is symmetric: only its lower triangle travels. Row i of the packed form starts at
// i (i + 1) / 2. pack: packed = weight * lower(h); unpack: h = packed mirrored to both triangles.
__global__ __launch_bounds__(256) void pack_lower_f64_kernel(const double* __restrict__ h, long long d, double weight,
                                                            double* __restrict__ packed) {
  const long long i = blockIdx.y;
  const long long row = i * (i + 1) / 2;
  for (long long j = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; j <= i; j += static_cast<long long>(gridDim.x) * 256)
    packed[row + j] = h[i * d + j] * weight;
}

__global__ __launch_bounds__(256) void unpack_lower_f64_kernel(const double* __restrict__ packed, long long d,
                                                              double* __restrict__ h) {
  __shared__ double tile[32][33];
  const int bi = blockIdx.y, bj = blockIdx.x;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  const bool upper = bj > bi;
  const int si = upper ? bj : bi, sj = upper ? bi : bj;         // source tile (lower triangle)
  for (int r = ty; r < 32; r += 8) {
    const long long i = si * 32LL + r, j = sj * 32LL + tx;
    double v = 0.0;
    if (i < d && j < d) {
      const long long a = i >= j ? i : j, b = i >= j ? j : i;
      v = packed[a * (a + 1) / 2 + b];
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const long long i = bi * 32LL + r, j = bj * 32LL + tx;
    if (i < d && j < d) h[i * d + j] = upper ? tile[tx][r] : tile[r][tx];
  }
}

// The same for the float32 product (lower triangle valid on both sides): pack, and unpack into the lower triangle.
__global__ __launch_bounds__(256) void pack_lower_f32_kernel(const float* __restrict__ p, long long d, float* __restrict__ packed) {
  const long long i = blockIdx.y;
  const long long row = i * (i + 1) / 2;
  for (long long j = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; j <= i; j += static_cast<long long>(gridDim.x) * 256)
    packed[row + j] = p != nullptr ? p[i * d + j] : 0.f;
}

__global__ __launch_bounds__(256) void unpack_lower_f32_kernel(const float* __restrict__ packed, long long d, float* __restrict__ p) {
  const long long i = blockIdx.y;
  const long long row = i * (i + 1) / 2;
  for (long long j = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; j <= i; j += static_cast<long long>(gridDim.x) * 256)
    p[i * d + j] = packed[row + j];
}

inline ncclComm_t as_comm(void* c) { return reinterpret_cast<ncclComm_t>(c); }

}  // namespace
}  // namespace mi355q

using namespace mi355q;

extern "C" int32_t mi355q_comm_unique_id(char* id_host) {
  clear_error();
  if (!id_host) return fail(MI355Q_BAD_ARG, "null pointer");
  const Rccl* R = rccl();
  if (!R) return no_rccl();
  ncclUniqueId id;
  MI355Q_RCCL(R->GetUniqueId(&id), "ncclGetUniqueId");
  static_assert(sizeof(id) == MI355Q_UNIQUE_ID_BYTES, "unique id size");
  memcpy(id_host, &id, sizeof(id));
  return MI355Q_OK;
}

extern "C" int32_t mi355q_comm_init_rank(void** comm_out_host, int32_t nranks, const char* id_host,
                                         int32_t rank) {
  clear_error();
  if (!comm_out_host || !id_host) return fail(MI355Q_BAD_ARG, "null pointer");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(MI355Q_BAD_ARG, "bad rank %d of %d", rank, nranks);
  const Rccl* R = rccl();
  if (!R) return no_rccl();
  ncclUniqueId id;
  memcpy(&id, id_host, sizeof(id));
  ncclComm_t comm = nullptr;
  MI355Q_RCCL(R->CommInitRank(&comm, nranks, id, rank), "ncclCommInitRank");
  *comm_out_host = comm;
  return MI355Q_OK;
}

extern "C" int32_t mi355q_comm_destroy(void* comm) {
  clear_error();
  if (!comm) return MI355Q_OK;
  const Rccl* R = rccl();
  if (!R) return no_rccl();
  MI355Q_RCCL(R->CommDestroy(as_comm(comm)), "ncclCommDestroy");
  return MI355Q_OK;
}

extern "C" int32_t mi355q_comm_info(void* comm, int32_t* nranks_host, int32_t* rank_host) {
  clear_error();
  if (!comm) return fail(MI355Q_BAD_ARG, "null communicator");
  const Rccl* R = rccl();
  if (!R) return no_rccl();
  int n = 0, r = 0;
  MI355Q_RCCL(R->CommCount(as_comm(comm), &n), "ncclCommCount");
  MI355Q_RCCL(R->CommUserRank(as_comm(comm), &r), "ncclCommUserRank");
  if (nranks_host) *nranks_host = n;
  if (rank_host) *rank_host = r;
  return MI355Q_OK;
}

extern "C" int32_t mi355q_allgather_minmax(void* comm, const float* local, int64_t n, float* out,
                                           void* stream) {
  clear_error();
  if (!comm) return fail(MI355Q_BAD_ARG, "null communicator");
  if (n < 0) return fail(MI355Q_BAD_ARG, "negative size");
  if (n == 0) return MI355Q_OK;
  if (!local || !out) return fail(MI355Q_BAD_ARG, "null pointer");
  const Rccl* R = rccl();
  if (!R) return no_rccl();
  MI355Q_RCCL(R->AllGather(local, out, static_cast<size_t>(n), ncclFloat32, as_comm(comm), as_stream(stream)),
              "ncclAllGather");
  return MI355Q_OK;
}

namespace {
template <ncclDataType_t DT>
int32_t allreduce(void* comm, void* buf, int64_t n, ncclRedOp_t op, void* stream) {
  clear_error();
  if (!comm) return fail(MI355Q_BAD_ARG, "null communicator");
  if (n < 0) return fail(MI355Q_BAD_ARG, "negative size");
  if (n == 0) return MI355Q_OK;
  if (!buf) return fail(MI355Q_BAD_ARG, "null pointer");
  const Rccl* R = rccl();
  if (!R) return no_rccl();
  MI355Q_RCCL(R->AllReduce(buf, buf, static_cast<size_t>(n), DT, op, as_comm(comm), as_stream(stream)),
              "ncclAllReduce");
  return MI355Q_OK;
}
}  // namespace

extern "C" int32_t mi355q_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream) {
  return allreduce<ncclFloat32>(comm, buf, n, ncclSum, stream);
}

extern "C" int32_t mi355q_allreduce_sum_f64(void* comm, double* buf, int64_t n, void* stream) {
  return allreduce<ncclFloat64>(comm, buf, n, ncclSum, stream);
}

extern "C" int32_t mi355q_allreduce_minmax_f32(void* comm, float* mins, float* maxs, int64_t n, void* stream) {
  clear_error();
  if (!comm) return fail(MI355Q_BAD_ARG, "null communicator");
  if (n < 0) return fail(MI355Q_BAD_ARG, "negative size");
  if (n == 0) return MI355Q_OK;
  if (!mins || !maxs) return fail(MI355Q_BAD_ARG, "null pointer");
  const Rccl* R = rccl();
  if (!R) return no_rccl();
  // the two reductions travel as one group: one launch, one trip around the ring
  MI355Q_RCCL(R->GroupStart(), "ncclGroupStart");
  ncclResult_t a = R->AllReduce(mins, mins, static_cast<size_t>(n), ncclFloat32, ncclMin, as_comm(comm), as_stream(stream));
  ncclResult_t b = R->AllReduce(maxs, maxs, static_cast<size_t>(n), ncclFloat32, ncclMax, as_comm(comm), as_stream(stream));
  MI355Q_RCCL(R->GroupEnd(), "ncclGroupEnd");
  MI355Q_RCCL(a, "ncclAllReduce(min)");
  MI355Q_RCCL(b, "ncclAllReduce(max)");
  return MI355Q_OK;
}

extern "C" size_t mi355q_hessian_exchange_workspace_bytes(int64_t d) {
  if (d <= 0) return 0;
  return static_cast<size_t>(d) * static_cast<size_t>(d + 1) / 2 * sizeof(double);
}

extern "C" int32_t mi355q_reduce_hessian_f64(void* comm, double* hessian, int64_t d, double weight, int32_t root,
                                             void* workspace, size_t workspace_bytes, void* stream) {
  clear_error();
  if (!comm) return fail(MI355Q_BAD_ARG, "null communicator");
  if (d < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  if (d == 0) return MI355Q_OK;
  if (d > 0x7FFFFFFF) return fail(MI355Q_UNSUPPORTED, "dimension too large");
  if (!hessian) return fail(MI355Q_BAD_ARG, "null pointer");
  if (!(weight >= 0.0 && weight <= 1.0)) return fail(MI355Q_BAD_ARG, "weight must be n_rank / N in [0, 1]");
  const size_t need = mi355q_hessian_exchange_workspace_bytes(d);
  if (!workspace || workspace_bytes < need) return fail(MI355Q_BAD_ARG, "workspace too small: need %zu bytes", need);
  const Rccl* R = rccl();
  if (!R) return no_rccl();
  int nranks = 0, me = 0;
  MI355Q_RCCL(R->CommCount(as_comm(comm), &nranks), "ncclCommCount");
  MI355Q_RCCL(R->CommUserRank(as_comm(comm), &me), "ncclCommUserRank");
  if (root >= nranks) return fail(MI355Q_BAD_ARG, "root %d of %d ranks", root, nranks);
  double* packed = static_cast<double*>(workspace);
  const size_t n = static_cast<size_t>(d) * static_cast<size_t>(d + 1) / 2;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(pack_lower_f64_kernel, dim3(static_cast<unsigned>((d + 2047) / 2048 < 1 ? 1 : (d + 2047) / 2048), static_cast<unsigned>(d)),
                     dim3(256), 0, st, hessian, static_cast<long long>(d), weight, packed);
  MI355Q_CHECK_LAUNCH("hessian pack launch");
  if (root < 0)
    MI355Q_RCCL(R->AllReduce(packed, packed, n, ncclFloat64, ncclSum, as_comm(comm), st), "ncclAllReduce(hessian)");
  else
    MI355Q_RCCL(R->Reduce(packed, packed, n, ncclFloat64, ncclSum, root, as_comm(comm), st), "ncclReduce(hessian)");
  if (root < 0 || root == me) {
    const unsigned t32 = static_cast<unsigned>((d + 31) / 32);
    hipLaunchKernelGGL(unpack_lower_f64_kernel, dim3(t32, t32), dim3(256), 0, st, packed, static_cast<long long>(d), hessian);
    MI355Q_CHECK_LAUNCH("hessian unpack launch");
  }
  return MI355Q_OK;
}

extern "C" size_t mi355q_product_exchange_workspace_bytes(int64_t d) {
  if (d <= 0) return 0;
  return static_cast<size_t>(d) * static_cast<size_t>(d + 1) / 2 * sizeof(float);
}

extern "C" int32_t mi355q_reduce_product_f32(void* comm, float* product, int64_t d, int32_t root, void* workspace,
                                             size_t workspace_bytes, void* stream) {
  clear_error();
  if (!comm) return fail(MI355Q_BAD_ARG, "null communicator");
  if (d < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  if (d == 0) return MI355Q_OK;
  if (d > 0x7FFFFFFF) return fail(MI355Q_UNSUPPORTED, "dimension too large");
  const size_t need = mi355q_product_exchange_workspace_bytes(d);
  if (!workspace || workspace_bytes < need) return fail(MI355Q_BAD_ARG, "workspace too small: need %zu bytes", need);
  const Rccl* R = rccl();
  if (!R) return no_rccl();
  int nranks = 0, me = 0;
  MI355Q_RCCL(R->CommCount(as_comm(comm), &nranks), "ncclCommCount");
  MI355Q_RCCL(R->CommUserRank(as_comm(comm), &me), "ncclCommUserRank");
  if (root >= nranks) return fail(MI355Q_BAD_ARG, "root %d of %d ranks", root, nranks);
  const bool receives = root < 0 || root == me;
  if (receives && !product) return fail(MI355Q_BAD_ARG, "the receiving rank needs a product buffer");
  float* packed = static_cast<float*>(workspace);
  const size_t n = static_cast<size_t>(d) * static_cast<size_t>(d + 1) / 2;
  hipStream_t st = as_stream(stream);
  const dim3 grid(static_cast<unsigned>((d + 2047) / 2048 < 1 ? 1 : (d + 2047) / 2048), static_cast<unsigned>(d));
  hipLaunchKernelGGL(pack_lower_f32_kernel, grid, dim3(256), 0, st, product, static_cast<long long>(d), packed);
  MI355Q_CHECK_LAUNCH("product pack launch");
  if (root < 0)
    MI355Q_RCCL(R->AllReduce(packed, packed, n, ncclFloat32, ncclSum, as_comm(comm), st), "ncclAllReduce(product)");
  else
    MI355Q_RCCL(R->Reduce(packed, packed, n, ncclFloat32, ncclSum, root, as_comm(comm), st), "ncclReduce(product)");
  if (receives) {
    hipLaunchKernelGGL(unpack_lower_f32_kernel, grid, dim3(256), 0, st, packed, static_cast<long long>(d), product);
    MI355Q_CHECK_LAUNCH("product unpack launch");
  }
  return MI355Q_OK;
}

extern "C" int32_t mi355q_allreduce_hessian_f64(void* comm, double* hessian, int64_t d, double weight,
                                                void* stream) {
  clear_error();
  if (!comm) return fail(MI355Q_BAD_ARG, "null communicator");
  if (d < 0) return fail(MI355Q_BAD_ARG, "negative shape");
  if (d == 0) return MI355Q_OK;
  if (!hessian) return fail(MI355Q_BAD_ARG, "null pointer");
  if (!(weight >= 0.0 && weight <= 1.0)) return fail(MI355Q_BAD_ARG, "weight must be n_rank / N in [0, 1]");
  const Rccl* R = rccl();
  if (!R) return no_rccl();
  const long long n = static_cast<long long>(d) * d;
  long long blocks = (n + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(scale_f64_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream),
                     hessian, n, weight);
  MI355Q_CHECK_LAUNCH("hessian weight launch");
  MI355Q_RCCL(R->AllReduce(hessian, hessian, static_cast<size_t>(n), ncclFloat64, ncclSum, as_comm(comm),
                           as_stream(stream)),
              "ncclAllReduce(hessian)");
  return MI355Q_OK;
}
