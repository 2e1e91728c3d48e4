// Status / error plumbing and device facts for libmi355q.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "common.h"

namespace mi355q {
namespace {
thread_local char g_err[512] = "";
}

void clear_error() { g_err[0] = '\0'; }

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int32_t fail(mi355q_status st, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return static_cast<int32_t>(st);
}
}  // namespace mi355q

extern "C" int32_t mi355q_version(void) { return MI355Q_VERSION; }

extern "C" const char* mi355q_last_error(void) { return mi355q::g_err; }

extern "C" int32_t mi355q_device_info(int32_t* cu_count_host, int32_t* wavefront_size_host,
                                      char* arch_name_host, int32_t arch_name_len) {
  mi355q::clear_error();
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  hipDeviceProp_t prop;
  if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return mi355q::fail(MI355Q_HIP_ERROR, "device query: %s", hipGetErrorString(e));
  if (cu_count_host) *cu_count_host = prop.multiProcessorCount;
  if (wavefront_size_host) *wavefront_size_host = prop.warpSize;
  if (arch_name_host && arch_name_len > 0) {
    std::strncpy(arch_name_host, prop.gcnArchName, static_cast<size_t>(arch_name_len) - 1);
    arch_name_host[arch_name_len - 1] = '\0';
  }
  return MI355Q_OK;
}
