// Which CUs does a stream created with hipExtStreamCreateWithCUMask use?  One record per workgroup:
// XCC id, SE id, CU id (s_getreg HW_ID / XCC_ID).   hipcc --offload-arch=gfx950 -O2 cumask_probe.hip -o cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

__global__ void where_kernel(unsigned* out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xF) << 16 | (hw & 0xFFFF);
}

int main(int argc, char** argv) {
  const int nblocks = 4096;
  unsigned* d;
  hipMalloc(&d, nblocks * sizeof(unsigned));
  std::vector<unsigned> h(nblocks);
  const char* masks[] = {"all", "low32", "word0", "word7", "even_words", "lowbyte_each"};
  for (const char* name : masks) {
    uint32_t m[8];
    for (int i = 0; i < 8; ++i) m[i] = 0xFFFFFFFFu;
    if (!strcmp(name, "low32")) { for (int i = 1; i < 8; ++i) m[i] = 0; }
    if (!strcmp(name, "word0")) { for (int i = 1; i < 8; ++i) m[i] = 0; m[0] = 0x0000FFFF; }
    if (!strcmp(name, "word7")) { for (int i = 0; i < 7; ++i) m[i] = 0; }
    if (!strcmp(name, "even_words")) { for (int i = 1; i < 8; i += 2) m[i] = 0; }
    if (!strcmp(name, "lowbyte_each")) { for (int i = 0; i < 8; ++i) m[i] = 0x000000FF; }
    hipStream_t s;
    hipError_t e = strcmp(name, "all") ? hipExtStreamCreateWithCUMask(&s, 8, m) : hipStreamCreate(&s);
    if (e != hipSuccess) { printf("%s: create failed: %s\n", name, hipGetErrorString(e)); continue; }
    hipMemsetAsync(d, 0xFF, nblocks * sizeof(unsigned), s);
    hipLaunchKernelGGL(where_kernel, dim3(nblocks), dim3(256), 0, s, d, 2000);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d, nblocks * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_xcc;
    std::map<unsigned, int> cus;
    for (unsigned v : h) {
      const unsigned xcc = v >> 16, se = (v >> 13) & 7, sh = (v >> 12) & 1, cu = (v >> 8) & 15;
      per_xcc[xcc]++;
      cus[xcc << 12 | se << 8 | sh << 4 | cu]++;
    }
    printf("%-14s distinct CUs %3zu  per XCC:", name, cus.size());
    for (auto& kv : per_xcc) printf(" x%u=%d", kv.first, kv.second);
    printf("\n");
    if (cus.size() <= 40) {
      printf("   (xcc.se.sh.cu):");
      for (auto& kv : cus) printf(" %u.%u.%u.%u", kv.first >> 12, (kv.first >> 8) & 15, (kv.first >> 4) & 15, kv.first & 15);
      printf("\n");
    }
    hipStreamDestroy(s);
  }
  return 0;
}
