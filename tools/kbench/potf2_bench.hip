// Phase timing of potf2_inv_kernel (one workgroup) with s_memtime stamps.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DMI355Q_POTF2_PROF -I include -I ai-edge-quantizer_amd/csrc \
//         tools/kbench/potf2_bench.hip ai-edge-quantizer_amd/csrc/{api.cpp,gemm.hip} -o /tmp/potf2_bench && /tmp/potf2_bench
#include <cstdio>
#include <vector>
#include <cmath>
#include "../../ai-edge-quantizer_amd/csrc/gptq.hip"

int main() {
  const int d = 64;
  std::vector<double> h(d * d, 0.0);
  for (int i = 0; i < d; ++i)
    for (int j = 0; j <= i; ++j) h[i * d + j] = (i == j ? d + 1.0 : 0.5 / (1 + i - j));
  double *a, *dinv;
  int* info;
  hipMalloc(&a, d * d * 8);
  hipMalloc(&dinv, (NB * NB + 64) * 8);
  hipMalloc(&info, 4);
  hipMemset(info, 0, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9;
  for (int it = 0; it < 200; ++it) {
    hipMemcpy(a, h.data(), d * d * 8, hipMemcpyHostToDevice);
    hipEventRecord(e0);
    hipLaunchKernelGGL(mi355q::potf2_inv_kernel, dim3(1), dim3(256), 0, 0, a, d, 0, d, dinv, info);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (it > 20 && ms < best) best = ms;
  }
  std::vector<long long> prof(64);
  hipMemcpy(prof.data(), dinv + NB * NB, 64 * 8, hipMemcpyDeviceToHost);
  std::vector<double> l(d * d), x(NB * NB);
  hipMemcpy(l.data(), a, d * d * 8, hipMemcpyDeviceToHost);
  hipMemcpy(x.data(), dinv, NB * NB * 8, hipMemcpyDeviceToHost);
  double err = 0;  // || L X - I ||
  for (int i = 0; i < d; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0;
      for (int m = j; m <= i; ++m) s += l[i * d + m] * x[m * NB + j];
      err = fmax(err, fabs(s - (i == j)));
    }
  printf("best %.2f us, |LX-I|max %.3e\n", best * 1e3, err);
  const char* names[] = {"load", "factor", "L->lds", "lvl1", "lvl2", "lvl4", "lvl8", "lvl16", "lvl32", "store"};
  for (int i = 1; i <= 10; ++i) printf("  %-8s %lld cycles\n", names[i - 1], prof[i] - prof[i - 1]);
  return 0;
}
