// Timing of the links of the Cholesky step chain: potf2_kernel (one workgroup; phases from
// cycle-counter stamps) and trsm_panel_kernel, on one 64 x 64 block with m rows below it.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DMI355Q_POTF2_PROF -I include -I ai-edge-quantizer_amd/csrc \
//         tools/kbench/potf2_bench.hip ai-edge-quantizer_amd/csrc/{api.cpp,gemm.hip,xtx_bf16x3.hip,xtx_f16x2.hip,file_io.hip} -lpthread -ldl \
//         -o tools/kbench/potf2_bench && tools/kbench/potf2_bench      (tools/refresh_profiles.sh runs the binary; it is git-ignored and travels with gpurun)
#include <cmath>
#include <cstdio>
#include <vector>
#include "../../ai-edge-quantizer_amd/csrc/gptq.hip"

int main() {
  const int nb = 64, m = 1984, d = nb + m;   // the first step of a d = 2048 factorization
  std::vector<double> h(static_cast<size_t>(d) * d, 0.0);
  for (int i = 0; i < d; ++i)
    for (int j = 0; j <= i && j < nb; ++j) h[static_cast<size_t>(i) * d + j] = (i == j ? nb + 1.0 : 0.5 / (1 + (i - j) % 97));
  double *a, *lt;
  long long* prof;
  int* info;
  hipMalloc(&a, h.size() * 8);
  hipMalloc(&lt, NB * NB * 8);
  hipMalloc(&prof, 64 * 8);
  hipMalloc(&info, 4);
  hipMemset(info, 0, 4);
  hipEvent_t e0, e1, e2;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventCreate(&e2);
  float best_potf2 = 1e9, best_trsm = 1e9;
  for (int it = 0; it < 200; ++it) {
    hipMemcpy(a, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipEventRecord(e0);
    hipLaunchKernelGGL(mi355q::potf2_kernel, dim3(1), dim3(mi355q::kPotf2Threads), 0, 0, a, d, 0, nb, info, lt, prof);
    hipEventRecord(e1);
    hipLaunchKernelGGL(mi355q::trsm_panel_kernel, dim3((m + 63) / 64), dim3(64), 0, 0, a, d, 0, nb, m, lt, prof);
    hipEventRecord(e2);
    hipEventSynchronize(e2);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (it > 20 && ms < best_potf2) best_potf2 = ms;
    hipEventElapsedTime(&ms, e1, e2);
    if (it > 20 && ms < best_trsm) best_trsm = ms;
  }
  std::vector<long long> stamps(64);
  hipMemcpy(stamps.data(), prof, 64 * 8, hipMemcpyDeviceToHost);
  std::vector<double> l(h.size());
  hipMemcpy(l.data(), a, h.size() * 8, hipMemcpyDeviceToHost);
  double err = 0;  // || L L^T - A || over the first nb columns
  for (int i = 0; i < d; ++i)
    for (int j = 0; j <= i && j < nb; ++j) {
      double s = 0;
      for (int c = 0; c <= j; ++c) s += l[static_cast<size_t>(i) * d + c] * l[static_cast<size_t>(j) * d + c];
      err = fmax(err, fabs(s - h[static_cast<size_t>(i) * d + j]));
    }
  int hinfo = -1;
  hipMemcpy(&hinfo, info, 4, hipMemcpyDeviceToHost);
  printf("potf2 %.2f us, trsm (%d rows) %.2f us, |L L^T - A|max %.3e, info %d\n", best_potf2 * 1e3, m, best_trsm * 1e3, err, hinfo);
  const char* names[] = {"load", "factor", "store"};
  for (int i = 1; i <= 3; ++i) printf("  %-8s %lld cycles\n", names[i - 1], stamps[i] - stamps[i - 1]);
  const char* tnames[] = {"stage", "solve", "store"};
  for (int i = 9; i <= 11; ++i) printf("  trsm %-8s %lld cycles\n", tnames[i - 9], stamps[i] - stamps[i - 1]);
  return 0;
}
