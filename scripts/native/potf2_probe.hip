// Developer probe: where the 128 x 128 diagonal-block kernel spends its cycles (per wave, panel and phase).
//   cd scripts/native && hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -DELFIHIP_POTF2_STAMP \
//       -I../../elfi_amd/csrc -I../../include -c -o /tmp/potf2_probe.o potf2_probe.hip && \
//   hipcc --offload-arch=gfx950 -o potf2_probe /tmp/potf2_probe.o $(ls ../../elfi_amd/csrc/build/*.o | grep -v gp_fit.o)
//   ./potf2_probe
// The kernel under test is compiled from gp_fit.hip itself (included below) with the stamps switched on.
#include "../../elfi_amd/csrc/gp_fit.hip"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace elfihip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const int n = NB;
  std::vector<double> A(n * n), W(n * n, 0.0);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) A[i * n + j] = std::exp(-0.5 * (i - j) * (i - j) / 400.0) + (i == j ? 0.05 : 0.0);
  for (int i = 0; i < n; ++i) W[i * n + i] = 1.0;
  double *dA, *dW, *dW11;
  int* dinfo;
  CK(hipMalloc(&dA, n * n * 8)); CK(hipMalloc(&dW, n * n * 8)); CK(hipMalloc(&dW11, n * n * 8)); CK(hipMalloc(&dinfo, 4));
  CK(hipMemset(dinfo, 0, 4));
  auto kern = potf2_tiles_kernel<1024>;
  const size_t lds = POTF2T_LDS_DOUBLES * sizeof(double);
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  std::vector<long long> st(16 * 8 * 8);
  for (int rep = 0; rep < 20; ++rep) {
    CK(hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, W.data(), n * n * 8, hipMemcpyHostToDevice));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, 0, dA, (int64_t)n, dW, (int64_t)n, dW11, dinfo, 0);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_potf2_stamp), st.size() * 8));
  int info; CK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
  printf("potf2_tiles_kernel<1024>: best %.2f us (event), info %d\n", best * 1e3f, info);
  auto S = [&](int w, int p, int s) { return st[(w * 8 + p) * 8 + s]; };
  const long long t0 = S(0, 0, 0);
  printf("cycle stamps relative to wave 0 / panel 0 / slot 0 (s_memtime ticks)\n");
  printf("phase-A wave 0:  panel | start  loaded  eliminated  stored  barrier1  barrier2 | elim  total\n");
  for (int p = 0; p < 8; ++p) {
    printf("   %d |", p);
    for (int s = 0; s < 6; ++s) printf(" %7lld", S(0, p, s) - t0);
    printf(" | %6lld %6lld\n", S(0, p, 2) - S(0, p, 1), S(0, p, 5) - S(0, p, 0));
  }
  for (int w : {3, 9, 15}) {
    printf("update wave %d:  panel | start  U2done  written  barrier1  U1done  barrier2\n", w);
    for (int p = 0; p < 8; ++p) {
      printf("   %d |", p);
      for (int s = 0; s < 6; ++s) printf(" %7lld", (p == 0 && (s == 1)) ? 0 : S(w, p, s) - t0);
      printf("\n");
    }
    printf("   end of the last write-out: %lld\n", S(w, 7, 6) - t0);
  }
  return 0;
}
