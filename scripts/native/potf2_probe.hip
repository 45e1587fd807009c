// Developer probe: the 128 x 128 diagonal-block kernel -- result against a host factorisation, event time, and where
// the waves spend their cycles (per wave, panel and phase).
//   sh scripts/native/build_potf2_probe.sh potf2_probe && scripts/native/potf2_probe
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
  // host: L and L^-1 in long double
  std::vector<long double> L(n * n, 0.0L), Li(n * n, 0.0L);
  for (int j = 0; j < n; ++j) {
    long double d = A[j * n + j];
    for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
    L[j * n + j] = sqrtl(d);
    for (int i = j + 1; i < n; ++i) {
      long double v = A[i * n + j];
      for (int k = 0; k < j; ++k) v -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = v / L[j * n + j];
    }
  }
  for (int c = 0; c < n; ++c) {
    Li[c * n + c] = 1.0L / L[c * n + c];
    for (int i = c + 1; i < n; ++i) {
      long double v = 0.0L;
      for (int k = c; k < i; ++k) v -= L[i * n + k] * Li[k * n + c];
      Li[i * n + c] = v / L[i * n + i];
    }
  }
  double *dA, *dW, *dW11;
  int* dinfo;
  CK(hipMalloc(&dA, n * n * 8)); CK(hipMalloc(&dW, n * n * 8)); CK(hipMalloc(&dW11, n * n * 8)); CK(hipMalloc(&dinfo, 4));
  CK(hipMemset(dinfo, 0, 4));
  CK(hipMemset(dW11, 0, n * n * 8));
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
#ifdef ELFIHIP_POTF2_STAMP
  CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_potf2_stamp), st.size() * 8));
#endif
  int info; CK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
  std::vector<double> gA(n * n), gW(n * n), gW11(n * n);
  CK(hipMemcpy(gA.data(), dA, n * n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(gW.data(), dW, n * n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(gW11.data(), dW11, n * n * 8, hipMemcpyDeviceToHost));
  double eL = 0, eW = 0, eW11 = 0, mW = 0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      eL = std::fmax(eL, std::fabs(gA[i * n + j] - (double)L[i * n + j]));
      eW = std::fmax(eW, std::fabs(gW[j * n + i] - (double)Li[i * n + j]));       // WT(j, i) = L^-T[j][i] = L^-1[i][j]
      eW11 = std::fmax(eW11, std::fabs(gW11[i * n + j] - (double)Li[i * n + j]));
      mW = std::fmax(mW, std::fabs((double)Li[i * n + j]));
    }
  printf("potf2_tiles_kernel<1024>: best %.2f us (event), info %d\n", best * 1e3f, info);
  printf("max |L - L_host| %.3g   max |WT - L^-T| %.3g   max |W11 - L^-1| %.3g   (max |L^-1| %.3g)\n", eL, eW, eW11, mW);
#ifdef ELFIHIP_POTF2_STAMP
  auto S = [&](int w, int p, int s) { return st[(w * 8 + p) * 8 + s]; };
  const long long t0 = S(8, 0, 0);
  printf("SIMD of tile waves 0..7 and of the factor wave (HW_ID[5:4]):");
  for (int w = 0; w < 9; ++w) printf(" %lld", (S(w, 0, 7) >> 4) & 3);
  printf("\ncycle stamps relative to the factor wave's start (shader clock)\n");
  printf("factor wave:  panel | start  tile in rows  eliminated  M out  after B(p)  flag seen\n");
  for (int p = 0; p < 8; ++p) {
    printf("   %d |", p);
    for (int s = 0; s < 6; ++s) printf(" %7lld", S(8, p, s) - t0);
    printf("\n");
  }
  printf("the wave whose diagonal tile comes next (R = p + 1):  panel | after B(p)  chain done (flag raised)  updates done  stores done\n");
  for (int p = 0; p < 7; ++p) {
    printf("   %d |", p);
    for (int s = 0; s < 4; ++s) printf(" %7lld", S(p + 1, p, s) - t0);
    printf("\n");
  }
  for (int w : {0, 5}) {
    printf("tile wave %d:  panel | after B(p)  chain done  updates done  stores done\n", w);
    for (int p = 0; p < 8; ++p) {
      printf("   %d |", p);
      for (int s = 0; s < 4; ++s) printf(" %7lld", S(w, p, s) - t0);
      printf("\n");
    }
  }
#endif
  return 0;
}
