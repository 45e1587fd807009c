// Developer probe: where the resident top-k selection (sel_resident_lds_kernel, the slice's keys in LDS) spends its time --
// wall-clock stamps (100 MHz) per workgroup and phase, for the k = 1000 smallest of 10^6 uniform doubles and for the
// prefix selection of configs[3]'s round (63 smallest of 312 500 values, stride 3).
//   sh scripts/native/build_sel_probe.sh && scripts/native/sel_probe
// The kernel under test is compiled from topk.hip itself (included below) with the stamps switched on.
#include "../../elfi_amd/csrc/topk.hip"
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
using namespace elfihip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static int run(int64_t n, int64_t stride, int64_t k) {
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  std::vector<double> h((size_t)n * stride);
  for (auto& x : h) x = U(g);
  double *d, *vals;
  int64_t* idx;
  SelWork* w;
  CK(hipMalloc(&d, h.size() * 8)); CK(hipMalloc(&vals, k * 8)); CK(hipMalloc(&idx, k * 8)); CK(hipMalloc(&w, sizeof(SelWork)));
  CK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  const int grid = (int)((n + SEL_RU * SEL_NT - 1) / (SEL_RU * SEL_NT));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(sel_resident_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEL_SLICE_LDS));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f, best_k = 1e9f;
  std::vector<unsigned long long> st(512 * 32), keep;
  for (int rep = 0; rep < 20; ++rep) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    CK(hipMemsetAsync(w, 0, sizeof(SelWork), 0));
    hipLaunchKernelGGL(sel_resident_lds_kernel, dim3(grid), dim3(SEL_NT), SEL_SLICE_LDS, 0, d + (stride - 1), n, stride, k, w, vals, idx);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_sel_stamp), st.size() * 8));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < grid; ++b) t0 = std::min(t0, st[b * 32]);
    for (int b = 0; b < grid; ++b)
      for (int s = 0; s < 32; ++s) t1 = std::max(t1, st[b * 32 + s]);
    const float kus = (t1 - t0) / 100.0f;
    if (ms < best) best = ms;
    if (kus < best_k) best_k = kus, keep = st;
    static unsigned long long zero[512 * 32];
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_sel_stamp), zero, sizeof zero));
  }
  std::vector<double> got(k);
  CK(hipMemcpy(got.data(), vals, k * 8, hipMemcpyDeviceToHost));
  std::vector<double> col(n);
  for (int64_t i = 0; i < n; ++i) col[i] = h[i * stride + stride - 1];
  std::sort(col.begin(), col.end());
  std::sort(got.begin(), got.end());
  int bad = 0;
  for (int64_t i = 0; i < k; ++i) bad += got[i] != col[i];
  printf("n = %lld (stride %lld), k = %lld: %d workgroups; memset + launch by events %.1f us, first stamp -> last stamp %.1f us; %d mismatches\n",
         (long long)n, (long long)stride, (long long)k, grid, best * 1e3, best_k, bad);
  unsigned long long t0 = ~0ull;
  for (int b = 0; b < grid; ++b) t0 = std::min(t0, keep[b * 32]);
  printf("  slot: earliest / median / latest workgroup, us after the first workgroup's start\n");
  for (int s = 0; s < 32; ++s) {
    std::vector<double> v;
    for (int b = 0; b < grid; ++b)
      if (keep[b * 32 + s]) v.push_back((keep[b * 32 + s] - t0) / 100.0);
    if (v.empty()) break;
    std::sort(v.begin(), v.end());
    printf("  %2d: %6.2f %6.2f %6.2f\n", s, v.front(), v[v.size() / 2], v.back());
  }
  CK(hipFree(d)); CK(hipFree(vals)); CK(hipFree(idx)); CK(hipFree(w));
  return 0;
}

int main() {
  if (run(1000000, 1, 1000)) return 1;
  if (run(312500, 3, 63)) return 1;
  if (run(39062, 3, 63)) return 1;
  return 0;
}
