// Developer probe: reject_merge_kernel alone -- event time for c candidates against an empty and a full state of k = 1000
// (the two habitats: the ~2000 candidates of an SMC round's first batch; a few hundred every 8th step of the distance bench).
//   sh scripts/native/build_merge_probe.sh && scripts/native/merge_probe
// The kernel under test is compiled from reject.hip itself (included below).
#include "../../elfi_amd/csrc/reject.hip"
#include <cstdio>
#include <random>
#include <vector>
using namespace elfihip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const int k = 1000, cap = 1 << 16;
  double *bv, *thr, *cvv;
  long long *br, *crr;
  unsigned int *count, *status;
  CK(hipMalloc(&bv, 2048 * 8)); CK(hipMalloc(&br, 2048 * 8)); CK(hipMalloc(&thr, 8));
  CK(hipMalloc(&cvv, cap * 8)); CK(hipMalloc(&crr, cap * 8)); CK(hipMalloc(&count, 4)); CK(hipMalloc(&status, 4));
  CK(hipMemset(status, 0, 4));
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  std::vector<double> hc(cap), hs(k);
  std::vector<long long> hr(cap), hsr(k);
  for (int i = 0; i < cap; ++i) hc[i] = U(g), hr[i] = 100000 + i;
  CK(hipMemcpy(cvv, hc.data(), cap * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(crr, hr.data(), cap * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double inf = __builtin_inf();
  for (int full = 0; full < 2; ++full)
    for (int c : {16, 100, 300, 1000, 1024, 1025, 1700, 2000, 2048, 3000, 4000, 8000}) {
      float best = 1e9f;
      std::vector<double> got(k);
      for (int rep = 0; rep < 12; ++rep) {
        for (int i = 0; i < k; ++i) hs[i] = full ? 0.5 * (i + 0.5) / k : inf, hsr[i] = full ? i : 0x7fffffffffffffffll;
        CK(hipMemcpy(bv, hs.data(), k * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(br, hsr.data(), k * 8, hipMemcpyHostToDevice));
        unsigned int cc = (unsigned int)c;
        CK(hipMemcpy(count, &cc, 4, hipMemcpyHostToDevice));
        RejArgs S{bv, br, thr, cvv, crr, count, status, (unsigned int)cap, k, -1, 0, nullptr, 0u, ~0u};
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(reject_merge_kernel, dim3(1), dim3(1024), REJ_MERGE_LDS, 0, S);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        CK(hipMemcpy(got.data(), bv, k * 8, hipMemcpyDeviceToHost));
      }
      // check against the host
      std::vector<double> all(hc.begin(), hc.begin() + c);
      if (full) for (int i = 0; i < k; ++i) all.push_back(0.5 * (i + 0.5) / k);
      std::sort(all.begin(), all.end());
      int bad = 0;
      for (int i = 0; i < k; ++i) {
        const double want = i < (int)all.size() ? all[i] : inf;
        if (got[i] != want) ++bad;
      }
      printf("state %s  c = %5d: %7.2f us  (%d mismatches)\n", full ? "full " : "empty", c, best * 1e3, bad);
    }
  return 0;
}
