// Developer probe: the timeline of the fused adaptive-distance pass (csrc/adaptive.hip) inside workgroups 0..7 --
// shader-clock stamps per wave, tile and phase.
//   sh scripts/native/build_ada_probe.sh && scripts/native/ada_probe [n] [m] [K] [stats 0/1]
#include "../../elfi_amd/csrc/adaptive.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CKE(x) do { int e_ = (x); if (e_ != 0) { printf("%s -> %d\n", #x, e_); return 1; } } while (0)

int main(int argc, char** argv) {
  const long long n = argc > 1 ? atoll(argv[1]) : 10000000;
  const int m = argc > 2 ? atoi(argv[2]) : 64, K = argc > 3 ? atoi(argv[3]) : 3, stats = argc > 4 ? atoi(argv[4]) : 1;
  elfihip_ctx* ctx;
  CKE(elfihip_ctx_create(0, &ctx));
  double *X, *y, *W, *out, *st;
  (void)hipMalloc(&X, (size_t)n * m * 8);
  (void)hipMalloc(&y, m * 8);
  (void)hipMalloc(&W, (size_t)K * m * 8);
  (void)hipMalloc(&out, (size_t)n * K * 8);
  (void)hipMalloc(&st, (1 + 2 * m) * 8);
  CKE(elfihip_randn_dev(ctx, 1, 0, n * m, 0.0, 1.0, X));
  std::vector<double> ones((size_t)K * m, 1.0);
  (void)hipMemcpy(W, ones.data(), ones.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemset(y, 0, m * 8);
  (void)hipMemset(st, 0, (1 + 2 * m) * 8);
  for (int rep = 0; rep < 3; ++rep) CKE(elfihip_adaptive_push_dev(ctx, nullptr, X, n, m, m, y, W, K, out, stats ? st : nullptr, 0));
  CKE(elfihip_ctx_synchronize(ctx));
  std::vector<long long> s(8 * 4 * 16 * 8);
  (void)hipMemcpyFromSymbol(s.data(), HIP_SYMBOL(elfihip::g_ada_stamp), s.size() * 8);
  const char* name[7] = {"top->barrier1", "commit", "fetch issue", "barrier2", "distances(+offer)", "statistics", "tile period"};
  printf("n %lld m %d K %d stats %d: s_memtime ticks (100 MHz on gfx950?) per phase; median over a workgroup's tiles 66..78, workgroups 0..7\n", n, m, K, stats);
  for (int w = 0; w < 4; ++w) {
    printf("wave %d:", w);
    for (int ph = 0; ph < 7; ++ph) {
      std::vector<long long> v;
      for (int b = 0; b < 8; ++b)
        for (int it = 2; it < 15; ++it) {
          const long long* e = &s[((b * 4 + w) * 16 + it) * 8];
          const long long* nx = &s[((b * 4 + w) * 16 + it + 1) * 8];
          if (ph < 6) v.push_back(e[ph + 1] - e[ph]);
          else v.push_back(nx[0] - e[0]);
        }
      std::sort(v.begin(), v.end());
      printf("  %s %lld", name[ph], v[v.size() / 2]);
    }
    printf("\n");
  }
  return 0;
}
