// Developer probe: the timeline of ONE fused step launch inside a rebuild (shader-clock stamps per workgroup).
//   sh scripts/native/build_step_timeline_probe.sh && scripts/native/step_timeline_probe [n] [d] [step]
#include "../../elfi_amd/csrc/gp_fit.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#define CKE(x) do { int e_ = (x); if (e_ != 0) { printf("%s -> %d\n", #x, e_); return 1; } } while (0)

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 4096, d = argc > 2 ? atoi(argv[2]) : 10;
  int step = argc > 3 ? atoi(argv[3]) : 16;
  elfihip_ctx* ctx; elfihip_gp* gp;
  CKE(elfihip_ctx_create(0, &ctx));
  CKE(elfihip_gp_create(ctx, d, n, &gp));
  std::mt19937_64 rng(1); std::uniform_real_distribution<double> U(-2, 2);
  std::vector<double> X((size_t)n * d), y(n);
  for (auto& v : X) v = U(rng);
  for (int i = 0; i < n; ++i) y[i] = std::sin(X[(size_t)i * d]) + 0.1 * U(rng);
  CKE(elfihip_gp_set_hyper(gp, 1.0, 1.5, 0.0, 0.1));
  CKE(elfihip_gp_set_data(gp, X.data(), y.data(), n));
  CKE(elfihip_gp_set_schedule(gp, argc > 4 ? atoi(argv[4]) : 3, 0));   // 3: the chained step launch (the stamps' subject)
  (void)hipMemcpyToSymbol(HIP_SYMBOL(elfihip::g_step_stamp_k), &step, sizeof(int));
  double lml;
  for (int rep = 0; rep < 5; ++rep) CKE(elfihip_gp_factorize(gp, &lml));
  std::vector<long long> st(1024 * 8);
  (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(elfihip::g_step_stamp), st.size() * 8);
  const int nwg = 256;
  // the clocks of different XCDs are not aligned: everything relative to the workgroup's own first stamp
  auto col = [&](int slot, int lo, int hi) {
    std::vector<long long> v;
    for (int b = lo; b < hi; ++b) if (st[b * 8 + slot] && st[b * 8]) v.push_back(st[b * 8 + slot] - st[b * 8]);
    std::sort(v.begin(), v.end());
    if (v.empty()) { printf("       -       -       -   (0)"); return; }
    printf(" %7lld %7lld %7lld   (%zu)", v.front(), v[v.size() / 2], v.back(), v.size());
  };
  printf("n %d d %d step %d (log marginal %.6f); ticks of s_memtime since the workgroup's own start: min median max (workgroups)\n", n, d, step, lml);
  printf("diagonal-block workgroup: tile arrived %lld  block done %lld\n", st[1] - st[0], st[2] - st[0]);
  printf("lead workgroups 1-8:\n  strip piece solved, arrived"); col(1, 1, 9);
  printf("\n  strip complete seen        "); col(2, 1, 9); printf("\n  tile pieces done, arrived  "); col(3, 1, 9);
  printf("\n  waits for the panel        "); col(4, 1, 9); printf("\n  panel seen                 "); col(5, 1, 9); printf("\n  end                        "); col(6, 1, 9);
  printf("\nother workgroups:\n  panel pieces solved, arrived"); col(1, 9, nwg);
  printf("\n  waits for the panel        "); col(4, 9, nwg); printf("\n  panel seen                 "); col(5, 9, nwg); printf("\n  end                        "); col(6, 9, nwg);
  printf("\n");
  elfihip_gp_free(gp); elfihip_ctx_destroy(ctx);
  return 0;
}
