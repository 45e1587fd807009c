// Test-only C wrapper around elfi_amd/csrc/lbfgsb.hpp (pure host code), built with g++ by tests/test_lbfgsb.py.
#include "../../elfi_amd/csrc/lbfgsb.hpp"

using elfihip::Lbfgsb;

extern "C" {
void* lb_new(int n, const double* lo, const double* hi, const double* x0, int maxiter) {
  Lbfgsb* s = new Lbfgsb();
  s->init(n, lo, hi, x0, maxiter);
  return s;
}
int lb_done(void* h) { return static_cast<Lbfgsb*>(h)->done() ? 1 : 0; }
const double* lb_x(void* h) { return static_cast<Lbfgsb*>(h)->x(); }
void lb_feed(void* h, double f, const double* g) { static_cast<Lbfgsb*>(h)->feed(f, g); }
void lb_result(void* h, double* x, double* f, int* it, int* nfev, int* status, int n) {
  Lbfgsb* s = static_cast<Lbfgsb*>(h);
  for (int i = 0; i < n; ++i) x[i] = s->best_x()[i];
  *f = s->best_f();
  *it = s->iterations();
  *nfev = s->evaluations();
  *status = static_cast<int>(s->status());
}
void lb_free(void* h) { delete static_cast<Lbfgsb*>(h); }
}


// ---- the host-thread pool of the multi-start search (elfi_amd/csrc/round_pool.hpp)
#include "../../elfi_amd/csrc/round_pool.hpp"
extern "C" long long pool_rounds_check(int nthreads, int rounds, long long n) {
  // every round: item i adds i + round to slot i exactly once; returns the number of slots with a wrong total
  elfihip::RoundPool pool(nthreads);
  std::vector<long long> slot((size_t)n, 0);
  long long expect_extra = 0;
  for (int r = 0; r < rounds; ++r) {
    const long long m = (r % 3 == 0) ? n : n / 2 + r % 7;     // changing sizes, some below the serial cut-off
    pool.run(m, [&](int64_t i) { slot[(size_t)i] += i + r; });
    (void)expect_extra;
    if (r % 40 == 39) std::this_thread::sleep_for(std::chrono::milliseconds(3));   // workers give up polling and block
  }
  long long bad = 0;
  for (long long i = 0; i < n; ++i) {
    long long want = 0;
    for (int r = 0; r < rounds; ++r) {
      const long long m = (r % 3 == 0) ? n : n / 2 + r % 7;
      if (i < m) want += i + r;
    }
    bad += slot[(size_t)i] != want;
  }
  return bad;
}
