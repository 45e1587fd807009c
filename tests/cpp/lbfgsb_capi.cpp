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
