// Multi-start minimisation of the LCB acquisition with every start advanced in lock-step.
//
// Replaces the inner optimisation of AcquisitionBase.acquire
// (elfi/methods/bo/acquisition.py:146-163), which calls minimize()
// (elfi/methods/bo/utils.py:40-111): the reference runs scipy's L-BFGS-B from each of the
// n_inits start points ONE AFTER THE OTHER, and every function / gradient evaluation inside it
// is a separate single-point GP prediction (three O(n^2) host products).
// Here all starts move together: one step = ONE batched device evaluation of value+gradient at
// the current trial point of every still-active start (gp_predict.hip, S columns through the
// triangular products at once), followed by a few dozen flops of quasi-Newton algebra per start
// on the host.  Per start the method is a bound-projected limited-memory BFGS:
//   * free set: coordinates not pinned at a bound by the sign of the gradient;
//   * direction: L-BFGS two-loop recursion (memory 10, as scipy's default) restricted to the
//     free set, steepest descent when no curvature pair is usable;
//   * step: projection arc x(a) = clip(x + a d) with Armijo backtracking (c1 = 1e-4, at most
//     20 trials, first step length min(1, 1/|d|) as L-BFGS-B does);
//   * stops: projected-gradient sup-norm <= 1e-5, relative decrease <= 2.22e-9 (scipy's
//     defaults pgtol / ftol), maxiter iterations, or a failed line search.
// It is not a transcription of L-BFGS-B (no Cauchy point / subspace step), so iterates differ
// from scipy's; both converge to stationary points of the same box-constrained problem and the
// parity tests compare the optima (tests/test_acquisition_gpu.py).
#include <algorithm>
#include <cmath>
#include <vector>

#include "gp.hpp"

namespace elfihip {

namespace {

constexpr int MEM = 10;
constexpr double PGTOL = 1e-5;
constexpr double FTOL = 2.220446049250313e-09;
constexpr double C1 = 1e-4;
constexpr int MAXLS = 20;

struct Start {
  std::vector<double> x, g, d, xt, S, Y;  // S, Y: MEM x dim ring (chronological, newest last)
  double f = 0.0, alpha = 1.0;
  int npairs = 0, iter = 0, nls = 0;
  bool done = false;
};

inline double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Direction for start s at its current (x, g); returns false when the projected gradient is small.
bool new_direction(Start& s, int dim, const double* lo, const double* hi) {
  std::vector<char> free_(dim);
  double pg = 0.0;
  for (int c = 0; c < dim; ++c) {
    const double step = s.x[c] - clipd(s.x[c] - s.g[c], lo[c], hi[c]);
    pg = std::max(pg, std::fabs(step));
    free_[c] = !((s.x[c] <= lo[c] && s.g[c] > 0.0) || (s.x[c] >= hi[c] && s.g[c] < 0.0));
  }
  if (pg <= PGTOL) return false;
  std::vector<double> q(dim);
  for (int c = 0; c < dim; ++c) q[c] = free_[c] ? s.g[c] : 0.0;
  auto mdot = [&](const double* a, const double* b) {
    double t = 0.0;
    for (int c = 0; c < dim; ++c)
      if (free_[c]) t += a[c] * b[c];
    return t;
  };
  double a[MEM], rho[MEM];
  double gamma = 0.0;
  for (int i = s.npairs - 1; i >= 0; --i) {
    const double* si = &s.S[(size_t)i * dim];
    const double* yi = &s.Y[(size_t)i * dim];
    const double sy = mdot(si, yi), yy = mdot(yi, yi);
    rho[i] = (sy > 1e-10 * yy && yy > 0.0) ? 1.0 / sy : 0.0;
    if (gamma == 0.0 && rho[i] > 0.0) gamma = sy / yy;  // newest usable pair sets the scaling
    a[i] = rho[i] * mdot(si, q.data());
    for (int c = 0; c < dim; ++c)
      if (free_[c]) q[c] -= a[i] * yi[c];
  }
  const bool have_curv = gamma > 0.0;
  if (have_curv)
    for (int c = 0; c < dim; ++c) q[c] *= gamma;
  for (int i = 0; i < s.npairs; ++i) {
    const double* si = &s.S[(size_t)i * dim];
    const double* yi = &s.Y[(size_t)i * dim];
    const double b = rho[i] * mdot(yi, q.data());
    for (int c = 0; c < dim; ++c)
      if (free_[c]) q[c] += si[c] * (a[i] - b);
  }
  double gd = 0.0, dn = 0.0;
  for (int c = 0; c < dim; ++c) {
    s.d[c] = free_[c] ? -q[c] : 0.0;
    gd += s.g[c] * s.d[c];
    dn += s.d[c] * s.d[c];
  }
  bool steepest = !have_curv;
  if (!(gd < 0.0) || !std::isfinite(gd)) {  // not a descent direction: drop the memory
    s.npairs = 0;
    steepest = true;
    dn = 0.0;
    for (int c = 0; c < dim; ++c) {
      s.d[c] = free_[c] ? -s.g[c] : 0.0;
      dn += s.d[c] * s.d[c];
    }
  }
  s.alpha = steepest ? std::min(1.0, 1.0 / std::sqrt(dn)) : 1.0;
  s.nls = 0;
  return true;
}

inline void make_trial(Start& s, int dim, const double* lo, const double* hi) {
  for (int c = 0; c < dim; ++c) s.xt[c] = clipd(s.x[c] + s.alpha * s.d[c], lo[c], hi[c]);
}

}  // namespace

static int lcb_minimize_impl(elfihip_gp* gp, const double* starts, int64_t S, const double* lo, const double* hi,
                             double beta, int maxiter, double* x_out, double* f_out, int* iters_out,
                             int64_t* n_eval_out) {
  elfihip_ctx* ctx = gp->ctx;
  const int dim = gp->d;
  ELFIHIP_REQUIRE(ctx, S >= 1 && starts && lo && hi && x_out, "bad arguments");
  ELFIHIP_REQUIRE(ctx, maxiter >= 0 && beta >= 0.0, "bad maxiter / beta");
  for (int c = 0; c < dim; ++c) ELFIHIP_REQUIRE(ctx, lo[c] <= hi[c], "empty bound interval in dimension %d", c);
  std::vector<Start> st((size_t)S);
  std::vector<double> px((size_t)S * dim), pv((size_t)S), pg((size_t)S * dim);
  std::vector<int64_t> who;
  who.reserve((size_t)S);
  for (int64_t i = 0; i < S; ++i) {
    Start& s = st[(size_t)i];
    s.x.resize(dim);
    s.g.resize(dim);
    s.d.assign(dim, 0.0);
    s.xt.resize(dim);
    s.S.assign((size_t)MEM * dim, 0.0);
    s.Y.assign((size_t)MEM * dim, 0.0);
    for (int c = 0; c < dim; ++c) px[(size_t)i * dim + c] = s.x[c] = clipd(starts[i * dim + c], lo[c], hi[c]);
  }
  int64_t n_eval = 0;
  ELFIHIP_TRY(predict_impl(gp, px.data(), S, 1, 1, beta, nullptr, nullptr, nullptr, nullptr, pv.data(), pg.data()));
  n_eval += S;
  for (int64_t i = 0; i < S; ++i) {
    Start& s = st[(size_t)i];
    s.f = pv[(size_t)i];
    std::copy(pg.begin() + i * dim, pg.begin() + (i + 1) * dim, s.g.begin());
    if (!std::isfinite(s.f) || maxiter == 0 || !new_direction(s, dim, lo, hi))
      s.done = true;
    else
      make_trial(s, dim, lo, hi);
  }
  for (;;) {
    who.clear();
    for (int64_t i = 0; i < S; ++i)
      if (!st[(size_t)i].done) who.push_back(i);
    if (who.empty()) break;
    const int64_t A = (int64_t)who.size();
    for (int64_t k = 0; k < A; ++k)
      std::copy(st[(size_t)who[k]].xt.begin(), st[(size_t)who[k]].xt.end(), px.begin() + k * dim);
    ELFIHIP_TRY(predict_impl(gp, px.data(), A, 1, 1, beta, nullptr, nullptr, nullptr, nullptr, pv.data(), pg.data()));
    n_eval += A;
    for (int64_t k = 0; k < A; ++k) {
      Start& s = st[(size_t)who[k]];
      const double ft = pv[(size_t)k];
      const double* gt = &pg[(size_t)k * dim];
      double gdx = 0.0;
      for (int c = 0; c < dim; ++c) gdx += s.g[c] * (s.xt[c] - s.x[c]);
      if (std::isfinite(ft) && ft <= s.f + C1 * gdx) {
        // accept: curvature pair, convergence tests, next direction
        if (s.npairs == MEM) {
          std::copy(s.S.begin() + dim, s.S.end(), s.S.begin());
          std::copy(s.Y.begin() + dim, s.Y.end(), s.Y.begin());
          --s.npairs;
        }
        double* sn = &s.S[(size_t)s.npairs * dim];
        double* yn = &s.Y[(size_t)s.npairs * dim];
        double sy = 0.0, yy = 0.0;
        for (int c = 0; c < dim; ++c) {
          sn[c] = s.xt[c] - s.x[c];
          yn[c] = gt[c] - s.g[c];
          sy += sn[c] * yn[c];
          yy += yn[c] * yn[c];
        }
        if (sy > 2.2e-16 * yy && yy > 0.0) ++s.npairs;
        const double fold = s.f;
        s.x = s.xt;
        s.f = ft;
        std::copy(gt, gt + dim, s.g.begin());
        ++s.iter;
        const double denom = std::max(std::max(std::fabs(fold), std::fabs(ft)), 1.0);
        if ((fold - ft) <= FTOL * denom || s.iter >= maxiter || !new_direction(s, dim, lo, hi))
          s.done = true;
        else
          make_trial(s, dim, lo, hi);
      } else {
        if (++s.nls >= MAXLS) {
          s.done = true;  // line search failed: keep the best point so far
        } else {
          s.alpha *= 0.5;
          make_trial(s, dim, lo, hi);
          bool moved = false;
          for (int c = 0; c < dim; ++c) moved |= (s.xt[c] != s.x[c]);
          if (!moved) s.done = true;
        }
      }
    }
  }
  for (int64_t i = 0; i < S; ++i) {
    const Start& s = st[(size_t)i];
    for (int c = 0; c < dim; ++c) x_out[i * dim + c] = s.x[c];
    if (f_out) f_out[i] = s.f;
    if (iters_out) iters_out[i] = s.iter;
  }
  if (n_eval_out) *n_eval_out = n_eval;
  return ELFIHIP_OK;
}

}  // namespace elfihip

using namespace elfihip;

extern "C" {

int elfihip_gp_lcb_minimize(elfihip_gp* gp, const double* starts, int64_t S, const double* lower,
                            const double* upper, double beta, int maxiter, double* x_out, double* f_out,
                            int* iters_out, int64_t* n_eval_out) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  if (!gp->factored)
    return fail(gp->ctx, ELFIHIP_ERR_STATE, "GP is not factorised (call elfihip_gp_factorize first)");
  DeviceGuard g(gp->ctx->device);
  return lcb_minimize_impl(gp, starts, S, lower, upper, beta, maxiter, x_out, f_out, iters_out, n_eval_out);
}

}  // extern "C"
