// Multi-start minimisation of the LCB acquisition with every start advanced in lock-step.
//
// Replaces the inner optimisation of AcquisitionBase.acquire
// (elfi/methods/bo/acquisition.py:146-163), which calls minimize()
// (elfi/methods/bo/utils.py:40-111): the reference runs scipy's L-BFGS-B from each of the
// n_inits start points ONE AFTER THE OTHER, and every function / gradient evaluation inside it
// is a separate single-point GP prediction (three O(n^2) host products).
// Here all starts move together: one step = ONE batched device evaluation of value+gradient at
// the point every still-running start is waiting for (gp_predict.hip, S columns through the
// triangular products at once), followed by a few hundred flops of quasi-Newton algebra per start
// on the host.  Each start owns an L-BFGS-B state machine (lbfgsb.hpp: Cauchy point, subspace
// minimisation, More-Thuente line search, SciPy's default tolerances), so a start follows the same
// iterates as scipy.optimize.minimize(method='L-BFGS-B') would from that point -- to rounding, since
// the device evaluation and the reference's host evaluation differ in the last bits -- and the
// number of lock-steps of a call is the largest evaluation count any single start needs
// (typically 10-25 at the BASELINE shapes).
#include <chrono>
#include <memory>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gp.hpp"
#include "lbfgsb.hpp"
#include "round_pool.hpp"

namespace elfihip {

// Host threads for the quasi-Newton algebra of many starts (256 starts x 5 us per state-machine step is longer than the
// device evaluation of the round): elfihip_gp_set_acq_options, default min(16, hardware threads, CPUs the cgroup grants).
// Measured at configs[4] (n = 8192, 256 starts, 16-CPU quota): host part of one acquisition 44 ms with 1 thread, 10.6 with
// 8, 8.1 with 16; 24 threads run into the quota (17-37 ms).
static int default_host_threads() {
  static const int v = [] {
    const unsigned hc = std::thread::hardware_concurrency();
    int t = (int)std::min<unsigned>(16u, hc ? hc : 1u);
    // cgroup v2 CPU quota ("max" or "<quota> <period>")
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
      long long q = 0, p = 0;
      if (std::fscanf(f, "%lld %lld", &q, &p) == 2 && q > 0 && p > 0) t = (int)std::min<long long>(t, std::max<long long>(1, q / p));
      std::fclose(f);
    }
    return t < 1 ? 1 : t;
  }();
  return v;
}
static int host_threads(const elfihip_gp* gp) {
  const int t = gp->acq_host_threads > 0 ? gp->acq_host_threads : default_host_threads();
  return t < 1 ? 1 : (t > 64 ? 64 : t);
}

static int lcb_minimize_impl(elfihip_gp* gp, const double* starts, int64_t S, const double* lo, const double* hi,
                             double beta, int maxiter, double* x_out, double* f_out, int* iters_out,
                             int64_t* n_eval_out) {
  elfihip_ctx* ctx = gp->ctx;
  const int dim = gp->d;
  ELFIHIP_REQUIRE(ctx, S >= 1 && starts && lo && hi && x_out, "bad arguments");
  ELFIHIP_REQUIRE(ctx, maxiter >= 0 && beta >= 0.0, "bad maxiter / beta");
  for (int c = 0; c < dim; ++c) ELFIHIP_REQUIRE(ctx, lo[c] <= hi[c], "empty bound interval in dimension %d", c);
  std::vector<Lbfgsb> opt((size_t)S);
  std::vector<double> px((size_t)S * dim), pv((size_t)S), pg((size_t)S * dim);
  std::vector<int64_t> who;
  who.reserve((size_t)S);
  for (int64_t i = 0; i < S; ++i) opt[(size_t)i].init(dim, lo, hi, starts + i * dim, maxiter);
  // the worker threads outlive the call (one pool per calling thread: starting 15 threads costs about half a millisecond,
  // a tenth of a short 64-start search); between calls they sleep on the pool's condition variable
  // (a pool made before a fork does not exist in the child, and one of another width is replaced)
  static thread_local RoundPool serial_pool(1);
  static thread_local std::unique_ptr<RoundPool> wide_pool;
  const int nth = host_threads(gp);
  if (S >= 64 && (!wide_pool || !wide_pool->alive() || wide_pool->threads() != nth)) wide_pool.reset(new RoundPool(nth));
  RoundPool& pool = S >= 64 ? *wide_pool : serial_pool;
  const int trace = gp->acq_trace;
  std::vector<std::pair<int, float>> per_round;
  double t_dev = 0.0, t_host = 0.0;
  int rounds = 0;
  int64_t n_eval = 0;
  for (;;) {
    const auto t0 = std::chrono::steady_clock::now();
    who.clear();
    for (int64_t i = 0; i < S; ++i)
      if (!opt[(size_t)i].done()) who.push_back(i);
    if (who.empty()) break;
    const int64_t A = (int64_t)who.size();
    for (int64_t k = 0; k < A; ++k) {
      const double* x = opt[(size_t)who[k]].x();
      std::copy(x, x + dim, px.begin() + k * dim);
    }
    const auto t1 = std::chrono::steady_clock::now();
    ELFIHIP_TRY(predict_impl(gp, px.data(), A, 1, 1, beta, nullptr, nullptr, nullptr, nullptr, pv.data(), pg.data()));
    const auto t2 = std::chrono::steady_clock::now();
    n_eval += A;
    // every start advances its own state machine: independent, so the starts are dealt to the host threads
    pool.run(A, [&](int64_t k) { opt[(size_t)who[(size_t)k]].feed(pv[(size_t)k], &pg[(size_t)k * dim]); });
    const auto t3 = std::chrono::steady_clock::now();
    t_dev += std::chrono::duration<double>(t2 - t1).count();
    t_host += std::chrono::duration<double>(t1 - t0).count() + std::chrono::duration<double>(t3 - t2).count();
    if (trace >= 2) per_round.emplace_back((int)A, (float)(1e3 * std::chrono::duration<double>(t2 - t1).count()));
    ++rounds;
  }
  if (trace)
    std::fprintf(stderr, "[elfihip acq] S=%lld n=%lld rounds=%d evals=%lld device %.3f ms host %.3f ms (threads %d)\n",
                 (long long)S, (long long)gp->n, rounds, (long long)n_eval, 1e3 * t_dev, 1e3 * t_host,
                 S >= 64 ? nth : 1);
  if (trace >= 2) {
    std::fprintf(stderr, "[elfihip acq rounds] (active points: device ms)");
    for (auto& pr : per_round) std::fprintf(stderr, " %d:%.3f", pr.first, pr.second);
    std::fprintf(stderr, "\n");
  }
  for (int64_t i = 0; i < S; ++i) {
    const Lbfgsb& o = opt[(size_t)i];
    for (int c = 0; c < dim; ++c) x_out[i * dim + c] = o.best_x()[c];
    if (f_out) f_out[i] = o.best_f();
    if (iters_out) iters_out[i] = o.iterations();
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
  try {
    return lcb_minimize_impl(gp, starts, S, lower, upper, beta, maxiter, x_out, f_out, iters_out, n_eval_out);
  } catch (const std::bad_alloc&) {
    return fail(gp->ctx, ELFIHIP_ERR_NOMEM, "out of host memory in the multi-start search (%lld starts)", (long long)S);
  } catch (const std::exception& e) {
    return fail(gp->ctx, ELFIHIP_ERR_STATE, "multi-start search failed: %s", e.what());
  }
}

int elfihip_gp_set_acq_options(elfihip_gp* gp, int host_threads, int trace) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  ELFIHIP_REQUIRE(gp->ctx, host_threads >= 0 && host_threads <= 64 && trace >= 0, "host_threads in [0, 64], trace >= 0");
  gp->acq_host_threads = host_threads;
  gp->acq_trace = trace;
  return ELFIHIP_OK;
}

// ---- reverse-communication form for objectives assembled on the host (include/elfihip.h) ----------
struct elfihip_lbfgsb {
  int d = 0;
  std::vector<elfihip::Lbfgsb> opt;
  std::vector<int64_t> waiting;  // searches handed out by the last pending() call
};

int elfihip_lbfgsb_create(int d, int64_t S, const double* lower, const double* upper, const double* starts,
                          int maxiter, elfihip_lbfgsb** out) {
  if (!out) return ELFIHIP_ERR_ARG;
  *out = nullptr;
  if (d < 1 || S < 1 || !lower || !upper || !starts || maxiter < 0) return ELFIHIP_ERR_ARG;
  for (int c = 0; c < d; ++c)
    if (!(lower[c] <= upper[c])) return ELFIHIP_ERR_ARG;
  elfihip_lbfgsb* h = new elfihip_lbfgsb();
  h->d = d;
  h->opt.resize((size_t)S);
  for (int64_t i = 0; i < S; ++i) h->opt[(size_t)i].init(d, lower, upper, starts + i * d, maxiter);
  *out = h;
  return ELFIHIP_OK;
}

int64_t elfihip_lbfgsb_pending(elfihip_lbfgsb* h, int64_t* idx, double* x) {
  if (!h || !idx || !x) return -1;
  h->waiting.clear();
  for (size_t i = 0; i < h->opt.size(); ++i)
    if (!h->opt[i].done()) {
      const int64_t k = (int64_t)h->waiting.size();
      idx[k] = (int64_t)i;
      std::copy(h->opt[i].x(), h->opt[i].x() + h->d, x + k * h->d);
      h->waiting.push_back((int64_t)i);
    }
  return (int64_t)h->waiting.size();
}

int elfihip_lbfgsb_feed(elfihip_lbfgsb* h, int64_t n, const double* f, const double* g) {
  if (!h || !f || !g || n != (int64_t)h->waiting.size()) return ELFIHIP_ERR_ARG;
  for (int64_t k = 0; k < n; ++k) h->opt[(size_t)h->waiting[(size_t)k]].feed(f[k], g + k * h->d);
  h->waiting.clear();
  return ELFIHIP_OK;
}

int elfihip_lbfgsb_result(const elfihip_lbfgsb* h, double* x, double* f, int* iters, int* status) {
  if (!h || !x) return ELFIHIP_ERR_ARG;
  for (size_t i = 0; i < h->opt.size(); ++i) {
    const elfihip::Lbfgsb& o = h->opt[i];
    std::copy(o.best_x(), o.best_x() + h->d, x + i * h->d);
    if (f) f[i] = o.best_f();
    if (iters) iters[i] = o.iterations();
    if (status) status[i] = static_cast<int>(o.status());
  }
  return ELFIHIP_OK;
}

int elfihip_lbfgsb_free(elfihip_lbfgsb* h) {
  delete h;
  return ELFIHIP_OK;
}

}  // extern "C"
