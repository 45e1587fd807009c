// Box-constrained limited-memory BFGS (the L-BFGS-B method) as a reverse-communication state machine.
//
// The reference minimises the acquisition with scipy.optimize.minimize(method='L-BFGS-B')
// (elfi/methods/bo/utils.py:97-103, called from acquisition.py:146-163) one start after the other.
// To advance every start in lock-step -- one batched device evaluation per step, gp_acq.hip -- each
// start needs its own optimiser that can be paused at "evaluate f and g here".  This is that optimiser:
// an independent implementation of the published algorithm,
//   [BLNZ95] Byrd, Lu, Nocedal, Zhu, "A limited memory algorithm for bound constrained optimization",
//            SIAM J. Sci. Comput. 16 (1995): generalised Cauchy point (Algorithm CP), direct primal
//            subspace minimisation (section 5.1), compact representation B = theta I - W M W^T;
//   [MN11]   Morales, Nocedal, "Remark on Algorithm 778" (2011): projection of the subspace minimiser;
//   [MT94]   More, Thuente, "Line search algorithms with guaranteed sufficient decrease" (1994):
//            the safeguarded cubic/quadratic step selection behind dcsrch/dcstep,
// with SciPy's defaults: memory 10, ftol = 1e7 * eps, pgtol = 1e-5, at most 20 evaluations per line
// search, sufficient-decrease 1e-3, curvature 0.9.  Problem sizes here are tiny (n <= 256 parameters,
// 2m = 20 columns), so the small dense systems are solved by plain Gaussian elimination.
//
// Pure host C++ (no HIP): tests/test_lbfgsb.py compiles it with g++ and compares it with SciPy.
#pragma once

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <vector>

namespace elfihip {

class Lbfgsb {
 public:
  static constexpr int MEM = 10;
  static constexpr int MAXLS = 20;
  static constexpr double EPS = 2.220446049250313e-16;
  static constexpr double FACTR = 1e7;     // ftol = FACTR * EPS = 2.22e-9
  static constexpr double PGTOL = 1e-5;
  static constexpr double LS_FTOL = 1e-3, LS_GTOL = 0.9, LS_XTOL = 0.1;

  enum Status { RUNNING = 0, CONV_PGTOL = 1, CONV_FTOL = 2, MAXITER = 3, ABNORMAL = 4, MAXFUN = 5 };
  // scipy.optimize.minimize(method='L-BFGS-B') default maxfun: checked once per new iterate, after maxiter
  static constexpr int MAXFUN_EVALS = 15000;

  // Start at x0 (projected into the box).  The first evaluation is requested at x().
  void init(int n, const double* lo, const double* hi, const double* x0, int maxiter) {
    n_ = n;
    maxiter_ = maxiter;
    lo_.assign(lo, lo + n);
    hi_.assign(hi, hi + n);
    x_.resize(n);
    for (int i = 0; i < n; ++i) x_[i] = std::min(std::max(x0[i], lo[i]), hi[i]);
    xk_ = x_;
    gk_.assign(n, 0.0);
    d_.assign(n, 0.0);
    z_.assign(n, 0.0);
    S_.clear();
    Y_.clear();
    theta_ = 1.0;
    iter_ = 0;
    nfev_ = 0;
    status_ = RUNNING;
    first_ = true;
  }

  bool done() const { return status_ != RUNNING; }
  Status status() const { return status_; }
  const double* x() const { return x_.data(); }         // where f, g are wanted next
  const double* best_x() const { return xk_.data(); }   // current iterate
  double best_f() const { return fk_; }
  int iterations() const { return iter_; }
  int evaluations() const { return nfev_; }

  // f, g at x().  Afterwards either done() or a new x() awaits evaluation.
  void feed(double f, const double* g) {
    ++nfev_;
    if (first_) {
      first_ = false;
      fk_ = f;
      gk_.assign(g, g + n_);
      xk_ = x_;
      if (!std::isfinite(f)) {
        status_ = ABNORMAL;
        return;
      }
      if (proj_grad_norm(xk_, gk_) <= PGTOL) {
        status_ = CONV_PGTOL;
        return;
      }
      if (maxiter_ <= 0) {
        status_ = MAXITER;
        return;
      }
      new_iteration();
      return;
    }
    // inside a line search: x_ = xk_ + stp_ d_
    double gd = 0.0;
    for (int i = 0; i < n_; ++i) gd += g[i] * d_[i];
    const double fin = std::isfinite(f) ? f : std::numeric_limits<double>::max();
    const int ls = ls_.step(stp_, fin, std::isfinite(gd) ? gd : 0.0);
    ++ls_evals_;
    if (ls == LineSearch::AGAIN) {
      if (ls_evals_ >= MAXLS) {
        failed_line_search();
        return;
      }
      set_trial();
      return;
    }
    // accepted (sufficient decrease + curvature, or one of the search's stop warnings)
    if (!std::isfinite(f)) {
      failed_line_search();
      return;
    }
    const double fold = fk_;
    std::vector<double>& s = snew_;   // (member scratch: a step of 256 state machines is a few thousand allocations otherwise)
    std::vector<double>& y = ynew_;
    s.resize(n_);
    y.resize(n_);
    double dr = 0.0, ddum = 0.0;  // s^T y and -g_k^T s
    for (int i = 0; i < n_; ++i) {
      s[i] = x_[i] - xk_[i];
      y[i] = g[i] - gk_[i];
      dr += s[i] * y[i];
      ddum -= gk_[i] * s[i];
    }
    xk_ = x_;
    fk_ = f;
    gk_.assign(g, g + n_);
    ++iter_;
    if (proj_grad_norm(xk_, gk_) <= PGTOL) {
      status_ = CONV_PGTOL;
      return;
    }
    if (fold - f <= FACTR * EPS * std::max(std::max(std::fabs(fold), std::fabs(f)), 1.0)) {
      status_ = CONV_FTOL;
      return;
    }
    if (iter_ >= maxiter_) {
      status_ = MAXITER;
      return;
    }
    if (nfev_ > MAXFUN_EVALS) {
      status_ = MAXFUN;
      return;
    }
    if (dr > EPS * ddum) {
      double rr = 0.0;
      for (int i = 0; i < n_; ++i) rr += y[i] * y[i];
      if (static_cast<int>(S_.size()) == MEM) {
        S_.erase(S_.begin());
        Y_.erase(Y_.begin());
        for (int i = 1; i < MEM; ++i)      // the cached inner products move with their pairs
          for (int j = 1; j < MEM; ++j) {
            sy_[(i - 1) * MEM + (j - 1)] = sy_[i * MEM + j];
            ss_[(i - 1) * MEM + (j - 1)] = ss_[i * MEM + j];
          }
      }
      S_.push_back(s);
      Y_.push_back(y);
      {
        // S_i . Y_j and S_i . S_j involving the new pair (index c - 1): every entry is the same left-to-right sum
        // build_middle used to recompute for all c^2 pairs in every iteration
        const int c = cols(), a = c - 1;
        for (int j = 0; j < c; ++j) {
          double sy_aj = 0.0, sy_ja = 0.0, ss_aj = 0.0, ss_ja = 0.0;
          for (int q = 0; q < n_; ++q) {
            sy_aj += S_[a][q] * Y_[j][q];
            sy_ja += S_[j][q] * Y_[a][q];
            ss_aj += S_[a][q] * S_[j][q];
            ss_ja += S_[j][q] * S_[a][q];
          }
          sy_[a * MEM + j] = sy_aj;
          sy_[j * MEM + a] = sy_ja;
          ss_[a * MEM + j] = ss_aj;
          ss_[j * MEM + a] = ss_ja;
        }
      }
      theta_ = rr / dr;
      if (!build_middle()) reset_memory();
    }
    new_iteration();
  }

 private:
  // ---- More-Thuente line search [MT94] on phi(stp) = f(xk + stp d) ----------------------------
  struct LineSearch {
    enum { AGAIN = 0, DONE = 1 };
    double finit, ginit, gtest, width, width1, stx, fx, gx, sty, fy, gy, stmin, stmax, stpmin, stpmax;
    bool brackt;
    int stage;

    void start(double f0, double g0, double stp0, double stpmax_) {
      finit = f0;
      ginit = g0;
      gtest = LS_FTOL * g0;
      stpmin = 0.0;
      stpmax = stpmax_;
      width = stpmax - stpmin;
      width1 = 2.0 * width;
      brackt = false;
      stage = 1;
      stx = sty = 0.0;
      fx = fy = f0;
      gx = gy = g0;
      stmin = 0.0;
      stmax = stp0 + 4.0 * stp0;
    }

    // f, g = phi, phi' at stp; on AGAIN stp holds the next trial.
    int step(double& stp, double f, double g) {
      const double ftest = finit + stp * gtest;
      if (stage == 1 && f <= ftest && g >= 0.0) stage = 2;
      bool stop = false;
      if (brackt && (stp <= stmin || stp >= stmax)) stop = true;                 // rounding errors
      if (brackt && stmax - stmin <= LS_XTOL * stmax) stop = true;               // interval too small
      if (stp == stpmax && f <= ftest && g <= gtest) stop = true;                // at the upper bound
      if (stp == stpmin && (f > ftest || g >= gtest)) stop = true;               // at the lower bound
      if (f <= ftest && std::fabs(g) <= LS_GTOL * (-ginit)) stop = true;         // strong Wolfe
      if (stop) return DONE;
      if (stage == 1 && f <= fx && f > ftest) {
        // auxiliary function psi(stp) = phi(stp) - phi(0) - ftol phi'(0) stp
        double fm = f - stp * gtest, fxm = fx - stx * gtest, fym = fy - sty * gtest;
        double gm = g - gtest, gxm = gx - gtest, gym = gy - gtest;
        pick(stx, fxm, gxm, sty, fym, gym, stp, fm, gm);
        fx = fxm + stx * gtest;
        fy = fym + sty * gtest;
        gx = gxm + gtest;
        gy = gym + gtest;
      } else {
        pick(stx, fx, gx, sty, fy, gy, stp, f, g);
      }
      if (brackt) {
        if (std::fabs(sty - stx) >= 0.66 * width1) stp = stx + 0.5 * (sty - stx);
        width1 = width;
        width = std::fabs(sty - stx);
        stmin = std::min(stx, sty);
        stmax = std::max(stx, sty);
      } else {
        stmin = stp + 1.1 * (stp - stx);
        stmax = stp + 4.0 * (stp - stx);
      }
      stp = std::min(std::max(stp, stpmin), stpmax);
      if ((brackt && (stp <= stmin || stp >= stmax)) || (brackt && stmax - stmin <= LS_XTOL * stmax)) stp = stx;
      return AGAIN;
    }

    // Safeguarded step from the best point so far (stx), the other end point (sty) and the trial (stp):
    // the four cases of [MT94] section 4; extrapolation is limited to the current interval [stmin, stmax].
    void pick(double& stx_, double& fx_, double& dx_, double& sty_, double& fy_, double& dy_, double& stp, double fp,
              double dp) {
      const double sgnd = dp * (dx_ / std::fabs(dx_));
      double stpf;
      if (fp > fx_) {  // higher value: the minimum is bracketed; cubic vs quadratic through (fx, dx, fp)
        const double theta = 3.0 * (fx_ - fp) / (stp - stx_) + dx_ + dp;
        const double s = std::max(std::max(std::fabs(theta), std::fabs(dx_)), std::fabs(dp));
        double gamma = s * std::sqrt((theta / s) * (theta / s) - (dx_ / s) * (dp / s));
        if (stp < stx_) gamma = -gamma;
        const double p = (gamma - dx_) + theta, q = ((gamma - dx_) + gamma) + dp, r = p / q;
        const double stpc = stx_ + r * (stp - stx_);
        const double stpq = stx_ + ((dx_ / ((fx_ - fp) / (stp - stx_) + dx_)) / 2.0) * (stp - stx_);
        stpf = std::fabs(stpc - stx_) < std::fabs(stpq - stx_) ? stpc : stpc + (stpq - stpc) / 2.0;
        brackt = true;
      } else if (sgnd < 0.0) {  // lower value, derivatives of opposite sign: bracketed; cubic vs secant
        const double theta = 3.0 * (fx_ - fp) / (stp - stx_) + dx_ + dp;
        const double s = std::max(std::max(std::fabs(theta), std::fabs(dx_)), std::fabs(dp));
        double gamma = s * std::sqrt((theta / s) * (theta / s) - (dx_ / s) * (dp / s));
        if (stp > stx_) gamma = -gamma;
        const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + dx_, r = p / q;
        const double stpc = stp + r * (stx_ - stp);
        const double stpq = stp + (dp / (dp - dx_)) * (stx_ - stp);
        stpf = std::fabs(stpc - stp) > std::fabs(stpq - stp) ? stpc : stpq;
        brackt = true;
      } else if (std::fabs(dp) < std::fabs(dx_)) {  // lower value, same sign, derivative shrinks
        const double theta = 3.0 * (fx_ - fp) / (stp - stx_) + dx_ + dp;
        const double s = std::max(std::max(std::fabs(theta), std::fabs(dx_)), std::fabs(dp));
        double gamma = s * std::sqrt(std::max(0.0, (theta / s) * (theta / s) - (dx_ / s) * (dp / s)));
        if (stp > stx_) gamma = -gamma;
        const double p = (gamma - dp) + theta, q = (gamma + (dx_ - dp)) + gamma, r = p / q;
        double stpc;
        if (r < 0.0 && gamma != 0.0)
          stpc = stp + r * (stx_ - stp);
        else
          stpc = stp > stx_ ? stmax : stmin;
        const double stpq = stp + (dp / (dp - dx_)) * (stx_ - stp);
        if (brackt) {
          stpf = std::fabs(stpc - stp) < std::fabs(stpq - stp) ? stpc : stpq;
          if (stp > stx_)
            stpf = std::min(stp + 0.66 * (sty_ - stp), stpf);
          else
            stpf = std::max(stp + 0.66 * (sty_ - stp), stpf);
        } else {
          stpf = std::fabs(stpc - stp) > std::fabs(stpq - stp) ? stpc : stpq;
          stpf = std::max(stmin, std::min(stmax, stpf));
        }
      } else {  // lower value, same sign, derivative does not shrink
        if (brackt) {
          const double theta = 3.0 * (fp - fy_) / (sty_ - stp) + dy_ + dp;
          const double s = std::max(std::max(std::fabs(theta), std::fabs(dy_)), std::fabs(dp));
          double gamma = s * std::sqrt((theta / s) * (theta / s) - (dy_ / s) * (dp / s));
          if (stp > sty_) gamma = -gamma;
          const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + dy_, r = p / q;
          stpf = stp + r * (sty_ - stp);
        } else {
          stpf = stp > stx_ ? stmax : stmin;
        }
      }
      if (fp > fx_) {
        sty_ = stp;
        fy_ = fp;
        dy_ = dp;
      } else {
        if (sgnd < 0.0) {
          sty_ = stx_;
          fy_ = fx_;
          dy_ = dx_;
        }
        stx_ = stp;
        fx_ = fp;
        dx_ = dp;
      }
      stp = stpf;
    }
  };

  // ---- small dense helpers -----------------------------------------------------------------
  // In-place LU with partial pivoting of the k x k matrix a (row-major); false if singular.
  static bool lu_factor(std::vector<double>& a, std::vector<int>& piv, int k) {
    piv.resize(k);
    for (int c = 0; c < k; ++c) {
      int p = c;
      for (int r = c + 1; r < k; ++r)
        if (std::fabs(a[r * k + c]) > std::fabs(a[p * k + c])) p = r;
      if (!(std::fabs(a[p * k + c]) > 0.0) || !std::isfinite(a[p * k + c])) return false;
      piv[c] = p;
      if (p != c)
        for (int j = 0; j < k; ++j) std::swap(a[c * k + j], a[p * k + j]);
      for (int r = c + 1; r < k; ++r) {
        const double m = a[r * k + c] / a[c * k + c];
        a[r * k + c] = m;
        for (int j = c + 1; j < k; ++j) a[r * k + j] -= m * a[c * k + j];
      }
    }
    return true;
  }
  static void lu_solve(const std::vector<double>& a, const std::vector<int>& piv, int k, std::vector<double>& b) {
    for (int c = 0; c < k; ++c)  // all row interchanges first: the stored multipliers are in final row order
      if (piv[c] != c) std::swap(b[c], b[piv[c]]);
    for (int c = 0; c < k; ++c)
      for (int r = c + 1; r < k; ++r) b[r] -= a[r * k + c] * b[c];
    for (int c = k - 1; c >= 0; --c) {
      for (int j = c + 1; j < k; ++j) b[c] -= a[c * k + j] * b[j];
      b[c] /= a[c * k + c];
    }
  }

  // The same solve for the m columns of B (k x m, row-major) at once: every entry goes through exactly the operations
  // lu_solve applies to its column, in the same order; the loops run along the rows of B (contiguous), so the compiler
  // vectorises what k separate calls walked with stride k.
  static void lu_solve_multi(const std::vector<double>& a, const std::vector<int>& piv, int k, std::vector<double>& B, int m) {
    for (int c = 0; c < k; ++c)
      if (piv[c] != c)
        for (int j = 0; j < m; ++j) std::swap(B[c * m + j], B[piv[c] * m + j]);
    for (int c = 0; c < k; ++c)
      for (int r = c + 1; r < k; ++r) {
        const double arc = a[r * k + c];
        double* br = &B[r * m];
        const double* bc = &B[c * m];
        for (int j = 0; j < m; ++j) br[j] -= arc * bc[j];
      }
    for (int c = k - 1; c >= 0; --c) {
      double* bc = &B[c * m];
      for (int q = c + 1; q < k; ++q) {
        const double acq = a[c * k + q];
        const double* bq = &B[q * m];
        for (int j = 0; j < m; ++j) bc[j] -= acq * bq[j];
      }
      const double acc = a[c * k + c];
      for (int j = 0; j < m; ++j) bc[j] /= acc;
    }
  }

  int cols() const { return static_cast<int>(S_.size()); }
  // row i of W = [Y, theta S]  (2c entries)
  void w_row(int i, double* w) const {
    const int c = cols();
    for (int j = 0; j < c; ++j) {
      w[j] = Y_[j][i];
      w[c + j] = theta_ * S_[j][i];
    }
  }
  // v <- M v, M = [[-D, L^T], [L, theta S^T S]]^-1  ([BLNZ95] eq. 3.4)
  void apply_M(std::vector<double>& v) const { lu_solve(mid_lu_, mid_piv_, 2 * cols(), v); }

  bool build_middle() {
    const int c = cols(), k = 2 * c;
    mid_lu_.assign(static_cast<size_t>(k) * k, 0.0);
    for (int i = 0; i < c; ++i)
      for (int j = 0; j < c; ++j) {
        const double sy = sy_[i * MEM + j], ss = ss_[i * MEM + j];   // S_i . Y_j, S_i . S_j (cached in feed())
        if (i == j) mid_lu_[i * k + j] = -sy;            // -D
        if (i > j) {
          mid_lu_[(c + i) * k + j] = sy;                 // L
          mid_lu_[j * k + (c + i)] = sy;                 // L^T
        }
        mid_lu_[(c + i) * k + (c + j)] = theta_ * ss;    // theta S^T S
      }
    return lu_factor(mid_lu_, mid_piv_, k);
  }

  void reset_memory() {
    S_.clear();
    Y_.clear();
    theta_ = 1.0;
  }

  double proj_grad_norm(const std::vector<double>& x, const std::vector<double>& g) const {
    double m = 0.0;
    for (int i = 0; i < n_; ++i) {
      const double gi = g[i] < 0.0 ? std::max(x[i] - hi_[i], g[i]) : std::min(x[i] - lo_[i], g[i]);
      m = std::max(m, std::fabs(gi));
    }
    return m;
  }

  // ---- generalised Cauchy point ([BLNZ95] Algorithm CP): z_ <- x^c, c_ <- W^T (x^c - x), free_ -------
  void cauchy() {
    const int c = cols(), k = 2 * c;
    std::vector<double>&t = t_, &dd = dd_, &p = p_, &w = w_, &tmp = tmp_;
    t.resize(n_);
    dd.resize(n_);
    p.assign(k, 0.0);
    w.resize(k);
    tmp.resize(k);
    c_.assign(k, 0.0);
    z_ = xk_;
    std::vector<int>& order = order_;
    order.clear();
    const double inf = std::numeric_limits<double>::infinity();
    for (int i = 0; i < n_; ++i) {
      const double g = gk_[i];
      t[i] = g < 0.0 ? (xk_[i] - hi_[i]) / g : (g > 0.0 ? (xk_[i] - lo_[i]) / g : inf);
      dd[i] = t[i] > 0.0 ? -g : 0.0;
      if (t[i] > 0.0) order.push_back(i);
    }
    std::sort(order.begin(), order.end(), [&](int a, int b) { return t[a] < t[b] || (t[a] == t[b] && a < b); });
    double fp = 0.0;
    for (int i = 0; i < n_; ++i) {
      fp -= dd[i] * dd[i];
      if (dd[i] != 0.0 && k) {
        w_row(i, w.data());
        for (int j = 0; j < k; ++j) p[j] += w[j] * dd[i];
      }
    }
    double fpp = -theta_ * fp;
    const double fpp0 = fpp;
    if (k) {
      tmp = p;
      apply_M(tmp);
      for (int j = 0; j < k; ++j) fpp -= p[j] * tmp[j];
    }
    fpp = std::max(EPS * fpp0, fpp);
    double dtm = fpp > 0.0 ? -fp / fpp : 0.0, told = 0.0;
    size_t pos = 0;
    bool at_end = true;  // every breakpoint consumed without finding the minimiser before it
    for (; pos < order.size(); ++pos) {
      const int b = order[pos];
      if (!std::isfinite(t[b])) {
        at_end = false;
        break;
      }
      const double dt = t[b] - told;
      if (dtm < dt) {
        at_end = false;
        break;
      }
      // variable b reaches its bound: fix it and update the directional derivatives of the model
      const double gb = gk_[b];
      z_[b] = dd[b] > 0.0 ? hi_[b] : lo_[b];
      const double zb = z_[b] - xk_[b];
      for (int j = 0; j < k; ++j) c_[j] += dt * p[j];
      fp += dt * fpp + gb * gb + theta_ * gb * zb;
      fpp -= theta_ * gb * gb;
      if (k) {
        w_row(b, w.data());
        tmp = c_;
        apply_M(tmp);
        double wmc = 0.0, wmp = 0.0, wmw = 0.0;
        for (int j = 0; j < k; ++j) wmc += w[j] * tmp[j];
        tmp = p;
        apply_M(tmp);
        for (int j = 0; j < k; ++j) wmp += w[j] * tmp[j];
        tmp = w;
        apply_M(tmp);
        for (int j = 0; j < k; ++j) wmw += w[j] * tmp[j];
        fp -= gb * wmc;
        fpp -= 2.0 * gb * wmp + gb * gb * wmw;
        for (int j = 0; j < k; ++j) p[j] += gb * w[j];
      }
      fpp = std::max(EPS * fpp0, fpp);
      dd[b] = 0.0;
      dtm = fpp > 0.0 ? -fp / fpp : 0.0;
      told = t[b];
    }
    if (at_end) dtm = 0.0;  // all variables fixed
    dtm = std::max(dtm, 0.0);
    told += dtm;
    free_.clear();
    for (size_t q = pos; q < order.size(); ++q) {
      const int i = order[q];
      z_[i] = xk_[i] + told * dd[i];
      free_.push_back(i);
    }
    std::sort(free_.begin(), free_.end());
    for (int j = 0; j < k; ++j) c_[j] += dtm * p[j];
  }

  // ---- subspace minimisation over the free variables ([BLNZ95] 5.1, projection of [MN11]): z_ <- x-bar ---
  void subspace() {
    const int c = cols(), k = 2 * c, nf = static_cast<int>(free_.size());
    if (k == 0 || nf == 0) return;
    std::vector<double>&mc = mc_, &r = r_, &v = v_, &w = w_;
    mc = c_;
    r.resize(nf);
    v.assign(k, 0.0);
    w.resize(k);
    apply_M(mc);
    for (int q = 0; q < nf; ++q) {
      const int i = free_[q];
      w_row(i, w.data());
      double wm = 0.0;
      for (int j = 0; j < k; ++j) wm += w[j] * mc[j];
      r[q] = gk_[i] + theta_ * (z_[i] - xk_[i]) - wm;
      for (int j = 0; j < k; ++j) v[j] += w[j] * r[q];
    }
    apply_M(v);
    // N = I - M (W_F^T W_F) / theta ;  v <- N^-1 v
    std::vector<double>&wtw = wtw_, &nmat = nmat_, &col = col_;
    wtw.assign(static_cast<size_t>(k) * k, 0.0);
    nmat.assign(static_cast<size_t>(k) * k, 0.0);
    col.resize(k);
    for (int q = 0; q < nf; ++q) {
      w_row(free_[q], w.data());
      for (int a = 0; a < k; ++a)
        for (int b = a; b < k; ++b) wtw[a * k + b] += w[a] * w[b];   // upper triangle: w[a] w[b] = w[b] w[a] exactly
    }
    for (int a = 0; a < k; ++a)
      for (int b = 0; b < a; ++b) wtw[a * k + b] = wtw[b * k + a];
    // N = I - M (W_F^T W_F / theta): all k columns through the factorised middle matrix at once
    for (int e = 0; e < k * k; ++e) nmat[e] = wtw[e] / theta_;
    lu_solve_multi(mid_lu_, mid_piv_, k, nmat, k);
    for (int a = 0; a < k; ++a)
      for (int b = 0; b < k; ++b) nmat[a * k + b] = (a == b ? 1.0 : 0.0) - nmat[a * k + b];
    std::vector<int>& piv = piv_;
    if (!lu_factor(nmat, piv, k)) return;  // keep the Cauchy point
    lu_solve(nmat, piv, k, v);
    std::vector<double>& du = du_;
    du.resize(nf);
    for (int q = 0; q < nf; ++q) {
      w_row(free_[q], w.data());
      double wv = 0.0;
      for (int j = 0; j < k; ++j) wv += w[j] * v[j];
      du[q] = -r[q] / theta_ - wv / (theta_ * theta_);
      if (!std::isfinite(du[q])) return;
    }
    // projected subspace minimiser [MN11]; if that is not a descent direction fall back to the
    // largest feasible step along du from the Cauchy point [BLNZ95]
    std::vector<double>& zc = zc_;
    zc = z_;
    bool clipped = false;
    for (int q = 0; q < nf; ++q) {
      const int i = free_[q];
      const double xi = zc[i] + du[q];
      z_[i] = std::min(std::max(xi, lo_[i]), hi_[i]);
      clipped |= (z_[i] != xi);
    }
    if (clipped) {
      double dd_p = 0.0;
      for (int i = 0; i < n_; ++i) dd_p += (z_[i] - xk_[i]) * gk_[i];
      if (dd_p > 0.0) {
        z_ = zc;
        double alpha = 1.0;
        int ibd = -1;
        for (int q = 0; q < nf; ++q) {
          const int i = free_[q];
          const double dk = du[q];
          double room;
          if (dk < 0.0) {
            room = lo_[i] - zc[i];
            if (room >= 0.0) {
              alpha = 0.0;
              ibd = q;
            } else if (dk * alpha < room) {
              alpha = room / dk;
              ibd = q;
            }
          } else if (dk > 0.0) {
            room = hi_[i] - zc[i];
            if (room <= 0.0) {
              alpha = 0.0;
              ibd = q;
            } else if (dk * alpha > room) {
              alpha = room / dk;
              ibd = q;
            }
          }
        }
        for (int q = 0; q < nf; ++q) z_[free_[q]] = zc[free_[q]] + alpha * du[q];
        if (alpha < 1.0 && ibd >= 0) z_[free_[ibd]] = du[ibd] > 0.0 ? hi_[free_[ibd]] : lo_[free_[ibd]];
      }
    }
  }

  // ---- one outer iteration up to the first trial point of its line search ------------------------
  void new_iteration() {
    for (;;) {
      cauchy();
      subspace();
      gd0_ = 0.0;
      for (int i = 0; i < n_; ++i) {
        d_[i] = z_[i] - xk_[i];
        gd0_ += gk_[i] * d_[i];
      }
      if (gd0_ < 0.0 && std::isfinite(gd0_)) break;
      if (cols() == 0) {  // not even the projected-gradient step descends: stop at the current iterate
        status_ = ABNORMAL;
        return;
      }
      reset_memory();  // discard the quasi-Newton model and retry from the steepest-descent model
    }
    // largest step that stays in the box (the box is bounded in every coordinate, so the first trial is 1)
    double stpmx = 1.0;
    if (iter_ > 0) {
      stpmx = 1e10;
      for (int i = 0; i < n_; ++i) {
        const double a1 = d_[i];
        if (a1 < 0.0) {
          const double a2 = lo_[i] - xk_[i];
          if (a2 >= 0.0)
            stpmx = 0.0;
          else if (a1 * stpmx < a2)
            stpmx = a2 / a1;
        } else if (a1 > 0.0) {
          const double a2 = hi_[i] - xk_[i];
          if (a2 <= 0.0)
            stpmx = 0.0;
          else if (a1 * stpmx > a2)
            stpmx = a2 / a1;
        }
      }
    }
    stp_ = std::min(1.0, stpmx);
    ls_.start(fk_, gd0_, stp_, stpmx);
    ls_evals_ = 0;
    set_trial();
  }

  void set_trial() {
    if (stp_ == 1.0) {
      x_ = z_;
    } else {
      for (int i = 0; i < n_; ++i) x_[i] = std::min(std::max(xk_[i] + stp_ * d_[i], lo_[i]), hi_[i]);
    }
  }

  void failed_line_search() {
    x_ = xk_;
    if (cols() == 0) {
      status_ = ABNORMAL;
      return;
    }
    reset_memory();
    new_iteration();
  }

  int n_ = 0, maxiter_ = 0, iter_ = 0, nfev_ = 0, ls_evals_ = 0;
  bool first_ = true;
  Status status_ = RUNNING;
  double fk_ = 0.0, theta_ = 1.0, gd0_ = 0.0, stp_ = 1.0;
  std::vector<double> lo_, hi_, x_, xk_, gk_, d_, z_, c_;
  std::vector<std::vector<double>> S_, Y_;
  std::vector<double> mid_lu_;
  std::vector<int> mid_piv_, free_;
  double sy_[MEM * MEM] = {}, ss_[MEM * MEM] = {};   // S_i . Y_j, S_i . S_j of the stored pairs
  // scratch of feed / cauchy / subspace (sized on use, kept between calls)
  std::vector<double> snew_, ynew_, t_, dd_, p_, w_, tmp_, mc_, r_, v_, wtw_, nmat_, col_, du_, zc_;
  std::vector<int> order_, piv_;
  LineSearch ls_;
};

}  // namespace elfihip
