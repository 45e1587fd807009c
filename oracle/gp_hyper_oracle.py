"""TEST INFRASTRUCTURE ONLY -- CPU restatement of GPyRegression.optimize() (hyper-parameter MAP fit).

PARITY UNPINNED: the reference delegates to GPy/paramz (elfi/methods/bo/gpy_regression.py:317-323
-> GPy model.optimize('scg', max_iters=50)); neither library is vendored under /root/reference
or installed here, and no reference test or fixture fixes an optimum, an objective value or a
trajectory (the only numbers are a 2017 notebook print-out,
docs/usage/BOLFI.rst:144-153).  This file restates the published algorithm independently of
elfi_amd/hyperopt.py (different code structure, NumPy/SciPy LAPACK on the host) so that the GPU
path can be checked against a second implementation of the same mathematics:

  objective(phi) = -( log Z(theta) + sum_{i in priored} [ln Gamma(theta_i | a_i, b_i)
                     + ln(1 - exp(-theta_i))] ),      theta = softplus(phi)      [GPy-upstream]
  optimiser: scaled conjugate gradients (M. Moller, Neural Networks 6, 1993) with the constants
  of paramz's SCG.

Only tests/ and bench.py's cpu_baseline leg import this.
"""
import numpy as np
from scipy.special import gammaln

import gp_oracle as G

ORDER = ('var', 'ls', 'bias', 'noise')


def softplus(phi):
    phi = np.asarray(phi, dtype=float)
    out = phi.copy()
    small = phi <= 36.0
    out[small] = np.log1p(np.exp(phi[small]))
    return out


def softplus_inv(theta):
    theta = np.asarray(theta, dtype=float)
    out = theta.copy()
    small = theta <= 36.0
    out[small] = np.log(np.expm1(theta[small]))
    return out


class MapObjective:
    """Negative log posterior of the hyper-parameters in the unconstrained coordinates."""

    def __init__(self, X, Y, priors):
        self.X, self.Y = np.asarray(X, float), np.asarray(Y, float).reshape(-1, 1)
        self.priors = priors  # name -> (a, b), Gamma shape / rate
        self.n_fits = 0
        self._key, self._post = None, None

    def posterior(self, phi):
        key = tuple(np.asarray(phi, float).tolist())
        if key != self._key:
            th = softplus(phi)
            self._post = G.Posterior(self.X, self.Y, th[0], th[1], th[2], th[3])
            self._key = key
            self.n_fits += 1
        return self._post

    def value(self, phi):
        th = softplus(phi)
        try:
            total = self.posterior(phi).log_marginal
        except np.linalg.LinAlgError:
            return np.inf
        for i, k in enumerate(ORDER):
            if k in self.priors:
                a, b = self.priors[k]
                total += a * np.log(b) - gammaln(a) + (a - 1) * np.log(th[i]) - b * th[i]
                total += np.log1p(-np.exp(-th[i]))      # ln |d theta / d phi|
        return -total

    def gradient(self, phi):
        th = softplus(phi)
        g = self.posterior(phi).log_marginal_grad().copy()
        for i, k in enumerate(ORDER):
            if k in self.priors:
                a, b = self.priors[k]
                g[i] += (a - 1) / th[i] - b + 1.0 / np.expm1(th[i])
        return -g * (1.0 - np.exp(-th))


def scaled_conjugate_gradient(value, gradient, x0, iterations=50, xtol=1e-6, ftol=1e-6, gtol=1e-5):
    """Moller's SCG.  Returns (x, list of objective values, status)."""
    x = np.array(x0, float)
    f_old = value(x)
    g_new = gradient(x)
    g_old = g_new.copy()
    direction = -g_new
    lam, lam_min, lam_max = 1.0, 1e-15, 1e15   # the scale parameter (paramz calls it beta)
    ok, n_ok = True, 0
    trace = [f_old]
    status = 'maxiter exceeded'
    for _ in range(iterations):
        if ok:   # second-order information along `direction` by a one-sided gradient difference
            slope = float(direction @ g_new)
            if slope >= 0:
                direction = -g_new
                slope = float(direction @ g_new)
            dd = float(direction @ direction)
            if not (dd > 0 and np.isfinite(dd)):
                status = 'zero search direction'
                break
            eps = 1e-7 / np.sqrt(dd)
            curvature = float(direction @ (gradient(x + eps * direction) - g_new)) / eps
        denom = curvature + lam * dd
        if denom <= 0:      # make the model Hessian positive definite
            denom = lam * dd
            lam = lam - curvature / dd
        step = -slope / denom
        x_try = x + step * direction
        f_try = value(x_try)
        ratio = 2.0 * (f_try - f_old) / (step * slope)   # actual / predicted reduction
        ok = ratio >= 0
        if ok:
            n_ok += 1
            x = x_try
        trace.append(f_try if ok else f_old)
        if ok:
            if abs(f_try - f_old) < ftol:
                status = 'converged - relative reduction in objective'
                break
            if np.max(np.abs(step * direction)) < xtol:
                status = 'converged - relative stepsize'
                break
            g_old, g_new = g_new, gradient(x)
            f_old = f_try
            if float(g_new @ g_new) <= gtol:
                status = 'converged - relative reduction in gradient'
                break
        if ratio < 0.25:
            lam = min(4.0 * lam, lam_max)
        if ratio > 0.75:
            lam = max(0.25 * lam, lam_min)
        if n_ok == x.size:          # restart with steepest descent every dim successes
            direction = -g_new
            lam, n_ok = 1.0, 0
        elif ok:
            direction = (float((g_old - g_new) @ g_new) / slope) * direction - g_new
    return x, trace, status


def optimize(X, Y, hyper0, priors, max_iters=50):
    """MAP fit from `hyper0` (dict var/ls/bias/noise); returns (hyper dict, info)."""
    obj = MapObjective(X, Y, priors)
    phi0 = softplus_inv(np.array([hyper0[k] for k in ORDER]))
    phi, trace, status = scaled_conjugate_gradient(obj.value, obj.gradient, phi0, iterations=max_iters)
    th = softplus(phi)
    return dict(zip(ORDER, th.tolist())), dict(objective=trace, status=status, n_fits=obj.n_fits)
