"""Hyper-parameter optimisation of the GP surrogate: what GPyRegression.optimize() does.

Reference: elfi/methods/bo/gpy_regression.py:317-323 calls
`self._gp.optimize(self.optimizer, max_iters=self.max_opt_iters)` with optimizer="scg",
max_opt_iters=50 (:30-31).  The arithmetic is third-party ([GPy-upstream] GPy.core.Model /
paramz, not vendored in the reference, not installed here); its published algorithm is
restated:

  * parameters theta = (rbf.variance, rbf.lengthscale, bias.variance, Gaussian_noise.variance),
    each constrained positive through paramz's Logexp transform theta = log(1 + exp(phi));
    the optimiser works on phi.
  * objective(phi) = -[ log Z(theta) + sum_priored ( ln Gamma(theta_i; a_i, b_i)
                        + ln(1 - exp(-theta_i)) ) ]
    -- the log marginal likelihood, the Gamma priors set in gpy_regression.py:270-278 (rbf
    variance, lengthscale, bias variance; none on the noise) and, for parameters that carry a
    prior, the log-Jacobian of the Logexp transform (paramz Parameterized.log_prior).
  * gradient w.r.t. phi = -(d log Z/d theta + d log prior/d theta [+ 1/(exp(theta)-1)]) *
    (1 - exp(-theta)).
  * optimiser: Moller's scaled conjugate gradient as paramz ships it (sigma0 = 1e-7, second-order
    information from one extra gradient per successful step, trust parameter beta in
    [1e-15, 1e15], x/f/g tolerances 1e-6 / 1e-6 / 1e-5, `max_iters` iterations).

Every objective / gradient evaluation is a full refit on the GPU through the C ABI
(elfihip_gp_set_hyper + elfihip_gp_factorize + elfihip_gp_nlml_grad: Gram matrix, Cholesky,
L^-T, and the K^-1 SYRK with the four kernel-derivative contractions fused into its epilogue);
this module only holds the four-dimensional optimiser state.  "Parity unpinned": no reference
test or fixture fixes the optimum; tests compare against the CPU oracle's
restatement of the same algorithm and check descent / stationarity.
"""
import numpy as np
from scipy.special import gammaln

_LIM_VAL = 36.0
_LOG_LIM_VAL = np.log(np.finfo(np.float64).max)
NAMES = ('var', 'ls', 'bias', 'noise')


def logexp(phi):
    """paramz Logexp.f: theta = log(1 + exp(phi)), linear above 36."""
    phi = np.asarray(phi, dtype=np.float64)
    return np.where(phi > _LIM_VAL, phi, np.log1p(np.exp(np.clip(phi, -_LOG_LIM_VAL, _LIM_VAL))))


def logexp_inv(theta):
    """paramz Logexp.finv."""
    theta = np.asarray(theta, dtype=np.float64)
    # (the branch above the limit never reads expm1: evaluate it on the clipped argument -- same values below 36, no overflow)
    return np.where(theta > _LIM_VAL, theta, np.log(np.expm1(np.minimum(theta, _LIM_VAL))))


def logexp_gradfactor(theta):
    """d theta / d phi expressed in theta (paramz Logexp.gradfactor)."""
    return np.where(theta > _LIM_VAL, 1., -np.expm1(-theta))


def gamma_lnpdf(x, a, b):
    return a * np.log(b) - gammaln(a) + (a - 1.) * np.log(x) - b * x


class MarginalObjective:
    """objective(phi) and its gradient, evaluated by refitting the device GP."""

    allowed_failures = 10  # paramz Model._allowed_failures

    def __init__(self, model):
        self.model = model          # HipGPRegression
        self.handle = model._handle
        self.priors = model._priors or {}
        self._phi = None
        self._logz = None
        self._fail_count = 0
        self._last_grad = np.zeros(4)
        self.n_fits = 0

    def _fit(self, phi):
        phi = np.asarray(phi, dtype=np.float64)
        if self._phi is not None and np.array_equal(phi, self._phi):
            return
        theta = logexp(phi)
        self._phi = None
        self.handle.set_hyper(*theta)
        self._logz = self.handle.factorize()   # LinAlgError when K is not positive definite
        self._phi = phi.copy()
        self.n_fits += 1

    def _prior_terms(self, theta):
        lp, dlp = 0.0, np.zeros(4)
        for i, name in enumerate(NAMES):
            if name in self.priors:
                a, b = self.priors[name]
                t = theta[i]
                lp += gamma_lnpdf(t, a, b) + (np.log(np.expm1(t)) - t if t <= _LIM_VAL else 0.0)
                with np.errstate(over='ignore'):        # expm1(t) = inf above 709.78 -> the term is exactly 0, as in paramz
                    dlp[i] = (a - 1.) / t - b + 1. / np.expm1(t)
        return lp, dlp

    def f(self, phi):
        try:
            self._fit(phi)
            self._fail_count = 0
        except (np.linalg.LinAlgError, ZeroDivisionError, ValueError):
            if self._fail_count >= self.allowed_failures:
                raise
            self._fail_count += 1
            return np.inf
        lp, _ = self._prior_terms(logexp(phi))
        return -(self._logz + lp)

    def grad(self, phi):
        try:
            self._fit(phi)
            _, g = self.handle.nlml_grad()
            self._fail_count = 0
        except (np.linalg.LinAlgError, ZeroDivisionError, ValueError):
            if self._fail_count >= self.allowed_failures:
                raise
            self._fail_count += 1
            return np.clip(self._last_grad, -1e10, 1e10)
        theta = logexp(phi)
        _, dlp = self._prior_terms(theta)
        self._last_grad = -(g + dlp) * logexp_gradfactor(theta)
        return self._last_grad


def scg(f, gradf, x, maxiters=50, xtol=1e-6, ftol=1e-6, gtol=1e-5):
    """Scaled conjugate gradients (Moller 1993) with the constants paramz uses.

    Returns (x, objective trace, function evaluations, status)."""
    sigma0 = 1.0e-7
    x = np.array(x, dtype=np.float64)
    fold = f(x)
    function_eval = 1
    fnow = fold
    gradnew = gradf(x)
    function_eval += 1
    current_grad = np.dot(gradnew, gradnew)
    gradold = gradnew.copy()
    d = -gradnew
    success = True
    nsuccess = 0
    beta, betamin, betamax = 1.0, 1.0e-15, 1.0e15
    status = "maxiter exceeded"
    flog = [fold]
    iteration = 0
    mu = kappa = theta = 0.0
    while iteration < maxiters:
        if success:
            mu = np.dot(d, gradnew)
            if mu >= 0:
                d = -gradnew
                mu = np.dot(d, gradnew)
            kappa = np.dot(d, d)
            if kappa <= 0 or not np.isfinite(kappa):
                status = "zero search direction"
                break
            sigma = sigma0 / np.sqrt(kappa)
            gplus = gradf(x + sigma * d)
            function_eval += 1
            theta = np.dot(d, (gplus - gradnew)) / sigma
        delta = theta + beta * kappa
        if delta <= 0:
            delta = beta * kappa
            beta = beta - theta / kappa
        alpha = -mu / delta
        xnew = x + alpha * d
        fnew = f(xnew)
        function_eval += 1
        Delta = 2. * (fnew - fold) / (alpha * mu)
        if Delta >= 0:
            success = True
            nsuccess += 1
            x = xnew
            fnow = fnew
        else:
            success = False
            fnow = fold
        flog.append(fnow)
        iteration += 1
        if success:
            if np.abs(fnew - fold) < ftol:
                status = "converged - relative reduction in objective"
                break
            if np.max(np.abs(alpha * d)) < xtol:
                status = "converged - relative stepsize"
                break
            gradold = gradnew
            gradnew = gradf(x)
            function_eval += 1
            current_grad = np.dot(gradnew, gradnew)
            fold = fnew
            if current_grad <= gtol:
                status = "converged - relative reduction in gradient"
                break
        if Delta < 0.25:
            beta = min(4.0 * beta, betamax)
        if Delta > 0.75:
            beta = max(0.25 * beta, betamin)
        if nsuccess == x.size:
            d = -gradnew
            beta = 1.
            nsuccess = 0
        elif success:
            Gamma = np.dot(gradold - gradnew, gradnew) / mu
            d = Gamma * d - gradnew
    return x, flog, function_eval, status


def _scipy_search(name, obj, phi0, max_iters):
    """The other optimisers GPy's `model.optimize(optimizer=...)` accepts ([GPy-upstream] paramz
    optimization.get_optimizer: 'lbfgsb' / 'lbfgs' / 'org-bfgs' -> fmin_l_bfgs_b, 'bfgs' / 'org-bfgs', 'tnc', 'simplex'),
    with the arguments paramz passes: the same objective and gradient on the transformed parameters, max_iters as the
    iteration / evaluation cap.  Parity unpinned (no reference test or print-out uses them)."""
    import scipy.optimize as so
    fg = lambda x: (obj.f(x), obj.grad(x))
    if name in ('lbfgsb', 'lbfgs', 'lbfgsb-ps'):
        x, f, d = so.fmin_l_bfgs_b(fg, phi0, maxfun=max_iters, maxiter=max_iters)
        return x, [f], d.get('funcalls', 0), d.get('task', '')
    if name in ('bfgs', 'org-bfgs'):
        r = so.fmin_bfgs(obj.f, phi0, obj.grad, maxiter=max_iters, disp=False, full_output=True)
        return r[0], [r[1]], r[4], 'warnflag %d' % r[6]
    if name == 'tnc':
        x, nfev, rc = so.fmin_tnc(fg, phi0, messages=0, maxfun=max_iters)
        return x, [obj.f(x)], nfev, 'rc %d' % rc
    if name == 'simplex':
        r = so.fmin(obj.f, phi0, disp=False, maxfun=max_iters, full_output=True)
        return r[0], [r[1]], r[3], 'warnflag %d' % r[4]
    raise ValueError("optimizer %r: expected one of 'scg' (the reference's default, gpy_regression.py:30), 'lbfgsb', "
                     "'bfgs', 'tnc', 'simplex'" % (name,))


def optimize_hyperparameters(model, max_iters=50):
    """MAP hyper-parameters for `model` (a fitted HipGPRegression); refits it at the optimum."""
    h0 = dict(model._hyper)
    obj = MarginalObjective(model)
    phi0 = logexp_inv(np.array([h0[k] for k in NAMES]))
    try:
        if model.optimizer == 'scg':
            phi, flog, nfev, status = scg(obj.f, obj.grad, phi0, maxiters=max_iters)
        else:
            phi, flog, nfev, status = _scipy_search(str(model.optimizer).lower(), obj, phi0, max_iters)
    except Exception:
        # whatever ended the search (LinAlgError, or the ValueError / ZeroDivisionError the objective re-raises after
        # ten failed evaluations in a row): the device GP holds the last TRIAL hyper-parameters, possibly
        # unfactorised -- put the starting values back before the caller sees the error, so that the next update()
        # does not extend a factorisation that belongs to other hyper-parameters
        model._hyper = h0
        model._refit()
        raise
    theta = logexp(phi)
    model._hyper = dict(zip(NAMES, (float(t) for t in theta)))
    if obj._phi is not None and np.array_equal(obj._phi, np.asarray(phi, dtype=np.float64)):
        # the search's last evaluation WAS the optimum (SCG stops right after a successful step): the device GP holds that
        # very factorisation -- same hyper-parameters, bit for bit -- so the refit would only repeat it (one rebuild in
        # seventeen of a configs[2] search)
        model._log_marginal = obj._logz
        model._fitted_hyper = dict(model._hyper)
    else:
        model._refit()
    # The factorisation made here serves every acquisition until the next search (update_interval of them, each a dozen or
    # more lock-steps, extended point by point in between): K^-1 for the one-product lock-step is formed now -- 2-4 % of
    # what the search just cost -- instead of after the first 64 lock-steps (include/elfihip.h: elfihip_gp_set_lockstep_form)
    if getattr(model, 'kinv_after_optimize', True) and model.n_evidence >= 256:
        try:
            model._handle.form_kinv()
        except Exception:          # (out of device memory for the n x n matrix: the triangular products serve)
            pass
    model._opt_info = dict(status=status, objective=flog, n_fits=obj.n_fits, n_eval=nfev)
    return model._opt_info
