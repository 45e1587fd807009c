"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the BOLFI GP-surrogate arithmetic.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Nothing under elfi_amd/ imports it; the product path has no CPU fallback.

PARITY: pinned to the reference's own closed forms and LCBSC at fixed hyper-parameters
(tests/test_oracle_pinning_gp.py against tests/golden/gp_*.npz, produced by executing the real
reference classes, oracle/make_golden_gp.py); UNPINNED for the hyper-parameter optimisation:

  The reference delegates this arithmetic to the third-party library GPy
  (requirements.txt:5 `GPy>=1.0.9`, un-vendored, NOT installed here; upstream latest
  1.13.x) and its dependency paramz (SCG optimiser).  Neither source is under
  /root/reference, so the GPy parts below restate GPy's published algorithm from
  knowledge of upstream and are marked [GPy-upstream].  What IS in the reference and
  is followed line by line:

    GPyRegression.predict, sampling-phase closed form      elfi/methods/bo/gpy_regression.py:127-140
    GPyRegression._cache_RBF_kernel (what is cached)         :151-160
    GPyRegression.predictive_gradients closed form           :206-218
    GPyRegression._init_gp / _default_kernel (defaults)      :242-280
    GPyRegression.update (append + full rebuild)             :286-315
    LCBSC._beta / evaluate / evaluate_gradient               elfi/methods/bo/acquisition.py:256-301
    minimize (multi-start L-BFGS-B driver)                   elfi/methods/bo/utils.py:40-111

  tests/test_oracle_pinning_gp.py pins predict / predictive_gradients / LCB of this
  oracle to the REFERENCE'S OWN CODE: the real reference classes GPyRegression and LCBSC
  are imported (under oracle/ref_shim.py), given a stand-in `_gp` object that carries
  this oracle's posterior quantities, and executed; tests/golden/gp_*.npz holds those
  outputs.  The reference's unit tests pin the same closed forms to GPy itself
  (tests/unit/test_methods.py:110-122, tests/functional/test_inference.py:192-204).
  No reference test pins log-marginal values, hyper-parameter optima or posterior numbers
  (SURVEY.md section 8c), hence "parity unpinned" for optimize().

[GPy-upstream] formulas restated:
  kern = RBF(variance s_f, lengthscale l) + Bias(s_b):
      r2 = clip(-2 X X'^T + |x|^2 + |x'|^2, 0, inf), diag(r2) = 0 when X' is X
      K = s_f * exp(-0.5 * r2 / l^2) + s_b
  ExactGaussianInference:  Ky = K + (s_n + 1e-8) I;  L = jitchol(Ky);
      alpha = Ky^-1 y; Wi = Ky^-1;  logZ = 0.5*(-n log 2pi - logdet - y^T alpha)
      dL_dK = 0.5*(alpha alpha^T - Wi);  d logZ / d s_n = trace(dL_dK)
  predict_noiseless: mu = Kx^T alpha; var = clip(kxx - sum(Wi Kx * Kx), 1e-15, inf)
  predictive_gradients: dmu/dx = sum_i alpha_i dk_i/dx ;
      dvar/dx = sum_i (-2 Kx^T Wi)_i dk_i/dx  with dk_i/dx of the RBF part only
  kernel gradients: d/d s_f = sum(dL_dK * K_rbf)/s_f ; d/d l = sum(dL_dK * K_rbf * r2/l^2)/l ;
      d/d s_b = sum(dL_dK)
  Gamma prior from_EV(E, V): a = E^2/V, b = E/V; lnpdf = a ln b - lnGamma(a) + (a-1) ln x - b x
"""
import numpy as np
import scipy.linalg as sl
from scipy.special import gammaln

JITTER = 1e-8  # [GPy-upstream] ExactGaussianInference adds 1e-8 to the noise on the diagonal


def rbf_r2(X, X2=None):
    """[GPy-upstream] Stationary._unscaled_dist squared."""
    if X2 is None:
        Xsq = np.sum(np.square(X), 1)
        r2 = -2. * X.dot(X.T) + (Xsq[:, None] + Xsq[None, :])
        r2[np.diag_indices_from(r2)] = 0.
    else:
        r2 = -2. * X.dot(X2.T) + (np.sum(np.square(X), 1)[:, None] + np.sum(np.square(X2), 1)[None, :])
    return np.clip(r2, 0, np.inf)


def kern_K(X, X2, var, ls, bias):
    return var * np.exp(-0.5 * rbf_r2(X, X2) / ls**2) + bias


def default_hyper(bounds, y, noise_var=None):
    """The heuristic values of elfi/methods/bo/gpy_regression.py:255,260-264.

    NB: in the reference these parametrise the Gamma PRIORS (from_EV(E, E), :270-278) and the
    initial noise; the kernel itself starts from GPy's defaults (see initial_hyper).  The parity
    tests and SURVEY.md's G-1 config use them as a convenient FIXED hyper-parameter set."""
    length_scale = (np.max(bounds) - np.min(bounds)) / 3.
    kernel_var = (np.max(y) / 3.)**2.
    bias_var = kernel_var / 4.
    noise = noise_var or np.max(y)**2. / 100.
    return dict(var=float(kernel_var), ls=float(length_scale), bias=float(bias_var), noise=float(noise))


def initial_hyper(y, noise_var=None):
    """Hyper-parameters of a freshly built reference GP: gpy_regression.py:267,275 call
    GPy.kern.RBF(input_dim) and GPy.kern.Bias(input_dim) with [GPy-upstream] defaults
    (variance=1, lengthscale=1; variance=1); noise from :255."""
    return dict(var=1.0, ls=1.0, bias=1.0, noise=float(noise_var or np.max(y)**2. / 100.))


def jitchol(A, maxtries=5):
    """[GPy-upstream] util.linalg.jitchol: LAPACK dpotrf first; on failure jitter = mean(diag) * 1e-6, times ten per
    failed try, at most `maxtries` tries (GPy's loop catches every exception of the retry, so a matrix with NaN in it also
    ends in "not positive definite, even with jitter")."""
    A = np.ascontiguousarray(A)
    L, info = sl.lapack.dpotrf(A, lower=1)
    if info == 0:
        return np.tril(L)
    diagA = np.diag(A)
    if np.any(diagA <= 0.):
        raise np.linalg.LinAlgError("not pd: non-positive diagonal elements")
    jitter = diagA.mean() * 1e-6
    num_tries = 1
    while num_tries <= maxtries and np.isfinite(jitter):
        try:
            return sl.cholesky(A + np.eye(A.shape[0]) * jitter, lower=True)
        except Exception:
            jitter *= 10
        finally:
            num_tries += 1
    raise np.linalg.LinAlgError("not positive definite, even with jitter.")


class Posterior:
    """Everything GPy's posterior object exposes to ELFI (gpy_regression.py:151-160)."""

    def __init__(self, X, Y, var, ls, bias, noise, need_inv=True):
        self.X = np.asarray(X, dtype=float)
        self.Y = np.asarray(Y, dtype=float).reshape(-1, 1)
        self.var, self.ls, self.bias, self.noise = float(var), float(ls), float(bias), float(noise)
        n = self.X.shape[0]
        self.K = kern_K(self.X, None, var, ls, bias)
        Ky = self.K.copy()
        Ky[np.diag_indices(n)] += noise + JITTER
        self.L = jitchol(Ky)                                    # woodbury_chol
        self.alpha = sl.cho_solve((self.L, True), self.Y)       # woodbury_vector
        self.logdet = 2. * np.sum(np.log(np.diag(self.L)))
        self.log_marginal = 0.5 * (-n * np.log(2 * np.pi) - self.logdet - float((self.Y.T @ self.alpha)[0, 0]))
        if need_inv:
            Linv = sl.solve_triangular(self.L, np.eye(n), lower=True)
            self.Linv = Linv
            self.Kinv = Linv.T @ Linv                           # woodbury_inv

    # --- [GPy-upstream] predict_noiseless / predict
    def predict(self, x, noiseless=False):
        x = np.asanyarray(x, dtype=float).reshape((-1, self.X.shape[1]))
        Kx = kern_K(self.X, x, self.var, self.ls, self.bias)    # (n, S)
        mu = Kx.T @ self.alpha
        var = (self.var + self.bias) - np.sum((self.Kinv @ Kx) * Kx, 0)
        var = np.clip(var, 1e-15, np.inf)[:, None]
        if not noiseless:
            var = var + self.noise
        return mu, var

    # --- [GPy-upstream] predictive_gradients (mean_jac[:, :, 0], dv_dX)
    def predictive_gradients(self, x):
        x = np.asanyarray(x, dtype=float).reshape((-1, self.X.shape[1]))
        Krbf = self.var * np.exp(-0.5 * rbf_r2(x, self.X) / self.ls**2)        # (S, n)
        Kx = Krbf + self.bias
        diff = x[:, None, :] - self.X[None, :, :]                               # (S, n, d)
        dk = -(Krbf / self.ls**2)[:, :, None] * diff                            # d k_i / d x
        grad_mu = np.einsum('snd,n->sd', dk, self.alpha[:, 0])
        a = -2. * Kx @ self.Kinv                                                 # (S, n)
        grad_var = np.einsum('snd,sn->sd', dk, a)
        return grad_mu, grad_var

    # --- the reference's sampling-phase closed forms, gpy_regression.py:127-140 / 206-218
    def predict_closed_form(self, x):
        x = np.asanyarray(x, dtype=float).reshape((-1, self.X.shape[1]))
        factor = -0.5 / self.ls**2
        x2sum = np.sum(self.X**2., 1)[None, :]
        r2 = np.sum(x**2., 1)[:, None] + x2sum - 2. * x.dot(self.X.T)
        kx = self.var * np.exp(r2 * factor) + self.bias
        mu = kx.dot(self.alpha)
        var = self.var + self.bias
        var = var - kx.dot(self.Kinv.dot(kx.T))
        var = var + self.noise
        return mu, var

    # --- objective GPy's optimizer minimises and its gradient w.r.t. (var, ls, bias, noise)
    def dL_dK(self):
        return 0.5 * (self.alpha @ self.alpha.T - self.Kinv)

    def log_marginal_grad(self):
        """d logZ / d (var, ls, bias, noise)  [GPy-upstream update_gradients_full]."""
        D = self.dL_dK()
        r2 = rbf_r2(self.X)
        Krbf = self.var * np.exp(-0.5 * r2 / self.ls**2)
        g_var = np.sum(D * Krbf) / self.var
        g_ls = np.sum(D * Krbf * r2) / self.ls**3
        g_bias = np.sum(D)
        g_noise = np.trace(D)
        return np.array([g_var, g_ls, g_bias, g_noise])


def gamma_from_EV(E, V):
    """[GPy-upstream] priors.Gamma.from_EV."""
    return E**2 / V, E / V


def gamma_lnpdf(x, a, b):
    return a * np.log(b) - gammaln(a) + (a - 1) * np.log(x) - b * x


def gamma_lnpdf_grad(x, a, b):
    return (a - 1.) / x - b


def default_priors(bounds, y):
    """Gamma priors of gpy_regression.py:270-278 as (a, b) per (var, ls, bias); noise has none."""
    h = default_hyper(bounds, y)
    return dict(var=gamma_from_EV(h['var'], h['var']), ls=gamma_from_EV(h['ls'], h['ls']),
                bias=gamma_from_EV(h['bias'], h['bias']))


# --- LCBSC, elfi/methods/bo/acquisition.py:256-301
def lcb_beta(t, d, exploration_rate=10.):
    t = t + 1
    delta = 1. / exploration_rate
    return 2 * np.log(t**(2 * d + 2) * np.pi**2 / (3 * delta))


def lcb_evaluate(post, x, t, exploration_rate=10.):
    mean, var = post.predict(x, noiseless=True)
    return mean - np.sqrt(lcb_beta(t, post.X.shape[1], exploration_rate) * var)


def lcb_evaluate_gradient(post, x, t, exploration_rate=10.):
    mean, var = post.predict(x, noiseless=True)
    grad_mean, grad_var = post.predictive_gradients(x)
    return grad_mean - 0.5 * grad_var * np.sqrt(lcb_beta(t, post.X.shape[1], exploration_rate) / var)


# --- multi-start driver, elfi/methods/bo/utils.py:40-111 (the sequential reference form)
def minimize_multistart(fun, grad, bounds, start_points, maxiter=1000):
    import scipy.optimize
    locs, vals = [], np.empty(len(start_points))
    for i, x0 in enumerate(start_points):
        res = scipy.optimize.minimize(fun, x0, method='L-BFGS-B', jac=grad, bounds=bounds,
                                      options={'maxiter': maxiter})
        locs.append(res['x'])
        vals[i] = res['fun']
    k = int(np.argmin(vals))
    x = locs[k]
    for i in range(len(bounds)):
        x[i] = np.clip(x[i], *bounds[i])
    return x, vals[k]


def synthetic_gp_problem(n=4096, d=10, seed=0):
    """BASELINE.md section 3 config 3 inputs (metric shape n=4096, d=10)."""
    X = np.random.RandomState(seed).uniform(-2, 2, (n, d))
    y = np.linalg.norm(X - 0.5, axis=1) + 0.1 * np.random.RandomState(seed + 1).randn(n)
    bounds = [(-2., 2.)] * d
    return X, y.reshape(-1, 1), bounds
