"""GP surrogate for elfi.BOLFI on the GPU: drop-in for GPyRegression.

`HipGPRegression` has the duck-type ELFI's Bayesian-optimisation code expects from
`target_model` (reference elfi v0.8.7):

    elfi/methods/bo/gpy_regression.py:15-364   GPyRegression (the class mirrored here)
    elfi/methods/inference/bolfi.py:35,86-87    BayesianOptimization(target_model=...)
    elfi/methods/bo/acquisition.py:37-41        what acquisition rules call on the model

Same constructor arguments, attributes (parameter_names, input_dim, bounds, n_evidence,
X, Y, noise, instance, is_sampling) and methods (predict, predict_mean,
predictive_gradients, predictive_gradient_mean, update, optimize, copy), same default
kernel / prior heuristics (gpy_regression.py:242-280).  The arithmetic GPy performs in the
reference (Gram matrix, Cholesky, K^-1, predict, gradients) runs in libelfihip.so through
the C ABI of include/elfihip.h; nothing here falls back to the CPU.

Use:  elfi.BOLFI(model, target_model=HipGPRegression(parameter_names, bounds=bounds), ...)
"""
import copy
import ctypes as C
import sys
import logging

import numpy as np

from . import _lib

logger = logging.getLogger(__name__)


class GPHandle:
    """Owner of one elfihip_gp (device-resident evidence, factor and alpha)."""

    def __init__(self, d, capacity, ctx=None):
        self.ctx = ctx or _lib.default_context()
        self.lib = self.ctx.lib
        self.d = int(d)
        h = C.c_void_p()
        self._check(self.lib.elfihip_gp_create(self.ctx.handle, self.d, int(capacity), C.byref(h)))
        self.h = h
        n, cap, dd = C.c_int64(), C.c_int64(), C.c_int()
        self._check(self.lib.elfihip_gp_size(self.h, C.byref(n), C.byref(cap), C.byref(dd)))
        self.capacity = cap.value
        self.n = 0
        self._n_int = 0   # integration points held by the device object (set_integration_points)

    def _check(self, rc):
        if rc != _lib.OK:
            _lib._raise(self.lib, self.ctx.handle, rc)

    def close(self):
        if getattr(self, 'h', None) is not None:
            self.lib.elfihip_gp_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_hyper(self, var, ls, bias, noise):
        self._check(self.lib.elfihip_gp_set_hyper(self.h, float(var), float(ls), float(bias), float(noise)))

    def set_data(self, X, y):
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, self.d)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        self._check(self.lib.elfihip_gp_set_data(self.h, _lib.ptr(X), _lib.ptr(y), X.shape[0]))
        self.n = X.shape[0]

    def append(self, X, y):
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, self.d)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        self._check(self.lib.elfihip_gp_append(self.h, _lib.ptr(X), _lib.ptr(y), X.shape[0]))
        self.n += X.shape[0]

    def extend(self, X, y):
        """Append evidence and update the factorisation by bordering (O(n^2) per point)."""
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, self.d)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        lz = C.c_double()
        self._check(self.lib.elfihip_gp_extend(self.h, _lib.ptr(X), _lib.ptr(y), X.shape[0], C.byref(lz)))
        self.n += X.shape[0]
        return lz.value

    def set_schedule(self, schedule=0, panel_group=0):
        """Sweep schedule of factorize(): 0 by size, 1 two-stream look-ahead, 2 fused steps, 3 fused steps chained inside
        one launch per block column, 4 panel solve + diagonal tile in one launch, 5 three concurrent launches
        (include/elfihip.h; 3-5 are kept for measurement)."""
        self._check(self.lib.elfihip_gp_set_schedule(self.h, int(schedule), int(panel_group)))

    def set_dense_threshold(self, min_points=0, tile_rows=0):
        """Calls with at least min_points query points use the dense (matrix-pipe-bound) form of the triangular products
        (include/elfihip.h: elfihip_gp_set_dense_threshold); 0 = default.  tile_rows: 0 = by size, else 64 / 32 / 16."""
        self._check(self.lib.elfihip_gp_set_dense_threshold(self.h, int(min_points), int(tile_rows)))

    def set_lockstep_form(self, form=0):
        """0: four launches per small prediction call (fused epilogues), acquisition lock-steps through ONE product with
        K^-1 once a factorisation has served 64 of them (default); 1: the six-launch form; 2: fused, triangular products
        only; 3: the K^-1 product from the first acquisition lock-step on (include/elfihip.h)."""
        self._check(self.lib.elfihip_gp_set_lockstep_form(self.h, int(form)))

    def lockstep_info(self):
        """(K^-1 in use by the acquisition lock-steps, lock-steps served by this factorisation, the estimate of cond(K) the
        K^-1 form is gated by: include/elfihip.h)."""
        use, steps, cond = C.c_int(0), C.c_int64(0), C.c_double(0.0)
        self._check(self.lib.elfihip_gp_lockstep_info(self.h, C.byref(use), C.byref(steps), C.byref(cond)))
        return bool(use.value), int(steps.value), float(cond.value)

    def set_acq_options(self, host_threads=0, trace=0):
        """Options of lcb_minimize (include/elfihip.h: elfihip_gp_set_acq_options): host threads of the multi-start
        search's quasi-Newton algebra (0: by the machine), trace level on stderr (0 none, 1 per search, 2 per round)."""
        self._check(self.lib.elfihip_gp_set_acq_options(self.h, int(host_threads), int(trace)))

    PHASES = ('gram', 'sweep', 'alpha', 'kstar', 'tri_first', 'tri_second', 'grad_finish', 'kinv_grad')

    def profile(self, enable=-1):
        """Device time per phase (include/elfihip.h: elfihip_gp_profile): enable 1 starts a fresh measurement, 0 stops,
        -1 only reads.  Returns {phase: (milliseconds summed, calls)} collected so far."""
        ms = np.zeros(len(self.PHASES))
        calls = np.zeros(len(self.PHASES), dtype=np.int64)
        self._check(self.lib.elfihip_gp_profile(self.h, int(enable), _lib.ptr(ms), _lib.ptr(calls)))
        return {k: (float(ms[i]), int(calls[i])) for i, k in enumerate(self.PHASES)}

    def factorize(self):
        lz = C.c_double()
        self._check(self.lib.elfihip_gp_factorize(self.h, C.byref(lz)))
        return lz.value

    def jitchol(self, maxtries=-1):
        """(jitter on the current factor's diagonal, retries of the latest factorize()) -- GPy's jitchol ladder
        (include/elfihip.h: elfihip_gp_jitchol); maxtries >= 0 sets the retries allowed (GPy: 5)."""
        j, t = C.c_double(0.0), C.c_int(0)
        self._check(self.lib.elfihip_gp_jitchol(self.h, int(maxtries), C.byref(j), C.byref(t)))
        return float(j.value), int(t.value)

    def get(self, which):
        n, d = self.n, self.d
        shape = {0: (n, n), 1: (n, n), 2: (n, 1), 3: (n, d), 4: (n, 1), 5: (n, n)}[which]
        out = np.empty(shape, dtype=np.float64)
        self._check(self.lib.elfihip_gp_get(self.h, which, _lib.ptr(out)))
        return out

    def _xs(self, x):
        return np.ascontiguousarray(x, dtype=np.float64).reshape(-1, self.d)

    def predict(self, x, noiseless=False):
        x = self._xs(x)
        S = x.shape[0]
        mu, var = np.empty((S, 1)), np.empty((S, 1))
        self._check(self.lib.elfihip_gp_predict(self.h, _lib.ptr(x), S, int(bool(noiseless)),
                                                _lib.ptr(mu), _lib.ptr(var)))
        return mu, var

    def predict_grad(self, x):
        x = self._xs(x)
        S = x.shape[0]
        mu, var = np.empty((S, 1)), np.empty((S, 1))
        dmu, dvar = np.empty((S, self.d)), np.empty((S, self.d))
        self._check(self.lib.elfihip_gp_predict_grad(self.h, _lib.ptr(x), S, _lib.ptr(mu), _lib.ptr(var),
                                                     _lib.ptr(dmu), _lib.ptr(dvar)))
        return mu, var, dmu, dvar

    def set_integration_points(self, points):
        """Fix the point set P (M, d) of the following cross_cov() calls (until the GP changes)."""
        pts = self._xs(points)
        self._check(self.lib.elfihip_gp_set_integration_points(self.h, _lib.ptr(pts), pts.shape[0]))
        self._n_int = pts.shape[0]

    def cross_cov(self, x):
        """Posterior covariance (M, S) between the integration points and x (S, d), noiseless variance (S,) of x."""
        if self._n_int == 0:
            raise RuntimeError('no integration points: call set_integration_points first')
        x = self._xs(x)
        S = x.shape[0]
        cov = np.empty((self._n_int, S))
        var = np.empty(S)
        self._check(self.lib.elfihip_gp_cross_cov(self.h, _lib.ptr(x), S, _lib.ptr(cov), _lib.ptr(var)))
        return cov, var

    def maxvar(self, x, eps, prior_pdf, prior_grad_logpdf):
        """MaxVar surface and gradient at x (S, d): one batched prediction + the device epilogue (elfihip_gp_maxvar)."""
        x = self._xs(x)
        S = x.shape[0]
        pdf = np.ascontiguousarray(prior_pdf, dtype=np.float64).reshape(S)
        glog = np.ascontiguousarray(prior_grad_logpdf, dtype=np.float64).reshape(S, self.d)
        val, grad = np.empty((S, 1)), np.empty((S, self.d))
        self._check(self.lib.elfihip_gp_maxvar(self.h, _lib.ptr(x), S, float(eps), _lib.ptr(pdf), _lib.ptr(glog),
                                               _lib.ptr(val), _lib.ptr(grad)))
        return val, grad

    def expintvar(self, x, eps, w_int, mean_int, var_int):
        """ExpIntVar loss of the candidates x (S, d) against the current integration points (elfihip_gp_expintvar)."""
        if self._n_int == 0:
            raise RuntimeError('no integration points: call set_integration_points first')
        x = self._xs(x)
        S, M = x.shape[0], self._n_int
        arrs = [np.ascontiguousarray(a, dtype=np.float64).reshape(M) for a in (w_int, mean_int, var_int)]
        loss = np.empty(S)
        self._check(self.lib.elfihip_gp_expintvar(self.h, _lib.ptr(x), S, float(eps), _lib.ptr(arrs[0]),
                                                  _lib.ptr(arrs[1]), _lib.ptr(arrs[2]), _lib.ptr(loss)))
        return loss

    def lcb(self, x, beta, with_grad=True):
        x = self._xs(x)
        S = x.shape[0]
        val = np.empty((S, 1))
        grad = np.empty((S, self.d)) if with_grad else None
        self._check(self.lib.elfihip_gp_lcb(self.h, _lib.ptr(x), S, float(beta), _lib.ptr(val), _lib.ptr(grad)))
        return val, grad

    def nlml_grad(self):
        """(log marginal likelihood, d logZ / d (var, ls, bias, noise)) at the current factorisation."""
        lz = C.c_double()
        g = np.empty(4)
        self._check(self.lib.elfihip_gp_nlml_grad(self.h, C.byref(lz), _lib.ptr(g)))
        return lz.value, g

    def form_kinv(self):
        self._check(self.lib.elfihip_gp_form_kinv(self.h))

    def kernel_matrix(self, A, B=None):
        """Prior covariance k(A, B) (k(A, A) when B is None) under the current hyper-parameters: GPy's kern.K."""
        A = self._xs(A)
        Bc = None if B is None else self._xs(B)
        out = np.empty((A.shape[0], A.shape[0] if Bc is None else Bc.shape[0]))
        self._check(self.lib.elfihip_gp_kernel_matrix(self.h, _lib.ptr(A), A.shape[0], _lib.ptr(Bc),
                                                      0 if Bc is None else Bc.shape[0], _lib.ptr(out)))
        return out

    def lcb_minimize(self, starts, bounds, beta, maxiter=1000):
        """Lock-step multi-start minimisation of the LCB; returns (x (S,d), f (S,), iters (S,), n_eval)."""
        starts = self._xs(starts)
        S = starts.shape[0]
        lo = np.ascontiguousarray([b[0] for b in bounds], dtype=np.float64)
        hi = np.ascontiguousarray([b[1] for b in bounds], dtype=np.float64)
        if lo.shape[0] != self.d:
            raise ValueError('bounds must have one (lower, upper) pair per input dimension')
        x = np.empty((S, self.d))
        f = np.empty(S)
        it = np.empty(S, dtype=np.int32)
        ne = C.c_int64()
        self._check(self.lib.elfihip_gp_lcb_minimize(self.h, _lib.ptr(starts), S, _lib.ptr(lo), _lib.ptr(hi),
                                                     float(beta), int(maxiter), _lib.ptr(x), _lib.ptr(f),
                                                     _lib.ptr(it), C.byref(ne)))
        return x, f, it, ne.value


# ---- the small part of the GPy object graph that ELFI code touches ---------------------------
class _Param:
    """Stands in for a paramz Param: float(p), p[0], p.values."""

    def __init__(self, value):
        self.values = np.array([float(value)])

    def __float__(self):
        return float(self.values[0])

    def __getitem__(self, i):
        return self.values[i]

    def __repr__(self):
        return repr(self.values)


class _Part:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _GPShim:
    """`target_model._gp` / `.instance` for code that reaches into GPy
    (tests/unit/test_bo.py:41 reads _gp.X; posteriors.py:296 reads _gp.param_array;
    gpy_regression.py:151-158 reads kern.rbf / kern.bias / posterior.woodbury_*)."""

    def __init__(self, model):
        self._m = model

    @property
    def X(self):
        return self._m._X

    @property
    def Y(self):
        return self._m._Y

    @property
    def num_data(self):
        return self._m._X.shape[0]

    @property
    def kern(self):
        m = self._m

        def K(X, X2=None):
            # acquisition.py:754,770 (the reference's ExpIntVar) evaluates the prior kernel matrix through GPy
            return m._handle.kernel_matrix(X, X2)

        return _Part(K=K, rbf=_Part(variance=_Param(m._hyper['var']), lengthscale=_Param(m._hyper['ls'])),
                     bias=_Part(variance=_Param(m._hyper['bias']),
                                K=lambda X, X2=None: np.full((len(X), len(X if X2 is None else X2)),
                                                             m._hyper['bias'])))

    @property
    def Gaussian_noise(self):
        return _Part(variance=_Param(self._m._hyper['noise']))

    likelihood = Gaussian_noise

    @property
    def param_array(self):
        h = self._m._hyper
        return np.array([h['var'], h['ls'], h['bias'], h['noise']])

    @property
    def posterior(self):
        m = self._m
        m._handle.form_kinv()
        return _Part(woodbury_vector=m._handle.get(2), woodbury_chol=m._handle.get(0),
                     woodbury_inv=m._handle.get(5))

    def log_likelihood(self):
        return self._m._log_marginal

    def __str__(self):
        h = self._m._hyper
        return ("HipGPRegression  n=%d  log-marginal=%.6g\n  rbf.variance %.6g\n  rbf.lengthscale %.6g\n"
                "  bias.variance %.6g\n  Gaussian_noise.variance %.6g" %
                (self.num_data, self._m._log_marginal, h['var'], h['ls'], h['bias'], h['noise']))


_REFERENCE_SUBCLASSES = {}


def _reference_base():
    """The reference's GPyRegression class if (and only if) the running program has imported it already."""
    mod = sys.modules.get('elfi.methods.bo.gpy_regression')
    return getattr(mod, 'GPyRegression', None) if mod is not None else None


def _rebuild_gp(origin, state):
    obj = origin.__new__(origin)
    obj.__setstate__(state)
    return obj


class HipGPRegression:
    """Gaussian-process regression on the GPU with the interface of GPyRegression
    (elfi/methods/bo/gpy_regression.py:15-364).

    Code that asks `isinstance(model, GPyRegression)` (elfi/methods/inference/bolfire.py:329-331) is satisfied as well:
    when the reference's class has been imported by the time an object is made, the object's class is a subclass of
    BOTH (this class first in the method resolution order, the reference's __init__ never runs); nothing of ELFI or
    GPy is imported from here."""

    def __new__(cls, *args, **kwargs):
        base = _reference_base()
        if base is not None and not issubclass(cls, base):
            sub = _REFERENCE_SUBCLASSES.get((cls, base))
            if sub is None:
                sub = type(cls.__name__, (cls, base), {'__module__': cls.__module__, '__doc__': cls.__doc__,
                                                       '_hip_origin': cls})
                _REFERENCE_SUBCLASSES[(cls, base)] = sub
            cls = sub
        return object.__new__(cls)

    _hip_origin = None

    def __reduce__(self):
        # pickle / copy by the importable class: the two-base class is re-derived where the object is rebuilt
        return _rebuild_gp, (type(self)._hip_origin or type(self), self.__getstate__())

    def __init__(self, parameter_names=None, bounds=None, optimizer="scg", max_opt_iters=50, gp=None,
                 device=-1, **gp_params):
        if parameter_names is None:
            input_dim = 1
        elif isinstance(parameter_names, (list, tuple)):
            input_dim = len(parameter_names)
        else:
            raise ValueError("Keyword `parameter_names` must be a list of strings")

        if bounds is None:
            logger.warning('Parameter bounds not specified. Using [0,1] for each parameter.')
            bounds = [(0, 1)] * input_dim
        elif len(bounds) != input_dim:
            raise ValueError('Length of `bounds` ({}) does not match the length of `parameter_names` ({}).'
                             .format(len(bounds), input_dim))
        elif isinstance(bounds, dict):
            if len(bounds) == 1:  # might be the case parameter_names=None
                bounds = [bounds[n] for n in bounds.keys()]
            else:
                bounds = [bounds[n] for n in parameter_names]
        else:
            raise ValueError("Keyword `bounds` must be a dictionary "
                             "`{'parameter_name': (lower, upper), ... }`")
        if gp_params.get('kernel') is not None or gp_params.get('mean_function') is not None:
            raise NotImplementedError('HipGPRegression implements the default RBF+Bias kernel with a zero '
                                      'mean function (gpy_regression.py:260-280); custom GPy kernels / mean '
                                      'functions need the reference GPyRegression')
        self.parameter_names = parameter_names
        self.input_dim = input_dim
        self.bounds = bounds
        self.gp_params = gp_params
        self.optimizer = optimizer
        self.max_opt_iters = max_opt_iters
        self.device = device
        self._gp = gp
        self._rbf_is_cached = False
        self.is_sampling = False
        self._kernel_is_default = True
        self._handle = None
        self._X = None
        self._Y = None
        self._hyper = None
        self._priors = None
        self._log_marginal = None

    # -- representation -----------------------------------------------------------------
    def __str__(self):
        return str(self._gp)

    __repr__ = __str__

    # -- prediction -----------------------------------------------------------------------
    def predict(self, x, noiseless=False):
        """GP (mean, var) at x, each (n, 1).  gpy_regression.py:98-147."""
        x = np.asanyarray(x).reshape((-1, self.input_dim))
        if self._gp is None:
            return np.zeros((x.shape[0], 1)), np.ones((x.shape[0], 1))
        if self.is_sampling and self._kernel_is_default:
            # the reference's sampling-phase closed form always includes the noise (:139)
            self._rbf_is_cached = True
            return self._handle.predict(x, noiseless=False)
        self._rbf_is_cached = False
        return self._handle.predict(x, noiseless=noiseless)

    def predict_mean(self, x):
        return self.predict(x)[0]

    def predictive_gradients(self, x):
        """(grad_mean, grad_var) at x, each (n, input_dim).  gpy_regression.py:179-223."""
        x = np.asanyarray(x).reshape((-1, self.input_dim))
        if self._gp is None:
            return np.zeros((x.shape[0], self.input_dim)), np.zeros((x.shape[0], self.input_dim))
        _, _, dmu, dvar = self._handle.predict_grad(x)
        return dmu, dvar

    def predictive_gradient_mean(self, x):
        return self.predictive_gradients(x)[0]

    # -- posterior covariance against a fixed point set (ExpIntVar, acquisition.py:775-808) -------------
    def set_integration_points(self, points):
        self._handle.set_integration_points(np.asanyarray(points).reshape((-1, self.input_dim)))

    def cross_cov(self, x):
        """(cov (M, S), var (S,)): covariance of the GP between the integration points and the rows of x, and the
        noiseless predictive variance of the rows of x."""
        return self._handle.cross_cov(np.asanyarray(x).reshape((-1, self.input_dim)))

    # -- surfaces of the MaxVar family (acquisition.py:392-463, 795-821), epilogues on the device ------
    def maxvar_surface(self, theta, eps, prior_pdf, prior_grad_logpdf):
        """(value (S, 1), gradient (S, d)) of the variance of the unnormalised approximate posterior at theta."""
        return self._handle.maxvar(np.asanyarray(theta).reshape((-1, self.input_dim)), eps, prior_pdf, prior_grad_logpdf)

    def expintvar_loss(self, theta, eps, w_int, mean_int, var_int):
        """Expected integrated variance (S,) for the candidates theta against the points of set_integration_points."""
        return self._handle.expintvar(np.asanyarray(theta).reshape((-1, self.input_dim)), eps, w_int, mean_int, var_int)

    # -- batched LCB used by elfi_amd.acquisition (one device pass for all start points) -------
    def lcb(self, x, beta, with_grad=True):
        x = np.asanyarray(x).reshape((-1, self.input_dim))
        return self._handle.lcb(x, beta, with_grad)

    # -- fitting --------------------------------------------------------------------------
    def _prior_expectations(self, y):
        # heuristics of gpy_regression.py:260-264: they parametrise the Gamma PRIORS only
        length_scale = (np.max(self.bounds) - np.min(self.bounds)) / 3.
        kernel_var = (np.max(y) / 3.)**2.
        bias_var = kernel_var / 4.
        return dict(var=float(kernel_var), ls=float(length_scale), bias=float(bias_var))

    def _default_hyper(self, y):
        # gpy_regression.py:267,275 construct GPy.kern.RBF(input_dim) / GPy.kern.Bias(input_dim) with
        # GPy's own defaults (variance 1, lengthscale 1 [GPy-upstream]); set_prior does not move the
        # values.  Only the noise gets a data-driven start (:255).
        noise_var = self.gp_params.get('noise_var') or np.max(y)**2. / 100.
        return dict(var=1.0, ls=1.0, bias=1.0, noise=float(noise_var))

    def _init_gp(self, x, y):
        self._kernel_is_default = self.gp_params.get('noise_var') is None
        self._hyper = self._default_hyper(y)
        # Gamma priors from_EV(E, V=E) -> shape a = E^2/V = E, rate b = E/V = 1
        # (gpy_regression.py:270-278) on lengthscale, rbf variance and bias variance; none on the noise
        self._priors = {k: (E, 1.0) for k, E in self._prior_expectations(y).items()}
        self._X = np.empty((0, self.input_dim))
        self._Y = np.empty((0, 1))
        self._gp = _GPShim(self)
        self._append_and_fit(x, y)

    def _ensure_capacity(self, n):
        if self._handle is not None and n <= self._handle.capacity:
            return False
        cap = max(512, int(2 ** np.ceil(np.log2(max(n, 1)))))
        old = self._handle
        self._handle = GPHandle(self.input_dim, cap, ctx=_lib.default_context(self.device))
        if self.acq_host_threads or self.acq_trace:
            self._handle.set_acq_options(self.acq_host_threads, self.acq_trace)
        if old is not None:
            old.close()
        return True

    # Options of the multi-start acquisition search handed to every device handle this model creates
    # (GPHandle.set_acq_options): host threads (0: by the machine), trace level on stderr.
    acq_host_threads = 0
    acq_trace = 0

    # After a hyper-parameter search the refit at the optimum also forms K^-1 (from 256 evidence points on), so that the
    # acquisitions of the interval run their lock-steps through ONE product from the first one on (hyperopt.py).
    kinv_after_optimize = True

    # Points added per update() up to which the factorisation is extended by bordering (O(n^2) each)
    # instead of rebuilt (O(n^3)); the reference always rebuilds (gpy_regression.py:304-312), the
    # results agree to rounding.  0 disables the incremental path.
    incremental_limit = 64

    def _append_and_fit(self, x, y):
        n_old = self._X.shape[0]
        self._host_append(x, y)
        if self._ensure_capacity(self._X.shape[0]) or n_old != self._handle.n:
            self._handle.set_data(self._X, self._Y)
        elif (0 < x.shape[0] <= self.incremental_limit and n_old > 0
              and getattr(self, '_fitted_hyper', None) == self._hyper):
            self._log_marginal = self._handle.extend(x, y)   # hyper-parameters unchanged since the last fit
            return
        else:
            self._handle.append(x, y)
        self._refit()

    def _host_append(self, x, y):
        """The host copy of the evidence (what `X` / `Y` and `_gp.X` show) in a buffer that grows geometrically: an update
        copies the new rows only (np.r_ of 4096 x 10 doubles was 50 us of every 1.7 ms rebuild)."""
        n, k = self._X.shape[0], x.shape[0]
        buf = getattr(self, '_Xbuf', None)
        if buf is None or buf.shape[0] < n + k or self._X.base is not buf:
            cap = max(256, int(1.5 * (n + k)))
            self._Xbuf = np.empty((cap, self.input_dim))
            self._Ybuf = np.empty((cap, 1))
            self._Xbuf[:n] = self._X
            self._Ybuf[:n] = self._Y
        self._Xbuf[n:n + k] = x
        self._Ybuf[n:n + k] = y
        self._X = self._Xbuf[:n + k]
        self._Y = self._Ybuf[:n + k]

    def fix_hyperparameters(self, var=None, ls=None, bias=None, noise=None, **hyper):
        """Set hyper-parameters by hand (signal variance, lengthscale, bias variance, noise variance; unnamed ones keep
        their values) and rebuild the GP with them -- what `m._gp.rbf.variance = v` etc. followed by a refit do on a
        GPy model.  Until the next `optimize()` (an `update(..., optimize=True)` of the BOLFI loop) every update keeps
        these values.  Returns the hyper-parameters in use."""
        if self._gp is None:
            raise RuntimeError('no evidence yet: hyper-parameters belong to a GP that exists (call update first)')
        new = dict(self._hyper)
        for k, v in dict(var=var, ls=ls, bias=bias, noise=noise, **hyper).items():
            if v is None:
                continue
            if k not in new:
                raise ValueError("unknown hyper-parameter %r (expected var, ls, bias, noise)" % (k,))
            new[k] = float(v)
        self._hyper = new
        self._refit()
        return dict(self._hyper)

    @property
    def device_handle(self):
        """The GPHandle (C-ABI object: elfihip_gp) this model computes with -- phase timers, schedules, raw entry points."""
        return self._handle

    @property
    def hyperparameters(self):
        """dict(var, ls, bias, noise) in use (GPy: rbf.variance, rbf.lengthscale, bias.variance, Gaussian_noise.variance)."""
        return None if self._gp is None else dict(self._hyper)

    def refit(self):
        """Rebuild the GP from all evidence with the current hyper-parameters (a full factorisation)."""
        if self._gp is None:
            raise RuntimeError('no evidence yet')
        self._refit()

    def _refit(self):
        h = self._hyper
        self._handle.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
        self._log_marginal = self._handle.factorize()
        self._fitted_hyper = dict(h)

    def update(self, x, y, optimize=False):
        """Add evidence and rebuild the GP (gpy_regression.py:286-315)."""
        x = np.asarray(x, dtype=np.float64).reshape((-1, self.input_dim))
        y = np.asarray(y, dtype=np.float64).reshape((-1, 1))
        if self._gp is None:
            self._init_gp(x, y)
        else:
            self._append_and_fit(x, y)
        if optimize:
            self.optimize()

    def optimize(self):
        """Optimise the hyper-parameters (gpy_regression.py:317-323)."""
        logger.debug("Optimizing GP hyperparameters")
        from .hyperopt import optimize_hyperparameters
        try:
            optimize_hyperparameters(self, max_iters=self.max_opt_iters)
        except np.linalg.LinAlgError:
            logger.warning("Numerical error in GP optimization. Stopping optimization")

    # -- attributes -------------------------------------------------------------------------
    @property
    def n_evidence(self):
        return 0 if self._gp is None else self._X.shape[0]

    @property
    def X(self):
        return self._gp.X

    @property
    def Y(self):
        return self._gp.Y

    @property
    def noise(self):
        return self._hyper['noise']

    @property
    def instance(self):
        return self._gp

    def copy(self):
        """Independent copy (own device state), as GPyRegression.copy (gpy_regression.py:352-364)."""
        kopy = copy.copy(self)
        kopy.gp_params = dict(self.gp_params)
        if self._gp is not None:
            kopy._hyper = dict(self._hyper)
            kopy._priors = dict(self._priors)
            kopy._X = self._X.copy()
            kopy._Y = self._Y.copy()
            kopy._gp = _GPShim(kopy)
            kopy._handle = None
            kopy._ensure_capacity(kopy._X.shape[0])
            kopy._handle.set_data(kopy._X, kopy._Y)
            kopy._refit()
        return kopy

    def __getstate__(self):
        st = dict(self.__dict__)
        st['_handle'] = None  # device state is rebuilt on first use after unpickling
        st.pop('_Xbuf', None)   # growth buffers: rebuilt on the next update
        st.pop('_Ybuf', None)
        if st.get('_X') is not None:
            st['_X'], st['_Y'] = np.array(st['_X']), np.array(st['_Y'])
        st['_gp'] = None if self._gp is None else True
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        if self._gp:
            self._gp = _GPShim(self)
            self._ensure_capacity(self._X.shape[0])
            self._handle.set_data(self._X, self._Y)
            self._refit()
