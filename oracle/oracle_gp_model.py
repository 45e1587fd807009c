"""TEST INFRASTRUCTURE ONLY -- a CPU stand-in for GPyRegression built on the oracle.

`OracleGPRegression` has the GPyRegression duck-type (elfi/methods/bo/gpy_regression.py:15-364) with
the arithmetic of oracle/gp_oracle.py and oracle/gp_hyper_oracle.py.  It exists so that the REAL
reference BOLFI loop (elfi/methods/inference/bolfi.py, with the reference's own LCBSC and
scipy L-BFGS-B) can be run in the build container, where GPy is not installable, to record golden
traces (oracle/make_golden_gp.py: gp_bolfi_trace.npz).  Never imported by elfi_amd/.
"""
import copy

import numpy as np

import gp_hyper_oracle as HO
import gp_oracle as G


class OracleGPRegression:
    def __init__(self, parameter_names=None, bounds=None, optimizer="scg", max_opt_iters=50, **gp_params):
        self.parameter_names = parameter_names
        self.input_dim = len(parameter_names)
        self.bounds = [bounds[n] for n in parameter_names]
        self.optimizer, self.max_opt_iters, self.gp_params = optimizer, max_opt_iters, gp_params
        self.is_sampling = False
        self._post = None
        self._X = self._Y = None
        self.hyper = None
        self.priors = None
        self.log = []        # ('update', x, y, optimize, hyper_after) records

    def _refit(self):
        self._post = G.Posterior(self._X, self._Y, **self.hyper)

    def predict(self, x, noiseless=False):
        x = np.asanyarray(x).reshape((-1, self.input_dim))
        if self._post is None:
            return np.zeros((x.shape[0], 1)), np.ones((x.shape[0], 1))
        return self._post.predict(x, noiseless=noiseless)

    def predict_mean(self, x):
        return self.predict(x)[0]

    def predictive_gradients(self, x):
        x = np.asanyarray(x).reshape((-1, self.input_dim))
        if self._post is None:
            return np.zeros((x.shape[0], self.input_dim)), np.zeros((x.shape[0], self.input_dim))
        return self._post.predictive_gradients(x)

    def predictive_gradient_mean(self, x):
        return self.predictive_gradients(x)[0]

    def update(self, x, y, optimize=False):
        x = np.asarray(x, float).reshape((-1, self.input_dim))
        y = np.asarray(y, float).reshape((-1, 1))
        if self._post is None:
            self._X, self._Y = x, y
            self.hyper = G.initial_hyper(y)
            self.priors = G.default_priors(self.bounds, y)
        else:
            self._X, self._Y = np.r_[self._X, x], np.r_[self._Y, y]
        if optimize:
            self.hyper, _ = HO.optimize(self._X, self._Y, self.hyper, self.priors, max_iters=self.max_opt_iters)
        self._refit()
        self.log.append((x.copy(), y.copy(), bool(optimize), dict(self.hyper)))

    def optimize(self):
        self.hyper, _ = HO.optimize(self._X, self._Y, self.hyper, self.priors, max_iters=self.max_opt_iters)
        self._refit()

    @property
    def n_evidence(self):
        return 0 if self._post is None else self._X.shape[0]

    @property
    def X(self):
        return self._X

    @property
    def Y(self):
        return self._Y

    @property
    def noise(self):
        return self.hyper['noise']

    @property
    def instance(self):
        return self._post

    @property
    def _gp(self):
        """What the reference's ExpIntVar reaches for (acquisition.py:754,770): `model._gp.kern.K(X, X2)`, the PRIOR kernel
        matrix [GPy-upstream: RBF + Bias]."""
        post = self._post

        class _Kern:
            @staticmethod
            def K(X, X2=None):
                X = np.atleast_2d(np.asarray(X, float))
                X2 = None if X2 is None else np.atleast_2d(np.asarray(X2, float))
                return G.kern_K(X, X2, post.var, post.ls, post.bias)

        class _Shim:
            kern = _Kern

        return _Shim

    def copy(self):
        return copy.deepcopy(self)
