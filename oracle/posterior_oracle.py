"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the BOLFI posterior for parity tests.

Restates elfi.methods.posteriors.BolfiPosterior (elfi/methods/posteriors.py:88-212) over the NumPy GP
of gp_oracle.Posterior: the likelihood F((h - mu) / sigma) with the NOISY predictive variance and its
gradient.  Pinned by tests/golden/bolfi_posterior.npz, which was recorded from the reference's own class
(oracle/make_golden_posterior.py); never imported by the product package.
"""
import numpy as np
import scipy.stats as ss


class BoxPrior:
    """Uniform prior on the bounds: what ModelPrior yields for independent elfi.Prior('uniform', ...) nodes
    (constant log density inside, -inf outside; its numerical gradient is zero, extensions.py:217-240)."""

    def __init__(self, bounds):
        self.lo = np.array([b[0] for b in bounds], float)
        self.hi = np.array([b[1] for b in bounds], float)

    def rvs(self, size=None, random_state=None):
        # parameter by parameter, as ModelPrior.rvs draws independent uniform nodes (extensions.py:156-174)
        rs = random_state or np.random
        return np.column_stack([rs.uniform(a, b, size or 1) for a, b in zip(self.lo, self.hi)])

    def logpdf(self, x):
        x = np.asarray(x, float).reshape((-1, len(self.lo)))
        inside = np.all((x >= self.lo) & (x <= self.hi), axis=1)
        return np.where(inside, -np.sum(np.log(self.hi - self.lo)), -np.inf)

    def pdf(self, x):
        return np.exp(self.logpdf(x))

    def gradient_logpdf(self, x):
        return np.zeros_like(np.asarray(x, float).reshape((-1, len(self.lo))))


class PosteriorOracle:
    def __init__(self, post, bounds, threshold, prior=None):
        self.post, self.threshold = post, float(threshold)
        self.lo = np.array([b[0] for b in bounds], float)
        self.hi = np.array([b[1] for b in bounds], float)
        self.prior = prior or BoxPrior(bounds)

    def _inside(self, x):
        return np.all((x >= self.lo) & (x <= self.hi), axis=1)

    # posteriors.py:135-156
    def loglik(self, x):
        x = np.asarray(x, float).reshape((-1, len(self.lo)))
        out = np.full(len(x), -np.inf)
        m = self._inside(x)
        if np.any(m):
            mean, var = self.post.predict(x[m], noiseless=False)
            out[m] = ss.norm.logcdf(self.threshold, mean, np.sqrt(var)).squeeze()
        return out

    # posteriors.py:158-190
    def grad_loglik(self, x):
        x = np.asarray(x, float).reshape((-1, len(self.lo)))
        out = np.zeros_like(x)
        m = self._inside(x)
        if np.any(m):
            mean, var = self.post.predict(x[m], noiseless=False)
            std = np.sqrt(var)
            gm, gv = self.post.predictive_gradients(x[m])
            factor = (-gm * std - (self.threshold - mean) * 0.5 * gv / std) / var
            term = (self.threshold - mean) / std
            out[m] = factor * ss.norm.pdf(term) / ss.norm.cdf(term)
        return out

    def logpdf_and_gradient(self, x):
        x = np.asarray(x, float).reshape((-1, len(self.lo)))
        return self.loglik(x) + self.prior.logpdf(x), self.grad_loglik(x) + self.prior.gradient_logpdf(x)
