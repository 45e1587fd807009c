"""Weighted statistics of an SMC-ABC population.

`weighted_var` and `weighted_sample_quantile` with the signatures and results of
elfi.methods.utils.weighted_var / weighted_sample_quantile (elfi/methods/utils.py:108-139, 379-411).
SMC calls the first once per round on the accepted parameter sample to set the proposal covariance
(elfi/methods/inference/samplers.py:521-534) and AdaptiveThresholdSMC the second on the distances of
the round.

* weighted_var: two streaming column reductions on the GPU (csrc/wstats.hip, fixed summation
  order).  Floating-point sums in a different order than NumPy's: parity within 1e-12 relative.
* weighted_sample_quantile: the order statistics decide, bit for bit, which sample is returned --
  the comparison `cum < alpha <= cum'` runs on NumPy's sequential cumulative sum of the sorted
  weights, and any other summation order could pick a neighbouring sample when alpha sits on a
  boundary (equal weights, alpha = k / n).  It is an O(n log n) host step on at most a population
  (10^4 - 10^5 values, once per round), kept in NumPy for that exactness; no device kernel is
  involved, so there is nothing it could fall back from.
"""
import numpy as np

from . import _lib


def weighted_var(x, weights=None, ctx=None):
    """Unbiased weighted variance of the columns of x (1-d: of x); weights default to ones."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim not in (1, 2):
        raise ValueError('x must be 1d or 2d with observations in rows')
    X = np.ascontiguousarray(x.reshape(len(x), -1))
    n, m = X.shape
    w = None
    if weights is not None:
        w = np.ascontiguousarray(weights, dtype=np.float64)
        if w.shape != (n,):
            raise ValueError('weights must be 1d with one entry per observation')
    out = np.empty(m, dtype=np.float64)
    ctx = ctx or _lib.default_context()
    ctx.call("elfihip_weighted_var", _lib.ptr(X), n, m, m, _lib.ptr(w) if w is not None else None, _lib.ptr(out))
    return out if x.ndim == 2 else out[0]


def weighted_sample_quantile(x, alpha, weights=None):
    """alpha-quantile of a weighted one-dimensional sample: the first sorted sample whose cumulative
    normalised weight reaches alpha (the last cumulative weight counts as exactly 1)."""
    x = np.asarray(x)
    order = np.argsort(x)
    if alpha == 0:
        return x[order[0]]
    w = np.ones(len(order)) if weights is None else np.asarray(weights)
    w = w / np.sum(w)
    upper = np.cumsum(w[order])
    lower = np.concatenate(([0.0], upper[:-1]))
    upper[-1] = 1.0
    hit = np.flatnonzero((lower < alpha) & (alpha <= upper))[0]
    return x[order][hit]
