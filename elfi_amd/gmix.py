"""Gaussian-mixture proposal density of SMC-ABC on the GPU.

`GMDistribution.pdf / logpdf` with the signature and shape conventions of
elfi.methods.utils.GMDistribution (elfi/methods/utils.py:139-198): `x` scalar, 1-d or 2-d with
observations in rows, `means` (N,) or (N, d), a shared covariance `cov` (scalar or matrix),
optional `weights`.  The reference loops over the N components on the host and calls
scipy.stats.multivariate_normal.pdf for each; here all M x N component densities are evaluated in
one kernel (csrc/gmix.hip).  The covariance is factored on the host exactly as SciPy does
(symmetric eigendecomposition, pseudo-determinant), so the only numerical difference is the
device exp().  Sampling (`rvs`) is host-side random-number work and stays with the reference.
"""
import ctypes as C

import numpy as np

from . import _lib


def _normalize_params(means, weights):
    # elfi/methods/utils.py:236-246
    means = np.atleast_1d(np.squeeze(means))
    if means.ndim > 2:
        raise ValueError('means.ndim = {} but must be at most 2.'.format(means.ndim))
    if weights is None:
        weights = np.ones(len(means))
    weights = np.asarray(weights, dtype=np.float64)
    weights = weights / np.sum(weights)
    return means, weights


def _psd_factor(cov, d):
    """U and log pdet as in scipy.stats._multivariate._PSD (allow_singular=False)."""
    cov = np.asarray(cov, dtype=np.float64)
    if cov.ndim == 0:
        cov = cov * np.eye(d)
    elif cov.ndim == 1:
        cov = np.diag(cov)
    if cov.shape != (d, d):
        raise ValueError("Array 'cov' must be square with the dimension of the means (%d)." % d)
    s, u = np.linalg.eigh(cov)
    # scipy.stats._multivariate._eigvalsh_to_eps for float64: 1e6 * eps * max|s|, no dimension factor
    eps = 1e6 * np.finfo(np.float64).eps * np.max(np.abs(s)) if s.size else 0.0
    eps = max(eps, 0.0)
    if np.min(s) < -eps:
        raise ValueError('the input matrix must be positive semidefinite')
    if np.any(s <= eps):
        raise np.linalg.LinAlgError('When `allow_singular is False`, the input matrix must be symmetric '
                                    'positive definite.')
    U = u * np.sqrt(1.0 / s)
    return np.ascontiguousarray(U), float(np.sum(np.log(s))), len(s)


class GMDistribution:
    """Gaussian mixture with a shared covariance; density on the GPU."""

    @classmethod
    def pdf(cls, x, means, cov=1, weights=None, ctx=None):
        means, weights = _normalize_params(means, weights)
        ndim = np.asanyarray(x).ndim
        if means.ndim == 1:
            xa = np.atleast_1d(np.asarray(x, dtype=np.float64)).reshape(-1, 1)
            mu = means.reshape(-1, 1)
        else:
            xa = np.atleast_2d(np.asarray(x, dtype=np.float64))
            mu = means
        d = mu.shape[1]
        if xa.shape[1] != d:
            raise ValueError('x has %d columns but the means have %d' % (xa.shape[1], d))
        if d > 64:
            raise NotImplementedError('GMDistribution on the GPU supports up to 64 dimensions')
        U, log_pdet, rank = _psd_factor(cov, d)
        xa = np.ascontiguousarray(xa)
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        out = np.empty(xa.shape[0], dtype=np.float64)
        ctx = ctx or _lib.default_context()
        ctx.call("elfihip_gm_pdf", _lib.ptr(xa), xa.shape[0], d, _lib.ptr(mu), mu.shape[0], _lib.ptr(weights),
                 _lib.ptr(U), C.c_double(rank * np.log(2 * np.pi) + log_pdet), _lib.ptr(out))
        if ndim == 0 or (ndim == 1 and means.ndim == 2):
            return out.squeeze()
        return out

    @classmethod
    def logpdf(cls, x, means, cov=1, weights=None, ctx=None):
        return np.log(cls.pdf(x, means=means, cov=cov, weights=weights, ctx=ctx))
