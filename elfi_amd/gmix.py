"""Gaussian-mixture proposal density of SMC-ABC on the GPU.

`GMDistribution.pdf / logpdf` with the signature and shape conventions of
elfi.methods.utils.GMDistribution (elfi/methods/utils.py:139-198): `x` scalar, 1-d or 2-d with
observations in rows, `means` (N,) or (N, d), a shared covariance `cov` (scalar or matrix),
optional `weights`.  The reference loops over the N components on the host and calls
scipy.stats.multivariate_normal.pdf for each; here all M x N component densities are evaluated in
one kernel (csrc/gmix.hip).  The covariance is factored on the host exactly as SciPy does
(symmetric eigendecomposition, pseudo-determinant), so the only numerical difference is the
device exp().  `rvs` draws on the device as well (csrc/gmix.hip gm_rvs_kernel: the library's counter-based
generator, so a different -- equally valid -- random realisation than the reference's MT19937 stream; the validity
check against the prior runs on the host exactly as in the reference).
"""
import ctypes as C

import numpy as np

from . import _lib
from .priors import PROPOSAL_STREAM_BASE


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

    @classmethod
    def rvs(cls, means, cov=1, weights=None, size=1, prior_logpdf=None, random_state=None, ctx=None):
        """elfi.methods.utils.GMDistribution.rvs (utils.py:199-262) with the draws on the device: `size` variates (one has
        the shape of a mean; size=None: one variate without the enclosing array), redrawn until `prior_logpdf` is finite
        for all of them.  The seed comes from `random_state` (one randint)."""
        means, weights = _normalize_params(means, weights)
        no_wrap = size is None
        size = 1 if no_wrap else int(size)
        d = 1 if means.ndim == 1 else means.shape[1]
        if d > 64:
            raise NotImplementedError('GMDistribution on the GPU supports up to 64 dimensions')
        mu = np.ascontiguousarray(means.reshape(len(means), d), dtype=np.float64)
        c = np.asarray(cov, dtype=np.float64)
        c = c * np.eye(d) if c.ndim == 0 else (np.diag(c) if c.ndim == 1 else c)
        if c.shape != (d, d):
            raise ValueError("Array 'cov' must be square with the dimension of the means (%d)." % d)
        s, u = np.linalg.eigh(c)
        A = np.ascontiguousarray(u * np.sqrt(np.maximum(s, 0.0)))        # A A^T = cov (semi-definite allowed)
        cumw = np.cumsum(weights)
        cumw[np.nonzero(weights)[0][-1]:] = 1.0                           # (rounding: the last component with mass ends at 1)
        rs = random_state or np.random
        seed = int(rs.randint(0, 2 ** 31 - 1))
        ctx = ctx or _lib.default_context()
        output = np.empty((size,) + means.shape[1:])
        n_accepted, trials = 0, 0
        while n_accepted < size:
            n_left = size - n_accepted
            x = np.empty((n_left, d), dtype=np.float64)
            ctx.call("elfihip_gm_rvs", C.c_uint64(seed), C.c_uint64(PROPOSAL_STREAM_BASE + 2 * trials), n_left, d, _lib.ptr(mu), mu.shape[0],
                     _lib.ptr(cumw), _lib.ptr(A), _lib.ptr(x))
            x = x.reshape((n_left,) + means.shape[1:])
            if prior_logpdf is not None:
                x = x[np.isfinite(prior_logpdf(x))]
            output[n_accepted:n_accepted + len(x)] = x
            n_accepted += len(x)
            trials += 1
            if trials == 100:
                import logging
                logging.getLogger(__name__).warning(
                    "SMC: It appears to be difficult to find enough valid proposals with prior pdf > 0. ELFI will keep "
                    "trying, but you may wish to kill the process and adjust the model priors.")
        return output[0] if no_wrap else output
