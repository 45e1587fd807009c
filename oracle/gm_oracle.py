"""TEST INFRASTRUCTURE ONLY -- CPU restatement of GMDistribution.pdf / logpdf
(elfi/methods/utils.py:139-198, _normalize_params :236-246).  The per-component density is
scipy.stats.multivariate_normal.pdf, the same third-party call the reference makes (SciPy 1.15.3 in
this image), so the restatement reproduces the reference by construction; tests/golden/gm_pdf.npz
holds outputs of the real reference class for pinning (oracle/make_golden.py gm)."""
import numpy as np
import scipy.stats as ss


def normalize_params(means, weights):
    means = np.atleast_1d(np.squeeze(means))
    if means.ndim > 2:
        raise ValueError('means.ndim = {} but must be at most 2.'.format(means.ndim))
    if weights is None:
        weights = np.ones(len(means))
    weights = weights / np.sum(weights)   # normalize_weights
    return means, weights


def pdf(x, means, cov=1, weights=None):
    means, weights = normalize_params(means, weights)
    ndim = np.asanyarray(x).ndim
    if means.ndim == 1:
        x = np.atleast_1d(x)
    if means.ndim == 2:
        x = np.atleast_2d(x)
    d = np.zeros(len(x))
    for m, w in zip(means, weights):
        d += w * ss.multivariate_normal.pdf(x, mean=m, cov=cov)
    if ndim == 0 or (ndim == 1 and means.ndim == 2):
        return d.squeeze()
    return d


def logpdf(x, means, cov=1, weights=None):
    return np.log(pdf(x, means=means, cov=cov, weights=weights))
