"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's weighted population statistics.

Pinned against the real reference through tests/golden/weighted_stats.npz (oracle/make_golden.py weighted).
Only tests/ may import this module; the product path (elfi_amd/weighted.py -> csrc/wstats.hip) never does.
"""
import numpy as np


def weighted_var(x, weights=None):
    """elfi/methods/utils.py:108-139: reliability-weights sample variance of every column.

    V1 = sum w, V2 = sum w^2 (:131-132); xbar = weighted column mean (:134); numerator = sum_i w_i (x_i - xbar)^2
    (:135); s2 = numerator / (V1 - V2 / V1) (:136)."""
    x = np.asarray(x, dtype=np.float64)
    w = np.ones(len(x)) if weights is None else np.asarray(weights, dtype=np.float64)
    v1 = w.sum()
    v2 = np.square(w).sum()
    cols = x.reshape(len(x), -1)
    xbar = (cols * w[:, None]).sum(axis=0) / v1
    num = (np.square(cols - xbar) * w[:, None]).sum(axis=0)
    s2 = num / (v1 - v2 / v1)
    return s2 if x.ndim == 2 else s2[0]


def weighted_sample_quantile(x, alpha, weights=None):
    """elfi/methods/utils.py:379-411: sort (:396); alpha == 0 -> smallest sample (:397-398); otherwise the first sorted
    sample with cum[j] < alpha <= cum[j + 1], cum = [0, cumsum(normalised sorted weights)] with its last entry forced
    to exactly 1 (:400-409)."""
    x = np.asarray(x)
    order = np.argsort(x)
    xs = x[order]
    if alpha == 0:
        return xs[0]
    w = np.ones(len(xs)) if weights is None else np.asarray(weights, dtype=np.float64)
    w = (w / w.sum())[order]
    cum = np.zeros(len(xs) + 1)
    cum[1:] = np.cumsum(w)
    cum[-1] = 1.0
    for j in range(len(xs)):       # plain loop: the oracle is the slow, obviously-right form
        if cum[j] < alpha <= cum[j + 1]:
            return xs[j]
    raise IndexError('no sample reaches alpha = %r' % (alpha,))
