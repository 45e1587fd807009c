"""CPU restatement of the Gaussian example model (TEST INFRASTRUCTURE ONLY -- the product path is elfi_amd/csrc/gauss.hip).

Follows elfi/examples/gauss.py:
    gauss(mu, sigma, n_obs, batch_size, random_state)   :11-35   ss.norm.rvs(loc, scale, size=(batch, n_obs), random_state)
    ss_mean(y) = np.mean(y, axis=1)                      :142-156
    ss_var(y)  = np.var(y, axis=1)                       :159-173
and the Distance node built at :133 (elfi.Distance('euclidean', ss_mean, ss_var): elfi/model/elfi_model.py Distance ->
elfi/model/utils.py:37-52 distance_as_discrepancy around scipy.spatial.distance.cdist).

SciPy's rv_continuous.rvs draws `random_state.standard_normal(size)` (norm_gen._rvs) and returns `vals * scale + loc`
[SciPy-upstream, scipy/stats/_distn_infrastructure.py: rvs]; restated here with the draw as an explicit input so that a
device kernel can be given the very same normals.  Pinned by tests/golden/gauss_example.npz (oracle/make_golden.py
make_gauss: the reference's functions run in this container)."""
import numpy as np


def draws(random_state, batch_size, n_obs):
    """The standard normals ss.norm.rvs consumes for size=(batch_size, n_obs)."""
    return random_state.standard_normal((batch_size, n_obs))


def gauss_from_draws(z, mu, sigma):
    mu = np.asanyarray(mu, dtype=float).reshape((-1, 1))          # gauss.py:27-28
    sigma = np.asanyarray(sigma, dtype=float).reshape((-1, 1))
    return z * sigma + mu                                         # rvs: vals * scale + loc


def ss_mean(y):
    return np.mean(y, axis=1)


def ss_var(y):
    return np.var(y, axis=1)


def euclidean_to_observed(sm, sv, observed):
    """distance_as_discrepancy(cdist euclidean): column_stack the summaries, one observed row, cdist's own loop order."""
    s = np.column_stack([sm, sv])
    o = np.asarray(observed, dtype=float).reshape(1, 2)
    out = np.empty(s.shape[0])
    for i in range(s.shape[0]):
        acc = 0.0
        for j in range(2):
            df = s[i, j] - o[0, j]
            acc += df * df
        out[i] = np.sqrt(acc)
    return out
