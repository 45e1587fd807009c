"""TEST INFRASTRUCTURE ONLY -- CPU restatement of ELFI's batched distance path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and only as the checker / the timed CPU baseline.  Nothing under elfi_amd/
imports it; the product path has no CPU fallback.

What is restated (reference = /root/reference, elfi v0.8.7):
  distance_as_discrepancy   elfi/model/utils.py:37-52
  elfi.Distance's dist_fn   elfi/model/elfi_model.py:1020-1039  (partial(cdist, metric, p/w/V/VI))
  AdaptiveDistance          elfi/model/elfi_model.py:1047-1151  (init_state, add_data,
                            update_distance, nested_distance)
  summaries of the configs  elfi/examples/ma2.py:40-59 (autocov),
                            elfi/examples/gauss.py:142-173 (ss_mean, ss_var)

The arithmetic itself is third-party in the reference: scipy.spatial.distance.cdist
(SciPy, C++), pinned in this image at scipy 1.15.3 -- the same library is called here,
so this oracle reproduces the reference bit for bit by construction.  Pinning:
tests/test_oracle_pinning.py checks it against the reference's own known answers
(tests/unit/test_elfi_model.py:139-153,186-253 restated, and the doc-printed values
docs/usage/tutorial.rst:396, docs/usage/adaptive_distance.rst:214,372-378 via the
fixtures in tests/golden/ that oracle/make_golden.py produced by running the real
reference under oracle/ref_shim.py).

`cdist_rows_sequential` is an independent plain-loop restatement of the C++ kernels'
accumulation order (left to right over the columns, no FMA); it equals cdist bit for
bit for euclidean(+w), sqeuclidean, cityblock(+w), chebyshev and minkowski(p,+w) and
documents the order the HIP kernels implement.
"""
from functools import partial

import numpy as np
import scipy.spatial.distance


# ----------------------------------------------------------------------------------
# elfi/model/utils.py:37-52
def distance_as_discrepancy(dist, *summaries, observed):
    summaries = np.column_stack(summaries)
    observed = np.concatenate([np.atleast_2d(o) for o in observed], axis=1)
    try:
        d = dist(summaries, observed)
    except ValueError as e:
        # same exception class and hint as the reference (utils.py:42-49), own wording
        raise ValueError('distance node: summary (XA) and observed (XB) data must be at most '
                         '2-d with matching widths (cdist said: {})'.format(e))
    if d.ndim == 2 and d.shape[1] == 1:
        d = d.reshape(-1)
    return d


# elfi/model/elfi_model.py:1020-1039
def make_distance(distance, **kwargs):
    """Return the discrepancy operation elfi.Distance(distance, ..., **kwargs) installs."""
    if isinstance(distance, str):
        cdist_kwargs = dict(metric=distance)
        if distance == 'seuclidean' and 'V' not in kwargs:
            raise ValueError('Parameter V must be specified for distance=seuclidean.')
        if distance == 'mahalanobis' and 'VI' not in kwargs:
            raise ValueError('Parameter VI must be specified for distance=mahalanobis.')
        for key in ['p', 'w', 'V', 'VI']:
            if key in kwargs:
                cdist_kwargs[key] = kwargs.pop(key)
        dist_fn = partial(scipy.spatial.distance.cdist, **cdist_kwargs)
    else:
        dist_fn = distance
    return partial(distance_as_discrepancy, dist_fn)


def cdist_rows(X, y, metric, p=2.0, w=None, V=None, VI=None):
    """cdist(X (n,m), y (1,m)) -> (n,) exactly as the reference calls it."""
    kw = {}
    if metric == 'minkowski':
        kw['p'] = p
    if w is not None:
        kw['w'] = w
    if V is not None:
        kw['V'] = V
    if VI is not None:
        kw['VI'] = VI
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    y = np.atleast_2d(np.asarray(y, dtype=np.float64))
    if X.shape[0] == 0:
        return np.empty(0)
    return scipy.spatial.distance.cdist(X, y, metric=metric, **kw)[:, 0]


def cdist_rows_sequential(X, y, metric, p=2.0, w=None):
    """Plain-loop restatement of SciPy's per-row kernels (small inputs only)."""
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    n, m = X.shape
    out = np.empty(n)
    if metric == 'minkowski':
        if p == 1:
            metric = 'cityblock'
        elif p == 2:
            metric = 'euclidean'
        elif np.isinf(p):
            metric = 'chebyshev'
    for i in range(n):
        s = 0.0
        for j in range(m):
            d = X[i, j] - y[j]
            if metric == 'euclidean':
                t = d * d
                s = s + (w[j] * t if w is not None else t)
            elif metric == 'sqeuclidean':  # SciPy associates (w*d)*d here, w*(d*d) above
                s = s + ((w[j] * d) * d if w is not None else d * d)
            elif metric == 'cityblock':
                t = abs(d)
                s = s + (w[j] * t if w is not None else t)
            elif metric == 'chebyshev':
                t = abs(d)
                if w is not None and w[j] == 0:
                    t = 0.0
                s = t if t > s else s
            elif metric == 'minkowski':
                t = abs(d) ** p
                s = s + (w[j] * t if w is not None else t)
            else:
                raise ValueError(metric)
        if metric == 'euclidean':
            s = np.sqrt(s)
        elif metric == 'minkowski':
            s = s ** (1.0 / p)
        out[i] = s
    return out


# ----------------------------------------------------------------------------------
# elfi/model/elfi_model.py:1047-1151
class AdaptiveDistanceOracle:
    """State machine of elfi.AdaptiveDistance without the node machinery."""

    def __init__(self):
        self.distance = partial(scipy.spatial.distance.cdist, metric='euclidean')
        self.init_state()

    def init_state(self):  # :1088-1094
        self.w = [None]
        self.distance_functions = [partial(self.distance, w=None)]
        self.store = 3 * [None]
        self.init_adaptation_round()

    def init_adaptation_round(self):  # :1096-1102
        self.store[0] = 0
        self.store[1] = 0
        self.store[2] = 0

    def add_data(self, *data):  # :1104-1125
        data = np.column_stack(data)
        self.store[0] += len(data)
        delta_1 = data - self.store[1]
        self.store[1] += np.sum(delta_1, axis=0) / self.store[0]
        delta_2 = data - self.store[1]
        self.store[2] += np.sum(delta_1 * delta_2, axis=0)
        self.scale = np.sqrt(self.store[2] / self.store[0])

    def update_distance(self):  # :1127-1133
        weis = 1 / self.scale
        self.w.append(weis)
        self.init_adaptation_round()
        self.distance_functions.append(partial(self.distance, w=weis**2))

    def nested_distance(self, u, v):  # :1135-1151
        return np.column_stack([d(u, v) for d in self.distance_functions])

    def __call__(self, *summaries, observed):
        return distance_as_discrepancy(self.nested_distance, *summaries, observed=observed)


# ----------------------------------------------------------------------------------
# summaries of the BASELINE configs
def autocov(x, lag=1):  # elfi/examples/ma2.py:40-59
    x = np.atleast_2d(x)
    return np.mean(x[:, lag:] * x[:, :-lag], axis=1)


def ss_mean(y):  # elfi/examples/gauss.py:142-156
    return np.mean(y, axis=1)


def ss_var(y):  # elfi/examples/gauss.py:159-173
    return np.var(y, axis=1)


def MA2(t1, t2, n_obs=100, batch_size=1, random_state=None):  # elfi/examples/ma2.py:11-37
    t1 = np.asanyarray(t1).reshape((-1, 1))
    t2 = np.asanyarray(t2).reshape((-1, 1))
    random_state = random_state or np.random
    w = random_state.randn(batch_size, n_obs + 2)
    return w[:, 2:] + t1 * w[:, 1:-1] + t2 * w[:, :-2]
