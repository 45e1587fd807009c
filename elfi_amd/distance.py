"""Host side of the batched distance path: ELFI's operation interface over the C ABI.

Mirrors, name for name, what reference ELFI builds for a distance node
(/root/reference, elfi v0.8.7):

  distance_as_discrepancy(dist, *summaries, observed)   elfi/model/utils.py:37-52
  elfi.Distance(distance, *summaries, p=, w=, V=, VI=)  elfi/model/elfi_model.py:974-1044
  elfi.AdaptiveDistance(*summaries)                     elfi/model/elfi_model.py:1047-1151

The arithmetic (SciPy's cdist in the reference) runs in libelfihip.so on the GPU;
this module only normalises shapes/dtypes the way the reference does and hands plain
pointers to the C ABI.  All callables here are picklable (they hold plain data and
create their HIP context lazily per process), because ELFI pickles operations into
worker processes (elfi/clients/multiprocessing.py:50) and into saved models
(elfi/model/elfi_model.py:401-438).

There is no CPU fallback: without libelfihip.so / a GPU the calls raise.
"""
import ctypes as C

import numpy as np

from . import _lib

_SHAPE_HINT = ('distance node: summary (XA) and observed (XB) data must be at most 2-d with '
               'matching widths ({})')


def _as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _metric_id(metric):
    try:
        return _lib.METRICS[metric]
    except KeyError:
        raise ValueError("Unknown Distance Metric: %s" % metric)


def _metric_aux(metric, m, p, w, V, VI):
    """Validate the optional cdist arguments the way SciPy does; return (aux, p)."""
    aux = None
    if metric == 'seuclidean':
        if V is None:
            raise ValueError('Parameter V must be specified for distance=seuclidean.')
        aux = _as_f64(V).reshape(-1)
        if aux.shape[0] != m:
            raise ValueError('V must be a 1-D array of the same dimension as the vectors.')
    elif metric == 'mahalanobis':
        if VI is None:
            raise ValueError('Parameter VI must be specified for distance=mahalanobis.')
        aux = _as_f64(VI)
        if aux.shape != (m, m):
            raise ValueError('VI must be a (m, m) matrix matching the vectors.')
    elif w is not None:
        aux = _as_f64(w).reshape(-1)
        if aux.shape[0] != m:
            raise ValueError('Weights must have same size as input vector. '
                             '%d vs. %d' % (aux.shape[0], m))
        if np.any(aux < 0):
            raise ValueError('Input weights should be all non-negative')
    if metric == 'minkowski':
        p = float(p)
        if p <= 0:
            raise ValueError('p must be greater than 0')
    else:
        p = 2.0
    return aux, p


def cdist_rows(X, y, metric='euclidean', p=2.0, w=None, V=None, VI=None, ctx=None):
    """Distances of the rows of X (n, m) to the single observed row y -> (n,) float64.

    GPU replacement for scipy.spatial.distance.cdist(X, Y (1, m), metric, ...)[:, 0] as
    called by elfi/model/elfi_model.py:1037.
    """
    X0 = X
    X = np.asarray(X)
    if X.ndim != 2:
        raise ValueError('XA must be a 2-dimensional array.')
    y = np.asarray(y)
    if y.ndim != 2:
        raise ValueError('XB must be a 2-dimensional array.')
    if y.shape[0] != 1:
        raise ValueError('the observed summaries must be a single row (got %d rows)' % y.shape[0])
    if X.shape[1] != y.shape[1]:
        raise ValueError('XA and XB must have the same number of columns '
                         '(i.e. feature dimension.)')
    n, m = X.shape
    if m < 1:
        raise ValueError('XA must have at least one column')
    mid = _metric_id(metric)
    aux, p = _metric_aux(metric, m, p, w, V, VI)
    # row-major with unit inner stride is passed as is (ldx = row pitch); anything else is copied
    if X.dtype != np.float64 or X.strides[1] != 8 or X.strides[0] % 8 or X.strides[0] < 8 * m:
        X = _as_f64(X)
    ldx = X.strides[0] // 8 if n > 1 else m
    y = _as_f64(y).reshape(-1)
    ctx = ctx or _lib.default_context()
    if metric == 'euclidean' and _lib.rows_epoch_of(X0, ctx) is not None:
        # rows a device-side simulator just returned (randn_rows): the pass runs on the device copy, no upload
        # (one weight row: sqrt(sum w (x - y)^2) in cdist's order, bit-identical to the row kernel)
        d, _ = adaptive_batch(X0, y, (np.ones(m) if aux is None else aux).reshape(1, m), ctx=ctx)
        return _lib.alias_kept(d.reshape(-1), d)
    out = np.empty(n, dtype=np.float64)
    ctx.call("elfihip_dist_rows", mid, _lib.ptr(X), n, m, ldx, _lib.ptr(y), _lib.ptr(aux),
             C.c_double(p), _lib.ptr(out))
    return _lib.remember_kept(out, ctx)


def cdist_cols(cols, y, metric='euclidean', p=2.0, w=None, V=None, ctx=None):
    """Same distances from m separate length-n columns (no column_stack on the host)."""
    m = len(cols)
    if m < 1:
        raise ValueError('at least one summary column is required')
    cols = [_as_f64(c).reshape(-1) for c in cols]
    n = cols[0].shape[0]
    for c in cols:
        if c.shape[0] != n:
            raise ValueError('all input arrays must have the same length')
    y = _as_f64(y).reshape(-1)
    if y.shape[0] != m:
        raise ValueError('XA and XB must have the same number of columns '
                         '(i.e. feature dimension.)')
    mid = _metric_id(metric)
    if metric == 'mahalanobis':
        raise ValueError('mahalanobis needs the stacked (row-major) form')
    aux, p = _metric_aux(metric, m, p, w, V, None)
    out = np.empty(n, dtype=np.float64)
    arr = (C.c_void_p * m)(*[c.ctypes.data for c in cols])
    ctx = ctx or _lib.default_context()
    ctx.call("elfihip_dist_cols", mid, arr, m, n, _lib.ptr(y), _lib.ptr(aux), C.c_double(p),
             _lib.ptr(out))
    return _lib.remember_kept(out, ctx)


def nested_weighted_euclidean(X, y, W, ctx=None):
    """(n, K) weighted euclidean distances, one column per weight vector in W (K, m).

    GPU replacement for AdaptiveDistance.nested_distance's K cdist calls + column_stack
    (elfi/model/elfi_model.py:1135-1151).
    """
    X = np.asarray(X)
    if X.ndim != 2:
        raise ValueError('XA must be a 2-dimensional array.')
    X = _as_f64(X)
    n, m = X.shape
    y = _as_f64(y).reshape(-1)
    if y.shape[0] != m:
        raise ValueError('XA and XB must have the same number of columns '
                         '(i.e. feature dimension.)')
    W = _as_f64(W)
    if W.ndim != 2 or W.shape[1] != m:
        raise ValueError('W must be (K, %d)' % m)
    K = W.shape[0]
    out = np.empty((n, K), dtype=np.float64)
    ctx = ctx or _lib.default_context()
    ctx.call("elfihip_dist_multiw", _lib.ptr(X), n, m, m, _lib.ptr(y), _lib.ptr(W), K, _lib.ptr(out))
    return _lib.remember_kept(out, ctx)


def welford_update(X, count, mean, M2, ctx=None):
    """One AdaptiveDistance.add_data step on the GPU; returns the new (count, mean, M2)."""
    X = _as_f64(X)
    if X.ndim != 2:
        raise ValueError('data must be 2-dimensional')
    n, m = X.shape
    mean = np.array(np.broadcast_to(np.asarray(mean, dtype=np.float64), (m,)))
    M2 = np.array(np.broadcast_to(np.asarray(M2, dtype=np.float64), (m,)))
    cnt = C.c_int64(int(count))
    ctx = ctx or _lib.default_context()
    ctx.call("elfihip_welford_update", _lib.ptr(X), n, m, m, C.byref(cnt), _lib.ptr(mean), _lib.ptr(M2))
    return cnt.value, mean, M2


def adaptive_batch(X, y, W, store=None, state=None, row_base=None, distances=True, ctx=None):
    """One AdaptiveDistance batch in ONE read of its rows on the GPU (csrc/adaptive.hip): the (n, K) nested distances
    under the weight rows W (K, m) (AdaptiveDistance.nested_distance, elfi/model/elfi_model.py:1135-1151), the batch
    folded into the running column statistics `store` = (count, mean, M2) (AdaptiveDistance.add_data, :1104-1125) and,
    with `state` (a selection.RunningBest), what Rejection._merge_batch keeps of the batch
    (elfi/methods/inference/samplers.py:209-237; rows numbered row_base + row, default: the rows pushed so far).
    Returns (distances or None, new store or None)."""
    X0 = X
    X = np.asarray(X)
    if X.ndim != 2:
        raise ValueError('XA must be a 2-dimensional array.')
    n, m = X.shape
    if X.dtype != np.float64 or X.strides[1] != 8 or X.strides[0] % 8 or X.strides[0] < 8 * m:
        X = _as_f64(X)
    ldx = X.strides[0] // 8 if n > 1 else m
    y = _as_f64(y).reshape(-1)
    if y.shape[0] != m:
        raise ValueError('XA and XB must have the same number of columns '
                         '(i.e. feature dimension.)')
    W = _as_f64(W)
    if W.ndim != 2 or W.shape[1] != m:
        raise ValueError('W must be (K, %d)' % m)
    K = W.shape[0]
    out = np.empty((n, K), dtype=np.float64) if distances else None
    cnt = mean = M2 = None
    if store is not None:
        cnt = C.c_int64(int(store[0]))
        mean = np.array(np.broadcast_to(np.asarray(store[1], dtype=np.float64), (m,)))
        M2 = np.array(np.broadcast_to(np.asarray(store[2], dtype=np.float64), (m,)))
    ctx = ctx or (state.ctx if state is not None else _lib.default_context())
    base = int(row_base if row_base is not None else (state.n_pushed if state is not None else 0))
    # rows that a device-side simulator (elfi_amd.randn_rows) just returned are still on the device: no upload
    epoch = _lib.rows_epoch_of(X0, ctx)
    rc = _lib.ERR_STATE
    if epoch is not None:
        rc = ctx.lib.elfihip_adaptive_push_kept(ctx.handle, state.h if state is not None else None, epoch, _lib.ptr(y),
                                               _lib.ptr(W), K, _lib.ptr(out), C.byref(cnt) if store is not None else None,
                                               _lib.ptr(mean), _lib.ptr(M2), base)
        if rc not in (_lib.OK, _lib.ERR_STATE):
            _lib.check(ctx.handle, rc)
    if rc != _lib.OK:
        ctx.call("elfihip_adaptive_push", state.h if state is not None else None, _lib.ptr(X), n, m, ldx, _lib.ptr(y),
                 _lib.ptr(W), K, _lib.ptr(out), C.byref(cnt) if store is not None else None, _lib.ptr(mean), _lib.ptr(M2),
                 base)
    if state is not None:
        state.n_pushed += n
    if out is not None:
        _lib.remember_kept(out, ctx)
    return out, ((cnt.value, mean, M2) if store is not None else None)


def randn_rows(loc, scale, seed=0, stream=0, ctx=None):
    """The synthetic Gaussian simulator of BASELINE configs[1] / [3] on the device: (n, m) = loc[:, None] + scale[None, :] *
    standard normals (Philox4x32-10 keyed by `seed`, counter stream `stream`; m even).  The rows come back as a NumPy array
    AND stay on the device for the distance call that follows (adaptive_batch recognises the array)."""
    loc = np.ascontiguousarray(np.asarray(loc, dtype=np.float64).reshape(-1))
    scale = np.ascontiguousarray(np.asarray(scale, dtype=np.float64).reshape(-1))
    n, m = loc.shape[0], scale.shape[0]
    if m < 2 or m % 2:
        raise ValueError('the number of columns must be even (got %d)' % m)
    out = _lib.pinned.array((n, m))      # page-locked (recycled): the rows come down at PCIe speed
    ctx = ctx or _lib.default_context()
    ctx.call("elfihip_randn_rows", C.c_uint64(int(seed)), C.c_uint64(int(stream)), n, m, _lib.ptr(loc), _lib.ptr(scale),
             _lib.ptr(out))
    return _lib.remember_rows(out, ctx)


class HipDistance:
    """Callable `dist(X (n,m), Y (1,m)) -> (n,)`: what elfi.Distance accepts as a callable
    metric (elfi/model/elfi_model.py:987-991).  Use as
    `elfi.Distance(elfi_amd.HipDistance('euclidean', w=...), S1, S2)`."""

    def __init__(self, metric='euclidean', p=2.0, w=None, V=None, VI=None, device=-1):
        _metric_id(metric)
        if metric == 'seuclidean' and V is None:
            raise ValueError('Parameter V must be specified for distance=seuclidean.')
        if metric == 'mahalanobis' and VI is None:
            raise ValueError('Parameter VI must be specified for distance=mahalanobis.')
        self.metric = metric
        self.p = p
        self.w = None if w is None else np.array(w, dtype=np.float64)
        self.V = None if V is None else np.array(V, dtype=np.float64)
        self.VI = None if VI is None else np.array(VI, dtype=np.float64)
        self.device = device

    def __call__(self, X, Y):
        return cdist_rows(X, Y, self.metric, self.p, self.w, self.V, self.VI,
                          ctx=_lib.default_context(self.device))

    def __repr__(self):
        return 'HipDistance(%r)' % self.metric


class HipDiscrepancy:
    """Operation for an elfi.Discrepancy node: `op(*summaries, observed) -> (n,)`.

    Drop-in for `partial(distance_as_discrepancy, dist_fn)` (elfi/model/utils.py:37-52,
    elfi/model/elfi_model.py:1041) that also removes the host-side np.column_stack:
    when every summary is a plain (n,) vector the columns go to the GPU as they are.
    """

    def __init__(self, metric='euclidean', p=2.0, w=None, V=None, VI=None, device=-1):
        self.dist = HipDistance(metric, p=p, w=w, V=V, VI=VI, device=device)

    def __call__(self, *summaries, observed):
        if not summaries:
            raise ValueError('This node requires that at least one parent is specified.')
        obs = [np.atleast_2d(o) for o in observed]
        if any(o.ndim > 2 for o in obs):
            raise ValueError(_SHAPE_HINT.format('observed has more than 2 dimensions'))
        observed = np.concatenate(obs, axis=1)
        d = self.dist
        summaries = [np.asarray(s) for s in summaries]
        try:
            if any(s.ndim > 2 for s in summaries):
                raise ValueError('XA must be a 2-dimensional array.')
            if d.metric != 'mahalanobis' and all(s.ndim == 1 for s in summaries):
                return cdist_cols(summaries, observed, d.metric, d.p, d.w, d.V,
                                  ctx=_lib.default_context(d.device))
            if len(summaries) == 1 and summaries[0].ndim == 2:
                X = summaries[0]
            else:
                X = np.column_stack(summaries)
            return d(X, observed)
        except ValueError as e:
            raise ValueError(_SHAPE_HINT.format(e))

    def __repr__(self):
        return 'HipDiscrepancy(%r)' % self.dist.metric


class AdaptiveDistanceState:
    """GPU twin of the state machine inside elfi.AdaptiveDistance
    (elfi/model/elfi_model.py:1088-1151): same method names, same `state` keys."""

    def __init__(self, device=-1):
        self.device = device
        self.state = {}
        self.init_state()

    def _ctx(self):
        return _lib.default_context(self.device)

    def init_state(self):
        self.state['w'] = [None]
        self.state['store'] = 3 * [None]
        self.init_adaptation_round()

    def init_adaptation_round(self):
        if 'store' not in self.state:
            self.init_state()
        self.state['store'][0] = 0
        self.state['store'][1] = 0
        self.state['store'][2] = 0

    def add_data(self, *data):
        data = np.column_stack(data)
        st = self.state['store']
        st[0], st[1], st[2] = welford_update(data, st[0], st[1], st[2], ctx=self._ctx())
        self.state['scale'] = np.sqrt(st[2] / st[0])

    def update_distance(self):
        weis = 1 / self.state['scale']
        self.state['w'].append(weis)
        self.init_adaptation_round()

    def weight_matrix(self, m):
        """(K, m) cdist weights: ones for the unweighted first function, then (1/scale)^2."""
        rows = [np.ones(m) if w is None else np.asarray(w, dtype=np.float64) ** 2
                for w in self.state['w']]
        return np.vstack(rows)

    def nested_distance(self, u, v):
        u = np.asarray(u)
        if u.ndim != 2:
            raise ValueError('XA must be a 2-dimensional array.')
        return nested_weighted_euclidean(u, v, self.weight_matrix(u.shape[1]), ctx=self._ctx())

    def __call__(self, *summaries, observed):
        """The node operation: distance_as_discrepancy(self.nested_distance, ...)."""
        obs = np.concatenate([np.atleast_2d(o) for o in observed], axis=1)
        try:
            X = summaries[0] if len(summaries) == 1 and np.ndim(summaries[0]) == 2 \
                else np.column_stack(summaries)
            d = self.nested_distance(X, obs)
        except ValueError as e:
            raise ValueError(_SHAPE_HINT.format(e))
        if d.ndim == 2 and d.shape[1] == 1:
            d = d.reshape(-1)
        return d
