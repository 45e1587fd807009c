"""Keeping the best samples of a batch on the GPU.

Reference: Rejection._merge_batch (elfi/methods/inference/samplers.py:209-237) copies the whole
batch behind its n_samples best rows and runs np.argsort over n_samples + batch_size distances,
every batch, on the host.  `smallest_k` returns the k smallest distances of a batch and their row
numbers (ascending by (distance, row); NaN last), computed by libelfihip.so's radix select, and
`merge_batch` is the sampler-state update built on it: same resulting sample (the n_samples
smallest so far, sorted), but only k rows of every batch array are touched on the host.

Tie-breaking differs from the reference only where NumPy's default quicksort leaves the order
unspecified (equal distances): here the lower row / the earlier batch wins.
"""
import ctypes as C

import numpy as np

from . import _lib


def smallest_k(d, k, ctx=None):
    """(values, rows) of the k smallest entries of d.  d: (n,) or nested (n, K) -- the LAST column
    decides, as in samplers.py:233."""
    d = np.asarray(d)
    if d.ndim == 2:
        d = d[:, -1]
    elif d.ndim != 1:
        raise ValueError('distances must be (n,) or (n, K)')
    d = np.ascontiguousarray(d, dtype=np.float64)
    n = d.shape[0]
    k = int(min(max(int(k), 0), n))
    vals = np.empty(k, dtype=np.float64)
    idx = np.empty(k, dtype=np.int64)
    if k:
        ctx = ctx or _lib.default_context()
        ctx.call("elfihip_topk_smallest", _lib.ptr(d), n, 1, k, _lib.ptr(vals), _lib.ptr(idx))
    return vals, idx


def merge_batch(samples, batch, discrepancy_name, n_samples, threshold=None, ctx=None):
    """One Rejection._merge_batch step on a sample state of exactly n_samples rows.

    samples: dict name -> array with n_samples rows (distances initialised to +inf), or None on the
    first call; batch: dict name -> array with batch_size rows.  Returns the new dict (sorted by the
    discrepancy, last column if nested)."""
    d = np.asarray(batch[discrepancy_name])
    key = d[:, -1] if d.ndim == 2 else d
    if threshold is not None:
        # acceptance condition of samplers.py:219-225 over the WHOLE batch first (every nested column <= threshold),
        # then the k smallest among the accepted rows: a rejected row's key becomes NaN, which the selection orders
        # last, so it can never displace an accepted row that ranks beyond n_samples by the last column alone
        accepted = np.all(np.atleast_2d(np.transpose(d <= threshold)), axis=0)
        key = np.where(accepted, key, np.nan)
        n_acc = int(np.count_nonzero(accepted))
    else:
        n_acc = key.shape[0]
    vals, rows = smallest_k(key, min(n_samples, n_acc), ctx=ctx)
    if samples is None:
        samples = {}
        for name, v in batch.items():
            v = np.asarray(v)
            shape = (n_samples,) + v.shape[1:]
            samples[name] = np.full(shape, np.inf, dtype=v.dtype) if name == discrepancy_name \
                else np.empty(shape, dtype=v.dtype)
    cur = np.atleast_2d(np.transpose(samples[discrepancy_name]))[-1]
    new = np.atleast_2d(np.transpose(d[rows]))[-1]
    both = np.concatenate([cur, new])
    order = np.argsort(both, kind='stable')[:n_samples]      # earlier rows win ties
    out = {}
    for name, v in samples.items():
        merged = np.concatenate([v, np.asarray(batch[name])[rows]], axis=0)
        out[name] = merged[order]
    return out


class RunningBest:
    """The sampler's running best-k on the GPU, updated by the distance pass itself (include/elfihip.h:
    elfihip_reject_*; csrc/reject.hip).

    What Rejection._merge_batch (samplers.py:209-237) maintains batch after batch -- the n_samples smallest distances
    seen so far -- as device state: `push(X, y)` computes a batch's distances (returned, as the Distance node must
    return them) and folds the batch in during the same pass; `result()` gives the k best (distance, row) pairs,
    ascending, ties to the earlier row.  Row numbers are global (row_base + row inside the batch): the host looks the
    accepted rows' parameters and summaries up in its own batch store.  k <= 2048: device state, asynchronous merges;
    larger k (up to 2^20): the candidates of every push are merged into a sorted host copy (csrc/reject.hip).
    `accept`: the acceptance threshold of a threshold objective (samplers.py:219-225) -- a row takes part only if every
    nested column of its distance is <= accept."""

    def __init__(self, k, metric='euclidean', w=None, p=2.0, ctx=None, accept=None):
        self.ctx = ctx or _lib.default_context()
        self.lib = self.ctx.lib
        self.k = int(k)
        if metric not in _lib.METRICS:
            raise ValueError('unknown metric %r' % (metric,))
        self.metric = _lib.METRICS[metric]
        self.aux = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
        self.p = float(p)
        h = C.c_void_p()
        self.ctx.call("elfihip_reject_create", self.k, C.byref(h))
        self.h = h
        self.n_pushed = 0
        self.accept = None
        if accept is not None:
            self.set_accept(accept)

    def close(self):
        if getattr(self, 'h', None) is not None:
            self.lib.elfihip_reject_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != _lib.OK:
            _lib._raise(self.lib, self.ctx.handle, rc)

    def reset(self):
        self._check(self.lib.elfihip_reject_reset(self.h))
        self.n_pushed = 0

    def set_accept(self, threshold):
        """Acceptance threshold (None removes it); before the first push or after reset().  A sequence is one threshold
        per nested column -- what AdaptiveDistanceSMC hands its Rejection (samplers.py:657-660: [inf, threshold of
        population 1, ...], compared column by column, samplers.py:222-223)."""
        if threshold is not None and np.ndim(threshold) > 0:
            t = np.ascontiguousarray(threshold, dtype=np.float64).reshape(-1)
            self._check(self.lib.elfihip_reject_set_accept_cols(self.h, t.shape[0], _lib.ptr(t)))
        else:
            self._check(self.lib.elfihip_reject_set_accept(self.h, 0 if threshold is None else 1,
                                                           0.0 if threshold is None else float(threshold)))
        self.accept = threshold

    def push_distances(self, d, row_base=None):
        """Fold a batch whose distances exist already (host array (n,) or nested (n, K): ranked by the last column).  When
        `d` is the very array a distance call on this context just returned, the device copy that call left is folded
        in instead (no upload: include/elfihip.h, elfihip_reject_push_kept)."""
        epoch = _lib.kept_epoch_of(d, self.ctx)
        if epoch is not None and self.ctx.kept_shape() != (epoch, len(d), 1 if np.ndim(d) == 1 else np.shape(d)[1]):
            epoch = None        # the device copy is not (or no longer) this array's: upload it
        if epoch is not None:
            base = self.n_pushed if row_base is None else int(row_base)
            rc = self.lib.elfihip_reject_push_kept(self.h, epoch, base)
            if rc == _lib.OK:
                self.n_pushed += len(d)
                self.kept_pushes = getattr(self, 'kept_pushes', 0) + 1
                return
            if rc != _lib.ERR_STATE:
                self._check(rc)
        d = np.ascontiguousarray(d, dtype=np.float64)
        if d.ndim == 1:
            d = d.reshape(-1, 1)
        elif d.ndim != 2:
            raise ValueError('distances must be (n,) or (n, K)')
        n, K = d.shape
        base = self.n_pushed if row_base is None else int(row_base)
        self._check(self.lib.elfihip_reject_push(self.h, _lib.ptr(d), n, K, base))
        self.n_pushed += n

    def meta(self):
        """(k-th distance so far (+inf while fewer than k rows are in), rows accepted since the previous call, in total)."""
        kth = C.c_double()
        last, total = C.c_int64(), C.c_int64()
        self._check(self.lib.elfihip_reject_meta(self.h, C.byref(kth), None, C.byref(last), C.byref(total)))
        return kth.value, last.value, total.value

    def push(self, X, y, row_base=None, return_distances=True):
        """Distances of the rows of X (n, m) to y (1, m) -- returned, shape (n,) -- with the state updated on the way.
        row_base: global number of the batch's first row (default: the rows pushed so far).  return_distances=False:
        nothing but the state's k rows ever leaves the GPU (returns None)."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        if X.ndim != 2 or X.shape[1] != y.shape[0]:
            raise ValueError('X must be (n, m) and y (1, m)')
        n, m = X.shape
        out = np.empty(n, dtype=np.float64) if return_distances else None
        base = self.n_pushed if row_base is None else int(row_base)
        self._check(self.lib.elfihip_reject_push_rows(self.h, self.metric, _lib.ptr(X), n, m, m, _lib.ptr(y),
                                                      _lib.ptr(self.aux), self.p, _lib.ptr(out), base))
        self.n_pushed += n
        return out

    def result(self):
        """(distances, rows) of the best min(k, rows pushed) rows so far, ascending."""
        vals = np.empty(self.k, dtype=np.float64)
        rows = np.empty(self.k, dtype=np.int64)
        cnt = C.c_int64()
        self._check(self.lib.elfihip_reject_result(self.h, _lib.ptr(vals), _lib.ptr(rows), C.byref(cnt)))
        return vals[:cnt.value], rows[:cnt.value]
