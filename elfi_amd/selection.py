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
