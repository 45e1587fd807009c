"""Summary-statistic operations of ELFI's example models on the GPU (bit-identical to NumPy).

Drop-ins for the callables the reference installs with elfi.Summary (a Summary operation is
`fn(*parents) -> array of length batch_size`, also called once on the observed data with a
leading dimension of 1 -- elfi/model/elfi_model.py:915-943, elfi/compiler.py:94-116):

    autocov(x, lag=1)    elfi/examples/ma2.py:40-59
    ss_mean(y)           elfi/examples/gauss.py:142-156
    ss_var(y)            elfi/examples/gauss.py:159-173
    ma2_distance(...)    the fused MA2 path: simulator arithmetic + both summaries + distance

    S1 = elfi.Summary(elfi_amd.autocov, Y);  S2 = elfi.Summary(elfi_amd.autocov, Y, 2)

Module-level functions (picklable), plain data in, NumPy arrays out; the arithmetic runs in
libelfihip.so (csrc/summaries.hip) with NumPy's pairwise summation order.  No CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _lib

MEAN, VAR, AUTOCOV = 0, 1, 2


def _rows(x):
    x = np.atleast_2d(np.asarray(x))
    if x.ndim != 2:
        raise ValueError('summary input must be at most 2-dimensional (batch, n_obs)')
    if x.dtype != np.float64 or x.strides[1] != 8 or x.strides[0] % 8 or x.strides[0] < 8 * x.shape[1]:
        x = np.ascontiguousarray(x, dtype=np.float64)
    return x


def _row_summary(kind, x, lag=0, ctx=None):
    x = _rows(x)
    n, L = x.shape
    out = np.empty(n, dtype=np.float64)
    ldx = x.strides[0] // 8 if n > 1 else L
    ctx = ctx or _lib.default_context()
    ctx.call("elfihip_row_summary", kind, _lib.ptr(x), n, L, ldx, int(lag), _lib.ptr(out))
    return out


def autocov(x, lag=1):
    """Autocovariance at `lag` (mean assumed zero), per row: np.mean(x[:, lag:] * x[:, :-lag], axis=1)."""
    x = _rows(x)
    if not 1 <= int(lag) < x.shape[1]:
        raise ValueError('lag must be in [1, n_obs)')
    return _row_summary(AUTOCOV, x, lag)


def ss_mean(y):
    """np.mean(y, axis=1)."""
    return _row_summary(MEAN, y)


def ss_var(y):
    """np.var(y, axis=1) (ddof = 0)."""
    return _row_summary(VAR, y)


def ma2_distance(w, t1, t2, observed, ctx=None):
    """Fused MA2 path.  w: (batch, n_obs + 2) white noise as `random_state.randn(batch, n_obs + 2)`
    draws it (elfi/examples/ma2.py:35); t1, t2: (batch,) or scalars; observed: the two observed
    summaries (autocov(y_obs, 1), autocov(y_obs, 2)).  Returns (S1, S2, d), each (batch,)."""
    w = _rows(w)
    n, L = w.shape
    if L < 5:
        raise ValueError('need n_obs >= 3')
    w = np.ascontiguousarray(w)
    t1 = np.ascontiguousarray(np.broadcast_to(np.asarray(t1, dtype=np.float64).reshape(-1), (n,)))
    t2 = np.ascontiguousarray(np.broadcast_to(np.asarray(t2, dtype=np.float64).reshape(-1), (n,)))
    o = np.asarray(observed, dtype=np.float64).reshape(-1)
    if o.shape[0] != 2:
        raise ValueError('observed must hold the two observed summaries')
    S1, S2, D = np.empty(n), np.empty(n), np.empty(n)
    ctx = ctx or _lib.default_context()
    ctx.call("elfihip_ma2_distance", _lib.ptr(w), n, L - 2, _lib.ptr(t1), _lib.ptr(t2), C.c_double(o[0]),
             C.c_double(o[1]), _lib.ptr(S1), _lib.ptr(S2), _lib.ptr(D))
    return S1, S2, _lib.remember_kept(D, ctx)


def ma2_draw_distance(t1, t2, observed, n_obs=100, seed=0, stream=0, ctx=None):
    """The whole MA2 example as ONE device operation (elfi/examples/ma2.py:11-59 + Distance('euclidean', S1, S2)): the
    white noise is drawn inside the kernel (Philox4x32-10 keyed by `seed`, counter stream `stream`), the series never
    exists in memory.  t1, t2: (batch,); observed: the two observed summaries.  Returns (S1, S2, d), each (batch,)."""
    t1 = np.asarray(t1, dtype=np.float64).reshape(-1)
    t2 = np.asarray(t2, dtype=np.float64).reshape(-1)
    n = max(t1.shape[0], t2.shape[0])
    t1 = np.ascontiguousarray(np.broadcast_to(t1, (n,)))
    t2 = np.ascontiguousarray(np.broadcast_to(t2, (n,)))
    o = np.asarray(observed, dtype=np.float64).reshape(-1)
    if o.shape[0] != 2:
        raise ValueError('observed must hold the two observed summaries')
    if int(n_obs) < 3:
        raise ValueError('need n_obs >= 3')
    S1, S2, D = np.empty(n), np.empty(n), np.empty(n)
    ctx = ctx or _lib.default_context()
    ctx.call("elfihip_ma2_draw_distance", C.c_uint64(int(seed)), C.c_uint64(int(stream)), n, int(n_obs), _lib.ptr(t1),
             _lib.ptr(t2), C.c_double(o[0]), C.c_double(o[1]), _lib.ptr(S1), _lib.ptr(S2), _lib.ptr(D))
    return S1, S2, _lib.remember_kept(D, ctx)


def gauss_distance(mu, sigma, observed, n_obs=50, z=None, seed=0, stream=0, return_y=False, ctx=None):
    """Fused Gaussian example (elfi/examples/gauss.py: gauss -> ss_mean, ss_var -> euclidean distance).

    mu, sigma: (batch,) or scalars; observed: the two observed summaries (ss_mean(y_obs), ss_var(y_obs)).
    z: (batch, n_obs) standard normals as `random_state.standard_normal((batch, n_obs))` draws them -- what
    ss.norm.rvs(loc=mu, scale=sigma, size=(batch, n_obs), random_state=...) consumes (gauss.py:31-33) -- for results
    bit-identical to the reference; z=None draws on the device (Philox4x32-10 keyed by `seed`, stream `stream`), in
    which case the batch size comes from mu / sigma.  Returns (ss_mean, ss_var, d) [, y]."""
    mu = np.asarray(mu, dtype=np.float64).reshape(-1)
    sigma = np.asarray(sigma, dtype=np.float64).reshape(-1)
    if z is not None:
        z = np.ascontiguousarray(_rows(z))
        n, n_obs = z.shape
    else:
        n = max(mu.shape[0], sigma.shape[0])
        n_obs = int(n_obs)
    mu = np.ascontiguousarray(np.broadcast_to(mu, (n,)))
    sigma = np.ascontiguousarray(np.broadcast_to(sigma, (n,)))
    o = np.asarray(observed, dtype=np.float64).reshape(-1)
    if o.shape[0] != 2:
        raise ValueError('observed must hold the two observed summaries')
    S1, S2, D = np.empty(n), np.empty(n), np.empty(n)
    Y = np.empty((n, n_obs)) if return_y else None
    ctx = ctx or _lib.default_context()
    ctx.call("elfihip_gauss_distance", _lib.ptr(z), C.c_uint64(int(seed)), C.c_uint64(int(stream)), n, n_obs,
             _lib.ptr(mu), _lib.ptr(sigma), C.c_double(o[0]), C.c_double(o[1]), _lib.ptr(Y), _lib.ptr(S1), _lib.ptr(S2),
             _lib.ptr(D))
    _lib.remember_kept(D, ctx)
    return (S1, S2, D, Y) if return_y else (S1, S2, D)
