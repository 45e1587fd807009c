"""Multi-start box-constrained minimisation with every start advanced in lock-step.

The reference's minimize() (elfi/methods/bo/utils.py:40-111) runs scipy's L-BFGS-B from each start
point in turn and the objective is called for one point at a time.  Here the S searches are the
L-BFGS-B state machines of libelfihip.so (elfihip_lbfgsb_*, include/elfihip.h; the same code that
drives the LCB search inside the library) and the objective is called ONCE per round with the
points all still-running searches wait for -- so an objective built on the batched device predictor
costs one device call per round.  Same algorithm and tolerances as SciPy: each start ends where
scipy.optimize.minimize(method='L-BFGS-B') would from that start (tests/test_lbfgsb.py).
"""
import ctypes as C

import numpy as np

from . import _lib
from .lcb_acquisition import draw_start_points


def minimize_lockstep(fun_and_grad, start_points, bounds, maxiter=1000):
    """fun_and_grad(X (k, d)) -> (f (k,), g (k, d)).  Returns dict(locs (S, d), vals (S), iters, status, rounds)."""
    lib = _lib.load_library()
    starts = np.ascontiguousarray(np.atleast_2d(np.asarray(start_points, dtype=np.float64)))
    S, d = starts.shape
    lo = np.ascontiguousarray([b[0] for b in bounds], dtype=np.float64)
    hi = np.ascontiguousarray([b[1] for b in bounds], dtype=np.float64)
    if len(lo) != d:
        raise ValueError('bounds must have one (lower, upper) pair per dimension of the start points')
    h = C.c_void_p()
    rc = lib.elfihip_lbfgsb_create(d, S, _lib.ptr(lo), _lib.ptr(hi), _lib.ptr(starts), int(maxiter), C.byref(h))
    if rc != 0:
        raise ValueError('bad arguments for the multi-start search (empty bound interval, maxiter < 0, ...)')
    try:
        idx = np.empty(S, dtype=np.int64)
        x = np.empty((S, d))
        rounds = 0
        while True:
            k = int(lib.elfihip_lbfgsb_pending(h, _lib.ptr(idx), _lib.ptr(x)))
            if k < 0:
                raise RuntimeError('elfihip_lbfgsb_pending failed')
            if k == 0:
                break
            f, g = fun_and_grad(x[:k].copy())
            f = np.ascontiguousarray(np.asarray(f, dtype=np.float64).reshape(k))
            g = np.ascontiguousarray(np.asarray(g, dtype=np.float64).reshape(k, d))
            if lib.elfihip_lbfgsb_feed(h, k, _lib.ptr(f), _lib.ptr(g)) != 0:
                raise RuntimeError('elfihip_lbfgsb_feed failed')
            rounds += 1
        locs, vals = np.empty((S, d)), np.empty(S)
        iters, status = np.empty(S, dtype=np.int32), np.empty(S, dtype=np.int32)
        lib.elfihip_lbfgsb_result(h, _lib.ptr(locs), _lib.ptr(vals), _lib.ptr(iters), _lib.ptr(status))
    finally:
        lib.elfihip_lbfgsb_free(h)
    return dict(locs=locs, vals=vals, iters=iters, status=status, rounds=rounds)


def minimize(fun_and_grad, bounds, prior=None, n_start_points=10, maxiter=1000, random_state=None):
    """minimize() of the reference (bo/utils.py:40-111) over a batched objective: start points drawn the same
    way, arg-min over the starts, clipped into the bounds.  Returns (location, value)."""
    starts = draw_start_points(bounds, n_start_points, prior, random_state)
    res = minimize_lockstep(fun_and_grad, starts, bounds, maxiter=maxiter)
    k = int(np.argmin(res['vals']))
    loc = res['locs'][k].copy()
    for i in range(len(bounds)):
        loc[i] = np.clip(loc[i], *bounds[i])
    return loc, res['vals'][k]
