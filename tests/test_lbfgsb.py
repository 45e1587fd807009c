"""The host-side L-BFGS-B state machine (elfi_amd/csrc/lbfgsb.hpp) against SciPy's L-BFGS-B.

The reference minimises the acquisition with scipy.optimize.minimize(method='L-BFGS-B')
(elfi/methods/bo/utils.py:97-103); the lock-step minimiser drives one of these state machines per
start point.  The header is pure C++, so it is compiled here with g++ (no GPU needed) behind a tiny
test-only C wrapper (tests/cpp/lbfgsb_capi.cpp) and driven from Python with the same objective
functions SciPy gets.  Same algorithm, same tolerances: iterates agree to rounding.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.optimize as so

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'oracle'))
import gp_oracle as G  # noqa: E402

DP = C.POINTER(C.c_double)


@pytest.fixture(scope='module')
def lb(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('lbfgsb') / 'liblbfgsb_test.so')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-pthread', '-o', out,
                           os.path.join(HERE, 'cpp', 'lbfgsb_capi.cpp')])
    lib = C.CDLL(out)
    lib.lb_new.restype = C.c_void_p
    lib.lb_new.argtypes = [C.c_int, DP, DP, DP, C.c_int]
    lib.lb_done.argtypes = [C.c_void_p]
    lib.lb_x.restype = DP
    lib.lb_x.argtypes = [C.c_void_p]
    lib.lb_feed.argtypes = [C.c_void_p, C.c_double, DP]
    lib.lb_result.argtypes = [C.c_void_p, DP, DP, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    lib.lb_free.argtypes = [C.c_void_p]
    return lib


def _p(a):
    return a.ctypes.data_as(DP)


def run(lib, fun, grad, x0, bounds, maxiter=1000):
    n = len(x0)
    lo = np.array([b[0] for b in bounds], float)
    hi = np.array([b[1] for b in bounds], float)
    x0 = np.ascontiguousarray(x0, float)
    h = lib.lb_new(n, _p(lo), _p(hi), _p(x0), maxiter)
    points = []
    while not lib.lb_done(h):
        x = np.ctypeslib.as_array(lib.lb_x(h), (n,)).copy()
        points.append(x)
        g = np.ascontiguousarray(grad(x), float)
        lib.lb_feed(h, float(fun(x)), _p(g))
        assert len(points) < 20000
    x = np.empty(n)
    f, it, nf, st = C.c_double(), C.c_int(), C.c_int(), C.c_int()
    lib.lb_result(h, _p(x), C.byref(f), C.byref(it), C.byref(nf), C.byref(st), n)
    lib.lb_free(h)
    return dict(x=x, fun=f.value, nit=it.value, nfev=nf.value, status=st.value, points=np.array(points))


def scipy_run(fun, grad, x0, bounds, maxiter=1000):
    return so.minimize(fun, x0, jac=grad, method='L-BFGS-B', bounds=bounds, options={'maxiter': maxiter})


def same_run(ours, ref, xtol=1e-7):
    """Same end point and (allowing one rounding-induced extra trial) the same amount of work."""
    assert np.max(np.abs(ours['x'] - ref.x)) <= xtol * (1 + np.max(np.abs(ref.x))), (ours['x'], ref.x)
    assert abs(ours['fun'] - ref.fun) <= 1e-9 * (1 + abs(ref.fun))
    assert abs(ours['nit'] - ref.nit) <= 1 and abs(ours['nfev'] - ref.nfev) <= 2, (ours['nit'], ref.nit,
                                                                                    ours['nfev'], ref.nfev)


@pytest.mark.parametrize('n', [2, 5, 10, 30])
def test_bounded_rosenbrock_follows_scipy(lb, n):
    rs = np.random.RandomState(n)
    bounds = [(-1.5, 0.8)] * n          # the unconstrained minimum (1, ..., 1) is outside the box
    x0 = rs.uniform(-1.5, 0.8, n)
    same_run(run(lb, so.rosen, so.rosen_der, x0, bounds), scipy_run(so.rosen, so.rosen_der, x0, bounds))


@pytest.mark.parametrize('box', [0.3, 2.0, 50.0])
def test_convex_quadratic_with_active_bounds(lb, box):
    rs = np.random.RandomState(7)
    A = rs.randn(10, 10)
    A = A @ A.T + np.eye(10)
    c = rs.randn(10) * 5
    fun = lambda x: 0.5 * x @ A @ x - c @ x
    grad = lambda x: A @ x - c
    bounds = [(-box, box)] * 10
    x0 = rs.uniform(-box, box, 10)
    ours, ref = run(lb, fun, grad, x0, bounds), scipy_run(fun, grad, x0, bounds)
    same_run(ours, ref)
    lo, hi = np.array(bounds).T
    assert np.all(ours['points'] >= lo) and np.all(ours['points'] <= hi), 'every evaluation is inside the box'


def test_lcb_acquisition_surface_every_start_matches_scipy(lb):
    """The actual use: LCB of a GP (CPU oracle) from ELFI's kind of start points (bo/utils.py:84-95)."""
    X, y, bounds = G.synthetic_gp_problem(300, 4)
    post = G.Posterior(X, y, **G.default_hyper(bounds, y))
    t = 300
    fun = lambda x: float(G.lcb_evaluate(post, x, t)[0, 0])
    grad = lambda x: G.lcb_evaluate_gradient(post, x, t)[0]
    nit = []
    for x0 in np.random.RandomState(3).uniform(-2, 2, (12, 4)):
        ours, ref = run(lb, fun, grad, x0, bounds), scipy_run(fun, grad, x0, bounds)
        same_run(ours, ref, xtol=1e-6)
        nit.append(ours['nit'])
    assert max(nit) < 100


def test_start_outside_the_box_and_degenerate_cases(lb):
    fun = lambda x: float(np.sum((x - 3.0) ** 2))
    grad = lambda x: 2.0 * (x - 3.0)
    bounds = [(-1.0, 1.0), (0.5, 0.5), (-4.0, 4.0)]
    r = run(lb, fun, grad, np.array([9.0, -9.0, 0.0]), bounds)
    assert np.allclose(r['x'], [1.0, 0.5, 3.0], atol=1e-6) and r['status'] in (1, 2)
    assert np.array_equal(r['points'][0], [1.0, 0.5, 0.0]), 'the start is projected into the box first'
    # already stationary: one evaluation, no iteration
    r = run(lb, fun, grad, np.array([1.0, 0.5, 3.0]), bounds)
    assert r['nit'] == 0 and r['nfev'] == 1 and r['status'] == 1
    # maxiter = 0: evaluates the start, moves nowhere
    r = run(lb, fun, grad, np.array([0.0, 0.5, 0.0]), bounds, maxiter=0)
    assert r['nit'] == 0 and r['nfev'] == 1 and np.array_equal(r['x'], [0.0, 0.5, 0.0])
    # maxiter honoured
    r = run(lb, so.rosen, so.rosen_der, np.array([-1.2, 1.0, -1.0]), [(-2, 2)] * 3, maxiter=3)
    assert r['nit'] == 3 and r['status'] == 3
    # a non-finite first value ends that start without iterating
    r = run(lb, lambda x: np.nan, grad, np.array([0.0, 0.5, 0.0]), bounds)
    assert r['nit'] == 0 and r['status'] == 4


def test_round_pool_every_item_once_per_round(lb):
    """The host-thread pool that advances the L-BFGS-B state machines of many starts between device evaluations
    (elfi_amd/csrc/round_pool.hpp): every item of every round is run exactly once, for any thread count."""
    lb.pool_rounds_check.restype = C.c_longlong
    lb.pool_rounds_check.argtypes = [C.c_int, C.c_int, C.c_longlong]
    for threads in (1, 2, 8):
        assert lb.pool_rounds_check(threads, 300, 256) == 0
    assert lb.pool_rounds_check(4, 50, 5000) == 0
