"""GPU: the lock-step multi-start LCB minimiser and HipLCBSC.acquire vs the reference recipe.

The reference minimises from each start with scipy's L-BFGS-B (elfi/methods/bo/utils.py:97-103).
The device path advances one L-BFGS-B state machine per start (csrc/lbfgsb.hpp, pinned against SciPy
on the CPU in tests/test_lbfgsb.py) over batched device evaluations; what must agree:
  * every end point is a stationary point of the box-constrained problem (projected gradient
    <= 1e-3 on the ORACLE's gradient) with a value not above its start's;
  * the best value over the starts is within 1e-6 (relative to the value scale) of, or below,
    the best value scipy finds from the same starts on the CPU oracle;
  * start by start, nearly every end point and iteration count equals scipy's;
  * identical seeds give identical acquisitions (determinism).
"""
import numpy as np
import pytest

import gp_oracle as G

pytestmark = pytest.mark.gpu


def _setup(n, d, seed):
    from elfi_amd import HipGPRegression
    X, y, bounds = G.synthetic_gp_problem(n, d, seed=seed)
    names = ['p%d' % i for i in range(d)]
    m = HipGPRegression(names, bounds=dict(zip(names, bounds)))
    m.update(X, y)
    ref = G.Posterior(X, y, **m._hyper)
    return m, ref, bounds


@pytest.mark.parametrize('n,d,S,t', [(150, 1, 5, 0), (300, 2, 10, 3), (600, 5, 10, 40), (500, 3, 37, 7)])
def test_lockstep_minimiser_vs_scipy_on_the_oracle(hip_ctx, n, d, S, t):
    m, ref, bounds = _setup(n, d, seed=n + S)
    beta = G.lcb_beta(t, d)
    starts = np.random.RandomState(S).uniform(-2, 2, (S, d))
    locs, vals, iters, n_eval = m._handle.lcb_minimize(starts, bounds, beta, maxiter=1000)
    f0 = G.lcb_evaluate(ref, starts, t)[:, 0]
    scale = np.max(np.abs(f0)) + 1.0
    lo, hi = np.array(bounds).T
    assert np.all(locs >= lo) and np.all(locs <= hi)
    assert np.all(vals <= f0 + 1e-9 * scale)
    np.testing.assert_allclose(vals, G.lcb_evaluate(ref, locs, t)[:, 0], rtol=0, atol=1e-8 * scale)
    g = G.lcb_evaluate_gradient(ref, locs, t)
    pg = locs - np.clip(locs - g, lo, hi)
    assert np.max(np.abs(pg)) <= 1e-3, 'end points must be stationary in the box'
    fun = lambda x: float(G.lcb_evaluate(ref, x, t)[0, 0])
    grad = lambda x: G.lcb_evaluate_gradient(ref, x, t)[0]
    xs, fs = G.minimize_multistart(fun, grad, bounds, starts)
    assert vals.min() <= fs + 1e-6 * scale, (vals.min(), fs)
    assert n_eval >= S and np.all(iters <= 1000)
    # start by start: same algorithm as scipy's L-BFGS-B (tests/test_lbfgsb.py), evaluated on the device instead of
    # the host, so nearly every start ends at scipy's end point after scipy's number of iterations; the odd start
    # may be tipped into a neighbouring optimum by last-bit differences of the two evaluations
    import scipy.optimize
    same, same_it = 0, 0
    for i, x0 in enumerate(starts):
        r = scipy.optimize.minimize(fun, x0, method='L-BFGS-B', jac=grad, bounds=bounds, options={'maxiter': 1000})
        close = np.max(np.abs(r.x - locs[i])) <= 1e-4
        same += close
        same_it += close and abs(int(r.nit) - int(iters[i])) <= 1
    assert same >= 0.8 * S and same_it >= 0.7 * S, (same, same_it, S)


def test_cfg5_shape_256_starts_n8192_d20(hip_ctx):
    """BASELINE.json configs[4]: n_evidence = 8192, d = 20, 256 parallel acquisition starts (acquisition.py:129-172
    with n_inits=256).  All starts in one lock-step call; every end point checked on the CPU oracle, a sample of the
    starts against scipy's L-BFGS-B from the same start on the oracle (what the reference would run)."""
    import scipy.optimize
    n, d, S, t = 8192, 20, 256, 128
    m, ref, bounds = _setup(n, d, seed=5)
    beta = G.lcb_beta(t, d)
    starts = np.random.RandomState(17).uniform(-2, 2, (S, d))
    locs, vals, iters, n_eval = m._handle.lcb_minimize(starts, bounds, beta, maxiter=1000)
    f0 = G.lcb_evaluate(ref, starts, t)[:, 0]
    scale = np.max(np.abs(f0)) + 1.0
    lo, hi = np.array(bounds).T
    assert np.all(locs >= lo) and np.all(locs <= hi)
    assert np.all(vals <= f0 + 1e-9 * scale)
    np.testing.assert_allclose(vals, G.lcb_evaluate(ref, locs, t)[:, 0], rtol=0, atol=1e-8 * scale)
    g = G.lcb_evaluate_gradient(ref, locs, t)
    pg = locs - np.clip(locs - g, lo, hi)
    assert np.max(np.abs(pg)) <= 1e-3, 'end points must be stationary in the box'
    assert n_eval >= S and np.all(iters <= 1000)
    fun = lambda x: float(G.lcb_evaluate(ref, x, t)[0, 0])
    grad = lambda x: G.lcb_evaluate_gradient(ref, x, t)[0]
    sample = [0, 31, 64, 100, 177, 255]
    same, best = 0, np.inf
    for i in sample:
        r = scipy.optimize.minimize(fun, starts[i], method='L-BFGS-B', jac=grad, bounds=bounds,
                                    options={'maxiter': 1000})
        same += np.max(np.abs(r.x - locs[i])) <= 1e-4 and abs(int(r.nit) - int(iters[i])) <= 1
        best = min(best, r.fun)
    assert same >= len(sample) - 1, (same, len(sample))
    assert vals.min() <= best + 1e-6 * scale
    # the sharded form of the same call (32 starts per rank on 8 GPUs) is the same computation per start
    part, pvals, piters, _ = m._handle.lcb_minimize(starts[3::8], bounds, beta, maxiter=1000)
    agree = np.max(np.abs(part - locs[3::8]), axis=1) <= 1e-4
    assert np.count_nonzero(agree) >= 30 and abs(pvals.min() - vals[3::8].min()) <= 1e-6 * scale


def test_acquire_end_to_end_and_determinism(hip_ctx):
    from elfi_amd import HipLCBSC
    m, ref, bounds = _setup(400, 2, seed=1)
    a1 = HipLCBSC(m, n_inits=10, noise_var=0.1, exploration_rate=10, seed=7)
    a2 = HipLCBSC(m, n_inits=10, noise_var=0.1, exploration_rate=10, seed=7)
    x1, x2 = a1.acquire(3, t=5), a2.acquire(3, t=5)
    assert x1.shape == (3, 2) and np.array_equal(x1, x2)
    lo, hi = np.array(bounds).T
    assert np.all(x1 >= lo) and np.all(x1 <= hi)
    info = a1.last_opt
    assert info['locs'].shape == (10, 2) and info['ind_min'] == int(np.argmin(info['vals']))
    # the point-wise interface agrees with the oracle's LCBSC (acquisition.py:262-301)
    xs = info['locs']
    np.testing.assert_allclose(a1.evaluate(xs, 5), G.lcb_evaluate(ref, xs, 5), rtol=1e-8)
    np.testing.assert_allclose(a1.evaluate_gradient(xs, 5), G.lcb_evaluate_gradient(ref, xs, 5), rtol=1e-6,
                               atol=1e-8)


def test_minimiser_edge_cases(hip_ctx):
    m, ref, bounds = _setup(100, 2, seed=3)
    beta = G.lcb_beta(0, 2)
    # maxiter = 0: returns the (clipped) starts and their values
    starts = np.array([[5.0, -7.0], [0.1, 0.2]])
    locs, vals, iters, n_eval = m._handle.lcb_minimize(starts, bounds, beta, maxiter=0)
    assert np.array_equal(locs, np.clip(starts, -2, 2)) and np.all(iters == 0) and n_eval == 2
    # degenerate box: nothing to optimise
    b0 = [(0.5, 0.5), (-1.0, 1.0)]
    locs, vals, iters, _ = m._handle.lcb_minimize(starts, b0, beta)
    assert np.all(locs[:, 0] == 0.5)
    with pytest.raises(ValueError):
        m._handle.lcb_minimize(starts, [(1.0, -1.0), (0, 1)], beta)


def test_against_the_reference_recorded_acquisition(hip_ctx):
    """tests/golden/gp_acquisition.npz: the reference's own acquire() (real LCBSC + minimize + scipy
    L-BFGS-B, oracle/make_golden_gp.py).  Same seed => same start points and same jitter stream;
    our optimum must be as good as the reference's best start (1e-6 of the value scale) and, when
    both land in the same basin, the acquired batch is the same to 1e-4."""
    import os
    from conftest import GOLDEN
    from elfi_amd import HipGPRegression, HipLCBSC
    g = np.load(os.path.join(GOLDEN, 'gp_acquisition.npz'))
    for tag in g['cases']:
        tag = str(tag)
        X, y = g['X_' + tag], g['y_' + tag]
        d = X.shape[1]
        names = ['p%d' % i for i in range(d)]
        m = HipGPRegression(names, bounds={k: (-2., 2.) for k in names})
        m.update(X, y)
        m._hyper = dict(zip(('var', 'ls', 'bias', 'noise'), (float(v) for v in g['hyper_' + tag])))
        m._refit()
        t, seed = int(g['t_' + tag]), int(g['seed_' + tag])
        acq = HipLCBSC(m, n_inits=8, noise_var=0.1, exploration_rate=10, seed=seed)
        x_acq = acq.acquire(3, t=t)
        info = acq.last_opt
        assert np.array_equal(info['starts'], g['starts_' + tag]), 'start points must be the reference draws'
        scale = np.max(np.abs(g['vals_' + tag])) + 1.0
        assert info['vals'].min() <= g['vals_' + tag].min() + 1e-6 * scale
        k_ref = int(np.argmin(g['vals_' + tag]))
        if np.max(np.abs(info['locs'][info['ind_min']] - g['locs_' + tag][k_ref])) <= 1e-3:
            np.testing.assert_allclose(x_acq, g['x_acq_' + tag], rtol=0, atol=1e-3)


class _QuadraticCost:
    """additive_cost of acquisition.py:278-279,299-300: evaluate(x) (n, 1), evaluate_gradient(x) (n, d)."""

    def __init__(self, centre, weight):
        self.c, self.w = np.asarray(centre, dtype=float), float(weight)

    def evaluate(self, x):
        x = np.atleast_2d(x)
        return self.w * np.sum((x - self.c) ** 2, axis=1, keepdims=True)

    def evaluate_gradient(self, x):
        x = np.atleast_2d(x)
        return 2.0 * self.w * (x - self.c)


def test_additive_cost_and_constraints_like_the_reference_rule(hip_ctx):
    """HipLCBSC(additive_cost=...) / HipLCBSC(constraints=...): what the reference's LCBSC does with these arguments
    (acquisition.py:146-172,256-301: criterion + cost; SLSQP under the constraints, bo/utils.py:97-103), the GP part of
    every evaluation on the device.  Checked against scipy on the CPU posterior from the same start points."""
    import scipy.optimize
    from elfi_amd import HipLCBSC
    m, ref, bounds = _setup(400, 2, seed=21)
    t = 5
    cost = _QuadraticCost([1.5, -1.0], 0.7)
    acq = HipLCBSC(m, n_inits=8, seed=3, additive_cost=cost)
    xs = np.random.RandomState(0).uniform(-2, 2, (6, 2))
    scale = np.max(np.abs(G.lcb_evaluate(ref, xs, t))) + 1.0
    np.testing.assert_allclose(acq.evaluate(xs, t), G.lcb_evaluate(ref, xs, t) + cost.evaluate(xs), rtol=0, atol=1e-8 * scale)
    np.testing.assert_allclose(acq.evaluate_gradient(xs, t), G.lcb_evaluate_gradient(ref, xs, t) + cost.evaluate_gradient(xs),
                               rtol=0, atol=1e-7 * scale)
    starts = np.random.RandomState(1).uniform(-2, 2, (8, 2))
    xhat, val = acq.minimize(t, start_points=starts)
    fun = lambda x: float(G.lcb_evaluate(ref, x, t)[0, 0] + cost.evaluate(x)[0, 0])
    grad = lambda x: G.lcb_evaluate_gradient(ref, x, t)[0] + cost.evaluate_gradient(x)[0]
    best = min(scipy.optimize.minimize(fun, x0, method='L-BFGS-B', jac=grad, bounds=bounds).fun for x0 in starts)
    assert val <= best + 1e-6 * scale and abs(fun(xhat) - val) <= 1e-8 * scale
    assert acq.acquire(3, t=t).shape == (3, 2)
    # a linear inequality constraint x0 + x1 <= -0.5, handed over as scipy's dict form
    cons = {'type': 'ineq', 'fun': lambda x: -0.5 - x[0] - x[1], 'jac': lambda x: np.array([-1.0, -1.0])}
    acq2 = HipLCBSC(m, n_inits=8, seed=3, constraints=cons)
    xhat2, val2 = acq2.minimize(t, start_points=starts)
    assert xhat2[0] + xhat2[1] <= -0.5 + 1e-6
    f2 = lambda x: float(G.lcb_evaluate(ref, x, t)[0, 0])
    g2 = lambda x: G.lcb_evaluate_gradient(ref, x, t)[0]
    best2 = min(scipy.optimize.minimize(f2, x0, method='SLSQP', jac=g2, bounds=bounds, constraints=cons).fun for x0 in starts)
    assert abs(val2 - best2) <= 1e-5 * scale
