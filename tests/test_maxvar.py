"""MaxVar / RandMaxVar mirrors (elfi_amd/maxvar_acquisition.py) against the reference's own classes.

tests/golden/maxvar.npz was recorded from elfi.methods.bo.acquisition.MaxVar / RandMaxVar over a GP with fixed
hyper-parameters and the real ModelPrior (oracle/make_golden_posterior.py).  Here the host logic (formulas,
lock-step multi-start search through libelfihip.so's L-BFGS-B machines, the single sampling chain) runs over the
NumPy GP of the oracle, so no GPU is needed; tests/test_maxvar_gpu.py repeats it on the device GP.
"""
import os

import numpy as np
import pytest

import acquisition_oracle as AO
import gp_oracle as G
import posterior_oracle as PO
from conftest import GOLDEN


class OracleModel:
    """What the acquisition classes touch of a GP regression object, answered by gp_oracle.Posterior."""

    def __init__(self, g):
        self.post = G.Posterior(g['X'], g['y'], *g['hyper'])
        self.bounds = [tuple(b) for b in g['bounds']]
        self.input_dim = g['X'].shape[1]
        self.parameter_names = ['p%d' % i for i in range(self.input_dim)]
        self.X, self.Y = g['X'], g['y'].reshape(-1, 1)
        self.noise = float(g['hyper'][3])
        self.n_evidence = len(self.X)

    def predict(self, x, noiseless=False):
        return self.post.predict(np.asarray(x, float).reshape(-1, self.input_dim), noiseless=noiseless)

    def predictive_gradients(self, x):
        return self.post.predictive_gradients(np.asarray(x, float).reshape(-1, self.input_dim))

    # what ExpIntVar.acquire / evaluate build from kern.K and cho_solve (acquisition.py:788-808)
    def set_integration_points(self, points):
        self._pts = np.asarray(points, float).reshape(-1, self.input_dim)

    def cross_cov(self, x):
        x = np.asarray(x, float).reshape(-1, self.input_dim)
        p = self.post
        K = lambda a, b: G.kern_K(a, b, p.var, p.ls, p.bias)
        cov = K(self._pts, x) - K(self._pts, self.X) @ p.Kinv @ K(self.X, x)
        return cov, p.predict(x, noiseless=True)[1][:, 0]

    # the two surfaces the acquisition classes ask their model for (HipGPRegression: one device call each), here by
    # the CPU restatement of the reference's formulas (oracle/acquisition_oracle.py)
    def maxvar_surface(self, theta, eps, prior_pdf, prior_grad_logpdf):
        mean, var = self.predict(theta, noiseless=True)
        gm, gv = self.predictive_gradients(theta)
        return (AO.maxvar_value(mean, var, self.noise, eps, prior_pdf),
                AO.maxvar_gradient(mean, var, gm, gv, self.noise, eps, prior_pdf, prior_grad_logpdf))

    def expintvar_loss(self, theta, eps, w_int, mean_int, var_int):
        cov, var_new = self.cross_cov(theta)
        return AO.expintvar_loss(cov.T, var_new, self.noise, eps, w_int, mean_int, var_int)


def check_against_fixture(make_model, value_tol, loc_tol):
    import elfi_amd
    g = np.load(os.path.join(GOLDEN, 'maxvar.npz'))
    model = make_model(g)
    prior = PO.BoxPrior(model.bounds)
    mv = elfi_amd.HipMaxVar(model, prior, quantile_eps=0.05, n_inits=8, seed=7)
    theta = mv.acquire(2)
    assert theta.shape == (2, 2) and np.array_equal(theta[0], theta[1])
    assert abs(mv.eps - float(g['eps'])) <= 1e-12
    # the surface at the reference's points, value and gradient, separately and fused
    v, gr = mv.value_and_gradient(g['xs'])
    scale = np.max(np.abs(g['value']))
    np.testing.assert_allclose(v, g['value'], rtol=value_tol, atol=value_tol * scale)
    np.testing.assert_allclose(gr, g['gradient'], rtol=100 * value_tol, atol=100 * value_tol * np.max(np.abs(g['gradient'])))
    np.testing.assert_allclose(mv.evaluate(g['xs']), v, rtol=1e-12, atol=0)
    np.testing.assert_allclose(mv.evaluate_gradient(g['xs']), gr, rtol=1e-12, atol=0)
    # the maximiser: same start points (same random stream), same L-BFGS-B, same arg-max
    np.testing.assert_allclose(theta[0], g['theta_max'][0], rtol=0, atol=loc_tol)
    assert mv.last_opt['rounds'] <= int(np.max(mv.last_opt['iters'])) * 3 + 25
    return g, model, prior


def test_maxvar_on_the_oracle_gp_equals_the_reference():
    check_against_fixture(OracleModel, 1e-10, 1e-6)


def test_randmaxvar_chains_equal_the_reference():
    import elfi_amd
    g = np.load(os.path.join(GOLDEN, 'maxvar.npz'))
    model = OracleModel(g)
    prior = PO.BoxPrior(model.bounds)
    for sampler in ('nuts', 'metropolis'):
        r1 = elfi_amd.HipRandMaxVar(model, prior, quantile_eps=0.05, sampler=sampler, n_samples=40, seed=9)
        np.testing.assert_allclose(r1.acquire(1), g['rand_%s_1' % sampler], rtol=0, atol=1e-8)
        r3 = elfi_amd.HipRandMaxVar(model, prior, quantile_eps=0.05, sampler=sampler, n_samples=40, seed=9)
        np.testing.assert_allclose(r3.acquire(3), g['rand_%s_3' % sampler], rtol=0, atol=1e-8)
    with pytest.raises(ValueError, match='number of acquisitions'):
        elfi_amd.HipRandMaxVar(model, prior, n_samples=10, seed=1).acquire(11)
    with pytest.raises(ValueError, match='Incompatible sampler'):
        elfi_amd.HipRandMaxVar(model, prior, sampler='gibbs', seed=1).acquire(1)
    with pytest.raises(ValueError, match='input as a dict'):
        elfi_amd.HipRandMaxVar(model, prior, sampler='metropolis', sigma_proposals=[0.1, 0.1])


def test_lockstep_multistart_equals_scipy_start_by_start():
    import scipy.optimize as so
    from elfi_amd import multistart
    rs = np.random.RandomState(0)
    bounds = [(-1.5, 0.8)] * 6
    starts = rs.uniform(-1.5, 0.8, (7, 6))
    calls = []

    def batch(X):
        calls.append(len(X))
        return np.array([so.rosen(x) for x in X]), np.array([so.rosen_der(x) for x in X])

    res = multistart.minimize_lockstep(batch, starts, bounds)
    for i, x0 in enumerate(starts):
        r = so.minimize(so.rosen, x0, jac=so.rosen_der, method='L-BFGS-B', bounds=bounds, options={'maxiter': 1000})
        assert np.max(np.abs(res['locs'][i] - r.x)) <= 1e-7 and abs(res['vals'][i] - r.fun) <= 1e-9 * (1 + abs(r.fun))
        assert abs(int(res['iters'][i]) - r.nit) <= 1
    assert calls[0] == 7 and res['rounds'] == len(calls) and sum(calls) > 2 * len(calls)
    loc, val = multistart.minimize(batch, bounds, n_start_points=5, random_state=np.random.RandomState(1))
    assert val <= np.min(batch(np.clip(loc[None, :], -1.5, 0.8))[0]) + 1e-12
    with pytest.raises(ValueError):
        multistart.minimize_lockstep(batch, starts, [(1.0, -1.0)] * 6)


def check_expintvar(make_model, loss_tol, min_tol):
    import elfi_amd
    g = np.load(os.path.join(GOLDEN, 'maxvar.npz'))
    model = make_model(g)
    prior = PO.BoxPrior(model.bounds)
    xs = g['xs']
    # grid integration
    ev = elfi_amd.HipExpIntVar(model, prior, quantile_eps=0.05, integration='grid', d_grid=0.4, n_inits=5, seed=11)
    np.testing.assert_array_equal(ev.points_int, g['eiv_grid_points'])
    th = ev.acquire(1, t=4)
    scale = np.max(np.abs(g['eiv_grid_loss']))
    np.testing.assert_allclose(ev.evaluate(xs), g['eiv_grid_loss'], rtol=0, atol=loss_tol * scale)
    # the minimiser runs on finite-difference gradients of a flat surface: compare the loss reached, and the location
    # where the surface is curved enough to pin it
    ours, ref = float(ev.evaluate(th)[0]), float(np.ravel(g['eiv_grid_loss_at_theta'])[0])
    assert ours <= ref + min_tol * scale, (ours, ref)
    assert th.shape == (1, 2) and np.all(np.abs(th) <= 2)
    # importance sampling: integration points from the RandMaxVar chain, weights 1 / MaxVar surface
    ei = elfi_amd.HipExpIntVar(model, prior, quantile_eps=0.05, integration='importance', n_samples_imp=24, iter_imp=2,
                               sampler='metropolis', n_samples=60, n_inits=4, seed=13)
    thi = ei.acquire(1, t=2)
    return g, ev, ei, thi


def test_expintvar_on_the_oracle_gp_equals_the_reference():
    g, ev, ei, thi = check_expintvar(OracleModel, 1e-9, 1e-6)
    np.testing.assert_allclose(ei.points_int, g['eiv_imp_points'], rtol=0, atol=1e-8)
    np.testing.assert_allclose(ei.omegas_int, g['eiv_imp_omegas'], rtol=1e-6)
    np.testing.assert_allclose(ei.evaluate(g['xs']), g['eiv_imp_loss'], rtol=0,
                               atol=1e-8 * np.max(np.abs(g['eiv_imp_loss'])))
    np.testing.assert_allclose(thi, g['eiv_imp_theta'], rtol=0, atol=5e-3)


def test_finite_difference_objective_matches_scipy_without_jacobian():
    import scipy.optimize as so
    from elfi_amd import multistart
    from elfi_amd.maxvar_acquisition import finite_difference_objective
    f = lambda X: np.array([so.rosen(x) + 0.1 * np.sin(3 * x[0]) for x in X])
    bounds = [(-1.5, 0.8)] * 4
    starts = np.random.RandomState(0).uniform(-1.5, 0.8, (4, 4))
    starts[0, 1] = 0.8   # on the upper bound: that coordinate's step turns backwards
    res = multistart.minimize_lockstep(finite_difference_objective(f, bounds), starts, bounds)
    for i, x0 in enumerate(starts):
        r = so.minimize(lambda x: float(f(x[None])[0]), x0, method='L-BFGS-B', bounds=bounds, options={'maxiter': 1000})
        assert np.max(np.abs(r.x - res['locs'][i])) <= 1e-5 and abs(r.fun - res['vals'][i]) <= 1e-9 * (1 + abs(r.fun))


def check_gradient_numerically(model):
    """tests/unit/test_bo.py:162-181 (GPy's GradientChecker, ratio tolerance 1e-4) on value_and_gradient."""
    import elfi_amd
    prior = PO.BoxPrior(model.bounds)
    mv = elfi_amd.HipMaxVar(model, prior, quantile_eps=0.05, seed=1)
    mv.eps = float(np.percentile(model.Y, 5))
    pts = np.random.RandomState(8).uniform(-1.8, 1.8, (20, model.input_dim))
    v, g = mv.value_and_gradient(pts)
    h = 1e-6
    num = np.empty_like(g)
    for i in range(model.input_dim):
        e = np.zeros(model.input_dim)
        e[i] = h
        num[:, i] = (mv.evaluate(pts + e)[:, 0] - mv.evaluate(pts - e)[:, 0]) / (2 * h)
    big = np.abs(num) > 1e-3 * np.max(np.abs(num))
    assert np.allclose(g[big] / num[big], 1.0, atol=1e-4), np.max(np.abs(g[big] / num[big] - 1))


def test_maxvar_gradient_is_the_derivative_of_the_value():
    g = np.load(os.path.join(GOLDEN, 'maxvar.npz'))
    check_gradient_numerically(OracleModel(g))
