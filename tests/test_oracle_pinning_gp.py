"""Pin oracle/gp_oracle.py to outputs of the REFERENCE'S OWN CODE (tests/golden/gp_*.npz, made by
oracle/make_golden_gp.py: real GPyRegression closed forms gpy_regression.py:127-140,206-218 and the
real LCBSC acquisition.py:256-301 executed under oracle/ref_shim.py).

The closed forms and the [GPy-upstream] restatement (predict_noiseless / predictive_gradients) are
different formulas for the same quantities; the reference's tests hold them equal with
np.allclose (tests/unit/test_methods.py:110-122).  Tolerance here: 1e-9 relative to the scale of
the quantity for values and mean gradients, 1e-7 for variance gradients (the reference's closed
form solves with the Cholesky factor, the restatement multiplies with K^-1).
"""
import os

import numpy as np
import pytest

import gp_oracle as G
from conftest import GOLDEN


def _post(g, tag):
    var, ls, bias, noise = g['hyper_' + tag]
    return G.Posterior(g['X_' + tag], g['y_' + tag], var, ls, bias, noise)


def _close(a, b, tol):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) <= tol * (np.max(np.abs(b)) + 1e-300)


def test_oracle_predictions_equal_the_reference_closed_forms():
    g = np.load(os.path.join(GOLDEN, 'gp_closed_forms.npz'))
    for tag in g['cases']:
        tag = str(tag)
        post = _post(g, tag)
        xs = g['xs_' + tag]
        mu, var = post.predict(xs, noiseless=False)
        assert _close(mu[:, 0], g['mu_' + tag], 1e-9), tag
        assert np.max(np.abs(var[:, 0] - g['var_' + tag])) <= 1e-9 * (post.var + post.bias + post.noise), tag
        gm, gv = post.predictive_gradients(xs)
        assert _close(gm, g['gmu_' + tag], 1e-9), tag
        assert _close(gv, g['gvar_' + tag], 1e-7), tag
        # the oracle's own copy of the closed form is the reference's, digit for digit
        for s in range(len(xs)):
            cm, cv = post.predict_closed_form(xs[s])
            assert cm[0, 0] == pytest.approx(g['mu_' + tag][s], rel=1e-13, abs=1e-15)
            assert cv[0, 0] == pytest.approx(g['var_' + tag][s], rel=1e-12)


def test_oracle_lcb_equals_the_reference_lcbsc():
    g = np.load(os.path.join(GOLDEN, 'gp_closed_forms.npz'))
    for tag in g['cases']:
        tag = str(tag)
        post = _post(g, tag)
        d = post.X.shape[1]
        xs = g['xs_' + tag]
        for t in (0, 17):
            assert G.lcb_beta(t, d) == pytest.approx(float(g['beta_%s_%d' % (tag, t)]), rel=1e-15)
            np.testing.assert_allclose(G.lcb_evaluate(post, xs, t), g['lcb_%s_%d' % (tag, t)], rtol=1e-13)
            np.testing.assert_allclose(G.lcb_evaluate_gradient(post, xs, t), g['lcbg_%s_%d' % (tag, t)], rtol=1e-12,
                                       atol=1e-14)


def test_oracle_multistart_reproduces_the_reference_acquire():
    """minimize() of bo/utils.py:40-111 as recorded from the reference: same per-start optima."""
    g = np.load(os.path.join(GOLDEN, 'gp_acquisition.npz'))
    for tag in g['cases']:
        tag = str(tag)
        post = _post(g, tag)
        d = post.X.shape[1]
        t = int(g['t_' + tag])
        bounds = [(-2., 2.)] * d
        fun = lambda x: float(G.lcb_evaluate(post, x, t)[0, 0])
        grad = lambda x: G.lcb_evaluate_gradient(post, x, t)[0]
        x, f = G.minimize_multistart(fun, grad, bounds, g['starts_' + tag])
        k = int(np.argmin(g['vals_' + tag]))
        np.testing.assert_allclose(x, g['locs_' + tag][k], rtol=0, atol=1e-7)
        assert f == pytest.approx(g['vals_' + tag][k], rel=1e-10)
        # the recorded acquisition is that optimum plus truncated-normal jitter inside the bounds
        assert g['x_acq_' + tag].shape == (3, d)
        assert np.all(np.abs(g['x_acq_' + tag]) <= 2.0)


def test_posterior_oracle_equals_the_reference_bolfi_posterior():
    """oracle/posterior_oracle.py against the real BolfiPosterior + ModelPrior (bolfi_posterior.npz)."""
    import posterior_oracle as PO
    g = np.load(os.path.join(GOLDEN, 'bolfi_posterior.npz'))
    post = G.Posterior(g['X'], g['y'], *g['hyper'])
    bounds = [tuple(b) for b in g['bounds']]
    po = PO.PosteriorOracle(post, bounds, float(g['threshold']))
    xs = g['xs']
    inside = np.all((xs >= -2) & (xs <= 2), axis=1)
    assert 0 < inside.sum() < len(xs), 'the fixture has points on both sides of the bounds'
    np.testing.assert_allclose(po.loglik(xs), g['loglik'], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(po.grad_loglik(xs), g['gradlik'], rtol=1e-8, atol=1e-10)
    np.testing.assert_array_equal(po.prior.logpdf(xs), g['prior_logpdf'])
    lp, gr = po.logpdf_and_gradient(xs)
    np.testing.assert_allclose(lp, g['logpdf'], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(gr, g['grad'], rtol=1e-8, atol=1e-8)   # the reference's prior gradient is numerical
    # the threshold the reference finds on its own is the minimum of the GP mean in the box
    assert float(g['threshold_auto']) <= float(g['min_of_mean_over_evidence']) + 1e-9


def test_documented_bolfi_run_is_reproduced_digit_for_digit():
    """docs/usage/BOLFI.rst:36-153 is the one place where the reference publishes GPy-side numbers: MA2 seed_obs=1,
    elfi.BOLFI(log_d, batch_size=1, initial_evidence=20, update_interval=10, bounds, acq_noise_var, seed=1)
    .fit(n_evidence=200) printed the MAP objective 151.866..., the four hyper-parameters, the Gamma priors and the
    posterior threshold.  tests/golden/bolfi_doc_run.npz is that recipe run HERE through the reference's real loop
    with oracle/oracle_gp_model.py as the surrogate (oracle/make_golden_gp.py:make_doc_run): 180 acquisitions and 18
    hyper-parameter searches later the restatement of GPy / paramz lands on the printed numbers.  This pins the whole
    a10 chain -- kernel, inference, priors and their Jacobian, SCG, LCB, multi-start L-BFGS-B -- to the reference's
    published output."""
    import gp_hyper_oracle as HO
    g = np.load(os.path.join(GOLDEN, 'bolfi_doc_run.npz'))
    printed, ours = g['hyper_printed'], g['hyper_oracle']
    # rbf.variance, rbf.lengthscale, bias.variance, Gaussian_noise.variance as printed (12 digits)
    np.testing.assert_allclose(ours[[0, 1, 3]], printed[[0, 1, 3]], rtol=1e-5)
    np.testing.assert_allclose(ours[2], printed[2], rtol=2e-4)          # the flat direction (Ga(0.006, 1) prior)
    assert abs(float(g['objective_oracle']) - float(g['objective_printed_in_doc'])) <= 1e-6 * 151.87
    assert abs(float(g['threshold_oracle']) - float(g['threshold_printed'])) <= 5e-5       # printed with 4 decimals
    # the printed priors: Ga(0.024, 1), Ga(1.3, 1), Ga(0.006, 1) for rbf.variance, lengthscale, bias.variance
    np.testing.assert_allclose(g['priors'][:, 0], [0.024, 1.3, 0.006], rtol=5e-2)        # printed with 2 digits
    assert np.all(g['priors'][:, 1] == 1.0)
    # the fixture is what the oracle computes on its evidence, and its search from the recorded start ends there
    pri = {k: tuple(g['priors'][i]) for i, k in enumerate(('var', 'ls', 'bias'))}
    obj = HO.MapObjective(g['X'], g['Y'], pri)
    assert abs(obj.value(HO.softplus_inv(ours)) - float(g['objective_oracle'])) <= 1e-9 * 151.87
    assert abs(obj.value(HO.softplus_inv(printed)) - float(g['objective_at_printed'])) <= 1e-9 * 151.87
    h, info = HO.optimize(g['X'], g['Y'], dict(zip(HO.ORDER, g['hyper_start'])), pri, max_iters=50)
    np.testing.assert_allclose([h[k] for k in HO.ORDER], ours, rtol=1e-9)
