"""HipBolfiPosterior and the lock-step posterior sampler on the GPU.

Pinned by tests/golden/bolfi_posterior.npz (the reference's own BolfiPosterior + ModelPrior on the
same GP, oracle/make_golden_posterior.py) and by the CPU restatement oracle/posterior_oracle.py;
the chain algorithms themselves are pinned bit-exactly on the CPU (tests/test_chains.py).
"""
import os

import numpy as np
import pytest

import gp_oracle as G
import posterior_oracle as PO
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _model(g):
    import elfi_amd
    X, y = g['X'], g['y']
    d = X.shape[1]
    names = ['p%d' % i for i in range(d)]
    bounds = [tuple(b) for b in g['bounds']]
    m = elfi_amd.HipGPRegression(names, bounds=dict(zip(names, bounds)))
    m.update(X, y)
    m._hyper = dict(zip(('var', 'ls', 'bias', 'noise'), (float(v) for v in g['hyper'])))
    m._refit()
    return m, bounds


def test_posterior_equals_the_reference_bolfi_posterior(hip_ctx):
    import elfi_amd
    g = np.load(os.path.join(GOLDEN, 'bolfi_posterior.npz'))
    m, bounds = _model(g)
    prior = PO.BoxPrior(bounds)        # = the ModelPrior of the fixture (pinned in test_oracle_pinning_gp.py)
    bp = elfi_amd.HipBolfiPosterior(m, threshold=float(g['threshold']), prior=prior)
    xs = g['xs']
    lp, gr = bp.logpdf_and_gradient(xs)
    np.testing.assert_allclose(lp, g['logpdf'], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(gr, g['grad'], rtol=1e-6, atol=1e-7)
    # point-wise interface: shapes and values as the reference's
    for i in (0, 3, 7):
        v = bp.logpdf(xs[i])
        assert np.ndim(v) == 0 or np.shape(v) == (1,)
        np.testing.assert_allclose(np.ravel(v)[0], g['logpdf'][i], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(np.ravel(bp.gradient_logpdf(xs[i])), g['grad'][i], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(bp._unnormalized_loglikelihood(xs), g['loglik'], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(bp._gradient_unnormalized_loglikelihood(xs), g['gradlik'], rtol=1e-6, atol=1e-7)
    outside = ~np.all((xs >= -2) & (xs <= 2), axis=1)
    assert np.all(np.isneginf(lp[outside])) and np.all(gr[outside] == 0)
    # threshold left to the posterior: the minimum of the GP mean, as the reference found it
    auto = elfi_amd.HipBolfiPosterior(m, threshold=None, prior=prior, n_inits=10, seed=0)
    assert abs(auto.threshold - float(g['threshold_auto'])) <= 1e-6 * max(1.0, abs(float(g['threshold_auto'])))


def test_lockstep_sampler_on_the_device_posterior(hip_ctx):
    import elfi_amd
    from elfi_amd import chains
    g = np.load(os.path.join(GOLDEN, 'bolfi_posterior.npz'))
    m, bounds = _model(g)
    prior = PO.BoxPrior(bounds)
    thr = float(g['threshold'])
    out, bp = elfi_amd.sample_posterior(m, prior, 200, n_chains=4, threshold=thr, seed=3)
    assert out.shape == (4, 200, 2) and np.all(np.isfinite(out))
    lo, hi = np.array(bounds).T
    assert np.all(out >= lo) and np.all(out <= hi), 'zero density outside the bounds: no sample can leave them'
    rounds, points = chains.run_lockstep.n_rounds, chains.run_lockstep.n_points
    assert points > 2.5 * rounds, 'four chains share each device call (%d points in %d calls)' % (points, rounds)
    again, _ = elfi_amd.sample_posterior(m, prior, 200, n_chains=4, threshold=thr, seed=3)
    assert np.array_equal(out, again), 'same seed, same chains (deterministic kernels)'
    # the same chains driven by the CPU restatement of the posterior: identical random draws, evaluations equal to
    # ~1e-9, so the trajectories agree until chaos amplifies the rounding differences
    post = G.Posterior(g['X'], g['y'], *g['hyper'])
    po = PO.PosteriorOracle(post, bounds, thr)
    pool = g['X'][np.argsort(g['y'][:, 0] if g['y'].ndim == 2 else g['y'])]
    seeds = [elfi_amd.posterior.sub_seed(3, ii) for ii in range(4)]
    cpu = chains.nuts(200, pool[:4], po.logpdf_and_gradient, seeds=seeds, n_adapt=100)
    np.testing.assert_allclose(out[:, :5], cpu[:, :5], rtol=1e-6, atol=1e-7)
    # ... and as samples of the same density afterwards
    a, b = out[:, 100:].reshape(-1, 2), cpu[:, 100:].reshape(-1, 2)
    assert np.all(np.abs(a.mean(0) - b.mean(0)) < 0.5 * np.maximum(a.std(0), b.std(0)) + 0.05)
    # Metropolis with the reference's default proposal widths
    met, _ = elfi_amd.sample_posterior(m, prior, 300, warmup=50, n_chains=3, threshold=thr, algorithm='metropolis',
                                       seed=4)
    assert met.shape == (3, 300, 2) and np.all(met >= lo) and np.all(met <= hi)
    with pytest.raises(ValueError):
        elfi_amd.sample_posterior(m, prior, 10, algorithm='gibbs')
    with pytest.raises(ValueError):
        elfi_amd.sample_posterior(m, prior, 10, n_chains=2, initials=np.zeros((3, 2)))


def test_sub_seed_matches_the_reference_formula():
    import elfi_amd
    # elfi.utils.get_sub_seed(123, i) for i = 0..4, recorded from the reference
    rs = np.random.RandomState(123)
    seen = []
    while len(set(seen)) < 5:
        seen.extend(rs.randint(2 ** 31, size=5 - len(set(seen)), dtype='uint32').tolist())
    assert int(elfi_amd.posterior.sub_seed(123, 0)) == seen[0]
    assert len({int(elfi_amd.posterior.sub_seed(123, i)) for i in range(5)}) == 5
