"""GPU: `elfi_amd.HipBOLFI` -- the reference's `elfi.BOLFI` (bolfi.py:386-580) with surrogate, acquisition and posterior
chains on the device -- through the reference's own entry points `fit` / `extract_posterior` / `sample`.

  * the documented run (docs/usage/BOLFI.rst:36-153,229-293: MA2 seed_obs=1, BOLFI(log_d, batch_size=1,
    initial_evidence=20, update_interval=10, bounds, acq_noise_var=0.1, seed=1).fit(200), then `sample(1000)`): with the
    run's 200 evidence points (tests/golden/bolfi_doc_run.npz, the reference's loop replayed in the build container) and
    the hyper-parameters the documentation prints, `HipBOLFI.sample(1000)` must find the printed threshold -1.6146 and the
    printed sample means t1 0.429, t2 0.0277 -- to the Monte-Carlo error of 2000 NUTS samples (the doc's own effective
    sample size is 2200: one standard error of a mean is 0.006-0.007; the tolerance below is 0.035 = five of them for
    the difference of two such runs), since NUTS trajectories are chaotic in the last bits of the density;
  * next to the reference's `BOLFI.sample` on the same evidence, hyper-parameters, threshold and seed (the reference's
    loop over the CPU oracle model, its own BolfiPosterior and mcmc.nuts): same initial points, the chains equal for their
    first iterations (1e-6: identical random draws, evaluations equal to ~1e-9) and equal as samples afterwards; the
    result object is the reference's BolfiSample with the reference's fields.
"""
import os
import sys

import numpy as np
import pytest

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle')
sys.path.insert(0, ORACLE)
import ref_shim  # noqa: E402
from conftest import GOLDEN  # noqa: E402

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref_shim.available(), reason='no reference package (run oracle/make_ref.sh)')]

BOUNDS = {'t1': (-2, 2), 't2': (-1, 1)}


@pytest.fixture(scope='module')
def elfi():
    e = ref_shim.install()
    import elfi.clients.native as native
    native.set_as_default()
    return e


def _doc_model(elfi):
    from elfi.examples import ma2
    m = ma2.get_model(seed_obs=1)
    log_d = elfi.Operation(np.log, m['d'], name='log_d')
    return m, log_d


def _set_hyper(model, h):
    model._hyper = dict(zip(('var', 'ls', 'bias', 'noise'), (float(v) for v in h)))
    model._refit()


def test_doc_run_posterior_sample_through_hip_bolfi(hip_ctx, elfi, capsys):
    import elfi_amd
    g = np.load(os.path.join(GOLDEN, 'bolfi_doc_run.npz'))
    m, log_d = _doc_model(elfi)
    pre = {'t1': g['X'][:, 0], 't2': g['X'][:, 1], 'log_d': g['Y'][:, 0]}
    bolfi = elfi_amd.HipBOLFI(log_d, batch_size=1, initial_evidence=pre, update_interval=10, bounds=BOUNDS,
                              acq_noise_var=0.1, seed=1)
    assert isinstance(bolfi, elfi.BOLFI)
    assert isinstance(bolfi.target_model, elfi_amd.HipGPRegression) and isinstance(bolfi.acquisition_method, elfi_amd.HipLCBSC)
    assert bolfi.state['n_evidence'] == 200 and bolfi.target_model.n_evidence == 200
    _set_hyper(bolfi.target_model, g['hyper_printed'])          # docs/usage/BOLFI.rst:144-153
    post = bolfi.extract_posterior()
    assert isinstance(post, elfi_amd.HipBolfiPosterior)
    assert abs(post.threshold - (-1.6146)) < 5e-4, post.threshold          # BOLFI.rst:236
    res = bolfi.sample(1000, n_evidence=200)
    from elfi.methods.results import BolfiSample
    assert isinstance(res, BolfiSample)
    assert res.n_samples == 2000 and res.n_sim == 200 and res.chains.shape == (4, 1000, 2)
    assert abs(res.threshold - (-1.6146)) < 5e-4
    assert abs(res.sample_means['t1'] - 0.429) < 0.035, res.sample_means     # BOLFI.rst:293
    assert abs(res.sample_means['t2'] - 0.0277) < 0.035, res.sample_means
    out = capsys.readouterr().out
    assert '4 chains of 1000 iterations acquired' in out and 't1 ' in out and 't2 ' in out
    # the triangle of the MA2 prior and the GP's box hold every sample
    t1, t2 = res.samples['t1'], res.samples['t2']
    assert np.all(np.abs(t1) <= 2) and np.all(np.abs(t2) <= 1) and np.all(t2 + t1 >= -1) and np.all(t2 - t1 >= -1)
    # same call, same seed: same chains
    again = bolfi.sample(1000, n_evidence=200)
    assert np.array_equal(again.chains, res.chains)
    # Metropolis through the same entry point, the reference's argument checks
    met = bolfi.sample(400, warmup=100, n_chains=2, algorithm='metropolis', sigma_proposals={'t1': 0.4, 't2': 0.2},
                       n_evidence=200)
    assert met.chains.shape == (2, 400, 2)
    with pytest.raises(ValueError, match='Unknown posterior sampler'):
        bolfi.sample(10, algorithm='gibbs', n_evidence=200)
    with pytest.raises(ValueError, match='shape of initials'):
        bolfi.sample(10, n_chains=2, initials=np.zeros((3, 2)), n_evidence=200)


def test_sample_next_to_the_reference_bolfi_sample(hip_ctx, elfi):
    import elfi_amd
    from oracle_gp_model import OracleGPRegression
    g = np.load(os.path.join(GOLDEN, 'bolfi_doc_run.npz'))
    n = 80
    pre = {'t1': g['X'][:n, 0], 't2': g['X'][:n, 1], 'log_d': g['Y'][:n, 0]}
    h = dict(zip(('var', 'ls', 'bias', 'noise'), (float(v) for v in g['hyper_printed'])))
    thr = -1.2
    # the reference: its BOLFI, its BolfiPosterior, its mcmc.nuts per chain, over the CPU oracle model
    m1, log_d1 = _doc_model(elfi)
    ref = elfi.BOLFI(log_d1, batch_size=1, initial_evidence=pre, update_interval=10, bounds=BOUNDS,
                     target_model=OracleGPRegression(['t1', 't2'], bounds=BOUNDS), acq_noise_var=0.1, seed=7)
    ref.target_model.hyper = dict(h)
    ref.target_model._refit()
    r_ref = ref.sample(60, n_chains=2, threshold=thr, n_evidence=n)
    # the device path through the same entry point
    m2, log_d2 = _doc_model(elfi)
    hip = elfi_amd.HipBOLFI(log_d2, batch_size=1, initial_evidence=pre, update_interval=10, bounds=BOUNDS,
                            acq_noise_var=0.1, seed=7)
    _set_hyper(hip.target_model, g['hyper_printed'])
    r_hip = hip.sample(60, n_chains=2, threshold=thr, n_evidence=n)
    assert type(r_hip) is type(r_ref)
    for attr in ('n_samples', 'n_sim', 'warmup', 'seed', 'threshold', 'n_chains', 'parameter_names', 'method_name'):
        assert getattr(r_hip, attr) == getattr(r_ref, attr), attr
    assert r_hip.chains.shape == r_ref.chains.shape == (2, 60, 2)
    np.testing.assert_allclose(r_hip.chains[:, 0], r_ref.chains[:, 0], rtol=0, atol=1e-12)  # same initial points, same first draws
    np.testing.assert_allclose(r_hip.chains[:, :4], r_ref.chains[:, :4], rtol=1e-6, atol=1e-7)
    a, b = r_hip.chains[:, 30:].reshape(-1, 2), r_ref.chains[:, 30:].reshape(-1, 2)
    assert np.all(np.abs(a.mean(0) - b.mean(0)) < 0.6 * np.maximum(a.std(0), b.std(0)) + 0.1)
    # the posterior objects agree point by point
    p_ref, p_hip = ref.extract_posterior(thr), hip.extract_posterior(thr)
    xs = np.random.RandomState(0).uniform([-1.5, -0.8], [1.5, 0.8], (20, 2))
    for x in xs:
        a, b = float(np.ravel(p_ref.logpdf(x))[0]), float(np.ravel(p_hip.logpdf(x))[0])
        assert (np.isinf(a) and np.isinf(b)) or abs(a - b) <= 1e-7 * max(1.0, abs(a))


def test_fit_then_sample_from_scratch(hip_ctx, elfi):
    """The whole user journey on the device defaults: HipBOLFI(...).fit(60) acquires through HipLCBSC, sample() runs the
    lock-step chains; the reference's statistical assertion for this model (tests/functional/test_inference.py:136-190:
    posterior mean within 0.2 of the data-generating (0.6, 0.2) ... at n=300; looser here at 60 points)."""
    import elfi_amd
    m, log_d = _doc_model(elfi)
    bolfi = elfi_amd.HipBOLFI(log_d, batch_size=1, initial_evidence=20, update_interval=10, bounds=BOUNDS,
                              acq_noise_var=0.1, seed=1)
    post = bolfi.fit(n_evidence=60, bar=False)
    assert isinstance(post, elfi_amd.HipBolfiPosterior)
    assert bolfi.target_model.n_evidence == 60 and bolfi.state['n_batches'] == 60
    res = bolfi.sample(200, n_chains=2)
    assert res.chains.shape == (2, 200, 2) and np.all(np.isfinite(res.chains))
    assert abs(res.sample_means['t1'] - 0.6) < 0.5 and abs(res.sample_means['t2'] - 0.2) < 0.5
