"""GPU: the REAL reference SMC loops driving the HIP objects (BASELINE.json configs[3]'s entry point).

elfi.AdaptiveDistanceSMC(d, ...).sample(n, rounds, quantile) (elfi/methods/inference/samplers.py:562-660) builds a
Rejection per round (:474-487), weighs every population with the Gaussian-mixture proposal density and the weighted
variance of the parameters (:505-534) and drives the AdaptiveDistance node's add_data / update_distance /
nested_distance (elfi/model/elfi_model.py:1104-1151).  Here the node is elfi_amd.HipAdaptiveDistance, the sampler
elfi_amd.HipAdaptiveDistanceSMC (subclasses of the running ELFI's classes; every round's sample state, the nested
distances, the column statistics, the mixture density and the weighted variance on the device) -- checked against

  * the documented runs of docs/usage/adaptive_distance.rst: weights [0.06940134, 0.0097677], threshold 0.462;
    n_sim 32000 -> 48000, thresholds 0.925 / 0.868, the seven printed weight vectors;
  * the reference's own classes on the same model and seed, population by population.

oracle/make_ref.sh copies the reference's Python package to the git-ignored oracle/_ref/ (it ships with the gpurun
snapshot); skipped when it is absent.
"""
import os
import sys

import numpy as np
import pytest
import scipy.stats as ss

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle')
sys.path.insert(0, ORACLE)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shim  # noqa: E402
from test_smc_host_logic import _run, _same, simulator1, simulator2  # noqa: E402

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref_shim.available(), reason='no reference package (run oracle/make_ref.sh)')]


@pytest.fixture(scope='module')
def elfi():
    e = ref_shim.install()
    import elfi.clients.native as native
    native.set_as_default()
    return e


@pytest.fixture()
def device_calls(monkeypatch):
    """Counts the device passes the node makes: `fused` = operation (distances + batch statistics in one read),
    `welford` = add_data having to look at the data itself."""
    import elfi_amd.adaptive as A
    calls = {'fused': 0, 'welford': 0}
    fused, welford = A.adaptive_batch, A.welford_update

    def count_fused(*a, **k):
        calls['fused'] += 1
        return fused(*a, **k)

    def count_welford(*a, **k):
        calls['welford'] += 1
        return welford(*a, **k)
    monkeypatch.setattr(A, 'adaptive_batch', count_fused)
    monkeypatch.setattr(A, 'welford_update', count_welford)
    return calls


def test_documented_run_1_through_the_device_classes(hip_ctx, elfi, device_calls):
    """adaptive_distance.rst:136-214."""
    args = (simulator1, (ss.uniform, 0, 50), np.array([20, 20])[None, :], 10000, [((100, 1), dict(quantile=0.01))])
    (got,), smc = _run(elfi, True, *args)
    assert isinstance(smc, elfi.AdaptiveDistanceSMC) and isinstance(smc.model['d'], elfi.AdaptiveDistance)
    assert type(smc.model['d']).__name__ == 'HipAdaptiveDistance' and type(smc._rejection).__name__ == 'HipRejection'
    assert np.allclose(got.adaptive_distance_w[0], [0.06940134, 0.0097677], rtol=0, atol=5e-9)
    assert abs(got.threshold - 0.462) < 5e-4 and got.n_sim == 10000 and got.n_samples == 100
    assert device_calls['welford'] == 0 and device_calls['fused'] >= 2
    (ref,), _ = _run(elfi, False, *args)
    _same(ref, got, 1e-11)


@pytest.mark.timeout(900)
def test_documented_run_2_continued_through_the_device_classes(hip_ctx, elfi, device_calls):
    """adaptive_distance.rst:258-378: sample(1000, 5), then sample(1000, 2) on the same sampler object."""
    doc = np.array([[0.01023228, 1.00584519], [0.00921258, 0.99287166], [0.01201937, 0.99365522],
                    [0.02217631, 0.98925365], [0.04355987, 1.00076738], [0.07863284, 0.9971017],
                    [0.13892778, 1.00929049]])
    args = (simulator2, (ss.norm, 0, 100), np.array([0, 0])[None, :], 2000, [((1000, 5), {}), ((1000, 2), {})])
    got, _ = _run(elfi, True, *args)
    assert got[0].n_sim == 32000 and got[1].n_sim == 48000
    assert abs(got[0].threshold - 0.925) < 5e-4 and abs(got[1].threshold - 0.868) < 5e-4
    assert np.allclose(np.array(got[1].adaptive_distance_w), doc, rtol=0, atol=5e-9)
    assert device_calls['welford'] == 0
    ref, _ = _run(elfi, False, *args)
    for a, b in zip(ref, got):
        _same(a, b, 1e-9, exact_rows=False)


def test_separate_summary_nodes_and_the_reference_sampler_over_the_hip_node(hip_ctx, elfi, device_calls):
    import elfi_amd
    args = (simulator1, (ss.uniform, 0, 50), np.array([20, 20])[None, :], 5000, [((200, 3), dict(quantile=0.25))])
    (ref,), _ = _run(elfi, False, *args, split=True)
    (got,), _ = _run(elfi, True, *args, split=True)
    _same(ref, got, 1e-9, exact_rows=False)
    assert device_calls['welford'] == 0
    # the reference's own AdaptiveDistanceSMC (host sample state, reference _merge_batch) over the device node
    def model(hip):
        m = elfi.new_model()
        theta = elfi.Prior(ss.uniform, 0, 50, model=m, name='theta')
        sim = elfi.Simulator(simulator1, theta, observed=np.array([20, 20])[None, :], name='sim')
        return (elfi_amd.HipAdaptiveDistance if hip else elfi.AdaptiveDistance)(sim, name='d')
    got = elfi.AdaptiveDistanceSMC(model(True), batch_size=2500, seed=3).sample(100, 2, quantile=0.1, bar=False)
    ref = elfi.AdaptiveDistanceSMC(model(False), batch_size=2500, seed=3).sample(100, 2, quantile=0.1, bar=False)
    _same(ref, got, 1e-10)       # (the reference's sampler: its own weighted variance, only the node differs)
    assert device_calls['welford'] == 0


def test_hip_smc_equals_elfi_smc(hip_ctx, elfi):
    """elfi.SMC (samplers.py:319-549) with thresholds and with quantiles on MA2, population by population; the HIP
    distance node in the model."""
    import elfi_amd
    from elfi.examples import ma2

    def model():
        m = ma2.get_model(seed_obs=4)
        m['d'].become(elfi.Distance(elfi_amd.HipDistance('euclidean'), m['S1'], m['S2'], model=m))
        return m['d']
    for kw in (dict(thresholds=[0.5, 0.3, 0.2]), dict(quantiles=[0.5, 0.5, 0.5])):
        ref = elfi.SMC(ma2.get_model(seed_obs=4)['d'], batch_size=2000, seed=5).sample(300, bar=False, **kw)
        got = elfi_amd.HipSMC(model(), batch_size=2000, seed=5).sample(300, bar=False, **kw)
        # population 0 comes from the prior: bit for bit; the later ones from proposals scaled by the previous
        # population's weighted variance (device summation order: ~1e-13), so to rounding
        assert got.n_sim == ref.n_sim
        np.testing.assert_allclose(got.threshold, ref.threshold, rtol=1e-9)
        for k in ('t1', 't2'):
            np.testing.assert_allclose(got.samples[k], ref.samples[k], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(got.discrepancies, ref.discrepancies, rtol=1e-8)
        np.testing.assert_allclose(got.weights, ref.weights, rtol=1e-8)
        p0r, p0g = ref.populations[0], got.populations[0]
        assert np.array_equal(p0r.samples['t1'], p0g.samples['t1']) and np.array_equal(p0r.discrepancies, p0g.discrepancies)
        for pa, pb in zip(ref.populations, got.populations):
            assert pa.n_sim == pb.n_sim
            np.testing.assert_allclose(pb.samples['t1'], pa.samples['t1'], rtol=1e-8, atol=1e-10)
            np.testing.assert_allclose(pb.cov, pa.cov, rtol=1e-9)


def test_device_proposals_and_device_priors_give_the_same_posterior(hip_ctx, elfi):
    """device_proposals = True (the proposals of rounds 2.. from elfi_amd.GMDistribution.rvs) with the prior drawn by
    elfi_amd.priors.uniform: another random realisation of the same sampler -- the population statistics agree with the
    reference classes' within Monte-Carlo error, thresholds shrink round by round, every sample lies in the prior's
    support."""
    import elfi_amd
    from elfi_amd import priors

    def noisy(mu, batch_size=1, random_state=None):
        mu = np.asarray(mu, dtype=float).reshape((-1, 1))
        return mu + random_state.randn(len(mu), 1)

    def build(device):
        m = elfi.new_model()
        mu = elfi.Prior(priors.uniform if device else ss.uniform, 0, 50, model=m, name='mu')
        y = elfi.Simulator(noisy, mu, observed=np.array([[20.0]]), name='y')
        d = elfi.Distance('euclidean', y, name='d')
        return m, d

    _, d_ref = build(False)
    ref = elfi.SMC(d_ref, batch_size=5000, seed=3).sample(2000, thresholds=[5, 2, 1], bar=False)
    _, d_dev = build(True)
    smc = elfi_amd.HipSMC(d_dev, batch_size=5000, seed=3)
    smc.device_proposals = True
    got = smc.sample(2000, thresholds=[5, 2, 1], bar=False)
    assert got.n_populations == 3 and got.threshold <= 1.0
    mu_g, mu_r = got.samples['mu'], ref.samples['mu']
    assert np.all((mu_g >= 0) & (mu_g <= 50))
    se = np.sqrt(np.var(mu_r) / len(mu_r) * 4)        # (weights: a generous effective sample size)
    assert abs(np.average(mu_g, weights=got.weights) - np.average(mu_r, weights=ref.weights)) < 6 * se + 0.05
    assert 0.5 < np.std(mu_g) / np.std(mu_r) < 2.0
    # with the option off the device class IS the reference class sample by sample (test_hip_smc_equals_elfi_smc)
    assert hip_smc_default_is_off(elfi_amd)


def hip_smc_default_is_off(elfi_amd):
    return elfi_amd.smc.hip_smc_class('SMC').device_proposals is False


def test_node_state_pickles_and_saves(hip_ctx, elfi, tmp_path):
    """A model with the device node saves and loads (elfi_model.py:401-438 pickles the node states, the node class by
    reference) and keeps computing."""
    import pickle
    import elfi_amd
    m = elfi.new_model()
    theta = elfi.Prior(ss.uniform, 0, 50, model=m, name='theta')
    sim = elfi.Simulator(simulator1, theta, observed=np.array([20, 20])[None, :], name='sim')
    d = elfi_amd.HipAdaptiveDistance(sim, name='d')
    a = d.generate(50)
    m2 = pickle.loads(pickle.dumps(m))
    assert type(m2['d']).__name__ == 'HipAdaptiveDistance' and isinstance(m2['d'], elfi.AdaptiveDistance)
    assert m2['d'].generate(50).shape == a.shape == (50,)
