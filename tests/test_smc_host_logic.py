"""CPU: the host side of elfi_amd.HipAdaptiveDistance / HipSMC / HipAdaptiveDistanceSMC (the node's state sharing between
a model and the sampler's copy, the hand-over of the batch statistics from the operation to add_data, per-column
acceptance, the round bookkeeping) with the device calls replaced by NumPy stand-ins of their contracts -- next to the
reference's own classes on the same model and seed, on the two documented runs of docs/usage/adaptive_distance.rst.
(The device calls themselves: tests/test_adaptive_gpu.py; the two together: tests/test_reference_smc_gpu.py.)"""
import os
import sys

import numpy as np
import pytest
import scipy.stats as ss
from scipy.spatial.distance import cdist

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle')
sys.path.insert(0, ORACLE)
import ref_shim  # noqa: E402
from test_sampler_host_logic import FakeRunningBest, fake_smallest_k  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason='no reference package')

CALLS = {'fused': 0, 'welford': 0}


class ColsRunningBest(FakeRunningBest):
    """+ one acceptance threshold per nested column (elfihip_reject_set_accept_cols)."""

    def push_distances(self, d, row_base=None):
        if self.accept is not None and np.ndim(self.accept) > 0:
            assert np.asarray(d).ndim == 2 and np.asarray(d).shape[1] == len(self.accept)
        super().push_distances(d, row_base=row_base)


def fake_adaptive_batch(X, y, W, store=None, state=None, row_base=None, distances=True, ctx=None):
    """Contract of elfi_amd.distance.adaptive_batch: cdist-exact nested distances; the batch folded into `store` by the
    two-pass statistics + Chan's update."""
    from elfi_amd.sharding import merge_welford
    X = np.asarray(X, dtype=float)
    if X.ndim != 2:
        raise ValueError('XA must be a 2-dimensional array.')
    CALLS['fused'] += 1
    d = np.column_stack([cdist(X, np.atleast_2d(y), 'euclidean', w=w) for w in W])
    out = None
    if store is not None:
        mean = X.mean(axis=0)
        out = merge_welford([store, (len(X), mean, ((X - mean) ** 2).sum(axis=0))])
    return d, out


def fake_welford_update(X, count, mean, M2, ctx=None):
    CALLS['welford'] += 1
    from distance_oracle import AdaptiveDistanceOracle
    a = AdaptiveDistanceOracle()
    a.store = [count, mean, M2]
    a.add_data(X)
    return a.store[0], a.store[1], a.store[2]


def fake_cdist_rows(X, y, metric='euclidean', p=2.0, w=None, V=None, VI=None, ctx=None):
    return cdist(np.asarray(X, dtype=float), np.atleast_2d(y), metric, w=w)[:, 0]


class FakeGM:
    @classmethod
    def logpdf(cls, x, means, cov=1, weights=None):
        from elfi.methods.utils import GMDistribution
        return GMDistribution.logpdf(x, means, cov, weights)


@pytest.fixture()
def elfi(monkeypatch):
    e = ref_shim.install()
    import elfi.clients.native as native
    native.set_as_default()
    import elfi_amd.sampler as S
    import elfi_amd.adaptive as A
    import elfi_amd.smc as M
    from elfi.methods.utils import weighted_var
    monkeypatch.setattr(S, 'RunningBest', ColsRunningBest)
    monkeypatch.setattr(S, 'smallest_k', fake_smallest_k)
    monkeypatch.setattr(A, 'adaptive_batch', fake_adaptive_batch)
    monkeypatch.setattr(A, 'welford_update', fake_welford_update)
    monkeypatch.setattr(A, 'cdist_rows', fake_cdist_rows)
    monkeypatch.setattr(M, 'GMDistribution', FakeGM)
    monkeypatch.setattr(M, 'weighted_var', weighted_var)
    CALLS['fused'] = CALLS['welford'] = 0
    return e


def simulator1(mu, batch_size=1, random_state=None):      # docs/usage/adaptive_distance.rst:43-51
    mu = np.asarray(mu).reshape((-1, 1))
    o1 = ss.norm.rvs(loc=mu, scale=1, random_state=random_state).reshape((-1, 1))
    o2 = ss.norm.rvs(loc=mu, scale=100, random_state=random_state).reshape((-1, 1))
    return np.hstack((o1, o2))


def simulator2(mu, batch_size=1, random_state=None):      # docs/usage/adaptive_distance.rst:231-238
    mu = np.asarray(mu).reshape((-1, 1))
    o1 = ss.norm.rvs(loc=mu, scale=0.1, random_state=random_state).reshape((-1, 1))
    o2 = ss.norm.rvs(loc=1, scale=1, size=batch_size, random_state=random_state).reshape((-1, 1))
    return np.hstack((o1, o2))


def _run(elfi, hip, simulator, prior, observed, batch_size, calls, split=False):
    import elfi_amd
    m = elfi.new_model()
    theta = elfi.Prior(*prior, model=m, name='theta')
    sim = elfi.Simulator(simulator, theta, observed=observed, name='sim')
    if split:       # two (n,) summaries instead of the simulator's (n, 2) output
        parents = (elfi.Summary(lambda y: y[:, 0], sim, name='S1'), elfi.Summary(lambda y: y[:, 1], sim, name='S2'))
    else:
        parents = (sim,)
    d = elfi.Distance('euclidean', *parents, name='d')
    # the documented route: an existing distance node BECOMES the adaptive one (adaptive_distance.rst:136)
    d.become((elfi_amd.HipAdaptiveDistance if hip else elfi.AdaptiveDistance)(*parents))
    cls = elfi_amd.HipAdaptiveDistanceSMC if hip else elfi.AdaptiveDistanceSMC
    smc = cls(d, batch_size=batch_size, seed=123)
    return [smc.sample(*a, bar=False, **k) for a, k in calls], smc


def _same(a, b, rtol, exact_rows=True):
    """exact_rows: the kept parameter values are the same numbers (one population, or the proposal covariance computed by
    the same code).  With the weighted variance on the device (summation order differs from NumPy's BLAS dot by ~1e-13)
    the proposals of the later rounds -- draws scaled by that covariance (samplers.py:441-451) -- differ in their last
    digits, and with them everything simulated from them: the same rows in the same order, values to rtol."""
    assert a.n_sim == b.n_sim and a.n_samples == b.n_samples
    np.testing.assert_allclose(np.array(b.adaptive_distance_w), np.array(a.adaptive_distance_w), rtol=rtol, atol=0)
    np.testing.assert_allclose(b.threshold, a.threshold, rtol=rtol)

    def same_rows(x, y):
        if exact_rows:
            assert np.array_equal(x, y)
        else:
            np.testing.assert_allclose(y, x, rtol=rtol, atol=rtol)
    same_rows(a.samples['theta'], b.samples['theta'])          # the same rows were kept, in the same order
    np.testing.assert_allclose(b.discrepancies, a.discrepancies, rtol=rtol)
    np.testing.assert_allclose(b.weights, a.weights, rtol=10 * rtol)
    for pa, pb in zip(a.populations, b.populations):
        assert pa.n_sim == pb.n_sim
        same_rows(pa.samples['theta'], pb.samples['theta'])
        np.testing.assert_allclose(pb.cov, pa.cov, rtol=10 * rtol)


def test_adaptive_distance_smc_example_1(elfi):
    """adaptive_distance.rst:136-214: one population, quantile 0.01 -> weights [0.06940134, 0.0097677], threshold 0.462."""
    args = (simulator1, (ss.uniform, 0, 50), np.array([20, 20])[None, :], 10000, [((100, 1), dict(quantile=0.01))])
    (ref,), _ = _run(elfi, False, *args)
    (got,), smc = _run(elfi, True, *args)
    assert isinstance(smc, elfi.AdaptiveDistanceSMC) and isinstance(smc.model['d'], elfi.AdaptiveDistance)
    assert type(smc._rejection).__name__ == 'HipRejection'
    assert np.allclose(got.adaptive_distance_w[0], [0.06940134, 0.0097677], rtol=0, atol=5e-9)
    assert abs(got.threshold - 0.462) < 5e-4 and got.n_sim == 10000
    _same(ref, got, 1e-12)
    # every batch's statistics came from the pass that computed its distances (no second look at the data)
    assert CALLS['welford'] == 0 and CALLS['fused'] >= 1


def test_adaptive_distance_smc_example_2_continued(elfi):
    """adaptive_distance.rst:258-378: 1000 samples in 5 rounds, then two more: n_sim 32000 -> 48000, seven weight vectors."""
    doc = np.array([[0.01023228, 1.00584519], [0.00921258, 0.99287166], [0.01201937, 0.99365522],
                    [0.02217631, 0.98925365], [0.04355987, 1.00076738], [0.07863284, 0.9971017],
                    [0.13892778, 1.00929049]])
    args = (simulator2, (ss.norm, 0, 100), np.array([0, 0])[None, :], 2000, [((1000, 5), {}), ((1000, 2), {})])
    ref, _ = _run(elfi, False, *args)
    got, _ = _run(elfi, True, *args)
    assert got[0].n_sim == 32000 and got[1].n_sim == 48000
    assert abs(got[0].threshold - 0.925) < 5e-4 and abs(got[1].threshold - 0.868) < 5e-4
    assert np.allclose(np.array(got[1].adaptive_distance_w), doc, rtol=0, atol=5e-9)
    for a, b in zip(ref, got):
        _same(a, b, 1e-11)
    assert CALLS['welford'] == 0


def test_adaptive_distance_smc_separate_summaries(elfi):
    """Two (n,) summary nodes (the operation stacks them; add_data receives the two arrays)."""
    args = (simulator1, (ss.uniform, 0, 50), np.array([20, 20])[None, :], 5000, [((200, 3), dict(quantile=0.25))])
    (ref,), _ = _run(elfi, False, *args, split=True)
    (got,), _ = _run(elfi, True, *args, split=True)
    _same(ref, got, 1e-11)
    assert CALLS['welford'] == 0


def test_plain_smc_and_reference_sampler_over_the_hip_node(elfi):
    """HipSMC (thresholds and quantiles) next to elfi.SMC; the reference's own AdaptiveDistanceSMC / Rejection over a
    HipAdaptiveDistance node (add_data then receives the arrays the operation saw, through the reference's _merge_batch)."""
    import elfi_amd
    from elfi.examples import ma2
    for kw in (dict(thresholds=[0.5, 0.3, 0.2]), dict(quantiles=[0.5, 0.5, 0.5])):
        ref = elfi.SMC(ma2.get_model(seed_obs=4)['d'], batch_size=2000, seed=5).sample(300, bar=False, **kw)
        got = elfi_amd.HipSMC(ma2.get_model(seed_obs=4)['d'], batch_size=2000, seed=5).sample(300, bar=False, **kw)
        assert got.n_sim == ref.n_sim and got.threshold == ref.threshold
        for k in ('t1', 't2'):
            assert np.array_equal(got.samples[k], ref.samples[k])
        assert np.array_equal(got.weights, ref.weights)
    m = elfi.new_model()
    theta = elfi.Prior(ss.uniform, 0, 50, model=m, name='theta')
    sim = elfi.Simulator(simulator1, theta, observed=np.array([20, 20])[None, :], name='sim')
    d = elfi_amd.HipAdaptiveDistance(sim, name='d')
    got = elfi.AdaptiveDistanceSMC(d, batch_size=2500, seed=3).sample(100, 2, quantile=0.1, bar=False)
    m2 = elfi.new_model()
    theta = elfi.Prior(ss.uniform, 0, 50, model=m2, name='theta')
    sim = elfi.Simulator(simulator1, theta, observed=np.array([20, 20])[None, :], name='sim')
    ref = elfi.AdaptiveDistanceSMC(elfi.AdaptiveDistance(sim, name='d'), batch_size=2500, seed=3).sample(
        100, 2, quantile=0.1, bar=False)
    _same(ref, got, 1e-11)
    assert CALLS['welford'] == 0
