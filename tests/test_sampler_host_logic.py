"""CPU: the host side of elfi_amd.HipRejection (sample gather, acceptable-row count, re-rank) with the device state
replaced by a NumPy stand-in of its contract -- next to the reference's Rejection on the same model and seed.
(The device state itself: tests/test_selection_gpu.py; the two together: tests/test_reference_loop_gpu.py.)"""
import os
import sys

import numpy as np
import pytest

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle')
sys.path.insert(0, ORACLE)
import ref_shim  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason='no reference package')


class FakeRunningBest:
    """Contract of elfi_amd.selection.RunningBest (csrc/reject.hip) in NumPy."""

    def __init__(self, k, accept=None, **kw):
        self.k, self.accept = k, accept
        self.vals, self.rows = np.empty(0), np.empty(0, dtype=np.int64)
        self.acc_total = self.acc_seen = 0

    def push_distances(self, d, row_base=None):
        d = np.asarray(d, dtype=float)
        D = d.reshape(len(d), -1)
        ok = np.ones(len(D), dtype=bool) if self.accept is None else np.all(D <= self.accept, axis=1)
        ok &= ~np.isnan(D[:, -1])
        self.acc_total += int(ok.sum()) if self.accept is not None else 0
        v = np.concatenate([self.vals, D[ok, -1]])
        r = np.concatenate([self.rows, row_base + np.nonzero(ok)[0]])
        order = np.lexsort((r, v))[:self.k]
        self.vals, self.rows = v[order], r[order]

    def result(self):
        return self.vals.copy(), self.rows.copy()

    def meta(self):
        last = self.acc_total - self.acc_seen
        self.acc_seen = self.acc_total
        return (self.vals[self.k - 1] if len(self.vals) == self.k else np.inf), last, self.acc_total


def fake_smallest_k(d, k, ctx=None):
    d = np.asarray(d, dtype=float)
    order = np.lexsort((np.arange(len(d)), np.where(np.isnan(d), np.inf, d), np.isnan(d)))[:k]
    return d[order], order


@pytest.fixture()
def elfi(monkeypatch):
    e = ref_shim.install()
    import elfi.clients.native as native
    native.set_as_default()
    import elfi_amd.sampler as S
    monkeypatch.setattr(S, 'RunningBest', FakeRunningBest)
    monkeypatch.setattr(S, 'smallest_k', fake_smallest_k)
    return e


@pytest.mark.parametrize('kwargs', [dict(n_sim=30000), dict(quantile=0.02), dict(threshold=0.25), dict(threshold=2.0)])
def test_hip_rejection_host_logic_equals_the_reference(elfi, kwargs):
    import elfi_amd.sampler as S
    from elfi.examples import ma2
    ref = elfi.Rejection(ma2.get_model(seed_obs=4)['d'], batch_size=1000, seed=1).sample(500, bar=False, **kwargs)
    got = S.HipRejection(ma2.get_model(seed_obs=4)['d'], batch_size=1000, seed=1).sample(500, bar=False, **kwargs)
    assert isinstance(S.hip_rejection_class()(ma2.get_model(seed_obs=4)['d'], batch_size=10), elfi.Rejection)
    assert got.n_sim == ref.n_sim
    assert got.threshold == ref.threshold
    assert np.array_equal(got.discrepancies, ref.discrepancies)
    for k in ('t1', 't2'):
        assert np.array_equal(got.samples[k], ref.samples[k])


def test_hip_rejection_adaptive_distance_rerank(elfi):
    import scipy.stats as ss
    import elfi_amd.sampler as S

    def sim(mu, batch_size=1, random_state=None):
        rs = random_state or np.random
        return np.column_stack([rs.normal(mu, 1.0, batch_size), rs.normal(mu, 30.0, batch_size)])

    def make():
        m = elfi.new_model()
        mu = elfi.Prior(ss.uniform, 0, 40, model=m, name='mu')
        Y = elfi.Simulator(sim, mu, observed=np.array([[20.0, 20.0]]), name='Y')
        S1 = elfi.Summary(lambda y: y[:, 0], Y, name='S1')
        S2 = elfi.Summary(lambda y: y[:, 1], Y, name='S2')
        return elfi.AdaptiveDistance(S1, S2, name='d')
    res = []
    for cls in (elfi.Rejection, S.HipRejection):
        rej = cls(make(), batch_size=2000, seed=7, output_names=['S1', 'S2'])
        assert rej.adaptive
        res.append(rej.sample(300, n_sim=20000, bar=False))
    a, b = res
    assert a.n_sim == b.n_sim == 20000
    assert np.array_equal(a.samples['mu'], b.samples['mu']) and np.array_equal(a.discrepancies, b.discrepancies)
    assert np.array_equal(a.outputs['S1'], b.outputs['S1'])
