"""Weighted population statistics: oracle pinned to the reference's outputs, host quantile
against the reference's outputs (CPU), device variance against oracle and reference (GPU)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'weighted_stats.npz')


@pytest.fixture(scope='module')
def gold():
    return np.load(GOLD)


def _w(gold, key):
    return gold[key] if key in gold.files else None


def test_oracle_weighted_var_matches_reference(gold):
    import weighted_oracle as O
    for k in gold['var_cases']:
        got = np.asarray(O.weighted_var(gold['x_%d' % k], _w(gold, 'w_%d' % k)))
        np.testing.assert_allclose(got, gold['var_%d' % k], rtol=1e-12, atol=0)


def test_oracle_and_host_quantile_equal_reference_exactly(gold):
    import weighted_oracle as O
    from elfi_amd.weighted import weighted_sample_quantile
    for k in gold['q_cases']:
        x, w = gold['qx_%d' % k], _w(gold, 'qw_%d' % k)
        alphas, ref = gold['qalpha_%d' % k], gold['q_%d' % k]
        step = 1 if len(x) <= 5000 else 3        # the oracle is a Python loop
        for a, r in list(zip(alphas, ref))[::step]:
            assert O.weighted_sample_quantile(x, a, w) == r
        for a, r in zip(alphas, ref):
            assert weighted_sample_quantile(x, a, weights=w) == r
    with pytest.raises(IndexError):                # as the reference: no sample reaches alpha > 1
        weighted_sample_quantile(np.arange(4.0), 1.5)


@pytest.mark.gpu
def test_device_weighted_var_matches_reference_and_oracle(hip_ctx, gold):
    import elfi_amd
    import weighted_oracle as O
    for k in gold['var_cases']:
        x, w = gold['x_%d' % k], _w(gold, 'w_%d' % k)
        got = np.asarray(elfi_amd.weighted_var(x, w))
        assert got.shape == np.asarray(gold['var_%d' % k]).shape
        np.testing.assert_allclose(got, gold['var_%d' % k], rtol=1e-12, atol=0)   # tolerance: summation order only
        np.testing.assert_allclose(got, O.weighted_var(x, w), rtol=1e-12, atol=0)
        again = np.asarray(elfi_amd.weighted_var(x, w))
        assert np.array_equal(got, again)                                          # deterministic


@pytest.mark.gpu
def test_device_weighted_var_large_and_strided(hip_ctx):
    import ctypes as C
    import elfi_amd
    import weighted_oracle as O
    from elfi_amd import _lib
    rs = np.random.RandomState(9)
    x = rs.randn(300000, 6) * [1, 10, 100, 0.1, 5, 50] + 1000.0
    w = rs.gamma(0.5, 1.0, len(x))
    np.testing.assert_allclose(elfi_amd.weighted_var(x, w), O.weighted_var(x, w), rtol=1e-11)
    np.testing.assert_allclose(elfi_amd.weighted_var(x), np.var(x, axis=0, ddof=1), rtol=1e-11)   # unit weights
    # a column block of a wider matrix through the pitch argument of the C ABI
    wide = np.ascontiguousarray(rs.randn(5000, 9))
    out = np.empty(4)
    ctx = _lib.default_context()
    ctx.call("elfihip_weighted_var", C.c_void_p(wide.ctypes.data + 2 * 8), 5000, 4, 9, None, _lib.ptr(out))
    np.testing.assert_allclose(out, np.var(wide[:, 2:6], axis=0, ddof=1), rtol=1e-12)
    with pytest.raises(ValueError):
        elfi_amd.weighted_var(x, w[:-1])


def test_host_quantile_equals_oracle_on_random_cases():
    """Property test (CPU): mirror == oracle for random samples with ties, zero weights and boundary alphas."""
    from hypothesis import given, settings, strategies as st
    import weighted_oracle as O
    from elfi_amd.weighted import weighted_sample_quantile

    @settings(max_examples=200, deadline=None)
    @given(st.integers(1, 40), st.integers(0, 2 ** 31 - 1), st.booleans(), st.booleans(),
           st.one_of(st.floats(0.0, 1.0), st.integers(0, 40)))
    def check(n, seed, ties, equal, a):
        rs = np.random.RandomState(seed)
        x = rs.randn(n)
        if ties:
            x = np.round(x)
        w = None if equal else rs.uniform(0, 1, n) * (rs.uniform(0, 1, n) > 0.2)
        if w is not None and w.sum() == 0:
            w[0] = 1.0
        alpha = float(a) if isinstance(a, float) else min(a, n) / n        # k / n sits on a cumulative-weight boundary
        try:
            ref = O.weighted_sample_quantile(x, alpha, w)
        except IndexError:
            with pytest.raises(IndexError):
                weighted_sample_quantile(x, alpha, weights=w)
            return
        assert weighted_sample_quantile(x, alpha, weights=w) == ref

    check()
