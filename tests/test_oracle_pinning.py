"""Pin the CPU oracle (oracle/distance_oracle.py) to the reference's own known answers.

Sources (reference = elfi v0.8.7):
  * tests/unit/test_elfi_model.py:139-153  Distance == np.linalg.norm(sim-obs), exact
  * tests/unit/test_elfi_model.py:186-218  Welford scale == np.std over 10+10+1 rows
  * tests/unit/test_elfi_model.py:220-253  w == 1/std, nested distances == sqrt(sum((s-o)/scale)^2)
  * docs/usage/tutorial.rst:396            threshold 0.116859716394976
  * docs/usage/adaptive_distance.rst:214,372-378  adaptive weights
through fixtures produced by oracle/make_golden.py from the REAL reference.
"""
import hashlib
import os

import numpy as np
import pytest

import distance_oracle as O
from conftest import GOLDEN


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


def parse_case(g, case):
    m, k, metric, kws = case.split('|')
    m = int(m)
    kw = {}
    for item in filter(None, kws.split(',')):
        a, b = item.split('=')
        if b in ('w', 'w0', 'V', 'VI'):
            kw[a] = g['%s_%d' % ({'V': 'w'}.get(b, b), m)]
        else:
            kw[a] = float(b)
    return m, int(k), metric, kw


def test_metrics_fixture_matches_oracle_bitwise():
    g = np.load(os.path.join(GOLDEN, 'metrics.npz'))
    for case in g['cases']:
        m, k, metric, kw = parse_case(g, str(case))
        X, y = g['X_%d' % m], g['y_%d' % m]
        ref = g['d_%d_%d' % (m, k)]
        got = O.make_distance(metric, **dict(kw))(X, observed=(y,))
        assert np.array_equal(got, ref), case


def test_sequential_restatement_matches_cdist_bitwise():
    g = np.load(os.path.join(GOLDEN, 'metrics.npz'))
    for case in g['cases']:
        m, k, metric, kw = parse_case(g, str(case))
        if metric in ('seuclidean', 'mahalanobis') or m > 33:
            continue
        X, y = g['X_%d' % m][:64], g['y_%d' % m]
        ref = g['d_%d_%d' % (m, k)][:64]
        got = O.cdist_rows_sequential(X, y, metric, p=kw.get('p', 2.0), w=kw.get('w'))
        if metric == 'minkowski' and kw.get('p') not in (1.0, 2.0, np.inf):
            np.testing.assert_allclose(got, ref, rtol=4e-16)  # libm pow vs numpy pow
        else:
            assert np.array_equal(got, ref), case


def test_reference_unit_test_become_restated():
    # tests/unit/test_elfi_model.py:139-153
    rs = np.random.RandomState(0)
    s1, s2 = rs.randn(100), rs.randn(100)
    o1, o2 = rs.randn(1), rs.randn(1)
    d = O.make_distance('euclidean')(s1, s2, observed=(o1, o2))
    ref = np.linalg.norm(np.column_stack((s1, s2)) - np.array([o1[0], o2[0]]), axis=1)
    assert np.array_equal(d, ref)


def test_reference_unit_test_adaptive_restated():
    # tests/unit/test_elfi_model.py:186-253
    rs = np.random.RandomState(1)
    a = O.AdaptiveDistanceOracle()
    d1, d2, d3 = rs.randn(10, 3) * [1, 10, 100], rs.randn(10, 3) * [1, 10, 100], rs.randn(1, 3)
    a.add_data(d1)
    a.add_data(d2)
    a.add_data(d3)
    allrows = np.vstack((d1, d2, d3))
    assert np.allclose(a.scale, np.std(allrows, axis=0))
    a.update_distance()
    assert np.allclose(a.w[1], 1 / np.std(allrows, axis=0))
    obs = rs.randn(1, 3)
    nd = a.nested_distance(d1, obs)
    assert nd.shape == (10, 2)
    assert np.allclose(nd[:, 0], np.sqrt(np.sum((d1 - obs) ** 2, axis=1)))
    assert np.allclose(nd[:, 1], np.sqrt(np.sum(((d1 - obs) / a.scale) ** 2, axis=1)))


def test_ma2_tutorial_known_answer():
    g = np.load(os.path.join(GOLDEN, 'ma2_tutorial.npz'))
    op = O.make_distance('euclidean')
    obs = (g['observed'][:, 0], g['observed'][:, 1])
    d = np.stack([op(g['S1'][b], g['S2'][b], observed=obs) for b in range(g['S1'].shape[0])])
    assert np.array_equal(d[0], g['d0'])
    assert sha(d) == str(g['d_sha'])
    # Rejection with quantile: threshold = the n_samples-th smallest of all distances
    thr = np.sort(d.reshape(-1))[999]
    assert repr(float(thr)) == '0.116859716394976'      # docs/usage/tutorial.rst:396
    assert float(thr) == float(g['threshold'])
    # observed summaries == autocov(y_obs)  (tests/unit/test_elfi_model.py:34-45)
    assert np.array_equal(g['observed'][0], [O.autocov(g['y_obs'])[0], O.autocov(g['y_obs'], 2)[0]])


@pytest.mark.parametrize('tag,doc', [
    ('adaptive_ex1', [[0.06940134, 0.0097677]]),
    ('adaptive_ex2', [[0.01023228, 1.00584519], [0.00921258, 0.99287166], [0.01201937, 0.99365522],
                      [0.02217631, 0.98925365], [0.04355987, 1.00076738], [0.07863284, 0.9971017],
                      [0.13892778, 1.00929049]]),
])
def test_adaptive_trace_replay(tag, doc):
    g = np.load(os.path.join(GOLDEN, tag + '.npz'))
    a = O.AdaptiveDistanceOracle()
    ws = []
    for i, kind in enumerate(g['kinds']):
        if kind == 'add_data':
            a.add_data(g['e%d_data' % i])
        elif kind == 'update_distance':
            a.update_distance()
            assert np.array_equal(a.w[-1], g['e%d_w' % i])
            ws.append(a.w[-1])
        else:
            out = a.nested_distance(g['e%d_u' % i], g['e%d_v' % i])
            assert tuple(g['e%d_shape' % i]) == out.shape
            assert np.array_equal(out[:64], g['e%d_head' % i])
            assert sha(out) == str(g['e%d_sha' % i])
    assert np.allclose(np.array(ws), np.array(doc), rtol=0, atol=5e-9)  # doc prints 8 decimals
    assert np.array_equal(np.array(ws), g['final_w'])


def test_gm_oracle_equals_the_reference_class(golden_dir):
    """oracle/gm_oracle.py vs outputs of the real elfi.methods.utils.GMDistribution."""
    import os
    import gm_oracle as GM
    g = np.load(os.path.join(golden_dir, 'gm_pdf.npz'))
    for k in g['cases']:
        cov = g['cov_%d' % k]
        cov = float(cov) if cov.ndim == 0 else cov
        args = (g['x_%d' % k], g['means_%d' % k])
        assert np.array_equal(GM.pdf(*args, cov=cov, weights=g['w_%d' % k]), g['pdf_%d' % k])
        assert np.array_equal(GM.logpdf(*args, cov=cov, weights=g['w_%d' % k]), g['logpdf_%d' % k])
        assert np.array_equal(GM.pdf(*args, cov=cov), g['pdf_now_%d' % k])
