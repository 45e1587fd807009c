"""Parity of the HIP distance path (through the C ABI) with the oracle and the reference's
golden fixtures.  Runs on an MI355X (`-m gpu`).

Tolerances (stated per SURVEY.md section 8c):
  * euclidean(+w), sqeuclidean(+w), cityblock(+w), chebyshev(+w0), minkowski p in {1,2,inf}:
    BIT-EXACT against SciPy's cdist (same left-to-right accumulation, no FMA).
  * minkowski general p (device pow vs libm pow), seuclidean, mahalanobis (SciPy's
    operation order is not reproducible from outside): relative 1e-14.
  * Welford column statistics (different but fixed summation order): mean relative 1e-12;
    M2 within 1e-13 of the sum of absolute terms (the update formula itself cancels when
    |mean| >> std, for the reference as for us); scale/weights relative 1e-12 on the
    reference's recorded traces.
"""
import hashlib
import os

import numpy as np
import pytest

import distance_oracle as O
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

EXACT = {'euclidean', 'sqeuclidean', 'cityblock', 'chebyshev'}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


def _parse_case(g, case):
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


def _is_exact(metric, kw):
    return metric in EXACT or (metric == 'minkowski' and kw.get('p') in (1.0, 2.0, np.inf))


def _check(got, ref, exact, what):
    assert got.shape == ref.shape, what
    if 'mahalanobis' in what:
        # d' VI d on the matrix cores: the additions come in the MFMA's order; SciPy's own come in its BLAS's
        np.testing.assert_allclose(got, ref, rtol=1e-13, atol=0, err_msg=what)
    elif exact:
        assert np.array_equal(got, ref), '%s: max rel %g' % (
            what, np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)))
    else:
        np.testing.assert_allclose(got, ref, rtol=1e-14, atol=0, err_msg=what)


def test_native_library_is_loaded(hip_ctx):
    import elfi_amd
    info = hip_ctx.device_info()
    assert 'gfx950' in info['name'], info
    assert info['cu_count'] == 256
    with open('/proc/self/maps') as f:
        assert 'libelfihip.so' in f.read()


def test_golden_metrics_rows(hip_ctx):
    """Every metric / keyword form elfi.Distance accepts, vs the reference's own outputs."""
    import elfi_amd
    g = np.load(os.path.join(GOLDEN, 'metrics.npz'))
    for case in g['cases']:
        m, k, metric, kw = _parse_case(g, str(case))
        X, y, ref = g['X_%d' % m], g['y_%d' % m], g['d_%d_%d' % (m, k)]
        got = elfi_amd.cdist_rows(X, y, metric, **kw)
        _check(got, ref, _is_exact(metric, kw), str(case))
        # the operation interface: op(*summaries, observed=...) with one 2-d summary
        op = elfi_amd.HipDiscrepancy(metric, **kw)
        _check(op(X, observed=(y,)), ref, _is_exact(metric, kw), 'op ' + str(case))


def test_golden_metrics_cols(hip_ctx):
    """Same fixtures through the structure-of-arrays path (separate summary columns)."""
    import elfi_amd
    g = np.load(os.path.join(GOLDEN, 'metrics.npz'))
    for case in g['cases']:
        m, k, metric, kw = _parse_case(g, str(case))
        if metric == 'mahalanobis':
            continue
        X, y, ref = g['X_%d' % m], g['y_%d' % m], g['d_%d_%d' % (m, k)]
        cols = [np.ascontiguousarray(X[:, j]) for j in range(m)]
        op = elfi_amd.HipDiscrepancy(metric, **kw)
        got = op(*cols, observed=tuple(y[:, j] for j in range(m)))
        _check(got, ref, _is_exact(metric, kw), 'cols ' + str(case))


def test_ma2_tutorial_known_answer(hip_ctx):
    """docs/usage/tutorial.rst:360-396: threshold 0.116859716394976 from GPU distances."""
    import elfi_amd
    g = np.load(os.path.join(GOLDEN, 'ma2_tutorial.npz'))
    op = elfi_amd.HipDiscrepancy('euclidean')
    obs = (g['observed'][:, 0], g['observed'][:, 1])
    d = np.stack([op(g['S1'][b], g['S2'][b], observed=obs) for b in range(g['S1'].shape[0])])
    assert np.array_equal(d[0], g['d0'])
    assert sha(d) == str(g['d_sha'])
    thr = np.sort(d.reshape(-1))[999]
    assert repr(float(thr)) == '0.116859716394976'
    # and the elfi.Distance(callable) form on the stacked matrix
    fn = elfi_amd.HipDistance('euclidean')
    X = np.column_stack((g['S1'][0], g['S2'][0]))
    assert np.array_equal(fn(X, g['observed']), g['d0'])


@pytest.mark.parametrize('tag', ['adaptive_ex1', 'adaptive_ex2'])
def test_adaptive_trace_replay(hip_ctx, tag):
    """Replay the AdaptiveDistance call trace recorded from the real reference run
    (docs/usage/adaptive_distance.rst examples 1 and 2) through the GPU state machine."""
    import elfi_amd
    g = np.load(os.path.join(GOLDEN, tag + '.npz'))
    a = elfi_amd.AdaptiveDistanceState()
    ref = O.AdaptiveDistanceOracle()
    ws = []
    for i, kind in enumerate(g['kinds']):
        if kind == 'add_data':
            a.add_data(g['e%d_data' % i])
            ref.add_data(g['e%d_data' % i])
            np.testing.assert_allclose(a.state['scale'], ref.scale, rtol=1e-12)
            assert a.state['store'][0] == ref.store[0]
        elif kind == 'update_distance':
            a.update_distance()
            ref.update_distance()
            np.testing.assert_allclose(a.state['w'][-1], g['e%d_w' % i], rtol=1e-12)
            ws.append(a.state['w'][-1])
            # keep the replay on the reference's exact weights so later distance checks are bit-exact
            a.state['w'][-1] = g['e%d_w' % i].copy()
        else:
            out = a.nested_distance(g['e%d_u' % i], g['e%d_v' % i])
            assert tuple(g['e%d_shape' % i]) == out.shape
            assert np.array_equal(out[:64], g['e%d_head' % i])
            assert sha(out) == str(g['e%d_sha' % i])
    np.testing.assert_allclose(np.array(ws), g['final_w'], rtol=1e-12)


@pytest.mark.parametrize('n', [0, 1, 2, 63, 64, 65, 255, 257, 1000, 4099])
@pytest.mark.parametrize('m', [1, 2, 5, 32, 64, 100])
def test_shapes_vs_oracle(hip_ctx, n, m):
    """Empty, ragged and tile-boundary sizes; weighted and unweighted."""
    import elfi_amd
    rs = np.random.RandomState(1000 * m + n)
    X, y, w = rs.randn(n, m), rs.randn(1, m), rs.uniform(0.1, 3, m)
    for metric, kw in [('euclidean', {}), ('euclidean', dict(w=w)), ('cityblock', dict(w=w)),
                       ('chebyshev', {})]:
        got = elfi_amd.cdist_rows(X, y, metric, **kw)
        assert np.array_equal(got, O.cdist_rows(X, y, metric, **kw)), (metric, n, m)
        if n:
            cols = [np.ascontiguousarray(X[:, j]) for j in range(m)]
            got = elfi_amd.cdist_cols(cols, y, metric, **kw)
            assert np.array_equal(got, O.cdist_rows(X, y, metric, **kw)), ('cols', metric, n, m)


@pytest.mark.parametrize('n', [1, 15, 64, 65, 1000, 4099])
@pytest.mark.parametrize('m', [1, 2, 3, 5, 16, 17, 32, 33, 64, 65, 100])
def test_mahalanobis_shapes_vs_oracle(hip_ctx, n, m):
    """sqrt(d' VI d): the matrix-core kernel (m <= 64, every padding case of its 4-deep / 16-wide tiles) and the
    lane-per-row kernel beyond, against SciPy's cdist through the oracle.  VI is a proper inverse covariance with
    negative off-diagonal entries, so the quadratic form's terms cancel."""
    import elfi_amd
    rs = np.random.RandomState(77 * m + n)
    Z = rs.randn(4 * m + 5, m) @ rs.randn(m, m)
    VI = np.linalg.inv(np.cov(Z.T).reshape(m, m) + 0.1 * np.eye(m))
    VI = 0.5 * (VI + VI.T)
    X, y = rs.randn(n, m) * 2, rs.randn(1, m)
    big = np.zeros((n, m + 3))
    big[:, 1:m + 1] = X                                     # a row pitch that is not the width
    ref = O.cdist_rows(X, y, 'mahalanobis', VI=VI)
    np.testing.assert_allclose(elfi_amd.cdist_rows(X, y, 'mahalanobis', VI=VI), ref, rtol=1e-13, atol=0)
    np.testing.assert_allclose(elfi_amd.cdist_rows(big[:, 1:m + 1], y, 'mahalanobis', VI=VI), ref, rtol=1e-13, atol=0)


def test_strided_rows_and_dtypes(hip_ctx):
    import elfi_amd
    rs = np.random.RandomState(5)
    big = rs.randn(777, 40)
    X = big[:, 3:20]                       # row pitch 40, odd width 17, unaligned start
    y = rs.randn(1, 17)
    assert np.array_equal(elfi_amd.cdist_rows(X, y), O.cdist_rows(X, y, 'euclidean'))
    X32 = rs.randn(300, 6).astype(np.float32)   # cdist casts to float64
    assert np.array_equal(elfi_amd.cdist_rows(X32, y[:, :6]), O.cdist_rows(X32, y[:, :6], 'euclidean'))
    Xi = rs.randint(-5, 5, (100, 3))
    assert np.array_equal(elfi_amd.cdist_rows(Xi, y[:, :3]), O.cdist_rows(Xi, y[:, :3], 'euclidean'))


def test_wide_rows_fallback(hip_ctx):
    import elfi_amd
    rs = np.random.RandomState(6)
    X, y = rs.randn(257, 700), rs.randn(1, 700)
    np.testing.assert_allclose(elfi_amd.cdist_rows(X, y), O.cdist_rows(X, y, 'euclidean'), rtol=1e-14)
    np.testing.assert_allclose(elfi_amd.cdist_rows(X, y, 'chebyshev'), O.cdist_rows(X, y, 'chebyshev'),
                               rtol=0)


def test_error_behaviour(hip_ctx):
    """Same exception classes as the reference path (ValueError from cdist / utils.py:42-49)."""
    import elfi_amd
    X, y = np.zeros((4, 3)), np.zeros((1, 2))
    with pytest.raises(ValueError):
        elfi_amd.cdist_rows(X, y)
    with pytest.raises(ValueError):
        elfi_amd.cdist_rows(np.zeros((2, 2, 2)), np.zeros((1, 2)))
    with pytest.raises(ValueError):
        elfi_amd.HipDistance('seuclidean')
    with pytest.raises(ValueError):
        elfi_amd.HipDistance('nonsense')
    with pytest.raises(ValueError):
        elfi_amd.cdist_rows(np.zeros((4, 2)), y, 'minkowski', p=-1)
    op = elfi_amd.HipDiscrepancy('euclidean')
    with pytest.raises(ValueError):
        op(np.zeros((4, 2, 2)), observed=(np.zeros((1, 2)),))


def test_full_size_properties(hip_ctx):
    """BASELINE config 2 size (10^6 x 32): size-independent properties, no CPU pass needed
    for the bulk -- a seeded 2^16-row sample is checked bit-exactly against cdist."""
    import elfi_amd
    rs = np.random.RandomState(0)
    n, m = 10 ** 6, 32
    X = rs.randn(n, m)
    y = np.random.RandomState(1).randn(1, m)
    w = 1 / np.random.RandomState(2).uniform(.5, 2, m) ** 2
    d = elfi_amd.cdist_rows(X, y)
    dw = elfi_amd.cdist_rows(X, y, w=w)
    d2 = elfi_amd.cdist_rows(X, y, 'sqeuclidean')
    idx = rs.choice(n, 1 << 16, replace=False)
    assert np.array_equal(d[idx], O.cdist_rows(X[idx], y, 'euclidean'))
    assert np.array_equal(dw[idx], O.cdist_rows(X[idx], y, 'euclidean', w=w))
    assert np.array_equal(np.sqrt(d2), d)                      # euclid == sqrt(sqeuclid), exactly
    # permutation equivariance: distances of permuted rows are the permuted distances
    perm = rs.permutation(n)
    assert np.array_equal(elfi_amd.cdist_rows(X[perm], y), d[perm])
    # scaling: dist(c*X, c*y) == c*dist for a power of two (exact in binary floating point)
    assert np.array_equal(elfi_amd.cdist_rows(4.0 * X, 4.0 * y), 4.0 * d)
    # a row equal to the observation has distance exactly 0; weights of zero drop columns
    X[123] = y[0]
    assert elfi_amd.cdist_rows(X[100:200], y)[23] == 0.0
    # SoA path agrees with AoS path bit for bit
    cols = [np.ascontiguousarray(X[:, j]) for j in range(m)]
    assert np.array_equal(elfi_amd.cdist_cols(cols, y), elfi_amd.cdist_rows(X, y))


def test_welford_reference_unit_test(hip_ctx):
    """tests/unit/test_elfi_model.py:186-253 restated on the GPU state machine."""
    import elfi_amd
    rs = np.random.RandomState(1)
    a = elfi_amd.AdaptiveDistanceState()
    d1, d2, d3 = rs.randn(10, 3) * [1, 10, 100], rs.randn(10, 3) * [1, 10, 100], rs.randn(1, 3)
    for d in (d1, d2, d3):
        a.add_data(d)
    allrows = np.vstack((d1, d2, d3))
    assert np.allclose(a.state['scale'], np.std(allrows, axis=0))
    a.update_distance()
    assert np.allclose(a.state['w'][1], 1 / np.std(allrows, axis=0))
    obs = rs.randn(1, 3)
    nd = a.nested_distance(d1, obs)
    assert nd.shape == (10, 2)
    assert np.allclose(nd[:, 0], np.sqrt(np.sum((d1 - obs) ** 2, axis=1)))
    assert np.allclose(nd[:, 1], np.sqrt(np.sum(((d1 - obs) / a.state['scale']) ** 2, axis=1)))


def _fold_and_check(batches, m):
    """Fold the batches into the running (count, mean, M2) on the GPU and in the oracle, checking after every batch
    with tolerances that are fixed IN ADVANCE by the data (no observed error is ever carried forward):

      mean   mean_new = mean_old + sum(x - mean_old) / N: the two sides sum in different (fixed) orders, allowed
             2e-13 of the mean absolute term + 8 ulp; a difference left by earlier batches enters the new mean scaled
             by N_old / N, so the allowance is the running sum  A <- A N_old / N + allowance_b  (it does not grow).
      M2     given OUR mean_new, our increment equals sum d1 d2 evaluated in extended precision to 1e-13 of the sum of
             absolute terms (formula check, no reference involved).  Against the reference: the increment has
             derivative -sum(d1) with respect to mean_new, so the means' allowance A is amplified by |sum d1| (up to
             N |mean| for the first batch, where mean_old = 0: the reference's formula is that ill-conditioned itself);
             increments add, so the allowance is the sum of the per-batch allowances -- linear in the number of
             batches, as the rounding of any running sum is.
    """
    import elfi_amd
    ref = O.AdaptiveDistanceOracle()
    cnt, mean, M2 = 0, np.zeros(m), np.zeros(m)
    M2_old = np.zeros(m)
    allow_mean, allow_m2 = np.zeros(m), np.zeros(m)
    for b, X in enumerate(batches):
        mean_old, n_old = mean.copy(), cnt          # OUR previous mean: what the kernel subtracts in d1
        ref.add_data(X)
        cnt, mean, M2 = elfi_amd.welford_update(X, cnt, mean, M2)
        assert cnt == ref.store[0]
        allow_mean = allow_mean * (n_old / cnt) + 2e-13 * np.mean(np.abs(X - mean_old), axis=0) \
            + 8 * np.spacing(np.abs(ref.store[1]))
        err = np.abs(mean - ref.store[1])
        worst = np.argmax(err / allow_mean)
        assert np.all(err <= allow_mean), ('mean', b, worst, mean[worst], ref.store[1][worst], allow_mean[worst])
        Xl = X.astype(np.longdouble)
        d1 = Xl - mean_old.astype(np.longdouble)
        terms = d1 * (Xl - mean.astype(np.longdouble))
        exact_inc = np.sum(terms, axis=0)
        scale = np.sum(np.abs(terms), axis=0).astype(float)
        inc = M2.astype(np.longdouble) - M2_old.astype(np.longdouble)
        e1 = np.abs((inc - exact_inc).astype(float))
        t1 = 1e-13 * scale + 4 * np.spacing(np.abs(M2))
        w1 = np.argmax(e1 / t1)
        assert np.all(e1 <= t1), ('formula', b, w1, e1[w1], t1[w1], scale[w1], M2[w1])
        allow_m2 = allow_m2 + 2e-13 * (scale + np.abs(M2)) + 2 * np.abs(np.sum(d1, axis=0).astype(float)) * allow_mean
        e2 = np.abs(M2 - ref.store[2])
        w2 = np.argmax(e2 / allow_m2)
        assert np.all(e2 <= allow_m2), ('vs reference', b, w2, e2[w2], allow_m2[w2])
        M2_old = M2.copy()
    return cnt, mean, M2


@pytest.mark.parametrize('n,m', [(1, 1), (7, 3), (10000, 2), (100003, 64), (5000, 300)])
def test_welford_vs_oracle(hip_ctx, n, m):
    rs = np.random.RandomState(n + m)
    _fold_and_check([rs.randn(n, m) * rs.uniform(0.1, 100, m) + rs.uniform(-1000, 1000, m) for _ in range(3)], m)


@pytest.mark.parametrize('nbatch,n,m', [(40, 700, 9), (200, 97, 3), (1000, 33, 2)])
def test_welford_many_batches(hip_ctx, nbatch, n, m):
    """elfi_model.py:1104-1125 over many batches (an SMC round folds in every batch of the round), same a-priori
    tolerances; at the end the scale is the population standard deviation of the union (test_elfi_model.py:197-218)."""
    rs = np.random.RandomState(nbatch + n)
    scale, shift = rs.uniform(0.1, 100, m), rs.uniform(-10, 10, m)
    batches = [rs.randn(n + b % 7, m) * scale + shift for b in range(nbatch)]
    cnt, mean, M2 = _fold_and_check(batches, m)
    np.testing.assert_allclose(np.sqrt(M2 / cnt), np.std(np.vstack(batches), axis=0), rtol=1e-11)


def test_welford_device_merge_equals_the_host_merge(hip_ctx):
    """elfihip_welford_merge_dev (multi-GPU adaptive distance: merge of the all-gathered rank states on the device)
    == elfi_amd.sharding.merge_welford bit for bit, weights N / M2 included."""
    import torch
    from elfi_amd import sharding as S
    rs = np.random.RandomState(4)
    world, m = 8, 64
    states = []
    for r in range(world):
        nb = 0 if r == 3 else int(rs.randint(1000, 200000))      # one rank without rows
        states.append(np.concatenate([[float(nb)], rs.randn(m) * 10, rs.uniform(1, 1e6, m) * (nb > 0)]))
    st = torch.from_numpy(np.vstack(states)).cuda()
    merged = torch.empty(1 + 2 * m, dtype=torch.float64, device='cuda')
    w2 = torch.empty(m, dtype=torch.float64, device='cuda')
    hip_ctx.call("elfihip_welford_merge_dev", st.data_ptr(), world, m, merged.data_ptr(), w2.data_ptr())
    hip_ctx.synchronize()
    N, mean, M2 = S.merge_welford([(v[0], v[1:1 + m], v[1 + m:]) for v in states])
    got = merged.cpu().numpy()
    assert got[0] == N and np.array_equal(got[1:1 + m], mean) and np.array_equal(got[1 + m:], M2)
    assert np.array_equal(w2.cpu().numpy(), 1.0 / (M2 / N))


def test_determinism(hip_ctx):
    import elfi_amd
    rs = np.random.RandomState(9)
    X, y = rs.randn(200000, 33), rs.randn(1, 33)
    a, b = elfi_amd.cdist_rows(X, y, 'minkowski', p=3), elfi_amd.cdist_rows(X, y, 'minkowski', p=3)
    assert np.array_equal(a, b)
    c1 = elfi_amd.welford_update(X, 0, np.zeros(33), np.zeros(33))
    c2 = elfi_amd.welford_update(X, 0, np.zeros(33), np.zeros(33))
    assert np.array_equal(c1[1], c2[1]) and np.array_equal(c1[2], c2[2])


def test_sharding_module_with_the_hip_backend(hip_ctx):
    """elfi_amd.sharding at world size 1 with the product backend == the oracle's AdaptiveDistance."""
    from elfi_amd import sharding as S
    rs = np.random.RandomState(3)
    data = [rs.randn(500 + 13 * b, 6) * np.linspace(0.5, 30, 6) + b for b in range(4)]
    y = rs.randn(1, 6)
    ad = S.ShardedAdaptiveDistance(6)
    ref = O.AdaptiveDistanceOracle()
    for rnd in range(2):
        for X in data:
            ad.add_data(X)
            ref.add_data(X)
        np.testing.assert_allclose(ad.sync_scale(), ref.scale, rtol=1e-12)
        ad.update_distance()
        ref.update_distance()
    got = np.vstack([ad.nested_distance(X, y) for X in data])
    W = ad.weight_matrix()
    exp = np.vstack([np.column_stack([O.cdist_rows(X, y, 'euclidean', w=w) for w in W]) for X in data])
    assert np.array_equal(got, exp)
    assert S.gather_rows(got)[0] is not None


def test_config4_shape_adaptive_round(hip_ctx):
    """BASELINE configs[3] per-GPU shape: 1.25e6 x 64 summaries, one adaptive-distance round with
    K = 3 nested weight vectors.  Scales vs np.std of the shard; every distance column bit-exact
    against cdist on a 2^15-row sample and through exact properties on the full shard."""
    import elfi_amd
    n, m = 1250000, 64
    X = np.random.RandomState(100).randn(n, m) * np.linspace(0.5, 20, m)
    y = np.random.RandomState(1).randn(1, m)
    ad = elfi_amd.AdaptiveDistanceState()
    ad.add_data(X)
    np.testing.assert_allclose(ad.state['scale'], np.std(X, axis=0), rtol=1e-11)
    ad.update_distance()
    ad.add_data(X[: n // 2])
    ad.update_distance()
    d = ad.nested_distance(X, y)
    assert d.shape == (n, 3)
    W = ad.weight_matrix(m)
    idx = np.random.RandomState(2).choice(n, 1 << 15, replace=False)
    for k in range(3):
        assert np.array_equal(d[idx, k], O.cdist_rows(X[idx], y, 'euclidean', w=W[k])), k
    assert np.array_equal(d[:, 0], elfi_amd.cdist_rows(X, y))              # unweighted column == plain euclidean
    assert np.array_equal(ad.nested_distance(X[::-1].copy(), y), d[::-1])  # row-order equivariance


@pytest.mark.parametrize('m', [16, 32, 64])
def test_lds_dma_row_stream_equals_the_register_pipeline_and_cdist(hip_ctx, m):
    """The LDS-DMA form of the row stream (csrc/distance.hip: dist_rows_dma_kernel, the default for 16 / 32 / 64 summaries)
    against SciPy's cdist and against the register-staged form it replaces, bit for bit: ragged and tiny n (slots of 64 / 32
    rows, rings of four), weights, every light metric, a row pitch that is not the width, and the sampler state fed by the
    same pass (the fused selection sees the same distances)."""
    import elfi_amd
    rs = np.random.RandomState(500 + m)
    y, w = rs.randn(1, m), rs.uniform(0.1, 3, m)
    w0 = w.copy()
    w0[::3] = 0.0
    try:
        for n in (1, 31, 32, 33, 63, 64, 65, 127, 129, 1000, 4099, 65536 + 17, 300007):
            X = rs.randn(n, m) * 1.5
            big = np.zeros((n, m + 6))
            big[:, 2:m + 2] = X                       # pitch m + 6 (even, rows stay 16-byte aligned), offset 2
            cases = [('euclidean', {}), ('euclidean', dict(w=w)), ('sqeuclidean', dict(w=w)), ('cityblock', {}),
                     ('cityblock', dict(w=w)), ('chebyshev', {}), ('chebyshev', dict(w=w0)), ('minkowski', dict(p=1.0)),
                     ('minkowski', dict(p=np.inf))]
            for metric, kw in cases:
                ref = O.cdist_rows(X, y, metric, **kw)
                hip_ctx.call('elfihip_dist_set_form', 0)
                dma = elfi_amd.cdist_rows(X, y, metric, **kw)
                dma_pitched = elfi_amd.cdist_rows(big[:, 2:m + 2], y, metric, **kw)
                hip_ctx.call('elfihip_dist_set_form', 1)
                reg = elfi_amd.cdist_rows(X, y, metric, **kw)
                assert np.array_equal(dma, ref), (metric, sorted(kw), n, m)
                assert np.array_equal(dma_pitched, ref), ('pitched', metric, n, m)
                assert np.array_equal(reg, ref), ('register form', metric, n, m)
            ref = O.cdist_rows(X, y, 'seuclidean', V=w)
            hip_ctx.call('elfihip_dist_set_form', 0)
            np.testing.assert_allclose(elfi_amd.cdist_rows(X, y, 'seuclidean', V=w), ref, rtol=1e-14, atol=0)
        # the sampler's running best-k through the same pass
        hip_ctx.call('elfihip_dist_set_form', 0)
        k = 300
        rb = elfi_amd.RunningBest(k, metric='euclidean', w=w)
        seen = []
        for n in (100, 150, 4000, 250000, 9, 70001):
            X = rs.randn(n, m)
            d = rb.push(X, y)
            assert np.array_equal(d, O.cdist_rows(X, y, 'euclidean', w=w))
            seen.append(d)
            vals, rows = rb.result()
            allv = np.concatenate(seen)
            order = np.lexsort((np.arange(len(allv)), allv))[:k]
            assert np.array_equal(vals, allv[order]) and np.array_equal(rows, order), n
    finally:
        hip_ctx.call('elfihip_dist_set_form', 0)


@pytest.mark.parametrize('m', [2, 4])
def test_narrow_row_kernels_equal_the_tile_kernels_and_cdist(hip_ctx, m):
    """Round 6: rows of 2 or 4 summaries (configs[0]'s own shape) are owned by lanes -- U 16-/32-byte loads per lane, no LDS
    (csrc/distance.hip: dist_rows_narrow_kernel, dist_rows_mahalanobis_narrow_kernel, dist_multiw_narrow_kernel) -- against
    SciPy's cdist and against the tile kernels they replace (form 1), bit for bit: every metric, weights, ragged and tiny n
    on both sides of the 1024-row granule, a row pitch that is not the width, the K-weight form, and the sampler state fed
    by the same pass."""
    import elfi_amd
    rs = np.random.RandomState(900 + m)
    y, w = rs.randn(1, m), rs.uniform(0.1, 3, m)
    w0 = w.copy()
    w0[0] = 0.0
    Z = rs.randn(40, m) @ rs.randn(m, m)
    VI = np.linalg.inv(np.cov(Z.T).reshape(m, m) + 0.1 * np.eye(m))
    VI = 0.5 * (VI + VI.T)
    W = np.vstack([np.ones(m), rs.uniform(0.2, 2, (4, m))])
    try:
        for n in (1, 2, 255, 256, 257, 1023, 1024, 1025, 4099, 65536 + 17, 1000003):
            X = rs.randn(n, m) * 1.5
            big = np.zeros((n, m + 6))
            big[:, 2:m + 2] = X                       # pitch m + 6 (even, rows stay 16-byte aligned), offset 2
            cases = [('euclidean', {}), ('euclidean', dict(w=w)), ('sqeuclidean', dict(w=w)), ('cityblock', {}),
                     ('cityblock', dict(w=w)), ('chebyshev', {}), ('chebyshev', dict(w=w0)), ('minkowski', dict(p=1.0)),
                     ('minkowski', dict(p=np.inf)), ('minkowski', dict(p=3.0)), ('minkowski', dict(p=2.5, w=w)),
                     ('seuclidean', dict(V=w)), ('mahalanobis', dict(VI=VI))]
            for metric, kw in cases:
                ref = O.cdist_rows(X, y, metric, **kw)
                hip_ctx.call('elfihip_dist_set_form', 0)
                nar = elfi_amd.cdist_rows(X, y, metric, **kw)
                nar_pitched = elfi_amd.cdist_rows(big[:, 2:m + 2], y, metric, **kw)
                hip_ctx.call('elfihip_dist_set_form', 1)
                til = elfi_amd.cdist_rows(X, y, metric, **kw)
                what = '%s %s n=%d m=%d' % (metric, sorted(kw), n, m)
                assert np.array_equal(nar, til), 'narrow form differs from the tile kernel: ' + what
                assert np.array_equal(nar_pitched, nar), 'pitched: ' + what
                _check(nar, ref, _is_exact(metric, kw), what)
            for K in (1, 3, 5):
                hip_ctx.call('elfihip_dist_set_form', 0)
                nar = elfi_amd.nested_weighted_euclidean(X, y, W[:K])
                hip_ctx.call('elfihip_dist_set_form', 1)
                til = elfi_amd.nested_weighted_euclidean(X, y, W[:K])
                ref = np.column_stack([O.cdist_rows(X, y, 'euclidean', w=W[k]) for k in range(K)])
                assert np.array_equal(nar, til) and np.array_equal(nar, ref), ('K-weight form', K, n, m)
        # the sampler's running best-k through the same pass
        hip_ctx.call('elfihip_dist_set_form', 0)
        k = 300
        rb = elfi_amd.RunningBest(k, metric='euclidean', w=w)
        seen = []
        for n in (100, 150, 4000, 250000, 9, 70001):
            X = rs.randn(n, m)
            d = rb.push(X, y)
            assert np.array_equal(d, O.cdist_rows(X, y, 'euclidean', w=w))
            seen.append(d)
            vals, rows = rb.result()
            allv = np.concatenate(seen)
            order = np.lexsort((np.arange(len(allv)), allv))[:k]
            assert np.array_equal(vals, allv[order]) and np.array_equal(rows, order), n
    finally:
        hip_ctx.call('elfihip_dist_set_form', 0)
