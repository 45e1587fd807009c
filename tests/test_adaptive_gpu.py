"""GPU: one AdaptiveDistance batch in one read (csrc/adaptive.hip, elfihip_adaptive_push[_dev]) vs the oracle.

  * nested distances: bit-identical to scipy's cdist per weight vector (oracle/distance_oracle.py:cdist_rows), as
    elfihip_dist_multiw;
  * column statistics: against the two-pass values in extended precision with tolerances fixed in advance, and against
    the reference's own batched Welford update (elfi/model/elfi_model.py:1104-1125, oracle AdaptiveDistanceOracle);
  * selection: what Rejection._merge_batch keeps (elfi/methods/inference/samplers.py:209-237) -- per-column acceptance,
    ranking by the last column, ties to the earlier row -- exact against NumPy, including the provisional threshold of
    a large first batch and the inputs that defeat it (sorted rows).
"""
import ctypes as C

import numpy as np
import pytest

import distance_oracle as O

pytestmark = pytest.mark.gpu


def _nested_ref(X, y, W):
    return np.column_stack([O.cdist_rows(X, y, 'euclidean', w=w) for w in W])


def _exact_stats(X):
    Xl = X.astype(np.longdouble)
    mean = Xl.mean(axis=0)
    return len(X), mean, ((Xl - mean) ** 2).sum(axis=0)


def _check_stats(store, X, fused=True):
    """(count, mean, M2) of the device against the exact two-pass values.  A-priori bounds: every Chan update rounds
    the running mean once (half an ulp of |mean|; chains are tens of updates long) and the mean of a tile carries
    1e-16 x the spread; M2 is a sum of non-negative terms (relative error ~ depth x eps) plus n (error of the mean)^2.
    fused=False: the shapes that take welford.hip, which follows the reference's own formula (first batch: sum of
    x (x - mean) about a zero mean) -- its terms, not M2, set the scale of the rounding."""
    n, mean, M2 = _exact_stats(X)
    cnt, dm, dq = store
    assert cnt == n
    spread = np.mean(np.abs(X - mean.astype(float)), axis=0)
    tol_mean = 1e-13 * spread + 64 * np.spacing(np.abs(mean.astype(float)))
    err = np.abs((dm.astype(np.longdouble) - mean).astype(float))
    assert np.all(err <= tol_mean), (err.max(), tol_mean[np.argmax(err / tol_mean)])
    tol_m2 = 2e-13 * M2.astype(float) + 4 * n * tol_mean ** 2 + 4 * np.spacing(M2.astype(float))
    if not fused:
        tol_m2 = tol_m2 + 2e-13 * np.sum(np.abs(X * (X - mean.astype(float))), axis=0)
    e2 = np.abs((dq.astype(np.longdouble) - M2).astype(float))
    assert np.all(e2 <= tol_m2), (e2.max(), tol_m2[np.argmax(e2 / tol_m2)])


@pytest.mark.parametrize('n,m,K', [(1, 2, 1), (7, 2, 2), (1000, 2, 3), (100003, 64, 3), (4097, 32, 5), (50000, 128, 2),
                                  (20000, 10, 9), (3000, 3, 2), (5000, 130, 2), (2048, 64, 40),
                                  # narrow rows (round 6: adaptive_narrow_kernel), both sides of its 1024-row granule; K = 9
                                  # is beyond its eight weight vectors
                                  (1023, 2, 3), (1024, 2, 8), (1025, 4, 1), (4099, 4, 8), (300007, 2, 3), (250001, 4, 5),
                                  (5000, 2, 9)])
def test_distances_and_statistics_vs_oracle(hip_ctx, n, m, K):
    """Fused shapes (even m <= 128) and the shapes that take the separate passes (odd m, m > 128, K m beyond LDS)."""
    import elfi_amd
    rs = np.random.RandomState(n + m + K)
    X = rs.randn(n, m) * rs.uniform(0.1, 100, m) + rs.uniform(-1000, 1000, m)
    y = rs.randn(1, m)
    W = np.vstack([np.ones(m)] + [rs.uniform(0.01, 4, m) for _ in range(K - 1)])
    d, store = elfi_amd.adaptive_batch(X, y, W, store=(0, 0.0, 0.0))
    assert d.shape == (n, K)
    assert np.array_equal(d, _nested_ref(X, y, W))
    assert np.array_equal(d, elfi_amd.nested_weighted_euclidean(X, y, W))
    if n > 1:
        _check_stats(store, X, fused=(m % 2 == 0 and m <= 128 and not (m == 2 and K > 8)))
    else:
        assert store[0] == 1 and np.array_equal(store[1], X[0]) and np.all(store[2] == 0)
    # no statistics / no distances requested
    d2, none = elfi_amd.adaptive_batch(X, y, W)
    assert none is None and np.array_equal(d2, d)
    none_d, store2 = elfi_amd.adaptive_batch(X, y, W, store=(0, 0.0, 0.0), distances=False)
    assert none_d is None and store2[0] == n and np.array_equal(store2[1], store[1]) and np.array_equal(store2[2], store[2])


def test_running_store_over_batches_equals_the_reference_update(hip_ctx):
    """add_data over the batches of a round (elfi_model.py:1104-1125): the store after every batch against the exact
    statistics of everything folded in so far, the final scale against the reference's update and np.std
    (tests/unit/test_elfi_model.py:197-218)."""
    import elfi_amd
    rs = np.random.RandomState(5)
    m = 64
    scale, shift = rs.uniform(0.1, 100, m), rs.uniform(-10, 10, m)
    y, W = rs.randn(1, m), np.ones((1, m))
    ref = O.AdaptiveDistanceOracle()
    store, seen = (0, 0.0, 0.0), []
    for b in range(12):
        X = rs.randn(3000 + 517 * (b % 5), m) * scale + shift
        seen.append(X)
        ref.add_data(X)
        _, store = elfi_amd.adaptive_batch(X, y, W, store=store, distances=False)
        _check_stats(store, np.vstack(seen))
    got = np.sqrt(store[2] / store[0])
    np.testing.assert_allclose(got, ref.scale, rtol=1e-12)
    np.testing.assert_allclose(got, np.std(np.vstack(seen), axis=0), rtol=1e-12)
    # bit-reproducible
    _, a = elfi_amd.adaptive_batch(seen[0], y, W, store=(0, 0.0, 0.0), distances=False)
    _, b = elfi_amd.adaptive_batch(seen[0], y, W, store=(0, 0.0, 0.0), distances=False)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_ill_conditioned_columns(hip_ctx):
    """mean >> spread (the reference's first-batch formula loses digits there; the tile-local two-pass form does not),
    a constant column, a column with one outlier."""
    import elfi_amd
    rs = np.random.RandomState(11)
    n, m = 40000, 8
    X = rs.randn(n, m)
    X[:, 0] = 1e8 + 1e-3 * rs.randn(n)
    X[:, 1] = 3.25
    X[:, 2] = 0.0
    X[17, 2] = 1e9
    X[:, 3] *= 1e-150
    _, store = elfi_amd.adaptive_batch(X, np.zeros((1, m)), np.ones((1, m)), store=(0, 0.0, 0.0), distances=False)
    _check_stats(store, X)
    assert store[2][1] == 0.0 and store[1][1] == 3.25


def _best_ref(D, k, acc=None, base=0):
    D = D.reshape(len(D), -1)
    ok = ~np.isnan(D[:, -1])
    if acc is not None:
        ok &= np.all(D <= np.asarray(acc), axis=1)
    rows = np.nonzero(ok)[0]
    order = np.lexsort((rows, D[rows, -1]))[:k]
    return D[rows[order], -1], rows[order] + base, int(ok.sum())


@pytest.mark.parametrize('n,m,K,k', [(5000, 64, 3, 100), (200000, 32, 2, 1000), (70000, 2, 3, 2048), (90000, 64, 2, 5000),
                                    (3000, 5, 2, 64), (120000, 4, 4, 1000), (1500, 2, 1, 300)])
def test_selection_state_over_batches(hip_ctx, n, m, K, k):
    """Several batches through one state: distances returned, statistics accumulated, running best-k exact."""
    import elfi_amd
    rs = np.random.RandomState(n + k)
    y = rs.randn(1, m)
    W = np.vstack([np.ones(m)] + [rs.uniform(0.1, 2, m) for _ in range(K - 1)])
    rb = elfi_amd.RunningBest(k)
    store, all_d, all_x = (0, 0.0, 0.0), [], []
    for b in range(5):
        X = rs.randn(n - 13 * b, m) * rs.uniform(0.5, 2, m)
        d, store = elfi_amd.adaptive_batch(X, y, W, store=store, state=rb)
        assert np.array_equal(d, _nested_ref(X, y, W))
        all_d.append(d)
        all_x.append(X)
        vals, rows = rb.result()
        rv, rr, _ = _best_ref(np.vstack(all_d), k)
        assert np.array_equal(vals, rv) and np.array_equal(rows, rr), b
    _check_stats(store, np.vstack(all_x))


@pytest.mark.parametrize('m', [64, 2, 4])
def test_per_column_acceptance(hip_ctx, m):
    """AdaptiveDistanceSMC's threshold list [inf, t1, t2] (samplers.py:657-660), column by column (:222-223); counts."""
    import elfi_amd
    rs = np.random.RandomState(2)
    n, K, k = 60000, 3, 500
    y = rs.randn(1, m)
    W = np.vstack([np.ones(m), rs.uniform(0.1, 2, m), rs.uniform(0.1, 2, m)])
    batches = [rs.randn(n, m) for _ in range(4)]
    D = [_nested_ref(X, y, W) for X in batches]
    acc = [np.inf, np.quantile(D[0][:, 1], 0.3), np.quantile(D[0][:, 2], 0.2)]
    rb = elfi_amd.RunningBest(k, accept=acc)
    total = 0
    for b, X in enumerate(batches):
        d, _ = elfi_amd.adaptive_batch(X, y, W, state=rb, row_base=b * n)
        assert np.array_equal(d, D[b])
        rv, rr, cnt = _best_ref(np.vstack(D[:b + 1]), k, acc)
        kth, last, tot = rb.meta()
        total += _best_ref(D[b], k, acc)[2]
        assert last == _best_ref(D[b], k, acc)[2] and tot == total
        vals, rows = rb.result()
        assert np.array_equal(vals, rv) and np.array_equal(rows, rr)
    # the same state fed with distances that exist already (what HipRejection._merge_batch pushes)
    rb2 = elfi_amd.RunningBest(k, accept=acc)
    for b in range(4):
        rb2.push_distances(D[b], row_base=b * n)
    v2, r2 = rb2.result()
    assert np.array_equal(v2, rv) and np.array_equal(r2, rr)
    # a strict list: nothing passes
    rb3 = elfi_amd.RunningBest(k, accept=[np.inf, 0.0, 0.0])
    elfi_amd.adaptive_batch(batches[0], y, W, state=rb3)
    assert len(rb3.result()[0]) == 0 and rb3.meta()[2] == 0


@pytest.mark.parametrize('order', ['random', 'ascending', 'descending', 'constant'])
@pytest.mark.parametrize('k', [1000, 3000])
def test_large_first_batch_provisional_threshold(hip_ctx, order, k):
    """n >= 2^20 rows into an empty state: the threshold taken from a prefix of the batch and its verification
    (csrc/reject.hip).  Sorted rows defeat the prefix (ascending: too few candidates; descending: every row qualifies;
    constant: all tie) -- the result stays exact, ties to the earlier row."""
    import torch
    n, m, K = (1 << 20) + 4099, 16, 2
    g = torch.Generator(device='cuda')
    g.manual_seed(k)
    X = torch.randn(n, m, dtype=torch.float64, device='cuda', generator=g)
    y = torch.zeros(1, m, dtype=torch.float64, device='cuda')
    W = torch.ones(K, m, dtype=torch.float64, device='cuda')
    W[1] = 0.5
    if order == 'constant':
        X[:] = X[0]
    elif order != 'random':
        key = torch.argsort((X * X).sum(1), descending=(order == 'descending'))
        X = X[key].contiguous()
    out = torch.empty(n, K, dtype=torch.float64, device='cuda')
    wel = torch.zeros(1 + 2 * m, dtype=torch.float64, device='cuda')
    import elfi_amd
    for keep in (True, False):          # with and without a destination for the distances
        rb = elfi_amd.RunningBest(k)
        wel.zero_()
        torch.cuda.synchronize()        # (torch's stream and the context's own stream are not ordered against each other)
        hip_ctx.call("elfihip_adaptive_push_dev", rb.h, X.data_ptr(), n, m, m, y.data_ptr(), W.data_ptr(), K,
                     out.data_ptr() if keep else None, wel.data_ptr(), 7 * n)
        vals, rows = rb.result()
        if keep:
            D = out.cpu().numpy()
            assert np.array_equal(D[::997], _nested_ref(X[::997].cpu().numpy(), y.cpu().numpy(), W.cpu().numpy()))
        rv, rr, _ = _best_ref(D, k, base=7 * n)
        assert np.array_equal(vals, rv) and np.array_equal(rows, rr), (order, keep)
        st = wel.cpu().numpy()
        if order != 'constant':
            _check_stats((int(st[0]), st[1:1 + m], st[1 + m:]), X.cpu().numpy())
    # a second batch against the now-full state
    X2 = torch.randn(300000, m, dtype=torch.float64, device='cuda', generator=g) * 0.9
    out2 = torch.empty(300000, K, dtype=torch.float64, device='cuda')
    torch.cuda.synchronize()
    hip_ctx.call("elfihip_adaptive_push_dev", rb.h, X2.data_ptr(), 300000, m, m, y.data_ptr(), W.data_ptr(), K,
                 out2.data_ptr(), None, 9 * n)
    vals, rows = rb.result()
    D2 = np.vstack([D, out2.cpu().numpy()])
    both = np.concatenate([np.arange(n) + 7 * n, np.arange(300000) + 9 * n])
    o = np.lexsort((both, D2[:, -1]))[:k]
    assert np.array_equal(vals, D2[o, -1]) and np.array_equal(rows, both[o])


def test_argument_errors(hip_ctx):
    import elfi_amd
    X, y = np.zeros((10, 4)), np.zeros((1, 4))
    with pytest.raises(ValueError):
        elfi_amd.adaptive_batch(X, np.zeros((1, 3)), np.ones((1, 4)))
    with pytest.raises(ValueError):
        elfi_amd.adaptive_batch(X, y, np.ones((2, 5)))
    with pytest.raises(ValueError):
        elfi_amd.adaptive_batch(np.zeros(10), y, np.ones((1, 4)))
    rb = elfi_amd.RunningBest(5, accept=[np.inf, 1.0])
    with pytest.raises(ValueError):          # two thresholds, three nested columns
        elfi_amd.adaptive_batch(X, y, np.ones((3, 4)), state=rb)
    d, _ = elfi_amd.adaptive_batch(np.zeros((0, 4)), y, np.ones((1, 4)))
    assert d.shape == (0, 1)


def test_device_simulator_rows_and_the_pass_on_the_kept_copy(hip_ctx):
    """elfi_amd.randn_rows: the synthetic Gaussian simulator on the device -- the draws of elfihip_randn_dev shaped into
    rows -- whose output stays on the device: adaptive_batch / cdist_rows on the returned array run on that copy (same
    results as on an uploaded copy of the same numbers), a later simulator call replaces it (the earlier array is uploaded)."""
    import torch
    import elfi_amd
    from elfi_amd import _lib
    n, m = 20001, 64
    rs = np.random.RandomState(0)
    loc, scale = rs.uniform(-3, 3, n), np.linspace(1.0, 20.0, m)
    X = elfi_amd.randn_rows(loc, scale, seed=11, stream=2)
    z = torch.empty(n * m, dtype=torch.float64, device='cuda')
    hip_ctx.call("elfihip_randn_dev", C.c_uint64(11), C.c_uint64(2), n * m, C.c_double(0.0), C.c_double(1.0), z.data_ptr())
    hip_ctx.synchronize()
    ref = z.cpu().numpy().reshape(n, m) * scale + loc[:, None]
    np.testing.assert_allclose(X, ref, rtol=0, atol=4 * np.spacing(np.abs(ref).max()))
    assert abs(np.mean((X - loc[:, None]) / scale)) < 5e-3 and abs(np.std((X - loc[:, None]) / scale) - 1) < 5e-3
    y = rs.randn(1, m)
    W = np.vstack([np.ones(m), rs.uniform(0.1, 2, m)])
    assert _lib.rows_epoch_of(X, hip_ctx) is not None
    d, st = elfi_amd.adaptive_batch(X, y, W, store=(0, 0.0, 0.0))             # on the kept copy
    d2, st2 = elfi_amd.adaptive_batch(X.copy(), y, W, store=(0, 0.0, 0.0))    # uploaded
    assert np.array_equal(d, d2) and np.array_equal(st[1], st2[1]) and np.array_equal(st[2], st2[2])
    assert np.array_equal(d, _nested_ref(X, y, W))
    assert np.array_equal(elfi_amd.cdist_rows(X, y), O.cdist_rows(X, y, 'euclidean'))
    assert np.array_equal(elfi_amd.cdist_rows(X, y, w=W[1]), O.cdist_rows(X, y, 'euclidean', w=W[1]))
    X2 = elfi_amd.randn_rows(loc[:500], scale, seed=12)
    assert _lib.rows_epoch_of(X, hip_ctx) != _lib.rows_epoch_of(X2, hip_ctx)
    assert np.array_equal(elfi_amd.adaptive_batch(X, y, W)[0], d)             # stale copy: the array is uploaded instead
    assert np.array_equal(elfi_amd.adaptive_batch(X2, y, W)[0], _nested_ref(X2, y, W))


@pytest.mark.parametrize('m,K', [(16, 1), (16, 8), (32, 3), (32, 7), (64, 1), (64, 3), (64, 8)])
def test_lds_dma_form_of_the_pass(hip_ctx, m, K):
    """The LDS-DMA form of the fused pass (csrc/adaptive.hip: adaptive_dma_kernel, elfihip_dist_set_form 2; 16 / 32 / 64
    summaries and up to 8 weight vectors; measured slower, not the default) next to the register-staged form (1): the same nested distances
    bit for bit (both equal to cdist), statistics inside the same a-priori bounds, the same selection -- on ragged sizes
    around the slot (64 / 32 rows) and ring boundaries."""
    import elfi_amd
    rs = np.random.RandomState(100 * m + K)
    y = rs.randn(1, m)
    W = np.vstack([np.ones(m)] + [rs.uniform(0.01, 4, m) for _ in range(K - 1)])
    try:
        for n in (1, 31, 33, 64, 65, 129, 4099, 250007):
            X = rs.randn(n, m) * rs.uniform(0.1, 100, m) + rs.uniform(-1000, 1000, m)
            ref = _nested_ref(X, y, W)
            out = {}
            for form in (2, 1):
                hip_ctx.call('elfihip_dist_set_form', form)
                rb = elfi_amd.RunningBest(50)
                d, store = elfi_amd.adaptive_batch(X, y, W, store=(0, 0.0, 0.0), state=rb)
                assert np.array_equal(d, ref), (form, n)
                if n > 1:
                    _check_stats(store, X)
                out[form] = (store, rb.result())
            assert out[2][0][0] == out[1][0][0] == n
            assert np.array_equal(out[2][1][0], out[1][1][0]) and np.array_equal(out[2][1][1], out[1][1][1])
            if n > 1:     # two different summation trees of the same statistics: equal to rounding
                np.testing.assert_allclose(out[2][0][1], out[1][0][1], rtol=1e-12, atol=1e-9)
                np.testing.assert_allclose(out[2][0][2], out[1][0][2], rtol=1e-11)
    finally:
        hip_ctx.call('elfihip_dist_set_form', 0)
