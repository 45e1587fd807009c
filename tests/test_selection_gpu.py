"""GPU: k smallest distances / sampler merge vs NumPy (exact: selection moves values, it computes none)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture
def topk_form(hip_ctx):
    """Select the implementation for one test (elfihip_topk_set_form), restore the default afterwards."""
    def choose(form):
        hip_ctx.call("elfihip_topk_set_form", {'resident': 0, 'nine-launch': 1, 'resident-memory': 2}[form])
    yield choose
    hip_ctx.call("elfihip_topk_set_form", 0)


def _ref(d, k):
    d = np.asarray(d, dtype=np.float64)
    order = np.lexsort((np.arange(len(d)), np.where(np.isnan(d), np.inf, d), np.isnan(d)))[:k]
    return d[order], order


@pytest.mark.parametrize('form', ['resident', 'nine-launch', 'resident-memory'])
@pytest.mark.parametrize('n,k', [(1, 1), (10, 3), (1000, 1000), (1000, 5000), (4097, 100), (10**6, 1000),
                                 (10**6, 1), (300000, 10000), (2 * 10**6 + 17, 64), (16384, 16384), (16385, 9000),
                                 (70000, 63), (3 * 10**6, 1000)])
def test_smallest_k_matches_numpy(hip_ctx, topk_form, form, n, k):
    import elfi_amd
    topk_form(form)
    rs = np.random.RandomState(n + k)
    d = np.abs(rs.randn(n)) * rs.uniform(0.1, 10)
    vals, idx = elfi_amd.smallest_k(d, k)
    rv, ri = _ref(d, min(k, n))
    assert np.array_equal(vals, rv) and np.array_equal(idx, ri)


@pytest.mark.parametrize('form', ['resident', 'nine-launch', 'resident-memory'])
def test_smallest_k_ties_negatives_nan_inf(hip_ctx, topk_form, form):
    import elfi_amd
    topk_form(form)
    rs = np.random.RandomState(0)
    d = rs.randint(-5, 5, 20000).astype(float)            # massive ties, negative values
    d[::97] = np.nan
    d[5::101] = np.inf
    d[7::103] = -np.inf
    for k in (1, 50, 777, 19999, 20000):
        vals, idx = elfi_amd.smallest_k(d, k)
        rv, ri = _ref(d, k)
        assert np.array_equal(idx, ri)
        assert np.array_equal(vals, rv, equal_nan=True)
    assert elfi_amd.smallest_k(d, 0)[0].shape == (0,)
    assert elfi_amd.smallest_k(np.empty(0), 5)[1].shape == (0,)
    nested = np.column_stack([rs.rand(5000), rs.rand(5000)])
    v, i = elfi_amd.smallest_k(nested, 10)                  # last column decides (samplers.py:233)
    assert np.array_equal(i, np.argsort(nested[:, 1], kind='stable')[:10])


def test_merge_batch_equals_the_reference_merge(hip_ctx):
    """Rejection._merge_batch semantics (samplers.py:209-237) over several batches."""
    import elfi_amd
    rs = np.random.RandomState(3)
    n_samples, bs = 200, 5000
    state = None
    ref = {'d': np.full(n_samples + bs, np.inf), 't1': np.empty(n_samples + bs), 'S': np.empty((n_samples + bs, 2))}
    for b in range(4):
        batch = {'d': np.abs(rs.randn(bs)), 't1': rs.rand(bs), 'S': rs.randn(bs, 2)}
        state = elfi_amd.merge_batch(state, batch, 'd', n_samples)
        for k_, v in ref.items():                           # the reference's steps, verbatim in spirit
            v[-bs:] = batch[k_]
        order = np.argsort(ref['d'])
        for k_, v in ref.items():
            v[:] = v[order]
        for k_ in ref:
            assert np.array_equal(state[k_], ref[k_][:n_samples]), (b, k_)
    assert np.all(np.diff(state['d']) >= 0)


@pytest.mark.parametrize('nested', [False, True])
def test_merge_batch_with_threshold_equals_the_reference_merge(hip_ctx, nested):
    """The acceptance condition (samplers.py:219-225: EVERY nested column <= threshold) is taken over the whole
    batch before the k smallest are selected: accepted rows that rank beyond n_samples by the last column must
    survive when higher-ranked rows fail the threshold in an earlier column."""
    import elfi_amd
    rs = np.random.RandomState(11)
    n_samples, bs, thr = 50, 4000, 0.9
    K = 3 if nested else 1
    state = None
    dshape = (n_samples + bs, K) if nested else (n_samples + bs,)
    ref = {'d': np.full(dshape, np.inf), 't1': np.empty(n_samples + bs)}
    kept = 0
    for b in range(5):
        d = np.abs(rs.randn(bs, K)) * np.array([3.0, 2.0, 1.0][:K])   # early columns fail the threshold most often
        if not nested:
            d = d[:, 0] * 0.3
        batch = {'d': d, 't1': rs.rand(bs)}
        state = elfi_amd.merge_batch(state, batch, 'd', n_samples, threshold=thr)
        accepted = np.all(np.atleast_2d(np.transpose(batch['d'] <= thr)), axis=0)   # the reference's steps
        num = int(np.sum(accepted))
        kept += num
        if num > 0:
            for k_, v in ref.items():
                v[-num:] = batch[k_][accepted]
        order = np.argsort(np.atleast_2d(np.transpose(ref['d']))[-1], kind='stable')
        for k_, v in ref.items():
            v[:] = v[order]
        for k_ in ref:
            assert np.array_equal(state[k_], ref[k_][:n_samples]), (b, k_)
    assert kept > n_samples and np.all(np.isfinite(state['d']))
    # nothing accepted: the state is unchanged
    same = elfi_amd.merge_batch(state, {'d': np.full_like(batch['d'], 5.0), 't1': batch['t1']}, 'd', n_samples,
                                threshold=thr)
    for k_ in ref:
        assert np.array_equal(same[k_], state[k_])


def test_smallest_k_many_equal_keys_large(hip_ctx):
    """10^6 keys with 7 distinct values: the prefix class stays large, so the resident kernel runs all its passes and
    the cut falls inside a run of equal keys (lowest rows win)."""
    import elfi_amd
    rs = np.random.RandomState(5)
    d = rs.randint(0, 7, 10**6).astype(float) * 0.25
    for k in (1, 1000, 150000):
        vals, idx = elfi_amd.smallest_k(d, k)
        rv, ri = _ref(d, k)
        assert np.array_equal(idx, ri) and np.array_equal(vals, rv)


def _best_ref(d_all, k):
    order = np.lexsort((np.arange(len(d_all)), d_all))[:k]
    return d_all[order], order


@pytest.mark.parametrize('metric,m,kw', [('euclidean', 32, {}), ('cityblock', 6, {}), ('euclidean', 7, {}),
                                         ('euclidean', 64, {'w': True}), ('minkowski', 10, {'p': 3.0})])
def test_running_best_equals_the_reference_merge_over_batches(hip_ctx, metric, m, kw):
    """elfihip_reject_*: the state after every batch is what Rejection._merge_batch (samplers.py:209-237) would hold
    -- the k smallest distances so far, ties to the earlier row -- and the distances returned are the Distance node's
    (bit-exact for these metrics except general Minkowski, where selection is still exact on the returned values)."""
    import elfi_amd
    import distance_oracle as O
    rs = np.random.RandomState(m)
    k = 500
    w = rs.uniform(0.5, 2, m) if kw.get('w') else None
    rb = elfi_amd.RunningBest(k, metric=metric, w=w, p=kw.get('p', 2.0))
    y = rs.randn(1, m)
    seen = []
    for n in (120, 300, 1000, 50000, 7, 200000, 33333):        # the first three fill the state (120 + 300 < 500)
        X = rs.randn(n, m) * (1.0 + 0.5 * rs.rand())
        d = rb.push(X, y)
        if metric != 'minkowski':
            assert np.array_equal(d, O.cdist_rows(X, y, metric, w=w))
        seen.append(d)
        vals, rows = rb.result()
        dv, dr = _best_ref(np.concatenate(seen), k)
        assert np.array_equal(vals, dv) and np.array_equal(rows, dr), (n, len(vals))
    rb.reset()
    d = rb.push(X[:10], y)
    vals, rows = rb.result()
    assert np.array_equal(vals, np.sort(d)) and len(rows) == 10


def test_running_best_ties_and_explicit_row_numbers(hip_ctx):
    import elfi_amd
    rs = np.random.RandomState(1)
    rb = elfi_amd.RunningBest(64, metric='cityblock')
    y = np.zeros((1, 4))
    alld, allr = [], []
    for b in range(6):
        X = rs.randint(0, 3, (5000, 4)).astype(float)           # distances in {0..8}: massive ties
        base = 1000000 * b                                      # batch index -> global row numbers
        alld.append(rb.push(X, y, row_base=base))
        allr.append(base + np.arange(5000))
    d, r = np.concatenate(alld), np.concatenate(allr)
    order = np.lexsort((r, d))[:64]
    vals, rows = rb.result()
    assert np.array_equal(vals, d[order]) and np.array_equal(rows, r[order])
    with pytest.raises(ValueError):
        elfi_amd.RunningBest((1 << 20) + 1)


def test_running_best_device_batches_nested_and_overflow(hip_ctx):
    """Device-resident batches: nested (n, K) distances ranked by the last column; a batch in which far more rows beat the
    threshold than round 2's fixed candidate list held is simply merged (the reference never fails there)."""
    import ctypes as C
    import torch
    import elfi_amd
    import distance_oracle as O
    from elfi_amd import _lib
    rs = np.random.RandomState(2)
    n, m, K, k = 300000, 64, 3, 1000
    lib = hip_ctx.lib
    h = C.c_void_p()
    hip_ctx.call("elfihip_reject_create", k, C.byref(h))
    W = np.vstack([np.ones(m), rs.uniform(0.5, 2, m), rs.uniform(0.5, 2, m)])
    y = rs.randn(1, m)
    dW, dy = torch.from_numpy(W).cuda(), torch.from_numpy(y).cuda()
    hip_ctx.synchronize()
    cols = []
    for b in range(3):
        X = rs.randn(n, m)
        dX = torch.from_numpy(X).cuda()
        out = torch.empty(n, K, dtype=torch.float64, device='cuda')
        torch.cuda.synchronize()
        rc = lib.elfihip_reject_push_multiw_dev(h, dX.data_ptr(), n, m, m, dy.data_ptr(), dW.data_ptr(), K,
                                                out.data_ptr(), b * n)
        assert rc == 0
        hip_ctx.synchronize()
        oh = out.cpu().numpy()
        assert np.array_equal(oh[:4096, 2], O.cdist_rows(X[:4096], y, 'euclidean', w=W[2]))
        cols.append(oh[:, K - 1].copy())
    vals, rows = np.empty(k), np.empty(k, dtype=np.int64)
    cnt = C.c_int64()
    assert lib.elfihip_reject_result(h, _lib.ptr(vals), _lib.ptr(rows), C.byref(cnt)) == 0 and cnt.value == k
    dv, dr = _best_ref(np.concatenate(cols), k)
    assert np.array_equal(vals, dv) and np.array_equal(rows, dr)
    # distances that exist already, then a batch in which far more than 65536 rows beat the threshold (every one of
    # its 600 000 rows does): no reset needed, the state is the best k of everything pushed
    d2h = np.concatenate(cols)[: 2 * n] * 1e-3
    d2 = torch.from_numpy(d2h).cuda()
    torch.cuda.synchronize()
    assert lib.elfihip_reject_push_dev(h, d2.data_ptr(), 2 * n, 1, 10 * n) == 0
    assert lib.elfihip_reject_result(h, _lib.ptr(vals), _lib.ptr(rows), C.byref(cnt)) == 0 and cnt.value == k
    dv, dr = _best_ref(d2h, k)
    assert np.array_equal(vals, dv) and np.array_equal(rows, dr + 10 * n)
    assert lib.elfihip_reject_reset(h) == 0
    assert lib.elfihip_reject_push_dev(h, d2.data_ptr(), 2 * n, 1, 0) == 0
    assert lib.elfihip_reject_result(h, _lib.ptr(vals), _lib.ptr(rows), C.byref(cnt)) == 0
    dv, dr = _best_ref(d2.cpu().numpy(), k)
    assert np.array_equal(vals, dv) and np.array_equal(rows, dr)
    lib.elfihip_reject_free(h)


def test_running_best_ignores_nan_distances(hip_ctx):
    """NaN distances (a simulator that failed) never enter the state: the count says how many entries are in use while
    fewer than k finite distances exist; afterwards the state equals the reference merge, which sorts NaN last."""
    import elfi_amd
    rs = np.random.RandomState(0)
    rb = elfi_amd.RunningBest(50, metric='euclidean')
    y = np.zeros((1, 3))
    X = rs.randn(40, 3)
    X[5] = np.nan
    X[17, 1] = np.nan
    d = rb.push(X, y)
    assert np.isnan(d).sum() == 2
    v, r = rb.result()
    assert len(v) == 38 and np.all(np.isfinite(v)) and np.array_equal(v, np.sort(d[np.isfinite(d)]))
    seen = [d]
    for n, scale in ((1000, 1.0), (5000, 0.1)):
        X = rs.randn(n, 3) * scale
        X[::97] = np.nan
        seen.append(rb.push(X, y))
        alld = np.concatenate(seen)
        order = np.lexsort((np.arange(len(alld)), np.where(np.isnan(alld), np.inf, alld), np.isnan(alld)))[:50]
        v, r = rb.result()
        assert np.array_equal(r, order) and np.array_equal(v, alld[order])


def test_running_best_small_first_batch_then_large_ones(hip_ctx):
    """The case that lost a whole run in round 2: a first batch of exactly k rows leaves a threshold every row of the
    next batches beats.  Pushes of 200 000 rows follow: each takes the radix selection while n k / rows_seen is large,
    the candidate list afterwards -- exact throughout."""
    import elfi_amd
    rs = np.random.RandomState(5)
    k = 500
    rb = elfi_amd.RunningBest(k)
    y = np.zeros((1, 8))
    seen = [rb.push(rs.randn(k, 8) * 3.0, y)]
    for n in (200000, 200000, 50000, 200000, 200000, 200000):
        seen.append(rb.push(rs.randn(n, 8), y))
        vals, rows = rb.result()
        dv, dr = _best_ref(np.concatenate(seen), k)
        assert np.array_equal(vals, dv) and np.array_equal(rows, dr)
    # without reset, a "new round" whose every row beats the state (an SMC round change)
    seen.append(rb.push(rs.randn(300000, 8) * 1e-3, y))
    vals, rows = rb.result()
    dv, dr = _best_ref(np.concatenate(seen), k)
    assert np.array_equal(vals, dv) and np.array_equal(rows, dr)
    assert rb.push(rs.randn(1000, 8), y, return_distances=False) is None     # nothing but the state leaves the GPU


@pytest.mark.parametrize('k', [3000, 10000])
def test_running_best_large_k_host_merge(hip_ctx, k):
    """n_samples beyond the device-resident merge (k > 2048): same contract, candidates merged on the host."""
    import elfi_amd
    rs = np.random.RandomState(k)
    rb = elfi_amd.RunningBest(k, metric='euclidean')
    y = rs.randn(1, 5)
    seen = []
    for n in (1500, 40000, 40000, 7, 100000, 40000):
        seen.append(rb.push(rs.randn(n, 5), y))
        vals, rows = rb.result()
        dv, dr = _best_ref(np.concatenate(seen), k)
        assert np.array_equal(vals, dv) and np.array_equal(rows, dr), (n, len(vals))
    kth, _, _ = rb.meta()
    assert kth == dv[k - 1]


def test_running_best_acceptance_threshold_nested(hip_ctx):
    """Threshold objective (samplers.py:219-225): a row of nested distances (n, K) takes part only if EVERY column is
    <= threshold; the state ranks the accepted rows by the last column; accepted rows are counted per push."""
    import elfi_amd
    rs = np.random.RandomState(9)
    k, K, thr = 200, 3, 1.2
    rb = elfi_amd.RunningBest(k, accept=thr)
    alld, total = [], 0
    for n in (5000, 20000, 100, 20000):
        d = np.abs(rs.randn(n, K)) + 0.05 * np.arange(K)
        rb.push_distances(d)
        alld.append(d)
        D = np.concatenate(alld)
        ok = np.all(D <= thr, axis=1)
        key = np.where(ok, D[:, -1], np.inf)
        order = np.lexsort((np.arange(len(D)), key))[:min(k, int(ok.sum()))]
        vals, rows = rb.result()
        assert np.array_equal(rows, order) and np.array_equal(vals, D[order, -1])
        kth, last, tot = rb.meta()
        total += int(np.all(d <= thr, axis=1).sum())
        assert last == int(np.all(d <= thr, axis=1).sum()) and tot == total
        assert kth == (vals[k - 1] if len(vals) == k else np.inf)
