"""GPU: k smallest distances / sampler merge vs NumPy (exact: selection moves values, it computes none)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ref(d, k):
    d = np.asarray(d, dtype=np.float64)
    order = np.lexsort((np.arange(len(d)), np.where(np.isnan(d), np.inf, d), np.isnan(d)))[:k]
    return d[order], order


@pytest.mark.parametrize('form', ['resident', 'nine-launch'])
@pytest.mark.parametrize('n,k', [(1, 1), (10, 3), (1000, 1000), (1000, 5000), (4097, 100), (10**6, 1000),
                                 (10**6, 1), (300000, 10000), (2 * 10**6 + 17, 64)])
def test_smallest_k_matches_numpy(hip_ctx, monkeypatch, form, n, k):
    import elfi_amd
    monkeypatch.setenv('ELFIHIP_TOPK_MULTI', '1' if form == 'nine-launch' else '0')
    rs = np.random.RandomState(n + k)
    d = np.abs(rs.randn(n)) * rs.uniform(0.1, 10)
    vals, idx = elfi_amd.smallest_k(d, k)
    rv, ri = _ref(d, min(k, n))
    assert np.array_equal(vals, rv) and np.array_equal(idx, ri)


@pytest.mark.parametrize('form', ['resident', 'nine-launch'])
def test_smallest_k_ties_negatives_nan_inf(hip_ctx, monkeypatch, form):
    import elfi_amd
    monkeypatch.setenv('ELFIHIP_TOPK_MULTI', '1' if form == 'nine-launch' else '0')
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
