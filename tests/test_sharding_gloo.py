"""world_size-2 tests of the multi-GPU host logic on CPU (gloo).

The sharding code (elfi_amd/sharding.py) is the one bench.py and a multi-GPU user run; here its
GPU calls are replaced by a backend built on the oracle so that the partitioning, the gather into
batch-index order and the fixed-order Welford merge can be checked without a GPU:
  * sharded result == single-process oracle result on the concatenated batches (distances
    bit-exact; scales to 1e-12 -- Chan's merge reorders the summation);
  * every rank ends with bit-identical weights.
"""
import os
import socket
import sys

import numpy as np
import pytest

import distance_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend:
    """CPU stand-in for sharding.HipBackend (tests only)."""

    def welford(self, X, count, mean, M2):
        count = count + len(X)
        d1 = X - mean
        mean = mean + np.sum(d1, axis=0) / count
        M2 = M2 + np.sum(d1 * (X - mean), axis=0)
        return count, mean, M2

    def nested(self, X, y, W):
        return np.column_stack([O.cdist_rows(X, y, 'euclidean', w=w) for w in W])


def _batches(n_batches, rows, m):
    return [np.random.RandomState(100 + b).randn(rows + 7 * b, m) * np.linspace(0.5, 30, m) + b
            for b in range(n_batches)]


def _worker(rank, world, port, n_batches, rows, m, out_dir):
    for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from elfi_amd import sharding as S
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        data = _batches(n_batches, rows, m)
        y = np.random.RandomState(7).randn(1, m)
        mine = S.owned_batches(n_batches, rank, world)
        ad = S.ShardedAdaptiveDistance(m, backend=OracleBackend())
        weights = []
        for rnd in range(2):
            for b in mine:
                ad.add_data(data[b])
            weights.append(ad.sync_scale().copy())
            ad.update_distance()
        local = [ad.nested_distance(data[b], y) for b in mine]
        stacked = np.vstack(local) if local else np.empty((0, 3))
        got = S.gather_rows(stacked, dst=0)
        np.save(os.path.join(out_dir, 'w_%d.npy' % rank), np.array(weights))
        if rank == 0:
            per_rank = []
            for r in range(world):
                lens = [len(data[b]) for b in S.owned_batches(n_batches, r, world)]
                per_rank.append(np.split(got[r], np.cumsum(lens)[:-1]) if lens else [])
            ordered = S.interleave_batches(per_rank, n_batches, world)
            np.save(os.path.join(out_dir, 'dist.npy'), np.vstack(ordered))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('n_batches', [5, 2, 1])
def test_sharded_adaptive_distance_matches_single_process(tmp_path, n_batches):
    import torch.multiprocessing as mp
    world, rows, m = 2, 300, 6
    mp.spawn(_worker, args=(world, _free_port(), n_batches, rows, m, str(tmp_path)), nprocs=world, join=True)
    w0, w1 = np.load(tmp_path / 'w_0.npy'), np.load(tmp_path / 'w_1.npy')
    assert np.array_equal(w0, w1), 'every rank must hold bit-identical scales'
    # single-process reference: the oracle's AdaptiveDistance over all batches in index order
    data = _batches(n_batches, rows, m)
    y = np.random.RandomState(7).randn(1, m)
    ref = O.AdaptiveDistanceOracle()
    for rnd in range(2):
        for X in data:
            ref.add_data(X)
        np.testing.assert_allclose(w0[rnd], ref.scale, rtol=1e-12)
        ref.update_distance()
    got = np.load(tmp_path / 'dist.npy')
    # distances with the sharded run's own weights are bit-exact cdist results, in batch order
    W = [np.ones(m)] + [(1.0 / w) ** 2 for w in w0]
    exp = np.vstack([np.column_stack([O.cdist_rows(X, y, 'euclidean', w=w) for w in W]) for X in data])
    assert got.shape == exp.shape and np.array_equal(got, exp)
    np.testing.assert_allclose(got, np.vstack([ref.nested_distance(X, y) for X in data]), rtol=1e-11)


def _strong_worker(rank, world, port, total, m, k, out_dir):
    for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from elfi_amd import sharding as S
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        X = np.random.RandomState(5).randn(total, m)          # the whole job; a rank touches only its rows
        y = np.random.RandomState(6).randn(1, m)
        row0, rows = S.strong_partition(total, world)[rank]
        best_v, best_r = np.full(k, np.inf), np.full(k, np.iinfo(np.int64).max)
        steps = 3
        for st in range(steps):                                # the rank's rows in `steps` pushes, a gather after each
            lo = row0 + rows * st // steps
            hi = row0 + rows * (st + 1) // steps
            d = O.cdist_rows(X[lo:hi], y, 'euclidean')
            best_v, best_r = S.merge_best([(best_v, best_r), (d, np.arange(lo, hi))], k)
            pad_v, pad_r = np.full(k, np.inf), np.zeros(k)
            pad_v[:len(best_v)], pad_r[:len(best_r)] = best_v, best_r
            got = S.gather_rows(np.column_stack([pad_v, pad_r]), dst=0)      # ONE exchange per step, as bench.py
            if rank == 0:
                gv, gr = S.merge_best([(g[:, 0], g[:, 1].astype(np.int64)) for g in got], k)
        if rank == 0:
            np.savez(os.path.join(out_dir, 'best.npz'), v=gv, r=gr)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('total', [1001, 64, 3])
def test_strong_scaling_partition_and_best_k_exchange(tmp_path, total):
    """--scaling strong: the job's rows are split over the ranks, every rank keeps the best k of ITS rows under global
    row numbers, rank 0 merges the gathered states: equal to the best k of the whole job."""
    import torch.multiprocessing as mp
    from elfi_amd.sharding import strong_partition
    world, m, k = 2, 5, 40
    parts = strong_partition(total, world)
    assert sum(r for _, r in parts) == total and parts[0][0] == 0 and parts[1][0] == parts[0][1]
    assert strong_partition(10, 4) == [(0, 3), (3, 3), (6, 3), (9, 1)] and strong_partition(2, 4)[2:] == [(2, 0), (2, 0)]
    mp.spawn(_strong_worker, args=(world, _free_port(), total, m, k, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / 'best.npz')
    X = np.random.RandomState(5).randn(total, m)
    y = np.random.RandomState(6).randn(1, m)
    d = O.cdist_rows(X, y, 'euclidean')
    order = np.lexsort((np.arange(total), d))[:k]
    assert np.array_equal(got['r'], order) and np.array_equal(got['v'], d[order])


def test_merge_welford_equals_pooled_statistics():
    from elfi_amd.sharding import merge_welford, batch_owner, owned_batches, interleave_batches
    rs = np.random.RandomState(0)
    parts = [rs.randn(n, 4) * 3 + 10 for n in (50, 1, 0, 200)]
    states = []
    for X in parts:
        states.append((len(X), X.mean(0) if len(X) else np.zeros(4), ((X - X.mean(0)) ** 2).sum(0) if len(X) else np.zeros(4)))
    N, mean, M2 = merge_welford(states)
    allx = np.vstack(parts)
    assert N == len(allx)
    np.testing.assert_allclose(mean, allx.mean(0), rtol=1e-13)
    np.testing.assert_allclose(np.sqrt(M2 / N), allx.std(0), rtol=1e-12)
    assert [batch_owner(b, 4) for b in range(6)] == [0, 1, 2, 3, 0, 1]
    assert owned_batches(7, 1, 3) == [1, 4]
    per_rank = [[('b', b) for b in owned_batches(7, r, 3)] for r in range(3)]
    assert interleave_batches(per_rank, 7, 3) == [('b', b) for b in range(7)]


# ---- sharded acquisition starts (configs[4]: starts split over the GPUs, all-gather of the optima)
class _FakeHandle:
    """Stands in for GPHandle.lcb_minimize: a deterministic 'optimiser' (projected gradient steps on a
    quadratic bowl) so that the sharding / gathering logic can run without a GPU."""

    def lcb_minimize(self, starts, bounds, beta, maxiter=1000):
        lo, hi = np.array(bounds).T
        x = np.clip(np.asarray(starts, float), lo, hi)
        c = np.linspace(-0.5, 0.7, x.shape[1])
        for _ in range(50):
            x = np.clip(x - 0.2 * 2 * (x - c), lo, hi)
        f = np.sum((x - c) ** 2, axis=1) + 0.01 * np.sin(7 * np.asarray(starts)[:, 0])   # start-dependent tie-breaker
        return x, f, np.full(len(x), 50, dtype=np.int32), 51 * len(x)


class _FakeModel:
    parameter_names = ['a', 'b', 'c']
    input_dim = 3
    bounds = [(-2, 2), (-1, 1), (0, 3)]
    n_evidence = 10
    _handle = _FakeHandle()


def _acq_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from elfi_amd import HipLCBSC
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        acq = HipLCBSC(_FakeModel(), n_inits=7, noise_var=0.05, seed=11)
        x = acq.acquire(3, t=4)
        np.save(os.path.join(out_dir, 'acq_%d.npy' % rank), x)
        np.save(os.path.join(out_dir, 'vals_%d.npy' % rank), acq.last_opt['vals'])
    finally:
        dist.destroy_process_group()


def test_sharded_acquisition_starts(tmp_path):
    import torch.multiprocessing as mp
    from elfi_amd import HipLCBSC
    mp.spawn(_acq_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    single = HipLCBSC(_FakeModel(), n_inits=7, noise_var=0.05, seed=11)
    x1 = single.acquire(3, t=4)
    a0, a1 = np.load(tmp_path / 'acq_0.npy'), np.load(tmp_path / 'acq_1.npy')
    assert np.array_equal(a0, a1), 'every rank must acquire the same points'
    assert np.array_equal(a0, x1), 'sharding the starts must not change the acquisition'
    assert np.array_equal(np.load(tmp_path / 'vals_0.npy'), single.last_opt['vals'])


# ---- the REAL elfi.BOLFI loop on two ranks, acquisition starts sharded (configs[4]'s split: every rank holds the same
# evidence and surrogate, start s belongs to rank s % world, ONE all-gather of the optima per acquisition)
class _OracleHandle:
    """GPHandle.lcb_minimize on the CPU oracle: scipy L-BFGS-B from every start handed in (what the device state
    machines do in lock-step), so that HipLCBSC's sharding logic runs in the reference's loop without a GPU."""

    def __init__(self, model):
        self.model = model

    def lcb_minimize(self, starts, bounds, beta, maxiter=1000):
        import gp_oracle as G
        import scipy.optimize
        post = self.model.instance
        fun = lambda x: float(post.predict(x[None, :], noiseless=True)[0][0, 0]
                              - np.sqrt(beta * post.predict(x[None, :], noiseless=True)[1][0, 0]))
        locs, vals = [], []
        for x0 in np.asarray(starts, float):
            r = scipy.optimize.minimize(fun, x0, method='L-BFGS-B', bounds=bounds, options={'maxiter': maxiter})
            locs.append(r.x)
            vals.append(r.fun)
        return np.array(locs), np.array(vals), np.zeros(len(locs), dtype=np.int32), 0


def _bolfi_run(n_inits, shard):
    import ref_shim
    elfi = ref_shim.install()
    import elfi.clients.native as native
    native.set_as_default()
    from elfi.examples import ma2
    from elfi.model.extensions import ModelPrior
    from elfi_amd import HipLCBSC
    from oracle_gp_model import OracleGPRegression
    m = ma2.get_model(seed_obs=4)
    log_d = elfi.Operation(np.log, m['d'], name='log_d')
    bounds = {'t1': (-2, 2), 't2': (-1, 1)}
    gp = OracleGPRegression(['t1', 't2'], bounds=bounds)
    gp._handle = _OracleHandle(gp)
    acq = HipLCBSC(gp, prior=ModelPrior(m, parameter_names=['t1', 't2']), n_inits=n_inits, noise_var=0.1,
                   exploration_rate=10, seed=1)
    acq.shard_starts = shard
    b = elfi.BOLFI(log_d, batch_size=1, initial_evidence=20, update_interval=5, bounds=bounds, acq_noise_var=0.1,
                   target_model=gp, acquisition_method=acq, seed=1)
    b.fit(n_evidence=32, bar=False)
    return gp.X.copy(), gp.Y.copy()


def _bolfi_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        X, Y = _bolfi_run(6, True)
        np.save(os.path.join(out_dir, 'bolfi_X_%d.npy' % rank), X)
        np.save(os.path.join(out_dir, 'bolfi_Y_%d.npy' % rank), Y)
    finally:
        dist.destroy_process_group()


def test_reference_bolfi_loop_with_sharded_starts(tmp_path):
    """elfi.BOLFI(...).fit on two ranks (elfi/methods/inference/bolfi.py:201-254) with HipLCBSC(shard_starts=True): both
    ranks acquire the same points as ONE process running all the starts -- the evidence stays identical on every rank,
    which is what lets every rank rebuild the same surrogate without exchanging it."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import ref_shim
    if not ref_shim.available():
        pytest.skip('no reference package')
    import torch.multiprocessing as mp
    mp.spawn(_bolfi_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    X1, Y1 = _bolfi_run(6, False)
    X0, Xb = np.load(tmp_path / 'bolfi_X_0.npy'), np.load(tmp_path / 'bolfi_X_1.npy')
    assert X0.shape == (32, 2)
    assert np.array_equal(X0, Xb) and np.array_equal(np.load(tmp_path / 'bolfi_Y_0.npy'), np.load(tmp_path / 'bolfi_Y_1.npy'))
    assert np.array_equal(X0, X1) and np.array_equal(np.load(tmp_path / 'bolfi_Y_0.npy'), Y1)
