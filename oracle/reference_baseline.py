"""TEST / BASELINE INFRASTRUCTURE ONLY -- the REAL reference (oracle/_ref or /root/reference through ref_shim) timed on
the host cores, for bench.py's `cpu_baseline` (kind "reference").  Never imported by elfi_amd/.

What is timed is what a user of the reference runs for BASELINE.json's distance configs:

  * the operation of a real `elfi.Distance('euclidean', S_1 .. S_m)` node -- `partial(distance_as_discrepancy,
    partial(cdist, metric='euclidean'))`, elfi/model/elfi_model.py:1037-1041, elfi/model/utils.py:37-52 -- on one
    batch of m summary columns (configs[1]: 10^6 x 32), one core: SciPy's cdist is single-threaded and the reference's
    native client (elfi/clients/native.py:55-65) executes batches one after the other on the caller's thread;
  * `elfi.Rejection(d, batch_size).sample(...)` (elfi/methods/inference/samplers.py:24-237) on that model under the
    native client and under the reference's multiprocessing client (elfi/clients/multiprocessing.py) with as many
    worker processes as the cgroup grants CPUs -- the whole loop: simulator draws, m Summary nodes, the Distance node,
    `_merge_batch`;
  * configs[0]: the MA2 example, `elfi.Rejection(d, batch_size=1000)`.
"""
import os
import time
from functools import partial

import numpy as np

import ref_shim


def cpu_quota():
    """CPUs this process may use: the cgroup quota if there is one, else the affinity mask."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:          # cgroup v2
            q = f.read().split()
        if q and q[0] != "max":
            quota = float(q[0]) / float(q[1])
    except (OSError, ValueError, IndexError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                qq, pp = float(f.read()), float(g.read())   # cgroup v1
            if qq > 0:
                quota = qq / pp
        except (OSError, ValueError):
            pass
    cores = avail if quota is None else max(1, min(avail, int(round(quota))))
    return cores, avail, quota


# ---- the synthetic Gaussian model of configs[1]: module-level callables, so that the multiprocessing client can
# pickle the compiled net into its workers (elfi/clients/multiprocessing.py:50)
def gauss_sim(mu, m=32, batch_size=1, random_state=None):
    rs = random_state or np.random
    return rs.randn(batch_size, m) + np.asarray(mu).reshape(-1, 1)


def column(x, j=0):
    return x[:, j]


def gauss_model(elfi, m):
    """mu ~ N(0, 1); Y = mu + N(0, I_m); m Summary nodes (one column each); Distance('euclidean') on all of them."""
    import scipy.stats as ss
    mod = elfi.new_model()
    mu = elfi.Prior(ss.norm, 0, 1, model=mod, name='mu')
    obs = np.random.RandomState(1).randn(1, m)
    Y = elfi.Simulator(partial(gauss_sim, m=m), mu, observed=obs, name='Y')
    S = [elfi.Summary(partial(column, j=j), Y, name='S%d' % j) for j in range(m)]
    d = elfi.Distance('euclidean', *S, name='d')
    return mod, d


def node_operation(n, m, budget_s=8.0):
    """The real Distance node's operation on one batch of m summary columns of n rows (one core)."""
    elfi = ref_shim.install()
    mod, d = gauss_model(elfi, m)
    op = d.state['attr_dict']['_operation']   # what the executor calls (elfi/executor.py:143-159)
    rs = np.random.RandomState(0)
    cols = [rs.randn(n) for _ in range(m)]
    obs = tuple(np.random.RandomState(1).randn(1, m)[:, j] for j in range(m))
    out = op(*cols, observed=obs)
    assert out.shape == (n,)
    best, reps, t_end = float("inf"), 0, time.perf_counter() + budget_s
    while time.perf_counter() < t_end or reps < 2:
        t0 = time.perf_counter()
        op(*cols, observed=obs)
        best = min(best, time.perf_counter() - t0)
        reps += 1
    return dict(value=n / best, unit="distances/s", cores=1, kind="reference",
                sample="the operation of a real elfi.Distance('euclidean', S_0..S_%d) node (reference package: "
                       "distance_as_discrepancy = np.column_stack + scipy cdist) on one %d x %d batch, best of %d; "
                       "native client = one core" % (m - 1, n, m, reps),
                seconds_per_batch=best)


def rejection_loop(n_batch, m, n_batches, workers=1):
    """elfi.Rejection(d, batch_size=n_batch).sample(1000, n_sim=n_batches * n_batch) on the synthetic Gaussian model:
    simulator + m summaries + distance + merge per batch.  workers > 1: the reference's multiprocessing client."""
    elfi = ref_shim.install()
    import elfi.client
    if workers > 1:
        import elfi.clients.multiprocessing as mpc
        elfi.client.set_client(mpc.Client(num_processes=workers))
    else:
        import elfi.clients.native as native
        native.set_as_default()
    try:
        mod, d = gauss_model(elfi, m)
        rej = elfi.Rejection(d, batch_size=n_batch, seed=1)
        t0 = time.perf_counter()
        res = rej.sample(1000, n_sim=n_batches * n_batch, bar=False)
        wall = time.perf_counter() - t0
        assert res.n_sim == n_batches * n_batch
    finally:
        if workers > 1:
            try:
                elfi.client.get_client().reset()
            except Exception:
                pass
            import elfi.clients.native as native
            native.set_as_default()
    return dict(value=n_batches * n_batch / wall, unit="distances/s", cores=workers, kind="reference",
                sample="elfi.Rejection(d, batch_size=%d).sample(1000, n_sim=%d): simulator + %d Summary nodes + "
                       "Distance('euclidean') + _merge_batch per batch, %s, %.2f s"
                       % (n_batch, n_batches * n_batch, m,
                          "multiprocessing client with %d workers" % workers if workers > 1 else "native client", wall))


def ma2_rejection(n_sim=200000, batch_size=1000):
    """configs[0]: the MA2 example under elfi.Rejection, batch_size=1000 (native client)."""
    elfi = ref_shim.install()
    import elfi.clients.native as native
    native.set_as_default()
    from elfi.examples import ma2
    mod = ma2.get_model(seed_obs=4)
    rej = elfi.Rejection(mod['d'], batch_size=batch_size, seed=1)
    t0 = time.perf_counter()
    res = rej.sample(1000, n_sim=n_sim, bar=False)
    wall = time.perf_counter() - t0
    return dict(value=res.n_sim / wall, unit="distances/s (simulations: MA2 simulator + 2 autocovariances + distance)",
                cores=1, kind="reference",
                sample="configs[0]: elfi.Rejection(ma2 d, batch_size=%d).sample(1000, n_sim=%d), native client, %.2f s"
                       % (batch_size, n_sim, wall))
