#!/usr/bin/env python
"""bench.py -- throughput of the ELFI hot path on MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Headline line (one JSON object on stdout, rank 0): BASELINE.json's scaling metric "ABC distances/sec".

Default workload (every N): configs[1] per GPU -- a synthetic Gaussian summary matrix of 10^6 samples x 32 summaries,
euclidean distance to the observed summaries (elfi.Distance('euclidean'), elfi/model/elfi_model.py:1037), inputs
resident in HBM.  A step is one ABC batch of the rank: the distance kernel on the main stream and, overlapped on a second
stream, what Rejection keeps of the previous batch (samplers.py:209-237): the device top-1000 (distance, row) pairs --
and, for N > 1, ONE RCCL gather of those 16 KB per rank to rank 0.  The exchange happens EVERY step, inside the timed
region.  Weak scaling: every rank owns independent batches (elfi/methods/parameter_inference.py:283-292); the work
per GPU is the same at every N, so value(N) / (N value(1)) is the cost of the per-step exchange.

--workload adaptive --scaling strong --total 10000000 --m 64 is BASELINE.json's configs[3] (SMC-ABC adaptive distance,
10^7 samples x 64 summaries sharded over the ranks; N = 1 holds all of it): per step the shard's Welford statistics,
an all-gather of the (1 + 2m)-double states, the fixed-order merge ON THE DEVICE, K = 3 nested weighted distances of every
row, the device top-1000 by the last column and one gather of those to rank 0.

The same line carries
  "roofline"      HBM roofline of the distance kernel (HIP events on the library's stream) and, under "phases", the
                  BOLFI phases (Gram, sweep, K^-1 gradient: FP64-matrix; prediction step: HBM) from the library's
                  own event timers, each against the bound that applies to it
  "cpu_baseline"  the oracle (column_stack + SciPy cdist, what the reference executes) on this box's host cores: one
                  core (the reference's native client) and all cores (its multiprocessing client), plus the BOLFI port
  "bolfi"         BASELINE.json's first metric, BOLFI iters/sec (GP fit + acquisition, n=4096, d=10) (rank 0, N=1)
"""
import argparse
import json
import os
import sys
import time

# the host driver of these boxes supports dmabuf IPC only: without this RCCL fails in hipIpcGetMemHandle (N > 1)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 measured)
FP64_MFMA_PEAK_TFLOPS = 78.6  # AMD public spec, FP64 matrix; measured 77.1 (scripts/native/mfma_probe.hip)
K_BEST = 1000


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["distance", "adaptive"], default="distance",
                    help="distance: configs[1] per GPU (default, the headline); adaptive: configs[3] "
                         "(AdaptiveDistance round with K=3 nested weights)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="weak: --n rows per GPU; strong: --total rows shared by the ranks "
                         "(default: weak for distance, strong for adaptive)")
    ap.add_argument("--n", type=int, default=None, help="samples per GPU per step (weak scaling)")
    ap.add_argument("--total", type=int, default=10 ** 7, help="samples per step over all GPUs (strong scaling)")
    ap.add_argument("--m", type=int, default=None, help="summaries per sample (32 distance / 64 adaptive)")
    ap.add_argument("--data", choices=["device", "torch"], default="device",
                    help="synthetic inputs from the library's own generator (elfihip_randn_dev, Philox4x32-10) or torch.randn")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bolfi", action="store_true")
    ap.add_argument("--bolfi-iters", type=int, default=50)
    return ap.parse_args()


# ----------------------------------------------------------------------------- CPU baselines (rank 0, N = 1)
def _oracle_batch(args):
    """One worker of the all-core baseline: `reps` batches of n rows x m summaries through the oracle's
    distance_as_discrepancy (np.column_stack + cdist), as a pool worker of the reference's multiprocessing client
    executes a batch (elfi/clients/multiprocessing.py:50)."""
    import numpy as np
    import distance_oracle as O
    seed, n, m, reps = args
    rs = np.random.RandomState(seed)
    cols = [rs.randn(n) for _ in range(m)]
    obs = tuple(np.random.RandomState(1).randn(1, m)[:, j] for j in range(m))
    op = O.make_distance('euclidean')
    op(*cols, observed=obs)
    t0 = time.perf_counter()
    for _ in range(reps):
        op(*cols, observed=obs)
    return time.perf_counter() - t0


def cpu_baseline_distance(n, m, budget_s=10.0):
    """Reference path on the host: m separate summary columns -> np.column_stack -> cdist
    (elfi/model/utils.py:37-52).  One core: SciPy's cdist is not threaded and the reference's native client runs
    batches one after the other; all cores: one batch stream per worker process, as its multiprocessing client."""
    import multiprocessing as mp
    import numpy as np
    import distance_oracle as O
    rs = np.random.RandomState(0)
    cols = [rs.randn(n) for _ in range(m)]
    obs = tuple(np.random.RandomState(1).randn(1, m)[:, j] for j in range(m))
    op = O.make_distance('euclidean')
    op(*cols, observed=obs)  # warm-up
    best, reps, t_end = float("inf"), 0, time.perf_counter() + budget_s
    while time.perf_counter() < t_end or reps < 2:
        t0 = time.perf_counter()
        op(*cols, observed=obs)
        best = min(best, time.perf_counter() - t0)
        reps += 1
    X = np.column_stack(cols)
    t0 = time.perf_counter()
    O.cdist_rows(X, np.array(obs).reshape(1, -1), 'euclidean')
    kern = time.perf_counter() - t0
    res = dict(value=n / best, unit="distances/s", cores=1, kind="port",
               sample="%d x %d batch through oracle distance_as_discrepancy (np.column_stack + "
                      "scipy cdist), best of %d; cdist alone %.0f Mdist/s" % (n, m, reps, n / kern / 1e6))
    # all cores: batches of 10^5 rows (a usual ELFI batch size) in C worker processes
    try:
        C_ = max(1, min(os.cpu_count() or 1, 128))
        nb, r = 100000, 3
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(C_) as pool:
            busy = pool.map(_oracle_batch, [(100 + i, nb, m, r) for i in range(C_)])
        wall = time.perf_counter() - t0
        res["all_cores"] = dict(value=C_ * r * nb / max(busy), unit="distances/s", cores=C_, kind="port",
                                sample="%d worker processes x %d batches of %d x %d (multiprocessing-client style); "
                                       "slowest worker %.2f s, pool wall %.2f s" % (C_, r, nb, m, max(busy), wall))
    except Exception as e:  # pragma: no cover - a baseline must never take the benchmark down
        res["all_cores"] = dict(error=repr(e))
    return res


def cpu_baseline_bolfi(n, d, S, budget_s=25.0):
    """BOLFI iteration on the host cores with the oracle (NumPy/SciPy restatement of the
    reference's GPy path): one GP rebuild (K, Cholesky, K^-1, alpha -- what GPyRegression.update
    triggers, gpy_regression.py:304-312) + the sequential multi-start L-BFGS-B acquisition of
    bo/utils.py:97-103, bounded to about `budget_s` seconds by limiting the start points run."""
    import numpy as np
    import gp_oracle as G
    from elfi_amd.bolfi_bench import heuristic_hyper, problem
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    X, y, bounds = problem(n, d)
    h = heuristic_hyper(bounds, y)
    G.Posterior(X[:512], y[:512], h['var'], h['ls'], h['bias'], h['noise'])   # BLAS warm-up
    t_fit = float("inf")
    for _ in range(2):
        t0 = time.perf_counter()
        post = G.Posterior(X, y, h['var'], h['ls'], h['bias'], h['noise'])
        t_fit = min(t_fit, time.perf_counter() - t0)
    starts = np.random.RandomState(2).uniform(-2, 2, (S, d))
    t = n
    fun = lambda x: float(G.lcb_evaluate(post, x, t)[0, 0])
    grad = lambda x: G.lcb_evaluate_gradient(post, x, t)[0]
    done, t_acq = 0, 0.0
    for s in range(S):
        t1 = time.perf_counter()
        G.minimize_multistart(fun, grad, bounds, starts[s:s + 1])
        t_acq += time.perf_counter() - t1
        done += 1
        if t_fit + t_acq > budget_s:
            break
    t_acq_full = t_acq * S / done
    return dict(value=1.0 / (t_fit + t_acq_full), unit="iters/s", cores=int(threads), kind="port",
                sample="GP rebuild at n=%d (best of 2: %.2f s) + L-BFGS-B from %d of %d starts (%.2f s, scaled to %d)"
                       % (n, t_fit, done, S, t_acq, S),
                ms_fit=1e3 * t_fit, ms_acquire=1e3 * t_acq_full)


# ----------------------------------------------------------------------------- the two workloads
class Job:
    """Streams, contexts and the per-step exchange shared by both workloads."""

    def __init__(self, dev, local_rank, world, rank):
        import torch
        import elfi_amd
        self.torch, self.dev, self.world, self.rank = torch, dev, world, rank
        # one explicit (non-null) stream shared by torch / RCCL and the library for the kernels of a step, a second one
        # for the per-step gather (N > 1)
        self.main = torch.cuda.Stream(dev)
        self.side = torch.cuda.Stream(dev)
        torch.cuda.set_stream(self.main)
        self.ctx = elfi_amd.Context(local_rank)
        self.ctx.set_stream(self.main.cuda_stream)
        # values and row numbers travel together: one buffer (the int64 rows viewed through the second half), ONE gather
        self.best = [torch.empty(2 * K_BEST, dtype=torch.float64, device=dev) for _ in range(2)]
        self.gath = [torch.empty(2 * K_BEST, dtype=torch.float64, device=dev) for _ in range(world)] \
            if (world > 1 and rank == 0) else None
        self.ev_done = [torch.cuda.Event() for _ in range(2)]   # the step that fills send buffer b is queued (main)
        self.ev_free = [torch.cuda.Event() for _ in range(2)]   # the gather has finished reading send buffer b (side)
        for e in self.ev_free:
            e.record(self.side)

    def randn(self, shape, seed, stream=0, device_gen=True):
        """Standard normals of the given shape on this rank's GPU: the library's counter-based generator (no torch kernel in
        the trace), or torch.randn."""
        import ctypes as C
        torch = self.torch
        if not device_gen:
            gen = torch.Generator(device=self.dev)
            gen.manual_seed(int(seed) * 1000 + int(stream))
            return torch.randn(*shape, dtype=torch.float64, device=self.dev, generator=gen)
        out = torch.empty(*shape, dtype=torch.float64, device=self.dev)
        self.ctx.call("elfihip_randn_dev", C.c_uint64(int(seed)), C.c_uint64(int(stream)), out.numel(), C.c_double(0.0),
                      C.c_double(1.0), out.data_ptr())
        return out

    def new_state(self, k):
        """The sampler state of this rank: the k best (distance, global row) pairs so far, on the device."""
        import ctypes as C
        h = C.c_void_p()
        self.ctx.call("elfihip_reject_create", k, C.byref(h))
        self.state = h
        return h

    def check(self, rc):
        from elfi_amd import _lib
        _lib.check(self.ctx.handle, rc)

    def claim(self, b):
        """Before a push: merges from now on leave the packed state in send buffer b (once the gather that last read
        the buffer has released it)."""
        if self.world > 1:
            self.main.wait_event(self.ev_free[b])
        self.check(self.ctx.lib.elfihip_reject_export_dev(self.state, self.best[b].data_ptr()))

    def publish(self, b):
        """After a push, N > 1: the rank's state as of its last merge (the library merges every 8th push; the final
        result is exact) goes to rank 0 -- ONE gather of the 16 KB send buffer on the second stream, every step."""
        if self.world > 1:
            import torch.distributed as dist
            self.ev_done[b].record(self.main)
            self.side.wait_event(self.ev_done[b])
            with self.torch.cuda.stream(self.side):
                dist.gather(self.best[b], self.gath, dst=0)
            self.ev_free[b].record(self.side)

    def barrier(self):
        import torch.distributed as dist
        if self.world > 1:
            dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def timed(self, step, steps, warmup):
        for i in range(warmup):
            step(i)
        self.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i)
        self.barrier()
        elapsed = time.perf_counter() - t0
        if self.world > 1:
            import torch.distributed as dist
            t = self.torch.tensor([elapsed], dtype=self.torch.float64, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed


def run_distance(args, job):
    """configs[1] per GPU (see the module docstring)."""
    import numpy as np
    torch, dev, world, rank, ctx = job.torch, job.dev, job.world, job.rank, job.ctx
    m = args.m or 32
    scaling = args.scaling or "weak"
    # weak: every rank its own n rows per step; strong: the job's --total rows split over the ranks (sharding.py)
    from elfi_amd.sharding import strong_partition
    n = (args.n or 10 ** 6) if scaling == "weak" else strong_partition(args.total, world)[rank][1]
    units = world * n if scaling == "weak" else args.total       # distances per step over all ranks
    # NBUF independent batches, visited round-robin: the working set (NBUF x 8nm bytes) exceeds the
    # 256 MiB Infinity Cache, so every step streams its batch from HBM instead of re-hitting the MALL
    NBUF = max(3, int(-(-(400 << 20) // (8 * n * m))))
    NBUF = min(NBUF, 8)
    Xs = [job.randn((n, m), 1234 + rank, b, args.data == "device") for b in range(NBUF)]
    y = torch.from_numpy(np.random.RandomState(1).randn(1, m)).to(dev)
    outs = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(2)]
    out_t = torch.empty(n, dtype=torch.float64, device=dev)   # target of the kernel-only timing loop
    k = min(K_BEST, n)
    state = job.new_state(k)
    lib = ctx.lib

    def step(i):
        # one ABC batch: distances + the sampler's running best-k in the same pass (rows numbered globally:
        # rank, batch, row), then the state's hand-over
        b = i & 1
        job.claim(b)
        job.check(lib.elfihip_reject_push_rows_dev(state, 0, Xs[i % NBUF].data_ptr(), n, m, m, y.data_ptr(), None, 2.0,
                                                   outs[b].data_ptr(), (rank * 1000003 + i) * n))
        job.publish(b)

    elapsed = job.timed(step, args.steps, args.warmup)
    # Kernel-only timing for the roofline: HIP events on the SAME stream the kernel runs on, nothing beside it
    torch.cuda.synchronize(dev)
    ctx.timer_start()
    for i in range(args.steps):
        ctx.call("elfihip_dist_rows_dev", 0, Xs[i % NBUF].data_ptr(), n, m, m, y.data_ptr(), None, 2.0,
                 out_t.data_ptr())
    kernel_ms = ctx.timer_stop() / args.steps
    last = (args.steps - 1) % NBUF
    alg_bytes = (8.0 * m + 8.0) * n          # 8m + 8 bytes per distance
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    if rank != 0:
        return None
    # parity guard on what was just timed (checker only)
    import distance_oracle as O
    idx = np.arange(0, n, max(1, n // 4096))[:4096]
    ref = O.cdist_rows(Xs[last][idx].cpu().numpy(), y.cpu().numpy(), 'euclidean')
    dh = out_t.cpu().numpy()
    assert np.array_equal(dh[idx], ref), "bench output differs from the oracle"
    # the sampler state after the run: the k best of everything this rank pushed (warm-up included) -- recomputed here
    # from the NBUF distinct batches, each of which was pushed several times under different global row numbers
    bsel = (args.warmup + args.steps - 1) & 1
    job.check(lib.elfihip_reject_flush(state))
    torch.cuda.synchronize(dev)
    bv = job.best[bsel][:k].cpu().numpy()
    bi = job.best[bsel][k:2 * k].view(torch.int64).cpu().numpy()
    total = args.warmup + args.steps
    dist_of, pool = [], []
    for j in range(min(NBUF, total)):
        ctx.call("elfihip_dist_rows_dev", 0, Xs[j].data_ptr(), n, m, m, y.data_ptr(), None, 2.0, out_t.data_ptr())
        dist_of.append(out_t.cpu().numpy().copy())
        pool.append(np.tile(np.sort(dist_of[j])[:k], len(range(j, total, NBUF))))
    assert np.array_equal(bv, np.sort(np.concatenate(pool))[:k]), "running best-k differs from numpy"
    step_of, row_of = np.divmod(bi - rank * 1000003 * n, n)
    assert np.all((step_of >= 0) & (step_of < total)) and len(set(bi.tolist())) == k
    assert np.array_equal(np.array([dist_of[s_ % NBUF][r_] for s_, r_ in zip(step_of, row_of)]), bv)
    order = np.lexsort((bi, bv))
    assert np.array_equal(order, np.arange(k)), "state must be ascending by (distance, row)"
    # (for N > 1 rank 0's slice of the last gather is its state as of the last merge before that gather)
    traffic, source = None, None
    try:   # HBM traffic per launch from this round's rocprofv3 PMC passes of this very command (profiles/)
        with open(os.path.join(ROOT, "profiles", "distance_pmc.json")) as f:
            pmc = json.load(f)
        if pmc["n"] == n and pmc["m"] == m:
            traffic, source = pmc["traffic_bytes_per_launch"], pmc.get("source")
    except (OSError, ValueError, KeyError):
        pass
    return {
        "metric": "ABC distances/sec", "value": units * args.steps / elapsed, "unit": "distances/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "f64",
        "data": "synthetic (device)" if args.data == "device" else "synthetic",
        "config": {"workload": "configs[1]: synthetic Gaussian summaries, %d samples x %d summaries per GPU per step, "
                               "elfi.Distance('euclidean'), inputs resident in HBM" % (n, m),
                   "samples_per_gpu": n, "summaries": m, "layout": "row-major (n,m) f64", "batches_in_rotation": NBUF,
                   "step": "ONE pass: distance kernel with the sampler's running top-%d fused in (rows below the "
                           "state's k-th distance are listed by the kernel itself; a one-workgroup merge every "
                           "8th step)" % k,
                   "exchange": ("every step: ONE RCCL gather of the rank's packed state (top-%d (distance, row) pairs as of "
                                "its last merge, 16 KB per rank) to rank 0, on a second stream" % k) if world > 1 else "none (N = 1)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": source,
                     "kernel": "dist_rows_pipe_kernel<euclidean>", "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "frac_of_measured_copy_peak_6290": achieved / 6290.0,
                     "step_rate_frac": (alg_bytes * args.steps / elapsed / 1e9) / HBM_PEAK_GBS},
    }


def run_adaptive(args, job):
    """configs[3]: one SMC-ABC adaptive-distance round on this rank's shard per step
    (elfi/model/elfi_model.py:1104-1151); nothing leaves the device inside a step."""
    import numpy as np
    import torch.distributed as dist
    torch, dev, world, rank, ctx = job.torch, job.dev, job.world, job.rank, job.ctx
    m, K = args.m or 64, 3
    scaling = args.scaling or "strong"
    from elfi_amd.sharding import strong_partition
    n = (args.n or 1250000) if scaling == "weak" else strong_partition(args.total, world)[rank][1]
    units = world * n if scaling == "weak" else args.total
    X = job.randn((n, m), 100 + rank, 0, args.data == "device") * torch.linspace(0.5, 20, m, device=dev, dtype=torch.float64)
    y = torch.from_numpy(np.random.RandomState(1).randn(1, m)).to(dev)
    outs = [torch.empty(n, K, dtype=torch.float64, device=dev) for _ in range(2)]
    out_t = torch.empty(n, K, dtype=torch.float64, device=dev)
    ns = 1 + 2 * m
    state = torch.zeros(ns, dtype=torch.float64, device=dev)
    states = torch.zeros(world, ns, dtype=torch.float64, device=dev)
    merged = torch.zeros(ns, dtype=torch.float64, device=dev)
    W = torch.ones(K, m, dtype=torch.float64, device=dev)
    k = min(K_BEST, n)
    rstate = job.new_state(k)
    lib = ctx.lib

    def step(i):
        b = i & 1
        state.zero_()
        ctx.call("elfihip_welford_update_dev", X.data_ptr(), n, m, m, state.data_ptr())
        if world > 1:
            dist.all_gather_into_tensor(states, state)       # (1 + 2m) doubles per rank, stays on the device
        else:
            states[0].copy_(state)
        # fixed-order Chan merge + weights 1/scale^2 on the device: row 1 of W; row 2 = a second adaptation round
        ctx.call("elfihip_welford_merge_dev", states.data_ptr(), world, m, merged.data_ptr(), W[1].data_ptr())
        torch.mul(W[1], 0.5, out=W[2])
        # an SMC round re-ranks from scratch (samplers.py:279-299): fresh state, then distances + running best-k (by the
        # LAST nested column, samplers.py:233) in one pass
        job.check(lib.elfihip_reject_reset(rstate))
        job.claim(b)
        job.check(lib.elfihip_reject_push_multiw_dev(rstate, X.data_ptr(), n, m, m, y.data_ptr(), W.data_ptr(), K,
                                                     outs[b].data_ptr(), rank * n))
        job.publish(b)

    elapsed = job.timed(step, args.steps, args.warmup)
    torch.cuda.synchronize(dev)
    ctx.timer_start()
    for _ in range(args.steps):
        ctx.call("elfihip_dist_multiw_dev", X.data_ptr(), n, m, m, y.data_ptr(), W.data_ptr(), K, out_t.data_ptr())
    kernel_ms = ctx.timer_stop() / args.steps
    alg_bytes = (8.0 * m + 8.0 * K) * n
    if rank != 0:
        return None
    import distance_oracle as O
    from elfi_amd import sharding
    idx = np.arange(0, n, max(1, n // 2048))[:2048]
    Wh = W.cpu().numpy()
    ref = np.column_stack([O.cdist_rows(X[idx].cpu().numpy(), y.cpu().numpy(), 'euclidean', w=Wh[kk]) for kk in range(K)])
    assert np.array_equal(out_t[idx].cpu().numpy(), ref), "adaptive bench output differs from the oracle"
    bsel = (args.warmup + args.steps - 1) & 1       # the state after the last timed step
    job.check(lib.elfihip_reject_flush(rstate))
    torch.cuda.synchronize(dev)
    bv = job.best[bsel][:k].cpu().numpy()
    bi = job.best[bsel][k:2 * k].view(torch.int64).cpu().numpy() - rank * n
    dl = outs[bsel][:, K - 1].cpu().numpy()
    assert np.array_equal(bv, np.sort(dl)[:k]) and np.array_equal(dl[bi], bv), "running best-k differs from numpy"
    sh = states.cpu().numpy()
    N_, mean_, M2_ = sharding.merge_welford([(v[0], v[1:1 + m], v[1 + m:]) for v in sh])
    assert N_ == world * n and np.array_equal(Wh[1], 1.0 / (M2_ / N_)), "device merge differs from the host merge"
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    return {
        "metric": "ABC distances/sec", "value": units * args.steps / elapsed,
        "unit": "distances/s (rows; K=3 nested each)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64",
        "data": "synthetic (device)" if args.data == "device" else "synthetic",
        "config": {"workload": "configs[3]: AdaptiveDistance round, %d samples x %d summaries %s, K=%d nested weights"
                               % (n * world if scaling == "strong" else n, m,
                                  "in total" if scaling == "strong" else "per GPU", K),
                   "samples_per_gpu": n, "summaries": m, "K": K,
                   "step": "Welford statistics of the shard + merge of the rank states on the device + K nested "
                           "distances of every row with the top-%d by the last column fused in" % k,
                   "exchange": ("every step: all_gather of %d doubles per rank + gather of the packed top-%d pairs "
                                "(16 KB per rank) to rank 0" % (ns, k)) if world > 1 else "none (N = 1)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "dist_multiw_pipe_kernel",
                     "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes,
                     "step_rate_frac": ((8.0 * m * 2 + 8.0 * K) * n * args.steps / elapsed / 1e9) / HBM_PEAK_GBS},
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        world = dist.get_world_size()

    job = Job(dev, local_rank, world, rank)
    result = (run_adaptive if args.workload == "adaptive" else run_distance)(args, job)
    if rank == 0:
        env = {"world_size_seen": world, "device": torch.cuda.get_device_name(dev)}
        try:
            env["rccl"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            pass
        result["env"] = env
        if world == 1 and not args.no_cpu_baseline and args.workload == "distance":
            result["cpu_baseline"] = cpu_baseline_distance(result["config"]["samples_per_gpu"],
                                                           result["config"]["summaries"])
            result["cpu_baseline"]["host_cores_available"] = os.cpu_count()
        if world == 1 and not args.no_bolfi and args.workload == "distance":
            from elfi_amd import bolfi_bench
            torch.cuda.synchronize(dev)
            b = bolfi_bench.run(iters=args.bolfi_iters)
            if not args.no_cpu_baseline:
                b["cpu_baseline"] = cpu_baseline_bolfi(4096, 10, 10)
                b["speedup_vs_cpu"] = b["value"] / b["cpu_baseline"]["value"]
                result["cpu_baseline"]["bolfi"] = b["cpu_baseline"]
            result["bolfi"] = b
            # the driver's parsed record keeps `roofline` and `cpu_baseline`: BASELINE.json's first metric and its
            # per-phase rooflines ride in them as well
            result["roofline"]["phases"] = b["roofline_phases"]
            result["roofline"]["bolfi"] = {"metric": b["metric"], "value": b["value"], "unit": b["unit"],
                                           "ms_fit": b["ms_fit"], "ms_acquire": b["ms_acquire"],
                                           "fit_only_frac": b["roofline"]["fit_only_frac"],
                                           "iter_frac_executed": b["roofline"]["frac"],
                                           "iterations_timed": b["iterations_timed"],
                                           "fit_large": b["fit_large"]}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
