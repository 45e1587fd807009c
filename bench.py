#!/usr/bin/env python
"""bench.py -- throughput of the ELFI hot path on MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Headline line (one JSON object on stdout, rank 0): BASELINE.json's metric
"ABC distances/sec" on configs[1] -- a synthetic Gaussian summary matrix of 10^6
samples x 32 summaries per GPU, euclidean distance to the observed summaries
(elfi.Distance('euclidean'), elfi/model/elfi_model.py:1037).  A step is one pass of
the distance path over the rank's batch, inputs already resident in HBM.  Weak
scaling: every rank owns an independent batch (independent ABC batches,
elfi/methods/parameter_inference.py:283-292); the only exchange, inside the timed
region, is a device top-k per rank and one RCCL gather of the k best (distance, row) pairs to rank 0.

The same line carries
  "roofline"      HBM roofline of the distance kernel (HIP events on the library's stream)
  "cpu_baseline"  the oracle (column_stack + SciPy cdist, what the reference executes)
                  timed on this box's host cores for the same batch shape
  "bolfi"         BASELINE.json's second metric, BOLFI iters/sec (GP fit + acquisition,
                  n=4096, d=10) with its own MFMA-fp64 roofline (rank 0, N=1 only)
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
FP64_MFMA_PEAK_TFLOPS = 78.6  # AMD public spec, FP64 matrix (not in the local guide)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=10 ** 6, help="samples per GPU per step")
    ap.add_argument("--m", type=int, default=32, help="summaries per sample")
    ap.add_argument("--workload", choices=["distance", "adaptive"], default="distance",
                    help="distance: configs[1] (default, the headline); adaptive: configs[3] shape per GPU "
                         "(1.25e6 x 64, AdaptiveDistance with K=3 nested weights, Welford merge + gather)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bolfi", action="store_true")
    ap.add_argument("--bolfi-iters", type=int, default=5)
    return ap.parse_args()


def cpu_baseline_distance(n, m, budget_s=12.0):
    """Reference path on the host: 32 separate summary columns -> np.column_stack -> cdist
    (elfi/model/utils.py:37-52).  Single thread (SciPy's cdist is not threaded)."""
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
    return dict(value=n / best, unit="distances/s", cores=1, kind="port",
                sample="%d x %d batch through oracle distance_as_discrepancy (np.column_stack + "
                       "scipy cdist), best of %d; cdist alone %.0f Mdist/s" % (n, m, reps, n / kern / 1e6))


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
    t0 = time.perf_counter()
    post = G.Posterior(X, y, h['var'], h['ls'], h['bias'], h['noise'])
    t_fit = time.perf_counter() - t0
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
                sample="1 GP rebuild at n=%d (%.2f s) + L-BFGS-B from %d of %d starts (%.2f s, scaled to %d)"
                       % (n, t_fit, done, S, t_acq, S),
                ms_fit=1e3 * t_fit, ms_acquire=1e3 * t_acq_full)


def run_adaptive(args, ctx, dev, world, rank):
    """configs[3] per GPU: one SMC-ABC adaptive-distance round on this rank's shard
    (elfi/model/elfi_model.py:1104-1151): Welford column statistics of the shard, all-gather +
    fixed-order Chan merge of the (count, mean, M2) triples, K = 3 nested weighted distances of
    every row, one gather of the (n, 3) distance shard to rank 0.  A step = that whole round."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from elfi_amd import sharding
    n, m, K = (args.n if args.n != 10 ** 6 else 1250000), (args.m if args.m != 32 else 64), 3
    gen = torch.Generator(device=dev)
    gen.manual_seed(100 + rank)
    X = torch.randn(n, m, dtype=torch.float64, device=dev, generator=gen) * torch.linspace(0.5, 20, m, device=dev,
                                                                                            dtype=torch.float64)
    y = torch.from_numpy(np.random.RandomState(1).randn(1, m)).to(dev)
    out = torch.empty(n, K, dtype=torch.float64, device=dev)
    state = torch.zeros(1 + 2 * m, dtype=torch.float64, device=dev)
    W = torch.ones(K, m, dtype=torch.float64, device=dev)
    gathered = [torch.empty(n, K, dtype=torch.float64, device=dev) for _ in range(world)] \
        if (world > 1 and rank == 0) else None
    states = [torch.empty_like(state) for _ in range(world)]

    def step():
        state.zero_()
        ctx.call("elfihip_welford_update_dev", X.data_ptr(), n, m, m, state.data_ptr())
        if world > 1:
            dist.all_gather(states, state)
            st = [s.cpu().numpy() for s in states]
        else:
            st = [state.cpu().numpy()]
        N, mean, M2 = sharding.merge_welford([(v[0], v[1:1 + m], v[1 + m:]) for v in st])
        w2 = 1.0 / (M2 / N)                      # (1/scale)^2, elfi_model.py:1129-1132
        W[1].copy_(torch.from_numpy(w2))
        W[2].copy_(torch.from_numpy(w2 * 0.5))   # a third, different weight vector (K = 3)
        ctx.call("elfihip_dist_multiw_dev", X.data_ptr(), n, m, m, y.data_ptr(), W.data_ptr(), K, out.data_ptr())
        if world > 1:
            dist.gather(out, gathered, dst=0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    torch.cuda.synchronize(dev)
    ctx.timer_start()
    for _ in range(args.steps):
        ctx.call("elfihip_dist_multiw_dev", X.data_ptr(), n, m, m, y.data_ptr(), W.data_ptr(), K, out.data_ptr())
    kernel_ms = ctx.timer_stop() / args.steps
    alg_bytes = (8.0 * m + 8.0 * K) * n
    if rank != 0:
        return None
    import distance_oracle as O
    idx = np.arange(0, n, max(1, n // 2048))[:2048]
    Wh = W.cpu().numpy()
    ref = np.column_stack([O.cdist_rows(X[idx].cpu().numpy(), y.cpu().numpy(), 'euclidean', w=Wh[k]) for k in range(K)])
    assert np.array_equal(out[idx].cpu().numpy(), ref), "adaptive bench output differs from the oracle"
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    return {
        "metric": "ABC distances/sec", "value": world * n * args.steps / elapsed, "unit": "distances/s (rows; K=3 nested each)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[3] shape per GPU: AdaptiveDistance round, %d samples x %d summaries, K=%d nested "
                               "weights, Welford + all-gather/merge + gather" % (n, m, K),
                   "samples_per_gpu": n, "summaries": m, "K": K,
                   "exchange": "all_gather of (1+2m) doubles + gather of the (n,K) shard" if world > 1 else "none"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "dist_multiw_pipe_kernel",
                     "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes},
    }


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import elfi_amd
    from elfi_amd import _lib
    ctx = elfi_amd.Context(local_rank)
    # one explicit (non-null) stream shared by torch/RCCL and the library, so the gather is
    # ordered after the last kernel without a host sync
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)

    if args.workload == "adaptive":
        result = run_adaptive(args, ctx, dev, world, rank)
        if rank == 0:
            print(json.dumps(result), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    n, m = args.n, args.m
    # Synthetic Gaussian simulator outputs: N(0,1) summaries,
    # one independent batch per rank (seeded by rank), observed row from a fixed seed.
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    # NBUF independent batches, visited round-robin: the working set (NBUF x 8nm bytes) exceeds the
    # 256 MiB Infinity Cache, so every step streams its batch from HBM instead of re-hitting the MALL
    NBUF = 3
    Xs = [torch.randn(n, m, dtype=torch.float64, device=dev, generator=gen) for _ in range(NBUF)]
    y = torch.from_numpy(np.random.RandomState(1).randn(1, m)).to(dev)
    out = torch.empty(n, dtype=torch.float64, device=dev)
    # The one exchange of a multi-GPU job: every rank selects its K_BEST smallest
    # distances on the GPU (elfihip_topk_smallest_dev, what Rejection keeps of a batch) and rank 0
    # gathers those (value, row) pairs -- 16 KB per rank instead of the 8 MB distance shard.
    K_BEST = min(1000, n)
    # values and row numbers travel together: one buffer (the int64 rows viewed through the second half), ONE gather
    best = torch.empty(2 * K_BEST, dtype=torch.float64, device=dev)
    best_v = best[:K_BEST]
    best_i = best[K_BEST:].view(torch.int64)
    gath = [torch.empty_like(best) for _ in range(world)] if (world > 1 and rank == 0) else None

    def exchange():
        ctx.call("elfihip_topk_smallest_dev", out.data_ptr(), n, 1, K_BEST, best_v.data_ptr(), best_i.data_ptr())
        dist.gather(best, gath, dst=0)

    counter = [0]

    def step():
        X = Xs[counter[0] % NBUF]
        counter[0] += 1
        ctx.call("elfihip_dist_rows_dev", 0, X.data_ptr(), n, m, m, y.data_ptr(), None,
                 2.0, out.data_ptr())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    if world > 1:  # warm the exchange path too
        exchange()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        exchange()  # the one exchange: each rank's best K_BEST (distance, row) pairs -> rank 0
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Kernel-only timing for the roofline: HIP events on the SAME stream the kernel runs on.
    torch.cuda.synchronize(dev)
    ctx.timer_start()
    for _ in range(args.steps):
        step()
    kernel_ms = ctx.timer_stop() / args.steps
    alg_bytes = (8.0 * m + 8.0) * n          # 8m + 8 bytes per distance
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9

    # parity guard on what was just timed (checker only)
    if rank == 0:
        import distance_oracle as O
        idx = np.arange(0, n, max(1, n // 4096))[:4096]
        X = Xs[(counter[0] - 1) % NBUF]   # the batch of the last step
        ref = O.cdist_rows(X[idx].cpu().numpy(), y.cpu().numpy(), 'euclidean')
        assert np.array_equal(out[idx].cpu().numpy(), ref), "bench output differs from the oracle"
        # ... and on the selection used by the multi-GPU exchange (exercised here at every N)
        ctx.call("elfihip_topk_smallest_dev", out.data_ptr(), n, 1, K_BEST, best_v.data_ptr(), best_i.data_ptr())
        torch.cuda.synchronize(dev)
        dh = out.cpu().numpy()
        assert np.array_equal(np.sort(best_v.cpu().numpy()), np.sort(dh)[:K_BEST]), "top-k differs from numpy"
        assert np.array_equal(dh[best_i.cpu().numpy()], best_v.cpu().numpy())

    # HBM traffic per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 +
    # WRITE_SIZE, profiles/r01_distance_pmc.md); PMC cannot be sampled from inside this process
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "distance_pmc.json")) as f:
            pmc = json.load(f)
        if pmc["n"] == n and pmc["m"] == m:
            traffic = pmc["traffic_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass

    result = None
    if rank == 0:
        value = world * n * args.steps / elapsed
        result = {
            "metric": "ABC distances/sec", "value": value, "unit": "distances/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic Gaussian summaries, %d samples x %d summaries per "
                                   "GPU per step, elfi.Distance('euclidean'), inputs resident in HBM" % (n, m),
                       "samples_per_gpu": n, "summaries": m, "layout": "row-major (n,m) f64",
                       "batches_in_rotation": NBUF,
                       "exchange": "per job: device top-%d of the last batch per rank + ONE RCCL gather of the packed "
                                   "(distance, row) pairs to rank 0" % K_BEST if world > 1 else "none"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "dist_rows_pipe_kernel<euclidean>", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "frac_of_measured_copy_peak_6290": achieved / 6290.0},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline_distance(n, m)
            result["cpu_baseline"]["host_cores_available"] = os.cpu_count()
        if world == 1 and not args.no_bolfi:
            from elfi_amd import bolfi_bench
            b = bolfi_bench.run(iters=args.bolfi_iters)
            if not args.no_cpu_baseline:
                b["cpu_baseline"] = cpu_baseline_bolfi(4096, 10, 10)
                b["speedup_vs_cpu"] = b["value"] / b["cpu_baseline"]["value"]
            result["bolfi"] = b
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
