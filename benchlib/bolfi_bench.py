"""BASELINE.json's first metric: BOLFI iterations per second (GP fit + acquisition, n=4096, d=10).

Workload (SURVEY.md section 8d "G-1"):
    X = RandomState(0).uniform(-2, 2, (4096, 10)),  y = |X - 0.5|_2 + 0.1 RandomState(1).randn(4096)
    hyper-parameters fixed at the heuristic values of gpy_regression.py:255,260-264
    LCBSC, exploration_rate = 10, S = 10 start points, L-BFGS run to convergence (maxiter 1000)
One iteration is what elfi.BOLFI does per acquired point with batch_size = 1
(elfi/methods/inference/bolfi.py:201-254): `target_model.update(x, y)` -- append one evidence
point and REBUILD the whole GP, as GPyRegression.update does (gpy_regression.py:304-312) --
followed by `acquisition_method.acquire(1, t)`.  It is driven through the same two objects a
user hands to elfi.BOLFI (HipGPRegression, HipLCBSC), so Python and ctypes overhead are inside
the timed region.

Flop model per iteration (FMA = 2 flops): Gram 2 n^2 d; factorisation n^3/3
(Cholesky) + n^3/3 (L^-T, which replaces GPy's dpotri + the per-point triangular solves);
acquisition E point-evaluations of value+gradient at 2 n^2 + 6 n d + 4 n each.  Reported against
the FP64 matrix peak (78.6 TFLOP/s) in two ways: `achieved_exec` counts the flops actually
executed (both n^3/3 terms), `achieved_model` SURVEY's reference count (n^3/3 only).
"""
import time

import numpy as np

FP64_MFMA_PEAK_TFLOPS = 78.6


def problem(n, d):
    X = np.random.RandomState(0).uniform(-2, 2, (n, d))
    y = np.linalg.norm(X - 0.5, axis=1) + 0.1 * np.random.RandomState(1).randn(n)
    return X, y.reshape(-1, 1), [(-2., 2.)] * d


def heuristic_hyper(bounds, y):
    ls = (np.max(bounds) - np.min(bounds)) / 3.
    var = (np.max(y) / 3.) ** 2
    return dict(var=float(var), ls=float(ls), bias=float(var / 4.), noise=float(np.max(y) ** 2 / 100.))


def _loop(gp, acq, X, y, n0, iters, warm):
    t_fit, t_acq, evals, steps = [], [], [], []
    x_next = None
    for i in range(warm + iters):
        k = n0 + i
        xi = X[k:k + 1] if x_next is None else x_next
        yi = np.linalg.norm(xi - 0.5, axis=1).reshape(1, 1)   # the synthetic discrepancy at the acquired point
        t0 = time.perf_counter()
        gp.update(xi, yi)
        t1 = time.perf_counter()
        x_next = acq.acquire(1, t=k)
        t2 = time.perf_counter()
        if i >= warm:
            t_fit.append(t1 - t0)
            t_acq.append(t2 - t1)
            evals.append(acq.last_opt['n_eval'])
            steps.append(int(np.max(acq.last_opt['iters'])))
    return float(np.mean(t_fit)), float(np.mean(t_acq)), float(np.mean(evals)), int(np.max(steps))


HBM_PEAK_GBS = 8000.0


def _phase_rooflines(prof, n, d):
    """Per-phase rooflines from the library's own HIP-event phase timers
    (elfihip_gp_profile): each phase against the bound that applies to it."""
    out = {}

    def per_call(k):
        ms, calls = prof[k]
        return ms / calls if calls else None
    g, sw, al = per_call('gram'), per_call('sweep'), per_call('alpha')
    if g:
        fl = 2.0 * n * n * d
        out['gram'] = {"bound": "mfma", "ms": g, "flops": fl, "achieved": fl / g / 1e9, "unit": "TFLOP/s",
                       "peak": FP64_MFMA_PEAK_TFLOPS, "frac": fl / g / 1e9 / FP64_MFMA_PEAK_TFLOPS,
                       "note": "2 n^2 d on the matrix cores + n^2/2 exp on the vector ALUs (the exp bounds it)"}
    if sw:
        fl = 2.0 * n ** 3 / 3.0
        out['sweep'] = {"bound": "mfma", "ms": sw, "flops": fl, "achieved": fl / sw / 1e9, "unit": "TFLOP/s",
                        "peak": FP64_MFMA_PEAK_TFLOPS, "frac": fl / sw / 1e9 / FP64_MFMA_PEAK_TFLOPS,
                        "note": "Cholesky n^3/3 + L^-T n^3/3 in one sweep"}
    if al:
        by = 8.0 * n * n / 2
        out['alpha_logdet'] = {"bound": "hbm", "ms": al, "bytes": by, "achieved": by / al / 1e6, "unit": "GB/s",
                               "peak": HBM_PEAK_GBS, "frac": by / al / 1e6 / HBM_PEAK_GBS}
    t1, t2 = per_call('tri_first'), per_call('tri_second')
    ks, gf = per_call('kstar'), per_call('grad_finish')
    if t1 and t2:
        by = 8.0 * n * n / 2     # one triangle of n^2/2 doubles streamed once per product
        out['predict'] = {"bound": "hbm", "bytes_per_step": 2 * by, "ms_first_product": t1, "ms_second_product": t2,
                          "ms_kernel_row": ks, "ms_gradient_and_assembly": gf,
                          "ms_per_step_device": t1 + t2 + (ks or 0) + (gf or 0),
                          "achieved": 2 * by / (t1 + t2) / 1e6, "unit": "GB/s", "peak": HBM_PEAK_GBS,
                          "frac": 2 * by / (t1 + t2) / 1e6 / HBM_PEAK_GBS,
                          "frac_whole_step": 2 * by / (t1 + t2 + (ks or 0) + (gf or 0)) / 1e6 / HBM_PEAK_GBS,
                          "steps_timed": prof['tri_second'][1],
                          "note": "two triangular products with 16 right-hand sides: 8 n^2 bytes per lock-step of the "
                                  "acquisition search (S <= 16 points: HBM/L2-bound, SURVEY.md 8d)"}
    kg = per_call('kinv_grad')
    if kg:
        fl = n ** 3 / 3.0
        out['kinv_grad'] = {"bound": "mfma", "ms": kg, "flops": fl, "achieved": fl / kg / 1e9, "unit": "TFLOP/s",
                            "peak": FP64_MFMA_PEAK_TFLOPS, "frac": fl / kg / 1e9 / FP64_MFMA_PEAK_TFLOPS,
                            "note": "K^-1 = L^-T L^-1 tiles with the dlogZ/dtheta contractions fused (one objective "
                                    "gradient of GPyRegression.optimize)"}
    return out


def run(n=4096, d=10, S=10, iters=50, warm=3):
    """Both update modes on the same workload: 'refactor' = full GP rebuild per update (what
    GPyRegression.update does), 'incremental' = bordering (elfihip_gp_extend), the product default
    between hyper-parameter changes.  The headline `value` is the refactor mode: it is the
    like-for-like of the reference's per-iteration work and the one with an MFMA roofline."""
    from elfi_amd.gp import HipGPRegression
    from elfi_amd.lcb_acquisition import HipLCBSC
    X, y, bounds = problem(n, d)
    names = ['t%d' % i for i in range(d)]
    n0 = n - (iters + warm)
    res = {}
    for mode in ('refactor', 'incremental'):
        gp = HipGPRegression(names, bounds=dict(zip(names, bounds)))
        gp.incremental_limit = 0 if mode == 'refactor' else 64
        gp.update(X[:n0], y[:n0])
        gp.fix_hyperparameters(**heuristic_hyper(bounds, y))
        acq = HipLCBSC(gp, n_inits=S, exploration_rate=10, seed=2)
        res[mode] = _loop(gp, acq, X, y, n0, iters, warm)
        if mode == 'refactor':
            # a second, short pass with the library's phase timers on (events cost a little, so not in the timed loop)
            gp.device_handle.profile(1)
            for i in range(6):
                gp.refit()
                acq.acquire(1, t=n0 + i)
            gp.device_handle.nlml_grad()
            gp.device_handle.nlml_grad()
            phases = _phase_rooflines(gp.device_handle.profile(0), gp.n_evidence, d)
        del gp, acq
    fit, ac, E, steps = res['refactor']
    it_s = 1.0 / (fit + ac)
    # evidence counts of the timed iterations: n - iters + 1 .. n (one point is appended per iteration); flops are
    # the means over exactly these sizes
    sizes = np.arange(n - iters + 1, n + 1, dtype=np.float64)
    fl_gram = float(np.mean(2.0 * sizes * sizes * d))
    fl_chol = float(np.mean(sizes ** 3 / 3.0))
    fl_eval = float(np.mean(E * (2.0 * sizes * sizes + 6.0 * sizes * d + 4.0 * sizes)))
    exec_flops = fl_gram + 2 * fl_chol + fl_eval
    model_flops = fl_gram + fl_chol + fl_eval
    ifit, iac, iE, isteps = res['incremental']
    out = {
        "metric": "BOLFI iters/sec (GP fit+acq, n=%d d=%d)" % (n, d), "value": it_s, "unit": "iters/s",
        "evidence_sizes_timed": [int(sizes[0]), int(sizes[-1])],
        "mode": "refactor (full GP rebuild per update, as GPyRegression.update)",
        "ms_fit": 1e3 * fit, "ms_acquire": 1e3 * ac, "starts": S,
        "point_evaluations_per_acquire": E, "max_lbfgs_iterations": steps,
        "flops_per_iter_executed": exec_flops, "flops_per_iter_model": model_flops, "iterations_timed": iters,
        "roofline_phases": phases,
        "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": FP64_MFMA_PEAK_TFLOPS,
                     "achieved": exec_flops * it_s / 1e12, "frac": exec_flops * it_s / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                     "achieved_model": model_flops * it_s / 1e12,
                     "fit_only_achieved": (fl_gram + 2 * fl_chol) / fit / 1e12,
                     "fit_only_frac": (fl_gram + 2 * fl_chol) / fit / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                     "note": "acquisition phase (S=%d columns) is HBM/L2-bound, not MFMA-bound "
                             "(SURVEY.md 8d roofline caveat)" % S},
        "incremental": {"value": 1.0 / (ifit + iac), "unit": "iters/s", "ms_fit": 1e3 * ifit,
                        "ms_acquire": 1e3 * iac, "point_evaluations_per_acquire": iE,
                        "mode": "bordering update (elfihip_gp_extend): two passes over L^-T per new point, "
                                "HBM-bound: %.0f MB per update" % (2 * 8.0 * n * n / 2 / 1e6)},
    }
    out["lockstep_us"] = lockstep_leg(n, d, S)
    out["fit_large"] = fit_only(8192, 20)
    # the sizes a real run's hyper-parameter searches rebuild at (configs[2]: 512 ... 4096): chain-bound, not MFMA-bound
    out["fit_small"] = {str(m): fit_only(m, d) for m in (1024, 2048)}
    out["cfg5"] = cfg5_leg()
    return out


def lockstep_leg(n=4096, d=10, S=10, reps=300):
    """One acquisition lock-step (value and gradient of the LCB at S points; host wall per elfihip_gp_lcb call, us): the
    two triangular products (four launches -- what a GP that is refactorised for every acquisition runs) against ONE
    product with K^-1 (three launches -- what a GP that is extended point by point switches to after 64 lock-steps)."""
    from elfi_amd.gp import GPHandle
    X, y, bounds = problem(n, d)
    h = heuristic_hyper(bounds, y)
    gp = GPHandle(d, n)
    gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
    gp.set_data(X, y)
    gp.factorize()
    xs = np.random.RandomState(2).uniform(-2, 2, (S, d))
    out = {}
    for name, form in (("triangular_products", 2), ("kinv_product", 3)):
        gp.set_lockstep_form(form)
        for _ in range(20):
            gp.lcb(xs, 3.0)
        t0 = time.perf_counter()
        for _ in range(reps):
            gp.lcb(xs, 3.0)
        out[name] = (time.perf_counter() - t0) / reps * 1e6
    out["kinv_in_use"], _, out["cond_estimate"] = gp.lockstep_info()
    gp.close()
    return out


def cfg5_leg(n=8192, d=20, S=256, refit_every=64, reps=3):
    """BASELINE configs[4] on ONE GPU: d = 20, n_evidence = 8192, 256 acquisition starts (all on this GPU: the 8-GPU form
    deals them round-robin, 32 per rank, lcb_acquisition.py), the GP refit every 64 acquisitions.  Reported: one rebuild,
    one acquisition with all 256 starts (16 lock-step passes of 16 columns per evaluation round), and the amortised
    iteration (acquisition + rebuild / 64) -- between refits a new evidence point costs one bordering update."""
    from elfi_amd.gp import HipGPRegression
    from elfi_amd.lcb_acquisition import HipLCBSC
    X, y, bounds = problem(n, d)
    names = ['t%d' % i for i in range(d)]
    gp = HipGPRegression(names, bounds=dict(zip(names, bounds)))
    gp.update(X, y)
    gp.fix_hyperparameters(**heuristic_hyper(bounds, y))
    t0 = time.perf_counter()
    for _ in range(reps):
        gp.refit()
    t_fit = (time.perf_counter() - t0) / reps
    acq = HipLCBSC(gp, n_inits=S, exploration_rate=10, seed=3)
    acq.acquire(1, t=n)
    t_acq, evals, steps = [], [], []
    for r in range(reps):
        t0 = time.perf_counter()
        acq.acquire(1, t=n + r)
        t_acq.append(time.perf_counter() - t0)
        evals.append(acq.last_opt['n_eval'])
        steps.append(int(np.max(acq.last_opt['iters'])))
    ta = float(np.median(t_acq))   # (the median: one acquisition that meets a busy host core would carry the mean)
    E = float(np.mean(evals))
    # the two triangular products stream the factor once per 128 points (8 passes share a launch): bytes per round
    rounds = float(np.max(steps))
    # S = 256 right-hand sides are 64 flop per byte of the factor -- the acquisition of this config is bound
    # by the FP64 matrix pipes, 2 n^2 + 6 n d + 4 n flops per point evaluation
    fl = E * (2.0 * n * n + 6.0 * n * d + 4.0 * n)
    return {"n": n, "d": d, "starts": S, "refit_every": refit_every, "ms_fit": 1e3 * t_fit, "ms_acquire": 1e3 * ta,
            "ms_acquire_each": [1e3 * x for x in t_acq],
            "point_evaluations_per_acquire": E, "max_lbfgs_iterations": int(np.max(steps)),
            "acquire_flops": fl, "acquire_tflops": fl / ta / 1e12, "acquire_frac_mfma": fl / ta / 1e12 / FP64_MFMA_PEAK_TFLOPS,
            "iters_per_s_amortised": 1.0 / (ta + t_fit / refit_every),
            "note": "one GPU holds all %d starts; sharded over 8 GPUs every rank searches %d of them" % (S, S // 8)}


def fit_only(n, d, reps=3):
    """GP rebuild alone at a larger shape (configs[4]: n_evidence = 8192, d = 20), where the trailing
    update dominates the sweep: executed flops 2 n^2 d + 2 n^3 / 3 against the FP64-matrix peak."""
    from elfi_amd.gp import GPHandle
    X, y, bounds = problem(n, d)
    h = heuristic_hyper(bounds, y)
    gp = GPHandle(d, n)
    gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
    gp.set_data(X, y)
    gp.factorize()
    t0 = time.perf_counter()
    for _ in range(reps):
        gp.factorize()
    t = (time.perf_counter() - t0) / reps
    flops = 2.0 * n * n * d + 2.0 * n ** 3 / 3.0
    gp.close()
    return {"n": n, "d": d, "ms_fit": 1e3 * t, "flops_executed": flops, "achieved": flops / t / 1e12,
            "unit": "TFLOP/s", "peak": FP64_MFMA_PEAK_TFLOPS, "frac": flops / t / 1e12 / FP64_MFMA_PEAK_TFLOPS}
