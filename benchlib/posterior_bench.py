"""Bench legs for SURVEY.md section 8f ranks 3 and 4: BOLFI posterior sampling and the MaxVar-family acquisitions.

Measured through the reference's own entry points (`elfi_amd.HipBOLFI(...).sample`, the acquisition classes' `acquire`) on
  * the documented BOLFI run (docs/usage/BOLFI.rst: MA2, 200 evidence points, the hyper-parameters it prints; evidence from
    tests/golden/bolfi_doc_run.npz) -- the run whose posterior sampling the documentation times at 55.1 s for 4 x 1000 NUTS
    iterations (BOLFI.rst:254-255, the reference on the docs' machine), and
  * a synthetic 10-parameter surrogate with 4096 evidence points (BASELINE.json's first-metric shape),
with the REFERENCE's code timed beside each on this box's host cores: `elfi.BOLFI.sample`, `MaxVar.acquire`,
`ExpIntVar.acquire` of oracle/_ref over the CPU oracle model (GPy is not installable here: the surrogate arithmetic under the
reference's classes is oracle/oracle_gp_model.py -- kind "reference loop over the CPU oracle surrogate"), on a bounded sample.
Only bench.py imports this; it needs the reference package (oracle/_ref) and is skipped without it.
"""
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOUNDS = {'t1': (-2, 2), 't2': (-1, 1)}


def _elfi():
    import ref_shim
    elfi = ref_shim.install()
    import elfi.clients.native as native
    native.set_as_default()
    return elfi


def _doc_model(elfi):
    from elfi.examples import ma2
    m = ma2.get_model(seed_obs=1)
    return m, elfi.Operation(np.log, m['d'], name='log_d')


def _doc_evidence():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'bolfi_doc_run.npz'))
    pre = {'t1': g['X'][:, 0].copy(), 't2': g['X'][:, 1].copy(), 'log_d': g['Y'][:, 0].copy()}
    hyper = dict(zip(('var', 'ls', 'bias', 'noise'), (float(v) for v in g['hyper_printed'])))
    return pre, hyper


def _hip_doc_bolfi(elfi):
    import elfi_amd
    pre, hyper = _doc_evidence()
    m, log_d = _doc_model(elfi)
    b = elfi_amd.HipBOLFI(log_d, batch_size=1, initial_evidence=pre, update_interval=10, bounds=BOUNDS, acq_noise_var=0.1,
                          seed=1)
    b.target_model.fix_hyperparameters(**hyper)
    return m, b


def _ref_doc_bolfi(elfi):
    from oracle_gp_model import OracleGPRegression
    pre, hyper = _doc_evidence()
    m, log_d = _doc_model(elfi)
    b = elfi.BOLFI(log_d, batch_size=1, initial_evidence=pre, update_interval=10, bounds=BOUNDS,
                   target_model=OracleGPRegression(['t1', 't2'], bounds=BOUNDS), acq_noise_var=0.1, seed=1)
    b.target_model.hyper = dict(hyper)
    b.target_model._refit()
    return m, b


def _silently(fn, *a, **k):
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def sample_leg(cpu=True, cpu_iters=120):
    """`BOLFI.sample(1000)` on the documented run: 4 chains x 1000 NUTS iterations."""
    from elfi_amd import chains
    elfi = _elfi()
    _, b = _hip_doc_bolfi(elfi)
    _silently(b.sample, 100, n_evidence=200)                      # warm-up: library load, workspaces, plans
    t0 = time.perf_counter()
    res = _silently(b.sample, 1000, n_evidence=200)
    wall = time.perf_counter() - t0
    # (the counters are those of the LAST run_lockstep call = the timed run; rounds 1-5 subtracted the warm-up run's counts
    # from them and so overstated the time per round by ~10 %)
    rounds, points = chains.run_lockstep.n_rounds, chains.run_lockstep.n_points
    out = {"config": "docs/usage/BOLFI.rst run: MA2, 200 evidence points, printed hyper-parameters; "
                     "HipBOLFI.sample(1000): 4 chains x 1000 NUTS iterations (500 warm-up), automatic threshold",
           "wall_s": wall, "chain_iterations_per_s": 4 * 1000 / wall, "lockstep_rounds": rounds,
           "point_evaluations": points, "us_per_round": 1e6 * wall / max(rounds, 1),
           "sample_means": {k: float(v) for k, v in res.sample_means.items()}, "threshold": float(res.threshold),
           "documented": {"wall_s": 55.1, "sample_means": {"t1": 0.429, "t2": 0.0277}, "threshold": -1.6146,
                          "where": "docs/usage/BOLFI.rst:236,254-255,293 (the reference with GPy on the docs' machine)"}}
    if cpu:
        _, rb = _ref_doc_bolfi(elfi)
        t0 = time.perf_counter()
        _silently(rb.sample, cpu_iters, n_chains=4, n_evidence=200)
        dt = time.perf_counter() - t0
        out["cpu_reference"] = {"kind": "reference loop (elfi.BOLFI.sample: one mcmc.nuts call per chain, bolfi.py:543-566) "
                                        "over the CPU oracle surrogate",
                                "sample": "4 chains x %d NUTS iterations (half of them warm-up, as in the timed run)" % cpu_iters,
                                "wall_s": dt, "cores": 1, "chain_iterations_per_s": 4 * cpu_iters / dt,
                                "scaled_to_4x1000_s": dt * 1000 / cpu_iters,
                                "scaled_is": "an EXTRAPOLATION of the measured rate to 4 x 1000 iterations, not a measurement"}
        out["speedup_vs_cpu_reference"] = out["chain_iterations_per_s"] / out["cpu_reference"]["chain_iterations_per_s"]
    return out


def sample_large_leg(n=4096, d=10, n_samples=300, n_chains=4, cpu=True):
    """The lock-step round at BASELINE's first-metric shape: every round is ONE batched value + gradient evaluation of all
    live chains on a 4096-point, 10-parameter surrogate."""
    import scipy.stats as ss
    import elfi_amd
    from elfi_amd import chains
    from benchlib.bolfi_bench import heuristic_hyper, problem
    elfi = _elfi()
    X, y, bounds = problem(n, d)
    names = ['p%02d' % i for i in range(d)]
    mdl = elfi.new_model()
    pri = [elfi.Prior(ss.uniform, -2, 4, model=mdl, name=nm) for nm in names]
    sim = elfi.Simulator(lambda *th, batch_size=1, random_state=None: np.column_stack([np.ravel(t) for t in th]), *pri,
                         observed=np.zeros((1, d)), name='sim')
    dist = elfi.Distance('euclidean', sim, name='d')
    pre = {nm: X[:, i].copy() for i, nm in enumerate(names)}
    pre['d'] = y[:, 0].copy()
    b = elfi_amd.HipBOLFI(dist, batch_size=1, initial_evidence=pre, bounds={nm: (-2, 2) for nm in names}, seed=1)
    b.target_model.fix_hyperparameters(**heuristic_hyper(bounds, y))
    thr = float(np.min(y) + 0.3)
    _silently(b.sample, 40, n_chains=n_chains, threshold=thr, n_evidence=n)
    t0 = time.perf_counter()
    _silently(b.sample, n_samples, n_chains=n_chains, threshold=thr, n_evidence=n)
    wall = time.perf_counter() - t0
    rounds, points = chains.run_lockstep.n_rounds, chains.run_lockstep.n_points   # of the timed run (see sample_leg)
    out = {"config": "synthetic surrogate, n = %d evidence points, d = %d: HipBOLFI.sample(%d), %d chains (NUTS)"
                     % (n, d, n_samples, n_chains),
           "wall_s": wall, "lockstep_rounds": rounds, "point_evaluations": points,
           "us_per_round": 1e6 * wall / max(rounds, 1), "chain_iterations_per_s": n_chains * n_samples / wall}
    # the DEVICE share of a round (the row's number: the rest of a round is ELFI's ModelPrior -- numeric gradient of the prior
    # -- and the NUTS Python, both outside the hot path): the library's event timers around the phases of every round of a
    # shorter run
    h = b.target_model._handle
    h.profile(1)
    _silently(b.sample, 60, n_chains=n_chains, threshold=thr, n_evidence=n)
    ph = h.profile(0)
    prof_rounds = chains.run_lockstep.n_rounds
    out["device_us_per_round"] = 1e3 * sum(ms for ms, _ in ph.values()) / max(prof_rounds, 1)
    out["device_phases_us_per_round"] = {k: 1e3 * ms / max(prof_rounds, 1) for k, (ms, calls) in ph.items() if calls}
    out["device_profile_rounds"] = prof_rounds
    out["device_share_of_round"] = out["device_us_per_round"] / out["us_per_round"]
    if cpu:
        ref = out["cpu_reference"] = sample_large_cpu_reference(n, d, thr)
        # per point evaluation: what a device round costs per live chain against one surrogate call of the reference
        out["us_per_point_evaluation"] = 1e6 * wall / max(points, 1)
        out["speedup_vs_cpu_reference_per_point_evaluation"] = 1e3 * ref["ms_per_surrogate_call"] / out["us_per_point_evaluation"]
    return out


class _Budget(Exception):
    pass


def sample_large_cpu_reference(n=4096, d=10, thr=None, cpu_iters=4, cpu_chains=2, budget_s=15.0):
    """The reference's own loop (elfi.BOLFI.sample: one mcmc.nuts call per chain, bolfi.py:543-566, posteriors.py:88-189) over
    the CPU oracle surrogate at sample_large_leg's shape, bounded: cpu_chains x cpu_iters NUTS iterations or budget_s seconds
    of sampling, whichever ends first (a NUTS iteration of the reference costs hundreds of surrogate calls -- the tree and the
    numeric gradient of ModelPrior -- at ~15 ms each at n = 4096).  The surrogate's point evaluations are counted: the number
    comparable with a device lock-step round is the time per point evaluation."""
    import scipy.stats as ss
    from benchlib.bolfi_bench import heuristic_hyper, problem
    from oracle_gp_model import OracleGPRegression
    elfi = _elfi()
    X, y, bounds = problem(n, d)
    names = ['p%02d' % i for i in range(d)]
    mdl = elfi.new_model()
    pri = [elfi.Prior(ss.uniform, -2, 4, model=mdl, name=nm) for nm in names]
    sim = elfi.Simulator(lambda *th, batch_size=1, random_state=None: np.column_stack([np.ravel(t) for t in th]), *pri,
                         observed=np.zeros((1, d)), name='sim')
    dist = elfi.Distance('euclidean', sim, name='d')
    pre = {nm: X[:, i].copy() for i, nm in enumerate(names)}
    pre['d'] = y[:, 0].copy()
    if thr is None:
        thr = float(np.min(y) + 0.3)
    gp = OracleGPRegression(names, bounds={nm: (-2, 2) for nm in names})
    gp.hyper = dict(heuristic_hyper(bounds, y))      # (set before the evidence arrives: ONE factorisation at n points)
    calls = {"n": 0}
    for meth in ("predict", "predictive_gradients"):
        def counted(x, *a, _f=getattr(gp, meth), **k):
            if calls["t0"] is not None and time.perf_counter() - calls["t0"] > budget_s:
                raise _Budget()
            calls["n"] += int(np.atleast_2d(x).shape[0])
            return _f(x, *a, **k)
        setattr(gp, meth, counted)
    calls["t0"] = None
    rb = elfi.BOLFI(dist, batch_size=1, initial_evidence=pre, bounds={nm: (-2, 2) for nm in names}, target_model=gp, seed=1)
    calls["n"] = 0
    t0 = calls["t0"] = time.perf_counter()
    done = True
    try:
        _silently(rb.sample, cpu_iters, n_chains=cpu_chains, threshold=thr, n_evidence=n)
    except _Budget:
        done = False
    dt = time.perf_counter() - t0
    return {"kind": "reference loop (elfi.BOLFI.sample: one mcmc.nuts call per chain) over the CPU oracle surrogate (GPy not "
                    "installable)", "cores": 1,
            "sample": "%d chains x %d NUTS iterations at n = %d, d = %d%s"
                      % (cpu_chains, cpu_iters, n, d, "" if done else ", stopped after %.0f s of sampling" % budget_s),
            "wall_s": dt, "surrogate_calls": calls["n"], "ms_per_surrogate_call": 1e3 * dt / max(calls["n"], 1),
            "chain_iterations_per_s": cpu_chains * cpu_iters / dt if done else None}


def acquisition_family_leg(cpu=True):
    """`acquire(1)` of MaxVar and ExpIntVar (grid integration) on the documented run's surrogate: device classes against the
    reference's own classes over the CPU oracle surrogate."""
    import elfi_amd
    from elfi.model.extensions import ModelPrior
    elfi = _elfi()
    out = {"config": "docs/usage/BOLFI.rst surrogate (200 evidence points, d = 2); acquire(1), quantile_eps = 0.01; "
                     "ExpIntVar with the reference's default grid (d_grid = 0.2)"}
    m, b = _hip_doc_bolfi(elfi)
    prior = ModelPrior(m, parameter_names=['t1', 't2'])
    for key, cls, kw in (("maxvar", elfi_amd.HipMaxVar, {}), ("expintvar", elfi_amd.HipExpIntVar, {})):
        acq = cls(b.target_model, prior, quantile_eps=0.01, seed=3, **kw)
        acq.acquire(1, t=0)                                       # warm-up
        ts = []
        for i in range(7):
            t0 = time.perf_counter()
            x = acq.acquire(1, t=i)
            ts.append(time.perf_counter() - t0)
        out[key + "_acquire_ms"] = 1e3 * float(np.median(ts))     # (median of 7: one 80 ms hiccup made a mean of 5 read 16.6 ms for 0.85)
        out[key + "_point"] = [float(v) for v in np.ravel(x)]
    if cpu:
        from elfi.methods.bo.acquisition import ExpIntVar, MaxVar
        m2, rb = _ref_doc_bolfi(elfi)
        prior2 = ModelPrior(m2, parameter_names=['t1', 't2'])
        ref = {"kind": "reference classes (acquisition.py:304-470, 629-821) over the CPU oracle surrogate", "cores": 1}
        for key, cls in (("maxvar", MaxVar), ("expintvar", ExpIntVar)):
            acq = cls(model=rb.target_model, prior=prior2, quantile_eps=0.01, seed=3)
            acq.acquire(1, t=0)                                   # the same warm-up call the device classes get
            ts = []
            for i in range(5 if key == "maxvar" else 3):
                t0 = time.perf_counter()
                acq.acquire(1, t=i)
                ts.append(time.perf_counter() - t0)
            ref[key + "_acquire_ms"] = 1e3 * float(np.median(ts))
        ref["sample"] = "acquire(1) after one warm-up call, median of 5 (MaxVar) / 3 (ExpIntVar) calls"
        out["cpu_reference"] = ref
    return out


def run(cpu=True):
    out = {"bolfi_sample": sample_leg(cpu=cpu), "bolfi_sample_n4096": sample_large_leg(cpu=cpu),
           "acquisition_family": acquisition_family_leg(cpu=cpu)}
    return out
