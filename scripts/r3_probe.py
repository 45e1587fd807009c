"""Round-3 developer probe (GPU): lock-step (fused / six-launch by R3_LOCKSTEP_FORM -> elfihip_gp_set_lockstep_form), dense / streaming form of the
many-point products, the configs[4] acquisition.  python scripts/r3_probe.py [lockstep|dense|cfg5|all]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
from benchlib import bolfi_bench
from elfi_amd.gp import GPHandle

what = sys.argv[1] if len(sys.argv) > 1 else 'all'
FORM = int(os.environ.get("R3_LOCKSTEP_FORM", "0"))      # 0 fused (default), 1 six launches
out = {"lockstep_form": FORM}


def fitted(n, d):
    X, y, bounds = bolfi_bench.problem(n, d)
    h = bolfi_bench.heuristic_hyper(bounds, y)
    gp = GPHandle(d, n)
    gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
    gp.set_data(X, y)
    gp.factorize()
    gp.set_lockstep_form(FORM)
    return gp


if what in ('lockstep', 'all'):
    for n, d in ((4096, 10), (2048, 2)):
        gp = fitted(n, d)
        xs = np.random.RandomState(1).uniform(-2, 2, (10, d))
        gp.lcb(xs, 3.0)
        R = 200
        t0 = time.perf_counter()
        for _ in range(R):
            gp.lcb(xs, 3.0)
        t = (time.perf_counter() - t0) / R
        gp.profile(1)
        for _ in range(50):
            gp.lcb(xs, 3.0)
        prof = gp.profile(0)
        out["lockstep_n%d_d%d" % (n, d)] = {"us_per_call_host": 1e6 * t,
                                           "phases_us": {k: 1e3 * v[0] / max(v[1], 1) for k, v in prof.items() if v[1]}}
        gp.close()

if what in ('dense', 'all'):
    for n, d in ((4096, 10), (8192, 20)):
        gp = fitted(n, d)
        res = {}
        for S in (64, 96, 128, 160, 192, 256, 512):
            xs = np.random.RandomState(S).uniform(-2, 2, (S, d))
            row = {}
            for name, thr in (("stream", 1 << 40), ("dense", 1)):
                gp.set_dense_threshold(thr)
                gp.lcb(xs, 3.0)
                R = 5
                t0 = time.perf_counter()
                for _ in range(R):
                    gp.lcb(xs, 3.0)
                row[name + "_ms"] = 1e3 * (time.perf_counter() - t0) / R
            fl = S * (2.0 * n * n)
            row["dense_tflops_on_2n2"] = fl / (row["dense_ms"] * 1e-3) / 1e12
            res["S%d" % S] = row
        gp.set_dense_threshold(1)
        gp.profile(1)
        xs = np.random.RandomState(0).uniform(-2, 2, (256, d))
        for _ in range(5):
            gp.lcb(xs, 3.0)
        prof = gp.profile(0)
        res["dense_S256_phases_ms"] = {k: v[0] / max(v[1], 1) for k, v in prof.items() if v[1]}
        out["dense_n%d_d%d" % (n, d)] = res
        gp.close()

if what in ('tiles',):
    for n, d in ((8192, 20), (4096, 10)):
        gp = fitted(n, d)
        res = {}
        for S in (128, 192, 256):
            xs = np.random.RandomState(S).uniform(-2, 2, (S, d))
            for tm in (64, 32, 16):
                gp.set_dense_threshold(1, tm)
                gp.lcb(xs, 3.0)
                t0 = time.perf_counter()
                for _ in range(5):
                    gp.lcb(xs, 3.0)
                res["S%d_tm%d_ms" % (S, tm)] = 1e3 * (time.perf_counter() - t0) / 5
        out["tiles_n%d" % n] = res
        gp.close()

if what in ('cfg5',):
    out["cfg5"] = bolfi_bench.cfg5_leg()

if what in ('bolfi', 'all'):
    b = bolfi_bench.run(iters=30)
    out["bolfi"] = {k: b[k] for k in ("value", "ms_fit", "ms_acquire", "point_evaluations_per_acquire", "max_lbfgs_iterations")}
    out["bolfi"]["predict_phase"] = b["roofline_phases"].get("predict")
    out["bolfi"]["incremental"] = b["incremental"]

print(json.dumps(out))
