"""Developer soak test: the factorisation is bit-reproducible -- the same evidence factored many times gives the same
log marginal, alpha and (sampled) factor entries every time, for the fused schedule in both forms.  A lost hand-off
inside the diagonal-block kernel (LDS flag, barriers) or the chained step launch would show up as a difference.
usage: python scripts/soak_fit.py [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from benchlib.bolfi_bench import problem, heuristic_hyper
from elfi_amd.gp import GPHandle

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t0 = time.time()
for n, d in ((130, 2), (384, 3), (1024, 2), (2100, 10), (4096, 10)):
    X, y, bounds = problem(n, d)
    h = heuristic_hyper(bounds, y)
    for sched in (2, 3):
        gp = GPHandle(d, n)
        gp.set_schedule(sched, 0)
        gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
        gp.set_data(X, y)
        lz0 = gp.factorize()
        a0 = gp.get(2).copy()
        L0 = gp.get(0).copy()
        W0 = gp.get(1).copy()
        bad = 0
        for r in range(reps):
            lz = gp.factorize()
            a = gp.get(2)
            if lz != lz0 or not np.array_equal(a, a0):
                bad += 1
            if r % 100 == 99:
                if not (np.array_equal(gp.get(0), L0) and np.array_equal(gp.get(1), W0)):
                    bad += 1
        print("n %5d d %2d schedule %d: %d rebuilds, %d differences (%.1f s)" % (n, d, sched, reps, bad, time.time() - t0),
              flush=True)
        assert bad == 0
        gp.close()
print("soak OK")
