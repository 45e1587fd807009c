"""Developer tool: one lock-step evaluation of the acquisition search (LCB value + gradient at S points, one call,
host round trip included) and a whole lcb_minimize call.  usage: python scripts/time_step.py [n d S]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from benchlib.bolfi_bench import problem, heuristic_hyper
from elfi_amd.gp import GPHandle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 10
S = int(sys.argv[3]) if len(sys.argv) > 3 else 10
X, y, bounds = problem(n, d)
h = heuristic_hyper(bounds, y)
gp = GPHandle(d, n)
gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
gp.set_data(X, y)
gp.factorize()
xs = np.random.RandomState(2).uniform(-2, 2, (S, d))
beta = 50.0
gp.lcb(xs, beta)
for with_grad in (True, False):
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(50):
            gp.lcb(xs, beta, with_grad=with_grad)
        best = min(best, (time.perf_counter() - t0) / 50)
    print("n=%d d=%d S=%d  lcb %s: %.1f us per call" % (n, d, S, "value+grad" if with_grad else "value only", best * 1e6))
starts = np.random.RandomState(3).uniform(-2, 2, (S, d))
ts = []
for r in range(10):
    t0 = time.perf_counter()
    locs, vals, iters, ne = gp.lcb_minimize(starts + 0.01 * r, bounds, beta)
    ts.append(time.perf_counter() - t0)
print("lcb_minimize: mean %.3f ms  min %.3f ms   (%d evaluations, max %d iterations in the last call)" %
      (np.mean(ts) * 1e3, np.min(ts) * 1e3, ne, iters.max()))
