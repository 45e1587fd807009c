"""Developer tool: `reps` GP rebuilds with one sweep schedule (for rocprofv3 runs).
usage: python scripts/fit_once.py n d schedule [reps] [group]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib.bolfi_bench import problem, heuristic_hyper
from elfi_amd.gp import GPHandle

n, d, sched = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
group = int(sys.argv[5]) if len(sys.argv) > 5 else 0
X, y, bounds = problem(n, d)
h = heuristic_hyper(bounds, y)
gp = GPHandle(d, n)
gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
gp.set_data(X, y)
gp.set_schedule(sched, group)
import numpy as np


def fact():
    try:
        gp.factorize()
    except np.linalg.LinAlgError:      # timing experiments with deliberately wrong operands
        pass


fact()
t0 = time.perf_counter()
for _ in range(reps):
    fact()
print("n=%d d=%d schedule=%d: %.3f ms per rebuild" % (n, d, sched, (time.perf_counter() - t0) / reps * 1e3))
