"""Developer tool: three rebuilds with the overlapped sweep (elfihip_gp_set_schedule(gp, 5, 0)), for profiler runs.
usage: python scripts/ov_once.py n d"""
import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
from benchlib.bolfi_bench import problem, heuristic_hyper
from elfi_amd.gp import GPHandle
n, d = int(sys.argv[1]), int(sys.argv[2])
X, y, bounds = problem(n, d)
h = heuristic_hyper(bounds, y)
gp = GPHandle(d, n)
gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
gp.set_data(X, y)
gp.set_schedule(5, 0)
for i in range(3):
    gp.factorize()
