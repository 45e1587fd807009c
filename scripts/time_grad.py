"""Developer tool: the log-marginal gradient kernel alone (device time from the library's phase timers).
usage: python scripts/time_grad.py [n d ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from benchlib.bolfi_bench import problem, heuristic_hyper
from elfi_amd.gp import GPHandle

shapes = [(int(a.split(':')[0]), int(a.split(':')[1])) for a in sys.argv[1:]] or [(1024, 2), (2048, 10), (4096, 10), (8192, 20)]
for n, d in shapes:
    X, y, bounds = problem(n, d)
    h = heuristic_hyper(bounds, y)
    gp = GPHandle(d, n)
    gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
    gp.set_data(X, y)
    gp.factorize()
    gp.nlml_grad()
    gp.profile(1)
    for _ in range(5):
        lz, g = gp.nlml_grad()
    prof = gp.profile(0)
    ms, calls = prof['kinv_grad']
    fl = n ** 3 / 3.0
    print("n=%5d d=%2d  gradient kernel %.3f ms  %.1f TFLOP/s (%.2f of 78.6)   grad %s" % (n, d, ms / calls, fl / (ms / calls * 1e-3) / 1e12,
          fl / (ms / calls * 1e-3) / 1e12 / 78.6, np.array2string(g, precision=6)))
    gp.close()
