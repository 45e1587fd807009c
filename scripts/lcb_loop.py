"""Repeated LCB value+gradient evaluations at fixed n (profiling target, developer tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import numpy as np
import gp_oracle as G
from elfi_amd.gp import GPHandle

n, d, S = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096), 10, (int(sys.argv[2]) if len(sys.argv) > 2 else 10)
X, y, bounds = G.synthetic_gp_problem(n, d)
h = G.default_hyper(bounds, y)
gp = GPHandle(d, n)
gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
gp.set_data(X, y)
gp.factorize()
xs = np.random.RandomState(2).uniform(-2, 2, (S, d))
for _ in range(200):
    gp.lcb(xs, 3.0)
