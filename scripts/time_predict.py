"""Batched-prediction timing against the number of query points (developer tool)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import numpy as np
import gp_oracle as G
from elfi_amd.gp import GPHandle

n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 10
X, y, bounds = G.synthetic_gp_problem(n, d)
h = G.default_hyper(bounds, y)
gp = GPHandle(d, n)
gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
gp.set_data(X, y)
gp.factorize()
for S in (10, 16, 32, 64, 128, 256, 1000, 4096):
    xs = np.random.RandomState(S).uniform(-2, 2, (S, d))
    v, g = gp.lcb(xs, 3.0)
    R = 10 if S <= 256 else 3
    t0 = time.perf_counter()
    for _ in range(R):
        gp.lcb(xs, 3.0)
    t = (time.perf_counter() - t0) / R
    t0 = time.perf_counter()
    for _ in range(R):
        gp.predict(xs)
    t2 = (time.perf_counter() - t0) / R
    print("n=%d S=%5d: value+grad %.3f ms (%.1f us/pt)   mean/var %.3f ms (%.1f us/pt)" %
          (n, S, t * 1e3, t * 1e6 / S, t2 * 1e3, t2 * 1e6 / S))
