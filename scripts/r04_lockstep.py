"""Developer tool: the acquisition lock-step (S <= 16 points) -- microseconds per call on the host and the library's phase
timers, the four-launch form (fused epilogues, form 0) against the six-launch form (form 1).  The one-product form on
K^-1 and its bound: scripts/r04_lockstep_bound.py.
usage: python scripts/r04_lockstep.py [n] [d] [S]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import gp_oracle as G  # noqa: E402
from elfi_amd.gp import GPHandle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 10
S = int(sys.argv[3]) if len(sys.argv) > 3 else 10
X, y, bounds = G.synthetic_gp_problem(n, d)
h = G.default_hyper(bounds, y)
gp = GPHandle(d, n)
gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
gp.set_data(X, y)
gp.factorize()
rs = np.random.RandomState(2)
for form in (0, 1):
    gp.set_lockstep_form(form)
    xs = rs.uniform(-2, 2, (S, d))
    for _ in range(20):
        gp.lcb(xs, 3.0)
    t0 = time.perf_counter()
    reps = 300
    for _ in range(reps):
        gp.lcb(xs, 3.0)
    us = (time.perf_counter() - t0) / reps * 1e6
    gp.profile(1)
    for _ in range(100):
        gp.lcb(xs, 3.0)
    ph = gp.profile(0)
    print("n=%d d=%d S=%d form %d (%s): %.1f us per call (host); device phases us: %s"
          % (n, d, S, form, "four launches" if form == 0 else "six launches", us,
             {k: round(1e3 * v[0] / max(1, v[1]), 1) for k, v in ph.items() if v[1]}))
