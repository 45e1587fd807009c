"""Developer tool: one configs[4]-shaped acquisition (n = 8192, d = 20, 256 starts) with the library's per-round trace
(active points : device ms per lock-step round) on stderr.   python scripts/cfg5_trace.py [n] [d] [S]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
from benchlib.bolfi_bench import heuristic_hyper, problem  # noqa: E402
from elfi_amd.gp import HipGPRegression  # noqa: E402
from elfi_amd.lcb_acquisition import HipLCBSC  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d = int(sys.argv[2]) if len(sys.argv) > 2 else 20
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
X, y, bounds = problem(n, d)
names = ['t%d' % i for i in range(d)]
gp = HipGPRegression(names, bounds=dict(zip(names, bounds)))
gp.update(X, y)
gp.fix_hyperparameters(**heuristic_hyper(bounds, y))
acq = HipLCBSC(gp, n_inits=S, exploration_rate=10, seed=3)
acq.acquire(1, t=n)
gp._handle.set_acq_options(0, 2)
for r in range(2):
    t0 = time.perf_counter()
    acq.acquire(1, t=n + r)
    print("acquire %d: %.2f ms, %d point evaluations, max iterations %d" %
          (r, 1e3 * (time.perf_counter() - t0), acq.last_opt['n_eval'], int(np.max(acq.last_opt['iters']))), flush=True)
