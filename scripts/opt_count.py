"""Developer tool: factorisations per HipGPRegression.optimize() call (the refit at the optimum is skipped when the search's
last evaluation was the optimum).   python scripts/opt_count.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import elfi_amd
from elfi_amd import hyperopt as H
rs = np.random.RandomState(4); n = 1500
X = rs.uniform(-2, 2, (n, 2)); y = (np.linalg.norm(X - 0.5, axis=1) + 0.1 * rs.randn(n))[:, None]
m = elfi_amd.HipGPRegression(['a', 'b'], bounds={'a': (-2, 2), 'b': (-2, 2)})
m.update(X, y)
h = m._handle
cnt = [0]
orig = h.factorize
def fz():
    cnt[0] += 1
    return orig()
h.factorize = fz
for rep in range(3):
    cnt[0] = 0
    t0 = time.perf_counter(); m.optimize(); dt = time.perf_counter() - t0
    print('optimize: %d factorisations, n_fits %d, status %s, %.1f ms' % (cnt[0], m._opt_info['n_fits'], m._opt_info['status'], dt * 1e3))
