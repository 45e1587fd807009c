"""Hyper-parameter MAP search (HipGPRegression.optimize: SCG, at most 50 iterations) timing.  Developer tool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import gp_oracle as G
import elfi_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 10
X, y, bounds = G.synthetic_gp_problem(n, d)
names = ['t%d' % i for i in range(d)]
m = elfi_amd.HipGPRegression(names, bounds=dict(zip(names, bounds)))
m.update(X, y)            # default hyper-parameters, no optimisation
for rep in range(2):
    m._hyper = dict(m._default_hyper) if hasattr(m, '_default_hyper') and isinstance(m._default_hyper, dict) else m._hyper
    t0 = time.perf_counter()
    m.optimize()
    dt = time.perf_counter() - t0
    info = m._opt_info
    print('n=%d d=%d optimize(): %.1f ms, %d factorisations, %d objective/gradient evaluations, status %s -> %.2f ms per factorisation+gradient; hyper %s'
          % (n, d, dt * 1e3, info['n_fits'], info['n_eval'], info['status'], dt * 1e3 / max(1, info['n_fits']),
             {k: round(v, 4) for k, v in m._hyper.items()}))
