"""Developer tool: one evaluation of the hyper-parameter search (rebuild + gradient) at the sizes a configs[2] run passes
through: device phases from the library's timers and the wall time per evaluation through the Python objective.
usage: python scripts/time_search_eval.py [n:d ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from benchlib.bolfi_bench import problem, heuristic_hyper
from elfi_amd.gp import GPHandle

shapes = [(int(a.split(':')[0]), int(a.split(':')[1])) for a in sys.argv[1:]] or [(512, 2), (1024, 2), (1536, 2), (2048, 2), (3072, 2), (4096, 2), (4096, 10)]
for n, d in shapes:
    X, y, bounds = problem(n, d)
    h = heuristic_hyper(bounds, y)
    gp = GPHandle(d, n)
    gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
    gp.set_data(X, y)
    gp.factorize()
    gp.nlml_grad()
    R = 20
    t0 = time.perf_counter()
    for _ in range(R):
        gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
        gp.factorize()
        gp.nlml_grad()
    wall = (time.perf_counter() - t0) / R
    gp.profile(1)
    for _ in range(5):
        gp.factorize()
        gp.nlml_grad()
    p = gp.profile(0)
    ph = {k: v[0] / max(v[1], 1) for k, v in p.items() if v[1]}
    print("n=%5d d=%2d  wall %.3f ms per evaluation | gram %.3f sweep %.3f alpha %.3f kinv_grad %.3f (sum %.3f)" %
          (n, d, 1e3 * wall, ph.get('gram', 0), ph.get('sweep', 0), ph.get('alpha', 0), ph.get('kinv_grad', 0),
           sum(ph.get(k, 0) for k in ('gram', 'sweep', 'alpha', 'kinv_grad'))))
    gp.close()
