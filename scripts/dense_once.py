"""Developer tool: `reps` value+gradient calls with S points on an n x d GP (for rocprofv3 runs of the dense predictor).
usage: python scripts/dense_once.py n d S [reps] [dense_min] [tile_rows]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from benchlib.bolfi_bench import problem, heuristic_hyper
from elfi_amd.gp import GPHandle

n, d, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dense_min = int(sys.argv[5]) if len(sys.argv) > 5 else 0
tile_rows = int(sys.argv[6]) if len(sys.argv) > 6 else 0
X, y, bounds = problem(n, d)
h = heuristic_hyper(bounds, y)
gp = GPHandle(d, n)
gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
gp.set_data(X, y)
gp.factorize()
gp.set_dense_threshold(dense_min, tile_rows)
xs = np.random.RandomState(0).uniform(-2, 2, (S, d))
for _ in range(reps):
    gp.lcb(xs, 3.0)
