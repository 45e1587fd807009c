"""Host wall per acquisition lock-step (value + gradient of the LCB at S points) by form: 1 = six launches, 2 = fused
triangular products (four launches), 3 = one product with K^-1 (three launches).
    python scripts/lockstep_forms.py [S] [d]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
from benchlib.bolfi_bench import heuristic_hyper, problem  # noqa: E402
from elfi_amd.gp import GPHandle  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 10
d = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for n in (1024, 2048, 3072, 4096, 8192):
    X, y, bounds = problem(n, d)
    h = heuristic_hyper(bounds, y)
    gp = GPHandle(d, n)
    gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
    gp.set_data(X, y)
    gp.factorize()
    xs = np.random.RandomState(2).uniform(-2, 2, (S, d))
    row = {}
    ref = None
    for name, form in (("six_launches", 1), ("triangular", 2), ("kinv", 3)):
        gp.set_lockstep_form(form)
        for _ in range(30):
            out = gp.lcb(xs, 3.0)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(200):
                gp.lcb(xs, 3.0)
            best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
        row[name] = round(best, 2)
        if form == 1:
            ref = out
        elif form == 2:
            assert np.array_equal(ref[0], out[0]), "fused form: LCB values differ from the six launches"
    print("n=%d d=%d S=%d us per lock-step: %s" % (n, d, S, row), flush=True)
    gp.close()
