"""Developer probe: per-workgroup wall-clock stamps inside the triangular products of one lock-step (stamped build of the
library: sh scripts/native/build_tri_stamp.sh).  Prints, for the fused triangular products (form 2) and the K^-1 product
(form 3), percentiles of each phase relative to the first stamp.
    python scripts/pair_timeline.py [n]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
from elfi_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, "scripts", "native", "libelfihip_stamp.so")
from benchlib.bolfi_bench import heuristic_hyper, problem  # noqa: E402
from elfi_amd.gp import GPHandle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d, S = 10, 10
X, y, bounds = problem(n, d)
h = heuristic_hyper(bounds, y)
gp = GPHandle(d, n)
gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
gp.set_data(X, y)
gp.factorize()
lib = _lib.load_library()
lib.elfihip_debug_tri_stamps.restype = C.c_int
lib.elfihip_debug_tri_stamps.argtypes = [C.c_void_p, C.c_int]
xs = np.random.RandomState(2).uniform(-2, 2, (S, d))
buf = np.zeros(8192 * 8, dtype=np.uint64)
names = ["start", "-", "B in LDS", "partials stored", "arrived", "epilogue end", "epilogue: sums in", "epilogue: round 0 reduced"]
for form in (2, 3):
    gp.set_lockstep_form(form)
    for _ in range(20):
        gp.lcb(xs, 3.0)
    assert lib.elfihip_debug_tri_stamps(None, 1) == 0
    gp.lcb(xs, 3.0)
    assert lib.elfihip_debug_tri_stamps(buf.ctypes.data, 0) == 0
    st = buf.reshape(8192, 8).astype(np.float64) / 100.0        # us
    live = st[:, 0] > 0
    live[7168:] = False
    t0 = st[live, 0].min()
    print("form %d: %d live workgroups" % (form, live.sum()))
    groups = [("first product / K^-1 product", live & (np.arange(8192) < 4096)), ("second product", live & (np.arange(8192) >= 4096))]
    ks = st[7168:7424]
    kl = ks[:, 0] > 0
    if kl.any():
        print("  kernel rows (kstar_kernel), %d workgroups, us after the FIRST product's first stamp:" % kl.sum())
        for j, nm in enumerate(["start", "query point in LDS", "rows done (before the last store)"]):
            v = ks[kl, j] - t0
            print("    %-34s min %6.2f  p50 %6.2f  max %6.2f" % (nm, v.min(), np.percentile(v, 50), v.max()))
        print("    own duration start -> rows done: p50 %.2f, of which the query point %.2f" %
              (np.percentile(ks[kl, 2] - ks[kl, 0], 50), np.percentile(ks[kl, 1] - ks[kl, 0], 50)))
    for gname, m in groups:
        print("  %s: %d workgroups" % (gname, m.sum()))
        for j, nm in enumerate(names):
            v = st[m, j]
            v = v[v > 0] - t0
            if len(v):
                print("    %-22s n=%5d  min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" %
                      (nm, len(v), v.min(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
        life = st[m, 4] - st[m, 0]
        life = life[(st[m, 4] > 0)]
        if len(life):
            print("    life start -> arrived: p50 %.2f  p90 %.2f us" % (np.percentile(life, 50), np.percentile(life, 90)))
