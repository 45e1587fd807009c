"""Developer tool: GP rebuild time per sweep schedule (elfihip_gp_set_schedule) and size.
usage: python scripts/time_schedules.py [n:d ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from benchlib.bolfi_bench import problem, heuristic_hyper
from elfi_amd.gp import GPHandle

shapes = [tuple(int(v) for v in a.split(':')) for a in sys.argv[1:]] or \
    [(512, 2), (1024, 2), (2048, 10), (3072, 10), (4096, 10), (5120, 10), (6144, 10), (8192, 20)]
print("%8s %4s | %-22s | %8s %8s" % ("n", "d", "schedule", "ms", "TFLOP/s"))
for n, d in shapes:
    X, y, bounds = problem(n, d)
    h = heuristic_hyper(bounds, y)
    gp = GPHandle(d, n)
    gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
    gp.set_data(X, y)
    flops = 2.0 * n * n * d + 2.0 * n ** 3 / 3.0
    ref = None
    for name, sched, group in (("streams (auto group)", 1, 0), ("fused steps (3 launches)", 2, 0),
                               ("fused steps, 2 launches", 4, 0), ("fused steps, chained", 3, 0),
                               ("overlapped (3 streams)", 5, 0)):
        if os.environ.get('ONLY_SCHEDULES') and str(sched) not in os.environ['ONLY_SCHEDULES'].split(','):
            continue
        gp.set_schedule(sched, group)
        lz = gp.factorize()
        if ref is None:
            ref = lz
        assert abs(lz - ref) <= 1e-9 * abs(ref), (name, lz, ref)
        if sched == 5:
            print("        overlapped: log Z %.12f against %.12f (three launches), equal bit for bit: %s" % (lz, lz2, lz == lz2), flush=True)
        if sched == 4:
            assert lz == lz2, "the two-launch step must give the three-launch step's log Z bit for bit"
        if sched == 2:
            lz2 = lz
        reps = 10 if n <= 4096 else 4
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                gp.factorize()
            best = min(best, (time.perf_counter() - t0) / reps)
        print("%8d %4d | %-24s | %8.3f %8.1f" % (n, d, name, best * 1e3, flops / best / 1e12), flush=True)
    gp.close()
