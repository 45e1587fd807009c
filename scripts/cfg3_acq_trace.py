"""Developer tool: BASELINE configs[2] through the reference's loop with the acquisition trace on -- rounds, point
evaluations, device and host time of every acquire() call (HipGPRegression.acq_trace = 1: lines on stderr), summarised.
usage: python scripts/cfg3_acq_trace.py [n_evidence]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("CFG3_CHILD") == "1":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bench
    import elfi_amd
    elfi_amd.HipGPRegression.acq_trace = 1
    r = bench.cfg3_end_to_end(n_evidence=int(sys.argv[1]))
    print("wall %.2f s" % r["wall_s"])
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else "2048"
env = dict(os.environ, CFG3_CHILD="1")
p = subprocess.run([sys.executable, os.path.abspath(__file__), n], env=env, capture_output=True, text=True)
rows = [tuple(float(v) for v in m) for m in re.findall(
    r"S=(\d+) n=(\d+) rounds=(\d+) evals=(\d+) device ([\d.]+) ms host ([\d.]+) ms", p.stderr)]
print(p.stdout.strip())
print("%d acquisitions" % len(rows))
if rows:
    import numpy as np
    a = np.array(rows)
    for lo, hi in ((0, 1024), (1024, 2048), (2048, 3072), (3072, 4097)):
        m = (a[:, 1] >= lo) & (a[:, 1] < hi)
        if m.any():
            b = a[m]
            print("n in [%4d, %4d): %4d calls | rounds %.1f (max %d) | evaluations %.1f | device %.3f ms = %.1f us per round | host %.3f ms"
                  % (lo, hi, m.sum(), b[:, 2].mean(), b[:, 2].max(), b[:, 3].mean(), b[:, 4].mean(),
                     1e3 * b[:, 4].sum() / b[:, 2].sum(), b[:, 5].mean()))
