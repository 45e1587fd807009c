"""Developer tool: the library's per-acquisition trace (rounds, point evaluations, device / host ms) inside BASELINE configs[4]
run through the reference's loop, shortened.   python scripts/cfg5_e2e_trace.py [n_initial]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench
import elfi_amd

n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 8128
elfi_amd.HipGPRegression.acq_trace = 1
r = bench.cfg5_end_to_end(n_evidence=8192, n_initial=n0)
print("wall %.2f s, %d acquisitions, ms per acquire %.2f" % (r["wall_s"], r["acquisitions"], r.get("ms_device_acquire", -1)))
