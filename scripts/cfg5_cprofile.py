"""Developer tool: cProfile of BASELINE configs[4] through the reference's loop, shortened (where the host time of an
acquisition goes).   python scripts/cfg5_cprofile.py [n_initial] [lines]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench

n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 8064
lines = int(sys.argv[2]) if len(sys.argv) > 2 else 45
bench.cfg5_end_to_end(n_evidence=8192, n_initial=8160)     # warm-up: library, plans, pools
pr = cProfile.Profile()
pr.enable()
r = bench.cfg5_end_to_end(n_evidence=8192, n_initial=n0)
pr.disable()
print("wall %.2f s (profiled), %d acquisitions" % (r["wall_s"], r["acquisitions"]))
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(lines)
st.sort_stats("tottime").print_stats(20)
