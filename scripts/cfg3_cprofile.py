"""Developer tool: cProfile of BASELINE configs[2] through the reference's loop (where the host time goes).
usage: python scripts/cfg3_cprofile.py [n_evidence] [lines]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
lines = int(sys.argv[2]) if len(sys.argv) > 2 else 45
pr = cProfile.Profile()
pr.enable()
r = bench.cfg3_end_to_end(n_evidence=n)
pr.disable()
print("wall %.2f s (profiled)" % r["wall_s"])
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(lines)
st.sort_stats("tottime").print_stats(25)
