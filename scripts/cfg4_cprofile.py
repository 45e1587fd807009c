"""Developer tool: BASELINE configs[3] through the reference's AdaptiveDistanceSMC loop (bench.cfg4_end_to_end's device leg),
repeated, then one run under cProfile (where the 0.1 s go).
usage: python scripts/cfg4_cprofile.py [repeats] [lines]"""
import cProfile
import gc
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
lines = int(sys.argv[2]) if len(sys.argv) > 2 else 40
elfi = bench._ref_elfi()
import elfi_amd
from elfi_amd import fused_models as F


def run():
    mdl = F.gauss_wide_model(m=64)
    smc = elfi_amd.HipAdaptiveDistanceSMC(mdl['d'], batch_size=10 ** 6, seed=1)
    smc.device_proposals = True
    t0 = time.perf_counter()
    res = smc.sample(1000, 3, quantile=0.25, bar=False)
    return time.perf_counter() - t0, res.n_sim


walls = []
for i in range(reps):
    gc.collect()
    w, n_sim = run()
    walls.append(w)
print("walls (s):", " ".join("%.4f" % w for w in walls), " n_sim", n_sim, " best %.1f M rows/s" % (n_sim / min(walls[1:]) / 1e6))
pr = cProfile.Profile()
pr.enable()
w, _ = run()
pr.disable()
print("profiled wall %.4f" % w)
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(lines)
st.sort_stats("tottime").print_stats(20)
