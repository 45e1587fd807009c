"""Developer tool: cProfile of configs[3] through the reference's AdaptiveDistanceSMC loop (bench.py cfg4_end_to_end,
device simulator): where ELFI's Python spends the batch once simulator and distance run on the GPU.
usage: python scripts/cfg4_cprofile.py [batch_size] [rounds]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import bench  # noqa: E402

elfi = bench._ref_elfi()
import elfi_amd  # noqa: E402
from elfi_amd import fused_models as F  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 10 ** 6
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def run():
    mdl = F.gauss_wide_model(m=64)
    smc = elfi_amd.HipAdaptiveDistanceSMC(mdl['d'], batch_size=bs, seed=1)
    return smc.sample(1000, rounds, quantile=0.25, bar=False)


run()
import gc  # noqa: E402
gc.collect()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
res = run()
pr.disable()
print("wall %.3f s, n_sim %d" % (time.perf_counter() - t0, res.n_sim))
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
