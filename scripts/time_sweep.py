"""Developer tool: the resident sweep kernel (ELFIHIP_SWEEP=1) against the default multi-launch sweep, panel groups
ELFIHIP_SWEEP_GROUP = 1, 2, 4."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, gp_oracle as G
from elfi_amd.gp import GPHandle

def fit_ms(gp, reps=5):
    gp.factorize()
    t0 = time.perf_counter()
    for _ in range(reps):
        gp.factorize()
    return (time.perf_counter() - t0) / reps * 1e3

for n in [int(a) for a in sys.argv[1:]] or [1024, 4096]:
    X, y, b = G.synthetic_gp_problem(n, 4)
    h = G.default_hyper(b, y)
    gp = GPHandle(4, n); gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise']); gp.set_data(X, y)
    os.environ['ELFIHIP_SWEEP'] = '0'
    lz0 = gp.factorize()
    line = 'n=%d default %.3f ms' % (n, fit_ms(gp))
    os.environ['ELFIHIP_SWEEP'] = '1'
    for g in (1, 2, 4):
        os.environ['ELFIHIP_SWEEP_GROUP'] = str(g)
        lz = gp.factorize()
        line += ' | G=%d %.3f ms (dlogZ %.1e)' % (g, fit_ms(gp), lz - lz0)
    print(line, flush=True)
