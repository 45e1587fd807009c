"""Posterior sampling rate: lock-step NUTS on the device posterior vs the reference's call pattern on the CPU
restatement (one chain after the other, separate logpdf / gradient calls per leapfrog step).  Developer tool.

    python scripts/time_posterior.py [n] [d] [n_chains] [n_samples]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import gp_oracle as G
import posterior_oracle as PO
import elfi_amd
from elfi_amd import chains

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 10
C = int(sys.argv[3]) if len(sys.argv) > 3 else 4
N = int(sys.argv[4]) if len(sys.argv) > 4 else 400
X, y, bounds = G.synthetic_gp_problem(n, d)
names = ['t%d' % i for i in range(d)]
m = elfi_amd.HipGPRegression(names, bounds=dict(zip(names, bounds)))
m.update(X, y)
h = G.default_hyper(bounds, y)
m._hyper = h
m._refit()
prior = PO.BoxPrior(bounds)
thr = float(np.min(y) + 0.3)
t0 = time.perf_counter()
out, bp = elfi_amd.sample_posterior(m, prior, N, n_chains=C, threshold=thr, seed=1)
dt = time.perf_counter() - t0
r, p = chains.run_lockstep.n_rounds, chains.run_lockstep.n_points
print('device  n=%d d=%d: %d chains x %d NUTS iterations in %.2f s  (%d device calls, %d point evaluations, %.0f us per call, %.0f samples/s)'
      % (n, d, C, N, dt, r, p, dt / r * 1e6, C * N / dt))
# CPU: the reference's pattern on the restated posterior, bounded sample of the same work
post = G.Posterior(X, y, h['var'], h['ls'], h['bias'], h['noise'])
po = PO.PosteriorOracle(post, bounds, thr)
calls = [0]
def target(x):
    calls[0] += 1
    return float(po.logpdf_and_gradient(x)[0][0])
def grad(x):
    calls[0] += 1
    return po.logpdf_and_gradient(x)[1][0]
def ev1(Xq):  # one point per call: what a sequential chain does
    lp = np.array([target(x) for x in Xq]); gr = np.array([grad(x) for x in Xq])
    return lp, gr
Ncpu = max(40, min(N, 40 * 4096 * 4096 // (n * n)))
x0 = X[int(np.argmin(y))]
t0 = time.perf_counter()
cpu = chains.nuts(Ncpu, x0[None, :], ev1, seeds=[elfi_amd.posterior.sub_seed(1, 0)], n_adapt=Ncpu // 2)
dtc = time.perf_counter() - t0
print('cpu port n=%d: 1 chain x %d NUTS iterations in %.2f s (%d single-point GP calls) -> %.2f samples/s; device/cpu = %.0fx'
      % (n, Ncpu, dtc, calls[0], Ncpu / dtc, (C * N / dt) / (Ncpu / dtc)))
