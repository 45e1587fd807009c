import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import numpy as np
from elfi_amd import HipGPRegression, HipLCBSC
from benchlib import bolfi_bench
n, d, S = 4096, 10, 10
X, y, bounds = bolfi_bench.problem(n, d)
names = ['t%d' % i for i in range(d)]
gp = HipGPRegression(names, bounds=dict(zip(names, bounds)))
gp.update(X[:4090], y[:4090]); gp.fix_hyperparameters(**bolfi_bench.heuristic_hyper(bounds, y))
acq = HipLCBSC(gp, n_inits=S, exploration_rate=10, seed=2)
for t in (4090, 4091, 4092):
    t0 = time.perf_counter(); acq.acquire(1, t=t); dt = time.perf_counter() - t0
    o = acq.last_opt
    print('t', t, 'ms %.2f' % (dt * 1e3), 'iters', o['iters'].tolist(), 'n_eval', o['n_eval'], 'vals', np.round(o['vals'], 4).tolist())
# scipy on oracle for the same starts: iteration counts
import gp_oracle as G, scipy.optimize
post = G.Posterior(gp._X, gp._Y, **gp._hyper)
tt = 4092
fun = lambda x: float(G.lcb_evaluate(post, x, tt)[0, 0]); grad = lambda x: G.lcb_evaluate_gradient(post, x, tt)[0]
its, nf, vals = [], [], []
for s in acq.last_opt['starts']:
    r = scipy.optimize.minimize(fun, s, method='L-BFGS-B', jac=grad, bounds=bounds, options={'maxiter': 1000})
    its.append(r.nit); nf.append(r.nfev); vals.append(round(float(r.fun), 4))
print('scipy L-BFGS-B its', its, 'nfev', nf, 'vals', vals)
