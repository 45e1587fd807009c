"""Distribution of lock-step counts of the acquisition search over many calls, with scipy's L-BFGS-B
evaluation counts for the slowest call (developer tool)."""
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
rows = []
for t in range(4090, 4090 + 40):
    t0 = time.perf_counter(); acq.acquire(1, t=t); dt = time.perf_counter() - t0
    o = acq.last_opt
    rows.append((t, dt * 1e3, o['iters'].copy(), int(o['n_eval']), o['starts'].copy(), o['vals'].copy()))
    print('t', t, 'ms %.2f' % (dt * 1e3), 'max it', int(o['iters'].max()), 'iters', o['iters'].tolist())
ms = np.array([r[1] for r in rows[1:]])
print('mean ms %.2f median %.2f max %.2f' % (ms.mean(), np.median(ms), ms.max()))
if '--scipy' in sys.argv:
    import gp_oracle as G, scipy.optimize
    post = G.Posterior(gp._X, gp._Y, **gp._hyper)
    worst = sorted(rows[1:], key=lambda r: -r[2].max())[:3]
    for t, _, iters, _, starts, vals in worst:
        fun = lambda x: float(G.lcb_evaluate(post, x, t)[0, 0]); grad = lambda x: G.lcb_evaluate_gradient(post, x, t)[0]
        its, nf, sv = [], [], []
        for s in starts:
            r = scipy.optimize.minimize(fun, s, method='L-BFGS-B', jac=grad, bounds=bounds, options={'maxiter': 1000})
            its.append(r.nit); nf.append(r.nfev); sv.append(round(float(r.fun), 4))
        print('t', t, 'ours iters', iters.tolist(), 'vals', np.round(vals, 4).tolist())
        print('   scipy its', its, 'nfev', nf, 'vals', sv)
