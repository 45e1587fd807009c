"""Developer soak test: many updates / refits / acquisitions / handle re-creations on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import numpy as np
import gp_oracle as G
from elfi_amd import HipGPRegression, HipLCBSC

rs = np.random.RandomState(0)
d = 3
names = ['a', 'b', 'c']
bounds = {k: (-2., 2.) for k in names}
f = lambda x: np.linalg.norm(x - 0.3, axis=1, keepdims=True) + 0.05 * rs.randn(len(x), 1)
t0 = time.time()
for rep in range(3):
    gp = HipGPRegression(names, bounds=bounds, max_opt_iters=15)
    acq = HipLCBSC(gp, noise_var=0.05, seed=rep)
    X = rs.uniform(-2, 2, (100, d))
    gp.update(X, f(X))
    for it in range(400):
        x = acq.acquire(1, t=it)
        gp.update(x, f(x), optimize=(it % 97 == 96))
        if it % 50 == 0:
            ref = G.Posterior(gp.X, gp.Y, **gp._hyper)
            xs = rs.uniform(-2, 2, (4, d))
            np.testing.assert_allclose(gp.predict(xs)[0], ref.predict(xs)[0], rtol=1e-7)
    k2 = gp.copy()
    assert k2.n_evidence == gp.n_evidence == 500
    np.testing.assert_allclose(k2.predict(xs)[0], gp.predict(xs)[0], rtol=1e-9)
    print('rep', rep, 'n', gp.n_evidence, 'hyper', {k: round(v, 4) for k, v in gp._hyper.items()}, 'elapsed %.1fs' % (time.time() - t0))
print('soak OK')
