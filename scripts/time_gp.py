"""Quick timing of the GP fit / LCB evaluation on the GPU (developer tool)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import numpy as np
import gp_oracle as G
from elfi_amd.gp import GPHandle

n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 10
X, y, bounds = G.synthetic_gp_problem(n, d)
h = G.default_hyper(bounds, y)
gp = GPHandle(d, n)
gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
gp.set_data(X, y)
gp.factorize()
t0 = time.perf_counter()
R = 5
for _ in range(R):
    gp.factorize()
t = (time.perf_counter() - t0) / R
fl = n**3 / 3 * 2
print("factorize n=%d: %.3f ms  (%.1f TFLOP/s on 2n^3/3)" % (n, t * 1e3, fl / t / 1e12))
xs = np.random.RandomState(2).uniform(-2, 2, (10, d))
gp.lcb(xs, 3.0)
t0 = time.perf_counter()
R = 20
for _ in range(R):
    gp.lcb(xs, 3.0)
t = (time.perf_counter() - t0) / R
print("lcb S=10 value+grad: %.3f ms" % (t * 1e3))
t0 = time.perf_counter()
for _ in range(R):
    gp.lcb(xs, 3.0, with_grad=False)
t = (time.perf_counter() - t0) / R
print("lcb S=10 value only: %.3f ms" % (t * 1e3))
t0 = time.perf_counter()
R = 5
for _ in range(R):
    gp.nlml_grad()
t = (time.perf_counter() - t0) / R
print("nlml_grad (K^-1 SYRK n^3/3 + contractions): %.3f ms  (%.1f TFLOP/s)" % (t * 1e3, n**3 / 3 / t / 1e12))
