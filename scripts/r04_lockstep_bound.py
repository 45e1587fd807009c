"""Developer tool: ONE product per lock-step -- measured and bounded.

  * two triangular products (four launches, form 2) against the value-only lock-step (one triangular product: the launch
    chain a symmetric product on ONE stored triangle of K^-1 would have, i.e. its bound) against the one-product form
    that was built (form 3: the full K^-1, three launches), with the difference of the results;
  * an extend (bordering) with and without K^-1 to carry;
  * the price of forming K^-1 after a rebuild (elfihip_gp_form_kinv).
usage: python scripts/r04_lockstep_bound.py [n] [d] [S]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import gp_oracle as G  # noqa: E402
from elfi_amd.gp import GPHandle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 10
S = int(sys.argv[3]) if len(sys.argv) > 3 else 10
X, y, bounds = G.synthetic_gp_problem(n, d)
h = G.default_hyper(bounds, y)
gp = GPHandle(d, n)
gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
gp.set_data(X, y)
gp.factorize()
rs = np.random.RandomState(2)
xs = rs.uniform(-2, 2, (S, d))


def timed(f, reps=300):
    for _ in range(20):
        f()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t0) / reps * 1e6


def phases(f):
    gp.profile(1)
    for _ in range(100):
        f()
    ph = gp.profile(0)
    return {k: round(1e3 * v[0] / max(1, v[1]), 1) for k, v in ph.items() if v[1]}


us = []
gp.set_lockstep_form(2)
for name, f in (("value + gradient (two triangular products)", lambda: gp.lcb(xs, 3.0)),
                ("value only (one triangular product)", lambda: gp.lcb(xs, 3.0, with_grad=False))):
    us.append(timed(f))
    print("n=%d d=%d S=%d %s: %.1f us per call (host); device phases us: %s" % (n, d, S, name, us[-1], phases(f)))
v2, g2 = gp.lcb(xs, 3.0)
gp.set_lockstep_form(3)
f = lambda: gp.lcb(xs, 3.0)
f()
assert gp.lockstep_info()[0], gp.lockstep_info()
us_k = timed(f)
v3, g3 = gp.lcb(xs, 3.0)
print("n=%d d=%d S=%d value + gradient (ONE product with K^-1): %.1f us per call (host); device phases us: %s; against the "
      "triangular form: value %.1e, gradient %.1e (scaled); cond(K) >= %.0f"
      % (n, d, S, us_k, phases(f), np.max(np.abs(v3 - v2)) / np.max(np.abs(v2)), np.max(np.abs(g3 - g2)) / np.max(np.abs(g2)),
         gp.lockstep_info()[2]))
gp.set_lockstep_form(2)
# bordering: an extend with and without K^-1 to carry (32 points inside one padded size)
for form in (2, 3):
    g2_ = GPHandle(d, n)
    g2_.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
    g2_.set_data(X[:n - 64], y[:n - 64])
    g2_.factorize()
    g2_.set_lockstep_form(form)
    g2_.lcb(xs, 3.0)
    g2_.extend(X[n - 64:n - 63], y[n - 64:n - 63])
    t0 = time.perf_counter()
    for i in range(n - 63, n - 31):
        g2_.extend(X[i:i + 1], y[i:i + 1])
    print("n=%d extend by one point, %s: %.1f us" % (n - 48, "K^-1 carried (rank-one bordering)" if form == 3 else
                                                     "triangular factors only", (time.perf_counter() - t0) / 32 * 1e6))
    g2_.close()

# the price of the matrix: K^-1 = W^T W after a rebuild (the gradient kernel's SYRK without the contractions)
t_fact = timed(gp.factorize, 20) / 1e3
def both():
    gp.factorize()
    gp.form_kinv()
t_both = timed(both, 20) / 1e3
print("n=%d rebuild %.3f ms, rebuild + K^-1 %.3f ms => K^-1 costs %.3f ms per rebuild = the saving of %.0f lock-steps (%.1f us each)"
      % (n, t_fact, t_both, t_both - t_fact, 1e3 * (t_both - t_fact) / max(1e-9, us[0] - us[1]), us[0] - us[1]))
