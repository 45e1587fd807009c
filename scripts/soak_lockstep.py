"""Developer tool (GPU): soak of the fused lock-step's in-launch hand-offs.  The last workgroup to arrive at a row block reads
the other workgroups' written-through partials after ONE agent-scope acquire; a stale read would show as a mean / variance
that differs from the six-launch form (bit-identical by construction).  Thousands of calls, sizes and point counts mixed,
a second context streaming on the same GPU meanwhile (uneven load, the consumer's L1 warm from the previous call).
usage: python scripts/soak_lockstep.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import elfi_amd
from benchlib.bolfi_bench import problem, heuristic_hyper
from elfi_amd.gp import GPHandle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
gps = []
for n, d in ((4096, 10), (1500, 2), (2304, 5)):
    X, y, bounds = problem(n, d)
    h = heuristic_hyper(bounds, y)
    gp = GPHandle(d, n)
    gp.set_hyper(h['var'], h['ls'], h['bias'], h['noise'])
    gp.set_data(X, y)
    gp.factorize()
    gps.append((gp, d))
other = elfi_amd.Context(0)                       # a second stream on the same GPU: distance passes beside the predictions
Xd = np.random.RandomState(0).randn(400000, 32)
yd = np.zeros((1, 32))
rs = np.random.RandomState(1)
t_end, calls, bad = time.time() + budget, 0, 0
while time.time() < t_end:
    gp, d = gps[rs.randint(len(gps))]
    S = int(rs.choice([1, 3, 10, 16, 17, 40, 100]))
    xs = rs.uniform(-2, 2, (S, d))
    if calls % 7 == 0:
        elfi_amd.cdist_rows(Xd, yd, 'euclidean', ctx=other)
    gp.set_lockstep_form(1)
    m0, v0, dm0, dv0 = gp.predict_grad(xs)
    gp.set_lockstep_form(0)
    for _ in range(3):
        m1, v1, dm1, dv1 = gp.predict_grad(xs)
        ok = np.array_equal(m0, m1) and np.array_equal(v0, v1) and np.allclose(dv0, dv1, rtol=0, atol=1e-10 * (np.max(np.abs(dv0)) + 1e-300))
        bad += not ok
        calls += 1
print("soak: %d fused calls compared with the six-launch form, %d mismatches" % (calls, bad))
sys.exit(1 if bad else 0)
