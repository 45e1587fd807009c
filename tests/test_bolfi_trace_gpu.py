"""GPU: replay of a BOLFI run recorded from the REAL reference loop (tests/golden/gp_bolfi_trace.npz).

The fixture (oracle/make_golden_gp.py:make_bolfi_trace) is the reference's own
BayesianOptimization loop on the MA2 example -- bolfi.py's update / prepare_new_batch /
_should_optimize, the reference's LCBSC + minimize + scipy L-BFGS-B with ModelPrior start points --
run with the CPU oracle as `target_model` (GPy is not installable).  It holds every update() call
the loop made (50, three of them with hyper-parameter optimisation) and every inner minimisation
(30: acquisition index, optimum, value).  Here HipGPRegression receives the same calls:

  * after an update with optimize=True the MAP hyper-parameters must agree with the recorded ones
    (parity unpinned: SCG on the GPU objective vs SCG on the CPU objective: objective value within
    2e-3, well-determined hyper-parameters within 10 %, the flat bias direction to its order of magnitude);
    the replay then continues from the RECORDED values so later steps compare like for like;
  * at every recorded optimum x* the GPU acquisition function equals the recorded value (1e-7 of
    the value scale) and x* is a stationary point of it (the lock-step minimiser started there
    does not improve the value by more than 1e-6 of the scale);
  * the incremental (bordering) updates of the 47 single-point steps track a fresh rebuild.
"""
import os

import numpy as np
import pytest

import gp_oracle as G
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
NAMES = ('var', 'ls', 'bias', 'noise')


def test_replay_of_the_reference_bolfi_loop(hip_ctx):
    from elfi_amd import HipGPRegression, HipLCBSC
    from elfi_amd import hyperopt as H
    g = np.load(os.path.join(GOLDEN, 'gp_bolfi_trace.npz'))
    bounds = {'t1': (-2, 2), 't2': (-1, 1)}
    gp = HipGPRegression(['t1', 't2'], bounds=bounds)
    acq = HipLCBSC(gp, noise_var=0.1, exploration_rate=10, seed=1)
    mins = {int(g['m%d_n' % i]): i for i in range(int(g['n_min']))}
    n_opt = 0
    for u in range(int(g['n_updates'])):
        x, y, opt = g['u%d_x' % u], g['u%d_y' % u], bool(g['u%d_opt' % u])
        rec = dict(zip(NAMES, (float(v) for v in g['u%d_hyper' % u])))
        gp.update(x, y, optimize=opt)
        if u == 0:
            assert gp._hyper == pytest.approx(rec, rel=1e-14)          # initial values of a fresh reference GP
        if opt:
            n_opt += 1
            # both runs stop near a stationary point of the same objective; compare objective values
            # (primary) and the well-determined hyper-parameters.  The bias variance sits in a flat
            # direction (values ~1e-3, Gamma prior with shape << 1): only its order of magnitude is fixed.
            obj = H.MarginalObjective(gp)
            f_gpu = obj.f(H.logexp_inv(np.array([gp._hyper[k] for k in NAMES])))
            f_rec = obj.f(H.logexp_inv(np.array([rec[k] for k in NAMES])))
            assert abs(f_gpu - f_rec) <= 2e-3 * max(1.0, abs(f_rec)), (u, f_gpu, f_rec, gp._hyper, rec)
            for k in ('var', 'ls', 'noise'):
                assert abs(np.log(gp._hyper[k] / rec[k])) <= 0.1, (u, k, gp._hyper, rec)
            assert abs(np.log(gp._hyper['bias'] / rec['bias'])) <= 1.5, (u, gp._hyper, rec)
            gp._hyper = dict(rec)                                       # continue from the recorded state
            gp._refit()
        else:
            assert gp._hyper == pytest.approx(rec, rel=1e-12)
        n = gp.n_evidence
        if n in mins:                                                   # the loop acquired at this state
            i = mins[n]
            xs, f_rec, t = g['m%d_x' % i], float(g['m%d_f' % i]), int(g['m%d_t' % i])
            val = acq.evaluate(xs, t)[0, 0]
            scale = abs(f_rec) + 1.0
            assert abs(val - f_rec) <= 1e-7 * scale, (n, val, f_rec)
            locs, vals, iters, _ = gp._handle.lcb_minimize(xs[None, :], gp.bounds, acq._beta(t))
            assert vals[0] >= f_rec - 1e-6 * scale and vals[0] <= f_rec + 1e-7 * scale, (n, vals[0], f_rec)
    assert n_opt == 3 and gp.n_evidence == 50
    assert np.array_equal(gp.X, g['final_X']) and np.array_equal(gp.Y, g['final_Y'])
    # 47 bordering updates later the factor still matches a fresh CPU posterior
    ref = G.Posterior(gp.X, gp.Y, **gp._hyper)
    xs = np.random.RandomState(0).uniform(-1, 1, (9, 2))
    np.testing.assert_allclose(gp.predict(xs, noiseless=True)[0], ref.predict(xs, noiseless=True)[0], rtol=1e-8)
    np.testing.assert_allclose(gp.predict(xs, noiseless=True)[1], ref.predict(xs, noiseless=True)[1], rtol=1e-7)
    assert abs(gp._log_marginal - ref.log_marginal) <= 1e-9 * abs(ref.log_marginal)
