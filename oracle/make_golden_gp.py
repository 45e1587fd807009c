"""TEST INFRASTRUCTURE ONLY -- GP / acquisition fixtures produced by the REFERENCE'S OWN CODE.

GPy is not installable here, so the reference's GP cannot be fitted; but everything ELFI itself
implements on top of GPy can be executed unmodified if it is handed the posterior quantities:

  * GPyRegression.predict / predictive_gradients, sampling-phase closed forms
    (elfi/methods/bo/gpy_regression.py:127-140, 206-218) -- the real class, real code, with a
    stand-in `_gp` object that carries X, the kernel parameters and woodbury_vector / _inv / _chol
    computed by oracle/gp_oracle.py at fixed hyper-parameters;
  * LCBSC._beta / evaluate / evaluate_gradient (elfi/methods/bo/acquisition.py:256-301) and
    AcquisitionBase.acquire -> minimize (acquisition.py:129-191, bo/utils.py:40-111) -- the real
    classes; in this (non-sampling) path they call `_gp.predict_noiseless / predictive_gradients`,
    which the stand-in answers with the oracle's [GPy-upstream] restatement.

The reference's own tests assert closed form == GPy (tests/unit/test_methods.py:110-122), so a
restatement that agrees with the closed forms agrees with GPy at that tolerance.

    python oracle/make_golden.py gp        ->  tests/golden/gp_closed_forms.npz, gp_acquisition.npz,
                                               gp_bolfi_trace.npz (the real BOLFI loop on MA2)
"""
import os

import numpy as np

import gp_oracle as G


class _P:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _Param(float):
    """float(p), p[0] -- what gpy_regression.py:151-155 does with paramz Params."""

    def __getitem__(self, i):
        return float(self)


def standin_gp(post):
    n = post.X.shape[0]
    bias_part = _P(variance=_Param(post.bias), K=lambda X, X2=None: np.full((len(X), len(X if X2 is None else X2)), post.bias))
    kern = _P(rbf=_P(variance=_Param(post.var), lengthscale=_Param(post.ls)), bias=bias_part,
              K=lambda X, X2=None: G.kern_K(np.asarray(X, float), None if X2 is None else np.asarray(X2, float),
                                            post.var, post.ls, post.bias))
    lik = _P(variance=[post.noise])
    posterior = _P(woodbury_vector=post.alpha, woodbury_inv=post.Kinv, woodbury_chol=post.L)

    def predictive_gradients(x):
        gm, gv = post.predictive_gradients(x)
        return gm[:, :, None], gv          # GPy returns (S, d, 1) for the mean

    return _P(X=post.X, Y=post.Y, num_data=n, kern=kern, likelihood=lik, Gaussian_noise=lik, posterior=posterior,
              predict_noiseless=lambda x: post.predict(x, noiseless=True),
              predict=lambda x: post.predict(x, noiseless=False), predictive_gradients=predictive_gradients)


def main(elfi, golden_dir):
    from elfi.methods.bo.acquisition import LCBSC
    from elfi.methods.bo.gpy_regression import GPyRegression

    out = {}
    cases = [(60, 1, 11), (200, 2, 12), (500, 5, 13)]
    for n, d, seed in cases:
        X, y, bounds = G.synthetic_gp_problem(n, d, seed=seed)
        h = G.default_hyper(bounds, y)
        post = G.Posterior(X, y, h['var'], h['ls'], h['bias'], h['noise'])
        names = ['p%d' % i for i in range(d)]
        ref = GPyRegression(names, bounds=dict(zip(names, bounds)))
        ref._gp = standin_gp(post)
        ref._kernel_is_default = True
        ref.is_sampling = True
        xs = np.random.RandomState(seed).uniform(-2, 2, (8, d))
        xs[0] = X[3]
        mu = np.array([ref.predict(x)[0][0, 0] for x in xs])
        var = np.array([ref.predict(x)[1][0, 0] for x in xs])           # includes the noise (:139)
        gmu = np.array([ref.predictive_gradients(x)[0][0] for x in xs])
        gvar = np.array([ref.predictive_gradients(x)[1][0] for x in xs])
        tag = '%d_%d' % (n, d)
        out.update({'X_' + tag: X, 'y_' + tag: y, 'xs_' + tag: xs, 'mu_' + tag: mu, 'var_' + tag: var,
                    'gmu_' + tag: gmu, 'gvar_' + tag: gvar,
                    'hyper_' + tag: np.array([h['var'], h['ls'], h['bias'], h['noise']])})
        # LCBSC through the reference class, non-sampling path
        ref.is_sampling = False
        acq = LCBSC(ref, n_inits=6, noise_var=0.05, exploration_rate=10, seed=seed)
        for t in (0, 17):
            out['beta_%s_%d' % (tag, t)] = np.float64(acq._beta(t))
            out['lcb_%s_%d' % (tag, t)] = acq.evaluate(xs, t)
            out['lcbg_%s_%d' % (tag, t)] = acq.evaluate_gradient(xs, t)
    out['cases'] = np.array(['%d_%d' % (n, d) for n, d, _ in cases])
    np.savez_compressed(os.path.join(golden_dir, 'gp_closed_forms.npz'), **out)
    print('gp_closed_forms:', len(cases), 'cases')

    # a full acquire() of the reference (start points, scipy L-BFGS-B per start, arg-min, jitter)
    acq_out = {}
    for n, d, seed in [(150, 2, 21), (300, 3, 22)]:
        X, y, bounds = G.synthetic_gp_problem(n, d, seed=seed)
        h = G.default_hyper(bounds, y)
        post = G.Posterior(X, y, h['var'], h['ls'], h['bias'], h['noise'])
        names = ['p%d' % i for i in range(d)]
        ref = GPyRegression(names, bounds=dict(zip(names, bounds)))
        ref._gp = standin_gp(post)
        ref._kernel_is_default = True
        acq = LCBSC(ref, n_inits=8, noise_var=0.1, exploration_rate=10, seed=seed)
        t = 9
        x_acq = acq.acquire(3, t=t)
        # replay the random stream to record the start points and the optimum before the jitter
        rs = np.random.RandomState(seed)
        starts = np.empty((8, d))
        for i in range(d):
            starts[:, i] = rs.uniform(*bounds[i], 8)
        fun = lambda x: acq.evaluate(x, t)
        grad = lambda x: acq.evaluate_gradient(x, t)
        import scipy.optimize
        locs, vals = [], []
        for s in starts:
            r = scipy.optimize.minimize(fun, s, method='L-BFGS-B', jac=grad, bounds=bounds, options={'maxiter': 1000})
            locs.append(r['x'])
            vals.append(float(np.ravel(r['fun'])[0]))
        tag = '%d_%d' % (n, d)
        acq_out.update({'X_' + tag: X, 'y_' + tag: y, 'hyper_' + tag: np.array([h['var'], h['ls'], h['bias'], h['noise']]),
                        'starts_' + tag: starts, 'locs_' + tag: np.array(locs), 'vals_' + tag: np.array(vals),
                        'x_acq_' + tag: x_acq, 't_' + tag: np.int64(t), 'seed_' + tag: np.int64(seed)})
    acq_out['cases'] = np.array(['150_2', '300_3'])
    np.savez_compressed(os.path.join(golden_dir, 'gp_acquisition.npz'), **acq_out)
    print('gp_acquisition: 2 cases')

    make_bolfi_trace(elfi, golden_dir)


def make_bolfi_trace(elfi, golden_dir):
    """The reference's whole BOLFI loop (bolfi.py: BayesianOptimization.update / prepare_new_batch /
    _should_optimize, the reference's LCBSC + minimize + scipy L-BFGS-B, ModelPrior start points) on
    the MA2 example, with OracleGPRegression as `target_model`.  Recorded: every update() call the
    loop makes (evidence, optimize flag, hyper-parameters afterwards) and every inner minimisation
    (acquisition index t, optimum, value)."""
    import elfi.methods.bo.acquisition as acq_mod
    from elfi.examples import ma2
    from oracle_gp_model import OracleGPRegression

    m = ma2.get_model(seed_obs=4)
    log_d = elfi.Operation(np.log, m['d'], name='log_d')
    bounds = {'t1': (-2, 2), 't2': (-1, 1)}
    tm = OracleGPRegression(['t1', 't2'], bounds=bounds, max_opt_iters=50)
    mins = []
    orig_minimize = acq_mod.minimize

    def recording_minimize(fun, *a, **k):
        x, f = orig_minimize(fun, *a, **k)
        mins.append((tm.n_evidence, np.array(x, float), float(np.ravel(f)[0])))
        return x, f

    acq_mod.minimize = recording_minimize
    try:
        bolfi = elfi.BOLFI(log_d, batch_size=1, initial_evidence=20, update_interval=10, bounds=bounds,
                           target_model=tm, acq_noise_var=0.1, seed=1)
        ts = []
        orig_acquire = bolfi.acquisition_method.acquire

        def recording_acquire(n, t=None):
            ts.append(t)
            return orig_acquire(n, t=t)

        bolfi.acquisition_method.acquire = recording_acquire
        bolfi.fit(n_evidence=50)
    finally:
        acq_mod.minimize = orig_minimize
    out = {'n_updates': np.int64(len(tm.log)), 'n_min': np.int64(len(mins))}
    for i, (x, y, opt, h) in enumerate(tm.log):
        out['u%d_x' % i], out['u%d_y' % i] = x, y
        out['u%d_opt' % i] = np.bool_(opt)
        out['u%d_hyper' % i] = np.array([h['var'], h['ls'], h['bias'], h['noise']])
    for i, (n_ev, x, f) in enumerate(mins):
        out['m%d_n' % i], out['m%d_x' % i], out['m%d_f' % i] = np.int64(n_ev), x, np.float64(f)
        out['m%d_t' % i] = np.int64(ts[i])
    out['final_X'], out['final_Y'] = tm.X, tm.Y
    np.savez_compressed(os.path.join(golden_dir, 'gp_bolfi_trace.npz'), **out)
    n_opt = sum(1 for r in tm.log if r[2])
    print('gp_bolfi_trace: %d updates (%d with optimisation), %d acquisitions, final hyper %s'
          % (len(tm.log), n_opt, len(mins), {k: round(v, 4) for k, v in tm.hyper.items()}))


# docs/usage/BOLFI.rst:36-153 -- the one place where the reference publishes GPy-side numbers of a BOLFI run:
# MA2 `seed_obs=1`, elfi.BOLFI(log_d, batch_size=1, initial_evidence=20, update_interval=10, bounds, acq_noise_var,
# seed=1).fit(n_evidence=200) on a 2017 GPy / SciPy / NumPy stack printed
DOC_PRINTED = dict(objective=151.86636065302943, var=0.321697451372, ls=0.541352150083, bias=0.021827430988,
                   noise=0.183562040169, threshold=-1.6146, prior_var=0.024, prior_ls=1.3, prior_bias=0.006)


def make_doc_run(elfi, golden_dir):
    """Replay of the documented run with the oracle model inside the reference's real loop.  The evidence cannot be
    the 2017 run's (other RNG consumers, other L-BFGS-B) but it is a run of the same recipe, so the printed MAP
    hyper-parameters must be near-optimal for it: the fixture records the evidence, the oracle's last MAP search
    (start, end, objective) and the objective the printed values reach on this evidence."""
    import gp_hyper_oracle as HO
    from elfi.examples import ma2
    from oracle_gp_model import OracleGPRegression
    seed = 1
    np.random.seed(seed)
    m = ma2.get_model(seed_obs=seed)
    log_d = elfi.Operation(np.log, m['d'], name='log_d')
    bounds = {'t1': (-2, 2), 't2': (-1, 1)}
    tm = OracleGPRegression(['t1', 't2'], bounds=bounds, max_opt_iters=50)
    bolfi = elfi.BOLFI(log_d, batch_size=1, initial_evidence=20, update_interval=10, bounds=bounds,
                       target_model=tm, acq_noise_var={'t1': 0.1, 't2': 0.1}, seed=seed)
    post = bolfi.fit(n_evidence=200, bar=False)
    opt_steps = [i for i, r in enumerate(tm.log) if r[2]]
    last = opt_steps[-1]
    assert last == len(tm.log) - 1 and tm.X.shape == (200, 2)
    start = tm.log[opt_steps[-2]][3]            # the hyper-parameters the last search started from
    obj = HO.MapObjective(tm.X, tm.Y, tm.priors)
    to_phi = lambda h: HO.softplus_inv(np.array([h[k] for k in HO.ORDER]))
    printed = {k: DOC_PRINTED[k] for k in HO.ORDER}
    out = dict(X=tm.X, Y=tm.Y,
               priors=np.array([[tm.priors[k][0], tm.priors[k][1]] for k in ('var', 'ls', 'bias')]),
               hyper_start=np.array([start[k] for k in HO.ORDER]),
               hyper_oracle=np.array([tm.hyper[k] for k in HO.ORDER]),
               objective_oracle=np.float64(obj.value(to_phi(tm.hyper))),
               objective_start=np.float64(obj.value(to_phi(start))),
               hyper_printed=np.array([printed[k] for k in HO.ORDER]),
               objective_at_printed=np.float64(obj.value(to_phi(printed))),
               objective_printed_in_doc=np.float64(DOC_PRINTED['objective']),
               threshold_oracle=np.float64(post.threshold), threshold_printed=np.float64(DOC_PRINTED['threshold']),
               n_optimisations=np.int64(len(opt_steps)))
    np.savez_compressed(os.path.join(golden_dir, 'bolfi_doc_run.npz'), **out)
    print('bolfi_doc_run: oracle MAP', {k: round(v, 4) for k, v in tm.hyper.items()}, 'objective %.4f' % out['objective_oracle'],
          '| printed values on this evidence: objective %.4f' % out['objective_at_printed'],
          '| priors (a, b):', {k: tuple(np.round(v, 4)) for k, v in tm.priors.items()},
          '| threshold %.4f (doc %.4f)' % (out['threshold_oracle'], DOC_PRINTED['threshold']))
