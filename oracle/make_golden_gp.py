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

    python oracle/make_golden.py gp        ->  tests/golden/gp_closed_forms.npz, gp_acquisition.npz
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
