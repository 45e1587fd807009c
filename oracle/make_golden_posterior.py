"""TEST INFRASTRUCTURE ONLY -- posterior / MCMC fixtures produced by the REFERENCE'S OWN CODE.

    python oracle/make_golden.py posterior   ->  tests/golden/mcmc_chains.npz, bolfi_posterior.npz, maxvar.npz

* mcmc_chains: elfi.methods.mcmc.nuts / metropolis (mcmc.py:114-429), the real functions, on an analytic
  bounded target (a correlated Gaussian with a quartic term, -inf outside a box), four chains each.
* bolfi_posterior: the real elfi.methods.posteriors.BolfiPosterior (posteriors.py:20-212) over the real
  GPyRegression class carrying the oracle's posterior quantities (the stand-in `_gp` of make_golden_gp.py),
  with the real ModelPrior of a two-parameter ElfiModel with uniform priors: logpdf / gradient_logpdf at
  points inside and outside the bounds, for a given threshold and for the threshold the reference finds
  itself (minimum of the GP mean, posteriors.py:66-79).
* maxvar: the real MaxVar / RandMaxVar classes (acquisition.py:304-626) over the same stand-in GP and ModelPrior:
  evaluate / evaluate_gradient at points, MaxVar.acquire (threshold, maximiser), RandMaxVar.acquire with both
  samplers (n = 1: the chain's last point; n = 3: a permutation of the post-warmup samples).
"""
import os

import numpy as np

import gp_oracle as G
from make_golden_gp import standin_gp

LO, HI = np.array([-3., -2.]), np.array([3., 4.])
A = np.array([[2.0, 0.6], [0.6, 1.0]])


def mcmc_target(x):
    x = np.asarray(x, float)
    if np.any(x < LO) or np.any(x > HI):
        return -np.inf
    return float(-0.5 * x @ A @ x - 0.1 * x[0] ** 4)


def mcmc_grad(x):
    x = np.asarray(x, float)
    if np.any(x < LO) or np.any(x > HI):
        return np.zeros_like(x)
    g = -A @ x
    g[0] -= 0.4 * x[0] ** 3
    return g


def main(elfi, golden_dir):
    from elfi.methods import mcmc
    from elfi.methods.bo.gpy_regression import GPyRegression
    from elfi.methods.posteriors import BolfiPosterior
    from elfi.model.extensions import ModelPrior

    inits = np.array([[0.5, 0.5], [-1., 2.], [2., -1.], [0., 3.5]])
    seeds = np.array([11, 12, 13, 14])
    sig = np.array([0.5, 0.7])
    nuts = np.array([mcmc.nuts(300, inits[c], mcmc_target, mcmc_grad, n_adapt=150, seed=int(seeds[c]))
                     for c in range(4)])
    nuts_fixed = np.array([mcmc.nuts(120, inits[c], mcmc_target, mcmc_grad, n_adapt=40, seed=int(seeds[c]),
                                     stepsize=0.3, max_depth=3, target_prob=0.7) for c in range(4)])
    metro = np.array([mcmc.metropolis(500, inits[c], mcmc_target, sig, warmup=100, seed=int(seeds[c]))
                      for c in range(4)])
    np.savez_compressed(os.path.join(golden_dir, 'mcmc_chains.npz'), inits=inits, seeds=seeds, sigma=sig, nuts=nuts,
                        nuts_fixed=nuts_fixed, metropolis=metro, lo=LO, hi=HI, A=A)
    print('mcmc_chains: nuts', nuts.shape, 'metropolis', metro.shape)

    out = {}
    n, d, seed = 200, 2, 31
    X, y, bounds = G.synthetic_gp_problem(n, d, seed=seed)
    h = G.default_hyper(bounds, y)
    post = G.Posterior(X, y, h['var'], h['ls'], h['bias'], h['noise'])
    names = ['p0', 'p1']
    ref = GPyRegression(names, bounds=dict(zip(names, bounds)))
    ref._gp = standin_gp(post)
    ref._kernel_is_default = True
    m = elfi.new_model()
    for nm, (a, b) in zip(names, bounds):
        elfi.Prior('uniform', a, b - a, model=m, name=nm)
    prior = ModelPrior(m, parameter_names=names)
    rs = np.random.RandomState(5)
    xs = rs.uniform(-2.4, 2.4, (24, d))          # a few rows fall outside the bounds [-2, 2]^2
    xs[0] = X[int(np.argmin(y))]
    thr = float(np.min(y) + 0.3)
    bp = BolfiPosterior(ref, threshold=thr, prior=prior)
    out.update(X=X, y=y, hyper=np.array([h['var'], h['ls'], h['bias'], h['noise']]), bounds=np.array(bounds), xs=xs,
               threshold=np.float64(thr),
               logpdf=np.array([bp.logpdf(x) for x in xs], dtype=float),
               grad=np.array([bp.gradient_logpdf(x) for x in xs], dtype=float),
               loglik=np.asarray(bp._unnormalized_loglikelihood(xs), dtype=float),
               gradlik=np.asarray(bp._gradient_unnormalized_loglikelihood(xs), dtype=float),
               prior_logpdf=np.asarray(prior.logpdf(xs), dtype=float))
    bp2 = BolfiPosterior(ref, threshold=None, prior=prior, n_inits=10, seed=0)
    out['threshold_auto'] = np.float64(bp2.threshold)
    out['min_of_mean_over_evidence'] = np.float64(np.min(post.predict(X)[0]))
    np.savez_compressed(os.path.join(golden_dir, 'bolfi_posterior.npz'), **out)
    print('bolfi_posterior: %d points, threshold %.4f, auto threshold %.6f' % (len(xs), thr, bp2.threshold))
    make_maxvar(elfi, golden_dir)


def make_maxvar(elfi, golden_dir):
    from elfi.methods.bo.acquisition import MaxVar, RandMaxVar
    from elfi.methods.bo.gpy_regression import GPyRegression
    from elfi.model.extensions import ModelPrior

    n, d, seed = 150, 2, 41
    X, y, bounds = G.synthetic_gp_problem(n, d, seed=seed)
    h = G.default_hyper(bounds, y)
    post = G.Posterior(X, y, h['var'], h['ls'], h['bias'], h['noise'])
    names = ['p0', 'p1']
    ref = GPyRegression(names, bounds=dict(zip(names, bounds)))
    ref._gp = standin_gp(post)
    ref._kernel_is_default = True
    m = elfi.new_model()
    for nm, (a, b) in zip(names, bounds):
        elfi.Prior('uniform', a, b - a, model=m, name=nm)
    prior = ModelPrior(m, parameter_names=names)
    out = dict(X=X, y=y, hyper=np.array([h['var'], h['ls'], h['bias'], h['noise']]), bounds=np.array(bounds))
    mv = MaxVar(ref, prior, quantile_eps=0.05, n_inits=8, seed=7)
    theta = mv.acquire(2)
    xs = np.random.RandomState(3).uniform(-2, 2, (12, d))
    out.update(eps=np.float64(mv.eps), theta_max=theta, xs=xs, value=mv.evaluate(xs), gradient=mv.evaluate_gradient(xs))
    for sampler in ('nuts', 'metropolis'):
        r1 = RandMaxVar(ref, prior, quantile_eps=0.05, sampler=sampler, n_samples=40, seed=9)
        out['rand_%s_1' % sampler] = r1.acquire(1)
        r3 = RandMaxVar(ref, prior, quantile_eps=0.05, sampler=sampler, n_samples=40, seed=9)
        out['rand_%s_3' % sampler] = r3.acquire(3)
    from elfi.methods.bo.acquisition import ExpIntVar
    ev = ExpIntVar(ref, prior, quantile_eps=0.05, integration='grid', d_grid=0.4, n_inits=5, seed=11)
    th = ev.acquire(1, t=4)
    out.update(eiv_grid_theta=th, eiv_grid_points=ev.points_int, eiv_grid_loss=ev.evaluate(xs),
               eiv_grid_loss_at_theta=ev.evaluate(th))
    ei = ExpIntVar(ref, prior, quantile_eps=0.05, integration='importance', n_samples_imp=24, iter_imp=2,
                   sampler='metropolis', n_samples=60, n_inits=4, seed=13)
    thi = ei.acquire(1, t=2)
    out.update(eiv_imp_theta=thi, eiv_imp_points=ei.points_int, eiv_imp_omegas=ei.omegas_int,
               eiv_imp_loss=ei.evaluate(xs))
    np.savez_compressed(os.path.join(golden_dir, 'maxvar.npz'), **out)
    print('maxvar: eps %.4f theta_max %s' % (mv.eps, theta[0]))
