"""MaxVar and RandMaxVar acquisition rules on the device GP.

Mirrors elfi.methods.bo.acquisition.MaxVar / RandMaxVar (elfi/methods/bo/acquisition.py:304-626;
Jarvenpaa et al. 2019).  The acquisition surface is the variance of the unnormalised approximate posterior

    Var(theta) = prior(theta)^2 [ Phi_skew(eps; mu, s, a) - Phi(eps; mu, s)^2 ],
    s^2 = sigma_n^2 + v(theta),   a = sigma_n / sqrt(sigma_n^2 + 2 v(theta)),

mu, v the GP mean and noiseless variance.  The reference evaluates value and gradient separately, one
point per call, three GP predictions per pair (acquisition.py:403-405,430-432), and MaxVar.acquire runs
scipy's L-BFGS-B from each start in turn.  Here ONE batched device call (elfihip_gp_predict_grad) per
round serves value AND gradient of all points of the round, the skew-normal / normal formulas run
vectorised on the host with the SciPy functions the reference uses, the starts of MaxVar.acquire advance in
lock-step (elfi_amd/multistart.py) and the chain of RandMaxVar.acquire spends one device call per
leapfrog step (elfi_amd/chains.py).  Same arguments, attributes, random streams and error texts.
"""
import logging

import numpy as np
import scipy.stats as ss

from . import chains as _chains
from . import multistart as _multistart

logger = logging.getLogger(__name__)


class HipMaxVar:
    """elfi.methods.bo.acquisition.MaxVar (acquisition.py:304-470) for a HipGPRegression."""

    def __init__(self, model, prior, quantile_eps=.01, n_inits=10, max_opt_iters=1000, noise_var=None,
                 exploration_rate=10, seed=None, constraints=None):
        if getattr(model, 'predictive_gradients', None) is None:
            raise TypeError('model must be a GP regression object (elfi_amd.HipGPRegression)')
        if constraints is not None:
            raise NotImplementedError('constraints need the reference MaxVar (SLSQP on the host); it accepts '
                                      'HipGPRegression as model')
        self.model = model
        self.prior = prior
        self.n_inits = int(n_inits)
        self.max_opt_iters = int(max_opt_iters)
        self.constraints = None
        self.noise_var = noise_var          # unused by this family (acquire() is overridden), kept as attribute
        self.exploration_rate = exploration_rate
        self.random_state = np.random if seed is None else np.random.RandomState(seed)
        self.seed = 0 if seed is None else seed
        self.name = 'max_var'
        self.label_fn = 'Variance of the Unnormalised Approximate Posterior'
        self.quantile_eps = quantile_eps
        self.eps = .1  # pre-set until the first acquire() (acquisition.py:346-347)
        self.last_opt = None

    # ---- value and gradient from one prediction ------------------------------------------------
    def _predict(self, theta):
        theta = np.asanyarray(theta, dtype=float).reshape((-1, self.model.input_dim))
        handle = getattr(self.model, '_handle', None)
        if handle is None or self.model.n_evidence == 0:
            mean, var = self.model.predict(theta, noiseless=True)
            grad_mean, grad_var = self.model.predictive_gradients(theta)
        else:
            mean, var, grad_mean, grad_var = handle.predict_grad(theta)  # noiseless variance
        return theta, mean, var, grad_mean, grad_var

    def _value(self, theta, mean, var):
        # acquisition.py:403-417 (the skew-normal cdf stands in for Owen's T function)
        sigma2_n = self.model.noise
        a = np.sqrt(sigma2_n) / np.sqrt(sigma2_n + 2. * var)
        scale = np.sqrt(sigma2_n + var)
        phi_skew = ss.skewnorm.cdf(self.eps, a, loc=mean, scale=scale)
        phi_norm = ss.norm.cdf(self.eps, loc=mean, scale=scale)
        var_p_a = phi_skew - phi_norm ** 2
        val_prior = self.prior.pdf(theta).ravel()[:, np.newaxis]
        return val_prior ** 2 * var_p_a

    def _gradient(self, theta, mean, var, grad_mean, grad_var):
        # acquisition.py:434-463
        phi = ss.norm.cdf
        sigma2_n = self.model.noise
        scale = np.sqrt(sigma2_n + var)
        a = (self.eps - mean) / scale
        b = np.sqrt(sigma2_n) / np.sqrt(sigma2_n + 2 * var)
        grad_a = (-1. / scale) * grad_mean - ((self.eps - mean) / (2. * (sigma2_n + var) ** (1.5))) * grad_var
        grad_b = (-np.sqrt(sigma2_n) / (sigma2_n + 2 * var) ** (1.5)) * grad_var
        _phi_a = phi(a)
        int_1 = _phi_a - _phi_a ** 2
        int_2 = phi(self.eps, loc=mean, scale=scale) - ss.skewnorm.cdf(self.eps, b, loc=mean, scale=scale)
        grad_int_1 = (1. - 2 * _phi_a) * (np.exp(-.5 * (a ** 2)) / np.sqrt(2. * np.pi)) * grad_a
        grad_int_2 = (1. / np.pi) * (((np.exp(-.5 * (a ** 2) * (1. + b ** 2))) / (1. + b ** 2)) * grad_b
                                     + (np.sqrt(np.pi / 2.) * np.exp(-.5 * (a ** 2)) * (1. - 2. * phi(a * b)) * grad_a))
        term_prior = self.prior.pdf(theta).ravel()[:, np.newaxis]
        grad_prior_log = self.prior.gradient_logpdf(theta)
        term_grad_prior = term_prior * grad_prior_log
        return 2. * term_prior * (int_1 - int_2) * term_grad_prior + term_prior ** 2 * (grad_int_1 - grad_int_2)

    def evaluate(self, theta_new, t=None):
        theta, mean, var = self._predict_value_only(theta_new)
        return self._value(theta, mean, var)

    def _predict_value_only(self, theta):
        theta = np.asanyarray(theta, dtype=float).reshape((-1, self.model.input_dim))
        mean, var = self.model.predict(theta, noiseless=True)
        return theta, mean, var

    def evaluate_gradient(self, theta_new, t=None):
        theta, mean, var, gm, gv = self._predict(theta_new)
        return self._gradient(theta, mean, var, gm, gv)

    def value_and_gradient(self, theta):
        """Both at theta (S, d) from one device prediction: (S, 1), (S, d)."""
        theta, mean, var, gm, gv = self._predict(theta)
        return self._value(theta, mean, var), self._gradient(theta, mean, var, gm, gv)

    # ---- acquisition.py:349-384 ----------------------------------------------------------------------
    def acquire(self, n, t=None):
        logger.debug('Acquiring the next batch of %d values', n)
        gp = self.model
        self.eps = np.percentile(gp.Y, self.quantile_eps * 100)

        def negated(theta):
            v, g = self.value_and_gradient(theta)
            return -v.ravel(), -g

        starts = _multistart.draw_start_points(gp.bounds, self.n_inits, self.prior, self.random_state)
        res = _multistart.minimize_lockstep(negated, starts, gp.bounds, maxiter=self.max_opt_iters)
        k = int(np.argmin(res['vals']))
        theta_max = res['locs'][k].copy()
        for i in range(len(gp.bounds)):
            theta_max[i] = np.clip(theta_max[i], *gp.bounds[i])
        self.last_opt = dict(starts=starts, ind_min=k, **res)
        return np.tile(theta_max, (n, 1))


class HipRandMaxVar(HipMaxVar):
    """elfi.methods.bo.acquisition.RandMaxVar (acquisition.py:472-626): the next point is a sample of the
    density proportional to the MaxVar surface, drawn with one NUTS / Metropolis chain."""

    def __init__(self, model, prior, quantile_eps=.01, sampler='nuts', n_samples=50, warmup=None,
                 limit_faulty_init=1000, init_from_prior=False, sigma_proposals=None, **opts):
        super(HipRandMaxVar, self).__init__(model, prior, quantile_eps, **opts)
        self.name = 'rand_max_var'
        self.name_sampler = sampler
        self._n_samples = n_samples
        self._warmup = warmup or n_samples // 2
        self._limit_faulty_init = limit_faulty_init
        self._init_from_prior = init_from_prior
        if self.name_sampler == 'metropolis':
            # resolve_sigmas (elfi/methods/utils.py:460-500)
            if sigma_proposals is None:
                self._sigma_proposals = [(b[1] - b[0]) / 10 for b in self.model.bounds]
            elif isinstance(sigma_proposals, dict):
                if len(sigma_proposals) != len(self.model.parameter_names):
                    raise ValueError("sigma_proposals' keys have to be identical to target_model.parameter_names.")
                self._sigma_proposals = [sigma_proposals[x] for x in self.model.parameter_names]
            else:
                raise ValueError("If provided, sigma_proposals need to be input as a dict.")

    def _log_density_and_gradient(self, theta):
        """log Var(theta) and its gradient for rows theta (acquisition.py:563-575): -inf where the surface is 0."""
        v, g = self.value_and_gradient(theta)
        v = v.ravel()
        with np.errstate(divide='ignore', invalid='ignore'):
            logp = np.where(v == 0, -np.inf, np.log(v))
            grad = np.where((v == 0)[:, None], -np.inf, g / v[:, None])
        return logp, grad

    def acquire(self, n, t=None):
        if n > self._n_samples:
            raise ValueError(("The number of acquisitions ({0}) has to be lower than the number "
                              "of the samples ({1}).").format(n, self._n_samples - self._warmup))
        logger.debug('Acquiring the next batch of %d values', n)
        gp = self.model
        self.eps = np.percentile(gp.Y, self.quantile_eps * 100)
        batch_theta = np.zeros(shape=len(gp.bounds))
        for i in range(self._limit_faulty_init + 1):
            if i == self._limit_faulty_init:
                raise SystemExit("Unable to find a suitable initial point.")
            if self._init_from_prior:
                theta_init = self.prior.rvs(random_state=self.random_state)
                for idx_param, bound in enumerate(gp.bounds):
                    theta_init[idx_param] = np.clip(theta_init[idx_param], bound[0], bound[1])
            else:
                theta_init = np.zeros(shape=len(gp.bounds))
                for idx_param, bound in enumerate(gp.bounds):
                    theta_init[idx_param] = self.random_state.uniform(bound[0], bound[1])
            if np.isinf(self._log_density_and_gradient(theta_init[None, :])[0][0]):
                continue  # a faulty initial point
            if self.name_sampler == 'metropolis':
                chain = _chains.metropolis_chain(self._n_samples, theta_init, np.asarray(self._sigma_proposals),
                                                 seed=self.seed)
            elif self.name_sampler == 'nuts':
                chain = _chains.nuts_chain(self._n_samples, theta_init, seed=self.seed)
            else:
                raise ValueError("Incompatible sampler. Please check the options in the documentation.")
            samples = _chains.run_lockstep([chain], self._log_density_and_gradient)[0]
            if n > 1:
                samples = samples[self._warmup:]
                batch_theta = self.random_state.permutation(samples)[:n]
            else:
                batch_theta = samples[-1:]
            break
        return batch_theta


def finite_difference_objective(fun_batch, bounds, step=1e-8):
    """fun_batch(X (k, d)) -> (k,) turned into the (value, gradient) batch objective of the lock-step search, with
    the forward differences SciPy's L-BFGS-B takes when it is given no gradient, as the reference does for ExpIntVar
    (acquisition.py:766-775 -> scipy.optimize.minimize(jac=None): absolute step 1e-8, the step of a coordinate
    turned backwards where it would leave the upper bound).  All k (d + 1) points of a round are ONE batch."""
    hi = np.array([b[1] for b in bounds], dtype=float)

    def value_and_gradient(X):
        k, d = X.shape
        pts = np.repeat(X, d + 1, axis=0)  # per search: the point, then its d displaced copies
        steps = np.full((k, d), step)
        steps[X + step > hi] = -step
        for i in range(d):
            pts[i + 1::d + 1, i] = X[:, i] + steps[:, i]
        f = np.asarray(fun_batch(pts), dtype=float).reshape(k, d + 1)
        moved = pts.reshape(k, d + 1, d)[:, 1:, :][:, np.arange(d), np.arange(d)] - X
        return f[:, 0], (f[:, 1:] - f[:, :1]) / moved

    return value_and_gradient


class HipExpIntVar(HipMaxVar):
    """elfi.methods.bo.acquisition.ExpIntVar (acquisition.py:629-821): the next point minimises the expected
    integrated variance of the unnormalised posterior over a set of integration points (a grid, or importance
    samples of the MaxVar surface).

    The reference's evaluate() factorises the n x n covariance matrix and solves with it on EVERY call
    (cho_factor + cho_solve, :806-808) to get the GP covariance between the integration points and the candidate.
    Here that covariance comes from the device factorisation already in place: the model keeps
    V_P = L^-1 k(X, P) for the point set (set_integration_points, once per acquire) and cross_cov streams it once
    per batch of candidates (elfihip_gp_cross_cov); all starts of the minimisation and the d + 1 points of each
    finite-difference gradient are one batch per round."""

    def __init__(self, model, prior, quantile_eps=.01, integration='grid', d_grid=.2, n_samples_imp=100, iter_imp=2,
                 sampler='nuts', n_samples=2000, sigma_proposals=None, **opts):
        super(HipExpIntVar, self).__init__(model, prior, quantile_eps, **opts)
        if getattr(model, 'cross_cov', None) is None:
            raise TypeError('model must provide set_integration_points / cross_cov (elfi_amd.HipGPRegression)')
        self.name = 'exp_int_var'
        self.label_fn = 'Expected Loss'
        self._integration = integration
        self._n_samples_imp = n_samples_imp
        self._iter_imp = iter_imp
        if self._integration == 'importance':
            self.density_is = HipRandMaxVar(model=self.model, prior=self.prior, n_inits=self.n_inits, seed=self.seed,
                                            quantile_eps=self.quantile_eps, sampler=sampler, n_samples=n_samples,
                                            sigma_proposals=sigma_proposals)
        elif self._integration == 'grid':
            grid_param = [slice(b[0], b[1], d_grid) for b in self.model.bounds]
            self.points_int = np.mgrid[grid_param].reshape(len(self.model.bounds), -1).T

    def acquire(self, n, t):
        logger.debug('Acquiring the next batch of %d values', n)
        gp = self.model
        self.sigma2_n = gp.noise
        self.eps = np.percentile(gp.Y, self.quantile_eps * 100)
        if self._integration == 'importance' and t % self._iter_imp == 0:
            self.points_int = self.density_is.acquire(self._n_samples_imp)
        self.mean_int, self.var_int = gp.predict(self.points_int, noiseless=True)
        self.priors_int = (self.prior.pdf(self.points_int) ** 2)[np.newaxis, :]
        if self._integration == 'importance' and t % self._iter_imp == 0:
            omegas_int_unnormalised = (1 / HipMaxVar.evaluate(self, self.points_int)).T
            self.omegas_int = omegas_int_unnormalised / np.sum(omegas_int_unnormalised, axis=1)[:, np.newaxis]
        elif self._integration == 'grid':
            self.omegas_int = np.empty(len(self.points_int))
            self.omegas_int.fill(1 / len(self.points_int))
        gp.set_integration_points(self.points_int)  # in place of K, k_int_old and the per-call Cholesky (:788-793)
        self.phi_int = ss.norm.cdf(self.eps, loc=self.mean_int.T, scale=np.sqrt(self.sigma2_n + self.var_int.T))
        starts = _multistart.draw_start_points(gp.bounds, self.n_inits, self.prior, self.random_state)
        res = _multistart.minimize_lockstep(finite_difference_objective(self.evaluate, gp.bounds), starts, gp.bounds,
                                            maxiter=self.max_opt_iters)
        k = int(np.argmin(res['vals']))
        theta_min = res['locs'][k].copy()
        for i in range(len(gp.bounds)):
            theta_min[i] = np.clip(theta_min[i], *gp.bounds[i])
        self.last_opt = dict(starts=starts, ind_min=k, **res)
        return np.tile(theta_min, (n, 1))

    def evaluate(self, theta_new, t=None):
        theta_new = np.asanyarray(theta_new, dtype=float).reshape((-1, self.model.input_dim))
        cov, var_new = self.model.cross_cov(theta_new)           # (M, S), (S,)
        cov_int = cov.T
        var_new = np.asarray(var_new).reshape(-1, 1)
        # acquisition.py:809-819
        delta_var_int = cov_int ** 2 / (self.sigma2_n + var_new)
        a = np.sqrt((self.sigma2_n + self.var_int.T - delta_var_int) / (self.sigma2_n + self.var_int.T + delta_var_int))
        phi_skew_imp = ss.skewnorm.cdf(self.eps, a, loc=self.mean_int.T, scale=np.sqrt(self.sigma2_n + self.var_int.T))
        w = ((self.phi_int - phi_skew_imp) / 2)
        loss_theta_new = 2 * np.sum(self.omegas_int * self.priors_int * w, axis=1)
        return np.where(self.prior.pdf(theta_new) == 0, np.finfo(float).max, loss_theta_new)
