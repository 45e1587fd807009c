"""MaxVar, RandMaxVar and ExpIntVar acquisition rules over the device GP.

Drop-ins for elfi.methods.bo.acquisition.MaxVar / RandMaxVar / ExpIntVar (elfi/methods/bo/acquisition.py:304-821;
Jarvenpaa et al. 2019): same constructor arguments, attributes (`eps`, `quantile_eps`, `points_int`, `omegas_int`,
`name`, `label_fn` ...), random streams and error behaviour, `acquire(n, t)` / `evaluate` / `evaluate_gradient`.

What this module holds is orchestration only.  The arithmetic of the three rules lives behind the model:

    model.maxvar_surface(theta, eps, prior_pdf, prior_grad_logpdf)   -> value (S, 1), gradient (S, d)
    model.expintvar_loss(theta, eps, w_int, mean_int, var_int)       -> loss (S,)
    model.set_integration_points(points)

which HipGPRegression answers with ONE batched device call each (elfihip_gp_maxvar / elfihip_gp_expintvar: prediction,
gradients and the skew-normal / Owen's-T epilogue on the GPU; the reference spends three single-point GP predictions
and SciPy's skewnorm per value / gradient pair, and a Cholesky of the n x n matrix per ExpIntVar evaluation).  The prior
is the caller's object (elfi ModelPrior): its density and log-gradient at a round's points are evaluated here, on the
host, and handed to the device call.  The searches run in lock-step (elfi_amd/multistart.py), the sampling chain of
RandMaxVar as a coroutine with one device call per leapfrog step (elfi_amd/chains.py).
"""
import logging

import numpy as np

from . import chains, multistart

logger = logging.getLogger(__name__)


def _as_points(model, theta):
    return np.asanyarray(theta, dtype=float).reshape((-1, model.input_dim))


def _clipped(point, bounds):
    lo, hi = np.array(bounds, dtype=float).T
    return np.minimum(np.maximum(point, lo), hi)


class _SurfaceRule:
    """What the three rules share: argument handling, the threshold eps (a quantile of the evidence) and the
    variance-of-the-unnormalised-posterior surface with its gradient, both from one model call."""

    name, label_fn = None, None

    def __init__(self, model, prior, quantile_eps=.01, n_inits=10, max_opt_iters=1000, noise_var=None,
                 exploration_rate=10, seed=None, constraints=None):
        for needed in ('maxvar_surface', 'expintvar_loss', 'set_integration_points'):
            if not callable(getattr(model, needed, None)):
                raise TypeError('model must evaluate the acquisition surfaces itself (elfi_amd.HipGPRegression); '
                                'missing: %s' % needed)
        self.model, self.prior = model, prior
        self.n_inits, self.max_opt_iters = int(n_inits), int(max_opt_iters)
        # stored as AcquisitionBase stores it (acquisition.py:65); this family's acquire() never hands it to its search
        # (acquisition.py:378-384 calls minimize() without constraints), so neither does this one
        self.constraints = constraints
        # kept as attributes; this family never jitters its acquisitions
        self.noise_var, self.exploration_rate = noise_var, exploration_rate
        self.seed, self.random_state = (0, np.random) if seed is None else (seed, np.random.RandomState(seed))
        self.quantile_eps, self.eps = quantile_eps, .1     # eps: until the first acquire() (acquisition.py:346-347)
        self.last_opt = None

    def _refresh_eps(self):
        self.eps = np.percentile(self.model.Y, self.quantile_eps * 100)

    def value_and_gradient(self, theta):
        """Surface and gradient at the rows of theta: (S, 1), (S, d) -- one device call."""
        theta = _as_points(self.model, theta)
        # (ELFI's ModelPrior: the density through a plan made once -- elfi_plans.py --, its numeric log-gradient, one
        # executor pass per ROW in the reference, for all rows in one pass; value for value the public calls)
        from .elfi_plans import prior_logpdf
        from .posterior import prior_logpdf_and_gradient
        pdf = np.ravel(prior_logpdf(self.prior, theta, log=False))
        return self.model.maxvar_surface(theta, self.eps, pdf, prior_logpdf_and_gradient(self.prior, theta)[1])

    def evaluate(self, theta_new, t=None):
        return self.value_and_gradient(theta_new)[0]

    def evaluate_gradient(self, theta_new, t=None):
        return self.value_and_gradient(theta_new)[1]

    def _search(self, objective):
        """Lock-step multi-start minimisation from the reference's start points; the arg-min, clipped."""
        bounds = self.model.bounds
        starts = multistart.draw_start_points(bounds, self.n_inits, self.prior, self.random_state)
        res = multistart.minimize_lockstep(objective, starts, bounds, maxiter=self.max_opt_iters)
        best = int(np.argmin(res['vals']))
        self.last_opt = dict(starts=starts, ind_min=best, **res)
        return _clipped(res['locs'][best], bounds)


class HipMaxVar(_SurfaceRule):
    """elfi.methods.bo.acquisition.MaxVar (acquisition.py:304-470): the next point maximises the surface."""

    name = 'max_var'
    label_fn = 'Variance of the Unnormalised Approximate Posterior'

    def acquire(self, n, t=None):
        logger.debug('MaxVar: acquiring %d point(s)', n)
        self._refresh_eps()

        def negated(theta):
            value, grad = self.value_and_gradient(theta)
            return -value.ravel(), -grad

        return np.tile(self._search(negated), (n, 1))


class HipRandMaxVar(_SurfaceRule):
    """elfi.methods.bo.acquisition.RandMaxVar (acquisition.py:472-626): the next point is a draw from the density
    proportional to the surface, taken from one NUTS or Metropolis chain."""

    name = 'rand_max_var'
    label_fn = HipMaxVar.label_fn

    def __init__(self, model, prior, quantile_eps=.01, sampler='nuts', n_samples=50, warmup=None,
                 limit_faulty_init=1000, init_from_prior=False, sigma_proposals=None, **opts):
        super().__init__(model, prior, quantile_eps, **opts)
        self.name_sampler, self._n_samples, self._warmup = sampler, n_samples, warmup or n_samples // 2
        self._limit_faulty_init, self._init_from_prior = limit_faulty_init, init_from_prior
        if sampler == 'metropolis':
            self._sigma_proposals = self._proposal_widths(sigma_proposals)

    def _proposal_widths(self, given):
        # what resolve_sigmas does for this caller (elfi/methods/utils.py:460-500): a tenth of each bound by default
        if given is None:
            return [(hi - lo) / 10 for lo, hi in self.model.bounds]
        if not isinstance(given, dict):
            raise ValueError("If provided, sigma_proposals need to be input as a dict.")
        names = self.model.parameter_names
        if len(given) != len(names):
            raise ValueError("sigma_proposals' keys have to be identical to target_model.parameter_names.")
        return [given[k] for k in names]

    def _log_density(self, theta):
        """(log surface, its gradient) per row; -inf where the surface vanishes (acquisition.py:563-575)."""
        value, grad = self.value_and_gradient(theta)
        value = value.ravel()
        dead = value == 0
        with np.errstate(divide='ignore', invalid='ignore'):
            return np.where(dead, -np.inf, np.log(value)), np.where(dead[:, None], -np.inf, grad / value[:, None])

    def _initial_point(self):
        bounds = self.model.bounds
        if self._init_from_prior:
            return _clipped(np.asarray(self.prior.rvs(random_state=self.random_state), dtype=float), bounds)
        return np.array([self.random_state.uniform(lo, hi) for lo, hi in bounds])

    def _chain(self, start):
        if self.name_sampler == 'metropolis':
            return chains.metropolis_chain(self._n_samples, start, np.asarray(self._sigma_proposals), seed=self.seed)
        if self.name_sampler == 'nuts':
            return chains.nuts_chain(self._n_samples, start, seed=self.seed)
        raise ValueError("Incompatible sampler. Please check the options in the documentation.")

    def acquire(self, n, t=None):
        if n > self._n_samples:
            raise ValueError(("The number of acquisitions ({0}) has to be lower than the number "
                              "of the samples ({1}).").format(n, self._n_samples - self._warmup))
        logger.debug('RandMaxVar: acquiring %d point(s) from one %s chain', n, self.name_sampler)
        self._refresh_eps()
        for _ in range(self._limit_faulty_init):
            start = self._initial_point()
            if np.isinf(self._log_density(start[None, :])[0][0]):
                continue                                   # the surface vanishes there: draw another start
            samples = chains.run_lockstep([self._chain(start)], self._log_density)[0]
            if n == 1:
                return samples[-1:]
            return self.random_state.permutation(samples[self._warmup:])[:n]
        raise SystemExit("Unable to find a suitable initial point.")


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


class HipExpIntVar(_SurfaceRule):
    """elfi.methods.bo.acquisition.ExpIntVar (acquisition.py:629-821): the next point minimises the expected
    integrated variance of the unnormalised posterior over a set of integration points -- a grid, or importance
    samples of the MaxVar surface refreshed every `iter_imp` acquisitions.

    Per acquire(): the point set goes to the model once (set_integration_points: V_P = L^-1 k(X, P) on the device),
    with the GP mean / variance there and the weights omega_i prior_i^2; every evaluation of the loss -- all starts of
    the minimisation and the d + 1 points of each finite-difference gradient in one batch -- is one expintvar_loss
    call (the reference factorises the n x n matrix and solves with it inside every evaluate(), :806-808)."""

    name = 'exp_int_var'
    label_fn = 'Expected Loss'

    def __init__(self, model, prior, quantile_eps=.01, integration='grid', d_grid=.2, n_samples_imp=100, iter_imp=2,
                 sampler='nuts', n_samples=2000, sigma_proposals=None, **opts):
        super().__init__(model, prior, quantile_eps, **opts)
        self._integration, self._n_samples_imp, self._iter_imp = integration, n_samples_imp, iter_imp
        if integration == 'importance':
            self.density_is = HipRandMaxVar(model=model, prior=prior, n_inits=self.n_inits, seed=self.seed,
                                            quantile_eps=quantile_eps, sampler=sampler, n_samples=n_samples,
                                            sigma_proposals=sigma_proposals)
        elif integration == 'grid':
            axes = [slice(lo, hi, d_grid) for lo, hi in model.bounds]
            self.points_int = np.mgrid[axes].reshape(len(model.bounds), -1).T

    def _prepare_integration(self, t):
        resample = self._integration == 'importance' and t % self._iter_imp == 0
        if resample:
            self.points_int = self.density_is.acquire(self._n_samples_imp)
        pts = self.points_int
        self.mean_int, self.var_int = self.model.predict(pts, noiseless=True)
        self.priors_int = (self.prior.pdf(pts) ** 2)[np.newaxis, :]
        if resample:
            inverse = (1 / self.value_and_gradient(pts)[0]).T
            self.omegas_int = inverse / np.sum(inverse, axis=1)[:, np.newaxis]
        elif self._integration == 'grid':
            self.omegas_int = np.full(len(pts), 1 / len(pts))
        self.model.set_integration_points(pts)
        self._w_int = np.ravel(self.omegas_int * self.priors_int)

    def acquire(self, n, t):
        logger.debug('ExpIntVar: acquiring %d point(s), t = %s', n, t)
        self.sigma2_n = self.model.noise
        self._refresh_eps()
        self._prepare_integration(t)
        return np.tile(self._search(finite_difference_objective(self.evaluate, self.model.bounds)), (n, 1))

    def evaluate(self, theta_new, t=None):
        theta_new = _as_points(self.model, theta_new)
        loss = self.model.expintvar_loss(theta_new, self.eps, self._w_int, self.mean_int, self.var_int)
        return np.where(self.prior.pdf(theta_new) == 0, np.finfo(float).max, loss)

    def evaluate_gradient(self, theta_new, t=None):
        raise NotImplementedError('ExpIntVar is minimised on finite differences, as in the reference '
                                  '(acquisition.py:766-775 passes no gradient)')
