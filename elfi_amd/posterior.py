"""The BOLFI posterior on the device GP, evaluated for many points at once, and its sampler.

Mirror of elfi.methods.posteriors.BolfiPosterior (elfi/methods/posteriors.py:20-212) and of the
chain loop of BOLFI.sample (elfi/methods/inference/bolfi.py:463-580) -- SURVEY.md 8f rank 3.

    L(theta)  ~  F((h - mu(theta)) / sigma(theta))           F = standard normal cdf,

mu, sigma^2 the GP mean and NOISY variance, h the threshold.  The reference evaluates logpdf and
gradient_logpdf one point at a time, and each of them predicts again (two to three single-point GP
predictions per call, posteriors.py:149,170,173).  Here ONE batched device call
(elfihip_gp_predict_grad: mean, variance and both gradients of all S points) feeds both, the few
normal-cdf formulas run vectorised on the host with the same SciPy functions the reference uses, and
the MCMC chains of a run advance in lock-step over it (elfi_amd/chains.py).
"""
import numpy as np
import scipy.special as sp

from . import chains as _chains
from .lcb_acquisition import draw_start_points


_SQRT_2PI = np.sqrt(2 * np.pi)


class _UniformBoxPrior:
    """Uniform density on the model bounds (what BolfiPosterior documents as its default prior)."""

    def __init__(self, bounds):
        self.lo = np.array([b[0] for b in bounds], dtype=float)
        self.hi = np.array([b[1] for b in bounds], dtype=float)
        self._logc = -float(np.sum(np.log(self.hi - self.lo)))

    def rvs(self, size=None, random_state=None):
        rs = random_state or np.random
        return rs.uniform(self.lo, self.hi, (size or 1, len(self.lo)))

    def logpdf(self, x):
        x = np.asanyarray(x, dtype=float).reshape((-1, len(self.lo)))
        inside = np.all((x >= self.lo) & (x <= self.hi), axis=1)
        return np.where(inside, self._logc, -np.inf)

    def gradient_logpdf(self, x):
        return np.zeros_like(np.asanyarray(x, dtype=float).reshape((-1, len(self.lo))))


def prior_logpdf_and_gradient(prior, x):
    """(prior.logpdf(x), prior.gradient_logpdf(x)) for the rows of x (S, d).

    ELFI's ModelPrior has no analytic gradient: gradient_logpdf (elfi/model/extensions.py:213-240) loops over the ROWS and
    takes numgrad (elfi/methods/utils.py:275-314) of logpdf for each -- one pass of ELFI's executor over the prior net
    per row (0.3-0.9 ms each), S + 1 passes per evaluation round of S chains where the GP part of the round is one device
    call of well under 0.1 ms.  Here all rows' stencils (3 d points each, the reference's step 1e-5) and the rows
    themselves go through ONE logpdf call; the central differences, the "any -inf in the stencil -> zero gradient" rule
    and the inf / nan replacement are the reference's, value for value (tests/test_host_logic.py).  Any other prior
    object: its own two methods."""
    x = np.asanyarray(x, dtype=float)
    if not (hasattr(prior, '_logpdf_net') and hasattr(prior, 'dim') and type(prior).__module__.startswith('elfi.')):
        return (np.asarray(prior.logpdf(x), dtype=float).reshape(-1),
                np.asarray(prior.gradient_logpdf(x), dtype=float).reshape(x.shape))
    S, d = x.shape
    h = 0.00001
    pts = np.empty((S, 3 * d + 1, d))
    pts[:, 0, :] = x
    for i in range(3):                      # numgrad: X[i d:(i + 1) d] = tile(x) with (i - 1) h added on the diagonal
        Xi = np.repeat(x[:, None, :], d, axis=1)
        idx = np.arange(d)
        Xi[:, idx, idx] = Xi[:, idx, idx] + (i - 1) * h
        pts[:, 1 + i * d:1 + (i + 1) * d, :] = Xi
    from .elfi_plans import prior_logpdf        # (the pass itself from a plan made once per batch size: elfi_plans.py)
    vals = np.asarray(prior_logpdf(prior, pts.reshape(-1, d)), dtype=float).reshape(S, 3 * d + 1)
    logp = vals[:, 0].copy()
    f = vals[:, 1:].reshape(S, 3, d)
    with np.errstate(invalid='ignore'):
        grad = np.gradient(f, h, axis=1)[:, 1, :]
    grad[np.any(np.isneginf(f), axis=(1, 2))] = 0.0
    grad[np.isinf(grad)] = 0
    grad[np.isnan(grad)] = 0
    return logp, grad


class HipBolfiPosterior:
    """BolfiPosterior (posteriors.py:20-212) on a HipGPRegression: same constructor, attributes and
    point-wise methods (logpdf, pdf, gradient_logpdf and the `_unnormalized_*` helpers, same output
    shapes), plus `logpdf_and_gradient` for a batch in one device call."""

    def __init__(self, model, threshold=None, prior=None, n_inits=10, max_opt_iters=1000, seed=0):
        if getattr(model, '_handle', None) is None:
            raise TypeError('HipBolfiPosterior needs a fitted elfi_amd.HipGPRegression (the device GP); '
                            'got %r' % (type(model).__name__,))
        self.threshold = threshold
        self.model = model
        self.random_state = np.random.RandomState(seed)
        self.n_inits = n_inits
        self.max_opt_iters = max_opt_iters
        self.prior = prior if prior is not None else _UniformBoxPrior(model.bounds)
        self.dim = self.model.input_dim
        self._lo = np.array([b[0] for b in model.bounds], dtype=float)
        self._hi = np.array([b[1] for b in model.bounds], dtype=float)
        if self.threshold is None:
            # minimum of the GP mean (posteriors.py:66-79): the reference's minimize() draws the start points
            # from the prior and runs L-BFGS-B from each; beta = 0 turns the LCB into the mean, so this is the
            # same multi-start search as an acquisition (gp_acq.hip), all starts in lock-step
            starts = draw_start_points(model.bounds, self.n_inits, prior, self.random_state)
            locs, vals, _, _ = model._handle.lcb_minimize(starts, model.bounds, 0.0, maxiter=self.max_opt_iters)
            self.threshold = float(vals[int(np.argmin(vals))])

    # ---- batched core -------------------------------------------------------------------------
    def _within_bounds(self, x):
        x = x.reshape((-1, self.dim))
        return np.all((x >= self._lo) & (x <= self._hi), axis=1)

    def _likelihood_terms(self, x, with_grad):
        """x (S, d) inside the bounds -> loglik (S,), grad (S, d) or None: posteriors.py:149-150,170-185."""
        if with_grad:
            mean, var, grad_mean, grad_var = self.model._handle.predict_grad(x)
            var = var + self.model.noise  # predict() includes the noise (gpy_regression.py:139)
        else:
            mean, var = self.model._handle.predict(x, noiseless=False)
            grad_mean = grad_var = None
        # scipy.stats.norm.logcdf / pdf / cdf are these special functions behind argument checking that costs more
        # than the device call: norm.logcdf(h, m, s) = log_ndtr((h - m) / s), norm.cdf = ndtr,
        # norm.pdf(t) = exp(-t**2 / 2) / sqrt(2 pi)  -- the same values, bit for bit
        std = np.sqrt(var)
        term = (self.threshold - mean) / std
        loglik = sp.log_ndtr(term).reshape(-1)
        if not with_grad:
            return loglik, None
        factor = -grad_mean * std - (self.threshold - mean) * 0.5 * grad_var / std
        factor = factor / var
        with np.errstate(divide='ignore', invalid='ignore'):
            grad = factor * (np.exp(-term ** 2 / 2.0) / _SQRT_2PI) / sp.ndtr(term)
        return loglik, grad

    def logpdf_and_gradient(self, x):
        """Unnormalised log posterior and its gradient at x (S, d): one device prediction for all rows.

        Rows outside the model bounds get -inf and a zero likelihood gradient (as the reference)."""
        x = np.ascontiguousarray(np.asanyarray(x, dtype=float).reshape((-1, self.dim)))
        logp = np.full(len(x), -np.inf)
        grad = np.zeros_like(x)
        inside = self._within_bounds(x)
        if np.any(inside):
            logp[inside], grad[inside] = self._likelihood_terms(x[inside], True)
        plog, pgrad = prior_logpdf_and_gradient(self.prior, x)
        logp = logp + plog
        grad = grad + pgrad
        return logp, grad

    # ---- the reference's point-wise interface ------------------------------------------------------
    def _scalar_out(self, ndim):
        return ndim == 0 or (ndim == 1 and self.dim > 1)

    def _unnormalized_loglikelihood(self, x):
        x = np.asanyarray(x)
        ndim = x.ndim
        x = np.ascontiguousarray(x.reshape((-1, self.dim)), dtype=float)
        logpdf = -np.ones(len(x)) * np.inf
        inside = self._within_bounds(x)
        if np.any(inside):
            logpdf[inside] = self._likelihood_terms(x[inside], False)[0]
        return logpdf[0] if self._scalar_out(ndim) else logpdf

    def _gradient_unnormalized_loglikelihood(self, x):
        x = np.asanyarray(x)
        ndim = x.ndim
        x = np.ascontiguousarray(x.reshape((-1, self.dim)), dtype=float)
        grad = np.zeros_like(x)
        inside = self._within_bounds(x)
        if np.any(inside):
            grad[inside] = self._likelihood_terms(x[inside], True)[1]
        return grad[0] if self._scalar_out(ndim) else grad

    def logpdf(self, x):
        return self._unnormalized_loglikelihood(x) + self.prior.logpdf(x)

    def pdf(self, x):
        return np.exp(self.logpdf(x))

    def gradient_logpdf(self, x):
        return self._gradient_unnormalized_loglikelihood(x) + self.prior.gradient_logpdf(x)

    def rvs(self, size=None, random_state=None):
        raise NotImplementedError('Currently not implemented. Please use a sampler to sample from the posterior.')

    def _unnormalized_likelihood(self, x):
        return np.exp(self._unnormalized_loglikelihood(x))

    def _neg_unnormalized_loglikelihood(self, x):
        return -1 * self._unnormalized_loglikelihood(x)

    def _gradient_neg_unnormalized_loglikelihood(self, x):
        return -1 * self._gradient_unnormalized_loglikelihood(x)

    def _neg_unnormalized_logposterior(self, x):
        return -1 * self.logpdf(x)

    def _gradient_neg_unnormalized_logposterior(self, x):
        return -1 * self.gradient_logpdf(x)


def sub_seed(seed, index, high=2 ** 31):
    """The index-th distinct sub seed of `seed`: elfi.utils.get_sub_seed (elfi/utils.py:71-127, no cache)."""
    if index >= high:
        raise ValueError("Sub seed index {} is out of range".format(index))
    rs = np.random.RandomState(seed)
    seen, last = set(), None
    while len(seen) != index + 1:
        draws = rs.randint(high, size=index + 1 - len(seen), dtype='uint32')
        seen.update(draws)
        last = draws
    return last[-1]


def sample_posterior(model, prior, n_samples, warmup=None, n_chains=4, threshold=None, initials=None,
                     algorithm='nuts', sigma_proposals=None, seed=0, **kwargs):
    """The chains of BOLFI.sample (bolfi.py:463-580) for a fitted HipGPRegression, all chains in lock-step.

    model / prior: what BOLFI holds as `target_model` and `ModelPrior(self.model, ...)`; the other arguments have
    the reference's meaning and defaults (n_samples per chain INCLUDING warmup; initials default to the evidence
    points of smallest discrepancy, points of zero posterior density skipped; chain ii runs with
    get_sub_seed(seed, ii); Metropolis proposal deviations default to 1/10 of the bound lengths).
    Returns (chains (n_chains, n_samples, d), posterior): `chains` is what the reference hands to BolfiSample.
    """
    if algorithm not in ['nuts', 'metropolis']:
        raise ValueError("Unknown posterior sampler.")
    posterior = HipBolfiPosterior(model, threshold=threshold, prior=prior)
    warmup = warmup or n_samples // 2
    d = model.input_dim
    if initials is not None:
        if np.asarray(initials).shape != (n_chains, d):
            raise ValueError("The shape of initials must be (n_chains, n_params).")
        pool = np.asarray(initials, dtype=float)
    else:
        pool = np.asarray(model.X[np.argsort(model.Y[:, 0])], dtype=float)
    # discard start points of zero density: one batched evaluation of the candidates instead of one call each
    usable = ~np.isinf(posterior.logpdf_and_gradient(pool[:max(4 * n_chains, 64)])[0])
    picks, k = [], 0
    for _ in range(n_chains):
        while True:
            if k == len(pool):
                raise ValueError("BOLFI.sample: Cannot find enough acceptable initialization points!")
            if k >= len(usable):  # beyond the first batch: the rare case, evaluate singly
                ok = not np.isinf(posterior.logpdf_and_gradient(pool[k:k + 1])[0][0])
            else:
                ok = bool(usable[k])
            if ok:
                break
            k += 1
        picks.append(k)
        k += 1
    starts = pool[picks]
    seeds = [sub_seed(seed, ii) for ii in range(n_chains)]
    model.is_sampling = True
    try:
        if algorithm == 'nuts':
            out = _chains.nuts(n_samples, starts, posterior.logpdf_and_gradient, seeds=seeds, n_adapt=warmup, **kwargs)
        else:
            # resolve_sigmas (elfi/methods/utils.py:460-500): a dict keyed by parameter name, default bound length / 10
            if sigma_proposals is None:
                sig = np.array([(b[1] - b[0]) / 10 for b in model.bounds])
            elif isinstance(sigma_proposals, dict):
                if len(sigma_proposals) != len(model.parameter_names):
                    raise ValueError("sigma_proposals' keys have to be identical to target_model.parameter_names.")
                sig = np.array([sigma_proposals[name] for name in model.parameter_names], dtype=float)
            else:
                raise ValueError("If provided, sigma_proposals need to be input as a dict.")
            out = _chains.metropolis(n_samples, starts, posterior.logpdf_and_gradient, sig, warmup=warmup, seeds=seeds)
    finally:
        model.is_sampling = False
    return out, posterior
