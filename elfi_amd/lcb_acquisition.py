"""LCBSC acquisition for elfi.BOLFI with the multi-start optimisation batched on the GPU.

Mirrors (reference elfi v0.8.7):
    AcquisitionBase.__init__ / acquire / _add_noise   elfi/methods/bo/acquisition.py:24-191
    LCBSC._beta / evaluate / evaluate_gradient        elfi/methods/bo/acquisition.py:226-301
    minimize (start points, arg-min, final clip)      elfi/methods/bo/utils.py:40-111

Use:  elfi.BOLFI(model, target_model=gp, acquisition_method=HipLCBSC(gp, prior=ModelPrior(model),
                 noise_var=..., exploration_rate=10, seed=...), ...)      (bolfi.py:36,103-107)

What changes against the reference is WHERE the work happens, not what is computed: the
reference minimises the acquisition from each of the `n_inits` start points one after the other
with scipy's L-BFGS-B, paying one single-point GP prediction per evaluation; here
`elfihip_gp_lcb_minimize` advances every start in lock-step, one batched device evaluation per
step (include/elfihip.h).  The random-number consumption is the reference's (start points
first, then the truncated-normal jitter), so a seeded run draws the same start points and the
same jitter.
"""
import logging

import numpy as np
import scipy.stats as ss

logger = logging.getLogger(__name__)


def _dist_rank_world():
    try:
        import torch.distributed as dist
    except Exception:
        return 0, 1
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _allgather_starts(locs, vals, iters, n_starts, world, device):
    """All-gather the per-rank optima and restore start order (start s came from rank s % world)."""
    import torch
    import torch.distributed as dist
    d = locs.shape[1]
    per = (n_starts + world - 1) // world
    buf = np.full((per, d + 2), np.inf)
    buf[:len(vals), :d] = locs
    buf[:len(vals), d] = vals
    buf[:len(vals), d + 1] = iters
    from .sharding import collective_device
    t = torch.from_numpy(buf).to(collective_device(device))
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    full_l, full_v, full_i = np.empty((n_starts, d)), np.empty(n_starts), np.empty(n_starts, dtype=np.int32)
    for r, o in enumerate(out):
        o = o.cpu().numpy()
        idx = np.arange(r, n_starts, world)
        full_l[idx], full_v[idx], full_i[idx] = o[:len(idx), :d], o[:len(idx), d], o[:len(idx), d + 1].astype(np.int32)
    return full_l, full_v, full_i


_TRUNCNORM_DIRECT = None   # None: not yet checked in this process; True / False: the direct form reproduces the public call


_PPF_LEAN = None       # None: being checked against SciPy's _ppf call by call; True: trusted; False: not used
_PPF_LEAN_CHECKS = 0
_PPF_LEAN_CALLS = 0    # uses since it became trusted: every 256th is checked again


def _logsumexp2(p, q):
    """scipy.special.logsumexp([p, q], axis=0) for two real 1-element arrays, operation by operation as
    scipy/special/_logsumexp.py forms it (the maximum is taken out, log1p of the rest over the multiplicity, then
    + log(multiplicity) + maximum) -- without the array-namespace machinery that makes up its 45 us per call."""
    if p[0] == q[0]:
        return np.log1p(np.zeros(1)) + np.log(np.full(1, 2.0)) + p
    hi, lo = (p, q) if p[0] > q[0] else (q, p)
    return np.log1p(np.exp(lo - hi)) + np.log(np.ones(1)) + hi


def _truncnorm_ppf_lean(q, a, b):
    """truncnorm_gen._ppf(q, a, b) (scipy/stats/_continuous_distns.py) for 1-element arrays whose interval contains 0
    (a <= 0 <= b, not both 0 -- the jitter of a point inside or on its bounds): the same special functions in the same
    order.  None where SciPy takes its complex-valued branch (an interval on one side of 0)."""
    import scipy.special as sc
    if not (a[0] <= 0.0 <= b[0]) or b[0] <= 0.0:
        return None
    log_mass = sc.log1p(-sc.ndtr(a) - sc.ndtr(-b))
    if a[0] < 0.0:
        return sc.ndtri_exp(_logsumexp2(sc.log_ndtr(a), np.log(q) + log_mass))
    return -sc.ndtri_exp(_logsumexp2(sc.log_ndtr(-b), np.log1p(-q) + log_mass))


def _truncnorm_direct(a, b, loc, scale, random_state):
    """One truncated-normal draw the way `ss.truncnorm.rvs(a, b, loc=loc, scale=scale, size=1, random_state=...)` makes
    it (scipy/stats/_continuous_distns.py, truncnorm_gen._rvs_scalar: U = random_state.uniform(0, 1, N), then the
    distribution's own _ppf, then * scale + loc) without rv_continuous.rvs' argument machinery in front of it
    (350 us per call on this box against 200; a BOLFI acquisition makes one call per parameter).  The quantile itself
    through the lean restatement above (15 us against 100) once that has agreed with SciPy's _ppf bit for bit on its
    first 64 uses in this process (and on every 256th use after that: a disagreement retires it for good)."""
    global _PPF_LEAN, _PPF_LEAN_CHECKS, _PPF_LEAN_CALLS
    U = random_state.uniform(low=0, high=1, size=1)
    x = None
    if _PPF_LEAN is not False:
        a1, b1 = np.asarray(a, dtype=float).reshape(1), np.asarray(b, dtype=float).reshape(1)
        x = _truncnorm_ppf_lean(U, a1, b1)
        if x is not None and _PPF_LEAN:
            _PPF_LEAN_CALLS += 1
        if x is not None and (_PPF_LEAN is None or _PPF_LEAN_CALLS % 256 == 0):
            ref = ss.truncnorm._ppf(U, a, b)
            if np.shape(ref) != np.shape(x) or not np.array_equal(ref, x):
                _PPF_LEAN = False
            elif _PPF_LEAN is None:
                _PPF_LEAN_CHECKS += 1
                if _PPF_LEAN_CHECKS >= 64:
                    _PPF_LEAN = True
            x = ref
    if x is None:
        x = ss.truncnorm._ppf(U, a, b)
    return x * scale + loc


def truncnorm_draw(a, b, loc, scale, size, random_state):
    """`ss.truncnorm.rvs(a, b, loc=loc, scale=scale, size=size, random_state=random_state)` -- for single draws from a
    generator object through the direct form above, once that has reproduced the public call bit for bit (value AND
    generator state) on its first use in this process; anything else goes through the public call."""
    global _TRUNCNORM_DIRECT
    public = lambda: ss.truncnorm.rvs(a, b, loc=loc, scale=scale, size=size, random_state=random_state)
    if size != 1 or _TRUNCNORM_DIRECT is False or not hasattr(random_state, 'get_state') or np.size(a) != 1 \
            or not (np.all(scale > 0) and np.all(a < b)):
        return public()
    if _TRUNCNORM_DIRECT:
        return _truncnorm_direct(a, b, loc, scale, random_state)
    before = None
    try:
        before = random_state.get_state()
        ref = public()
        after = random_state.get_state()
        random_state.set_state(before)
        got = _truncnorm_direct(a, b, loc, scale, random_state)
        now = random_state.get_state()
        same = (np.shape(got) == np.shape(ref) and np.array_equal(got, ref) and now[0] == after[0]
                and np.array_equal(now[1], after[1]) and tuple(now[2:]) == tuple(after[2:]))
        _TRUNCNORM_DIRECT = bool(same)
        if not same:
            random_state.set_state(after)
        return ref
    except Exception:   # a SciPy without these internals: the public call from now on
        _TRUNCNORM_DIRECT = False
        if before is not None:
            random_state.set_state(before)
        return public()


from .elfi_plans import prior_rvs  # noqa: E402  (ModelPrior.rvs replayed from a plan made once: same draws, same generator state)


def draw_start_points(bounds, n, prior=None, random_state=None):
    """Start points of a multi-start search, drawn as the reference's minimize() draws them
    (elfi/methods/bo/utils.py:72-88) so that the random stream is consumed identically: uniform in the bounds
    without a prior (one uniform() call per dimension), else prior.rvs clipped to the bounds."""
    ndim = len(bounds)
    if prior is None:
        rs = random_state or np.random
        pts = np.empty((n, ndim))
        for i in range(ndim):
            pts[:, i] = rs.uniform(*bounds[i], n)
        return pts
    pts = np.asarray(prior_rvs(prior, n, random_state), dtype=float)
    if pts.ndim == 1:
        pts = pts[:, None]
    lo = np.array([b[0] for b in bounds], dtype=float)
    hi = np.array([b[1] for b in bounds], dtype=float)
    return np.clip(pts, lo, hi)


class HipLCBSC:
    """Lower Confidence Bound Selection Criterion (GP-LCB of Srinivas et al.), interface of
    elfi.methods.bo.acquisition.LCBSC."""

    def __init__(self, model, prior=None, n_inits=10, max_opt_iters=1000, noise_var=None,
                 exploration_rate=10, seed=None, constraints=None, delta=None, additive_cost=None):
        if delta is not None:
            if delta <= 0 or delta >= 1:
                logger.warning('Parameter delta should be in the interval (0,1)')
            exploration_rate = 1 / delta
        self.model = model
        self.prior = prior
        self.n_inits = int(n_inits)
        self.max_opt_iters = int(max_opt_iters)
        # constraints: scipy constraint definitions, as the reference hands them to SLSQP (acquisition.py:159-160);
        # additive_cost: an object with evaluate(x) / evaluate_gradient(x) added to the criterion (acquisition.py:278-279,
        # 299-300) -- host callbacks of the user; the GP part of every evaluation still runs on the device
        self.constraints = constraints
        self.additive_cost = additive_cost
        self.noise_var = None if noise_var is None else self._noise_spec(noise_var)
        self.exploration_rate = exploration_rate
        self.random_state = np.random if seed is None else np.random.RandomState(seed)
        self.seed = 0 if seed is None else seed
        self.name = 'lcbsc'
        self.label_fn = 'Confidence Bound'
        self.last_opt = None   # diagnostics of the latest acquire(): per-start optima, iterations
        # Multi-GPU: when torch.distributed is initialised (one process per GPU, every rank holding the
        # same evidence and therefore the same deterministic factorisation), the start points are dealt
        # round-robin over the ranks and the optima are all-gathered (SURVEY.md section 8e, configs[4]).
        self.shard_starts = True
        self.dist_device = None    # None: by the process group's backend (sharding.collective_device)

    # -- argument handling, same accepted forms and error texts as acquisition.py:75-109
    def _noise_spec(self, noise_var):
        """Validate `noise_var`; return a scalar or a per-parameter list in parameter_names order."""
        number = (int, float)
        if isinstance(noise_var, number):
            if noise_var < 0:
                raise ValueError("Acquisition noise should be non-negative int or float.")
            return noise_var
        if not isinstance(noise_var, dict):
            raise ValueError("Either acquisition noise is a float or it is a dictionary of floats "
                             "defining variance for each parameter dimension.")
        names = self.model.parameter_names
        if set(noise_var) != set(names):
            raise ValueError("Acquisition noise dictionary should contain all parameters.")
        values = [noise_var[k] for k in names]
        if not all(isinstance(v, number) for v in values):
            raise ValueError("Acquisition noise dictionary values should all be int or float.")
        if min(values) < 0:
            raise ValueError("Acquisition noises values should all be non-negative int or float.")
        return values

    @property
    def delta(self):
        return 1 / self.exploration_rate

    def _beta(self, t):
        t += 1   # start from 0
        d = self.model.input_dim
        return 2 * np.log(t**(2 * d + 2) * np.pi**2 / (3 * self.delta))

    # -- point-wise interface (plots, tests, other optimisers): acquisition.py:262-301
    def evaluate(self, x, t=None):
        mean, var = self.model.predict(x, noiseless=True)
        value = mean - np.sqrt(self._beta(t) * var)
        if self.additive_cost is not None:
            value += self.additive_cost.evaluate(x)
        return value

    def evaluate_gradient(self, x, t=None):
        mean, var = self.model.predict(x, noiseless=True)
        grad_mean, grad_var = self.model.predictive_gradients(x)
        value = grad_mean - 0.5 * grad_var * np.sqrt(self._beta(t) / var)
        if self.additive_cost is not None:
            value += self.additive_cost.evaluate_gradient(x)
        return value

    def _value_and_gradient(self, X, t):
        """Criterion and gradient at the rows of X: ONE batched device call for the GP part + the user's cost."""
        if getattr(self.model, '_handle', None) is None or self.model.n_evidence == 0:
            # no evidence yet: the reference's GP predicts (0, 1) everywhere (gpy_regression.py:113-116)
            X = np.atleast_2d(X)
            val, grad = np.full(len(X), -np.sqrt(self._beta(t))), np.zeros((len(X), self.model.input_dim))
        else:
            val, grad = self.model.lcb(X, self._beta(t), with_grad=True)
            val, grad = val.reshape(-1), np.array(grad, dtype=float)
        if self.additive_cost is not None:
            val = val + np.asarray(self.additive_cost.evaluate(X), dtype=float).reshape(-1)
            grad = grad + np.asarray(self.additive_cost.evaluate_gradient(X), dtype=float).reshape(grad.shape)
        return val, grad

    def _minimize_on_host(self, t, start_points):
        """The two forms whose search cannot stay inside the library.  additive_cost: the library's L-BFGS-B state
        machines in reverse-communication form (multistart.py), every round one batched device evaluation + the cost.
        constraints: scipy's SLSQP from every start in turn, as the reference does (bo/utils.py:97-103 with
        method='SLSQP', acquisition.py:159), each evaluation one device call."""
        bounds = self.model.bounds
        if self.constraints is None:
            from .multistart import minimize_lockstep
            res = minimize_lockstep(lambda X: self._value_and_gradient(X, t), start_points, bounds,
                                    maxiter=self.max_opt_iters)
            return res['locs'], res['vals'], res['iters'], None
        import scipy.optimize
        locs, vals, iters = [], np.empty(len(start_points)), np.empty(len(start_points), dtype=np.int32)
        fun = lambda x: float(self._value_and_gradient(x[None, :], t)[0][0])
        jac = lambda x: self._value_and_gradient(x[None, :], t)[1][0]
        for i, x0 in enumerate(start_points):
            r = scipy.optimize.minimize(fun, x0, method='SLSQP', jac=jac, bounds=bounds, constraints=self.constraints,
                                        options={'maxiter': self.max_opt_iters})
            locs.append(r['x'])
            vals[i] = r['fun']
            iters[i] = r.get('nit', 0)
        return np.array(locs), vals, iters, None

    def _start_points(self):
        return draw_start_points(self.model.bounds, self.n_inits, self.prior, self.random_state)

    def minimize(self, t=None, start_points=None):
        """All starts in lock-step on the device; returns (xhat, value) like utils.minimize."""
        if start_points is None:
            start_points = self._start_points()
        bounds = self.model.bounds
        no_evidence = getattr(self.model, '_handle', None) is None or self.model.n_evidence == 0
        host_form = self.constraints is not None or self.additive_cost is not None
        if no_evidence and not host_form:
            # no evidence yet: the reference's GP predicts (0, 1) everywhere, every start is a minimum
            self.last_opt = None
            return np.array(start_points[0], dtype=float), float(-np.sqrt(self._beta(t)))
        rank, world = _dist_rank_world() if self.shard_starts else (0, 1)
        mine = np.arange(rank, len(start_points), world)     # start s belongs to rank s % world
        if len(mine) and host_form:   # (also without evidence: the user's cost / constraints still shape the search)
            locs, vals, iters, n_eval = self._minimize_on_host(t, start_points[mine])
        elif len(mine):
            locs, vals, iters, n_eval = self.model._handle.lcb_minimize(start_points[mine], bounds, self._beta(t),
                                                                       maxiter=self.max_opt_iters)
        else:
            locs, vals = np.empty((0, len(bounds))), np.empty(0)
            iters, n_eval = np.empty(0, dtype=np.int32), 0
        if world > 1:
            # ONE exchange: every rank's end points and values (S x (d+2) doubles in total), put
            # back into start order so that the arg-min (ties: first) is the same on every rank
            locs, vals, iters = _allgather_starts(locs, vals, iters, len(start_points), world, self.dist_device)
        ind_min = np.argmin(vals)
        xhat = locs[ind_min].copy()
        for i in range(len(bounds)):
            xhat[i] = np.clip(xhat[i], *bounds[i])
        self.last_opt = dict(starts=start_points, locs=locs, vals=vals, iters=iters, n_eval=n_eval,
                             ind_min=int(ind_min))
        return xhat, float(vals[ind_min])

    def acquire(self, n, t=None):
        """Next batch of acquisition points, (n, input_dim).  acquisition.py:129-172."""
        logger.debug('Acquiring the next batch of %d values', n)
        xhat, _ = self.minimize(t)
        x = np.tile(xhat, (n, 1))
        return self._add_noise(x)

    def _add_noise(self, x):
        """Truncated-normal jitter inside the bounds, one draw call per dimension in index order
        (the reference's random-number consumption, acquisition.py:174-191)."""
        if self.noise_var is None:
            return x
        std = np.sqrt(np.broadcast_to(np.asarray(self.noise_var, dtype=float), (self.model.input_dim,)))
        for i, (lo, hi) in enumerate(self.model.bounds):
            if std[i] == 0:
                continue
            centre = x[:, i]
            x[:, i] = truncnorm_draw((lo - centre) / std[i], (hi - centre) / std[i], centre, std[i], len(x),
                                     self.random_state)
        return x
