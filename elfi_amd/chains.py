"""MCMC chains over a batched log-density: every chain of a run advances in lock-step, so that
one device call evaluates the point each chain is waiting for.

Replaces the way BOLFI.sample runs its chains (elfi/methods/inference/bolfi.py:542-580): the
reference calls mcmc.nuts / mcmc.metropolis (elfi/methods/mcmc.py:114-311, 379-429) once per
chain, and inside a chain every leapfrog step costs two gradient_logpdf and one or two logpdf
calls, each of them a separate single-point GP prediction (posteriors.py:88-189, SURVEY.md 8f
rank 3).  Here a chain is a coroutine that yields the point it needs and is sent
``(logpdf, gradient)`` back; `run_lockstep` gathers the requests of all live chains into ONE
batched evaluation per round (the batched predictor of gp_predict.hip).  A chain keeps the value
and gradient with every tree edge, so a leapfrog step needs exactly one new evaluation.

The samplers follow the reference's algorithms step for step -- NUTS after Hoffman & Gelman
(2014, Algorithm 6) with dual-averaging step-size adaptation, random-walk Metropolis -- with the
reference's argument meaning, defaults, error behaviour and ORDER OF RANDOM DRAWS
(one numpy RandomState per chain: momentum, slice variable, direction, subtree and tree
acceptance), so chain c reproduces ``mcmc.nuts(..., seed=seed_c)`` on the same target
(tests/test_chains.py compares with fixtures recorded from the reference).
"""
import logging

import numpy as np

logger = logging.getLogger(__name__)


def run_lockstep(chains, evaluate):
    """Drive coroutines that yield points and expect ``(logpdf, gradient)``.

    evaluate(X (S, d)) -> (logpdf (S,), gradient (S, d)): one batched call per round.
    Returns the list of the coroutines' return values.  `n_rounds` / `n_points` of the last run are
    kept on the function for diagnostics.
    """
    results = [None] * len(chains)
    waiting = {}
    for i, c in enumerate(chains):
        try:
            waiting[i] = np.asarray(next(c), dtype=float)
        except StopIteration as stop:
            results[i] = stop.value
    rounds = points = 0
    while waiting:
        order = sorted(waiting)
        X = np.stack([waiting[i] for i in order])
        logp, grad = evaluate(X)
        rounds += 1
        points += len(order)
        for k, i in enumerate(order):
            try:
                waiting[i] = np.asarray(chains[i].send((float(logp[k]), np.array(grad[k], dtype=float))),
                                        dtype=float)
            except StopIteration as stop:
                results[i] = stop.value
                del waiting[i]
    run_lockstep.n_rounds, run_lockstep.n_points = rounds, points
    return results


class _Edge:
    """A point of a trajectory with everything a leapfrog step from it needs."""
    __slots__ = ('q', 'p', 'logp', 'grad')

    def __init__(self, q, p, logp, grad):
        self.q, self.p, self.logp, self.grad = q, p, logp, grad


def _leapfrog(edge, step):
    """One leapfrog step of size `step` (signed) from `edge`; a coroutine: one evaluation."""
    p_half = edge.p + 0.5 * step * edge.grad
    q1 = edge.q + step * p_half
    logp1, grad1 = yield q1
    p1 = p_half + 0.5 * step * grad1
    return _Edge(q1, p1, logp1, grad1)


def _subtree(edge, log_u, step, depth, log_joint0, rs, count):
    """Balanced binary tree of 2**depth leapfrog steps from `edge` (mcmc.py:314-376).

    Returns (left, right, proposal, n_ok, ok, mh_sum, n_steps, diverged, outside); the proposal is an
    edge too, so an accepted point brings its value and gradient along.
    """
    if depth == 0:
        new = yield from _leapfrog(edge, step)
        count[0] += 1
        log_joint = new.logp - 0.5 * np.inner(new.p, new.p)
        n_ok = float(log_u <= log_joint)
        ok = log_u < (1000. + log_joint)  # otherwise: the integration error diverged
        outside = False
        if not ok:
            outside = bool(np.isinf(new.logp))  # stepped where the density is zero: not a divergence
            mh = 0.
        else:
            mh = min(1., np.exp(log_joint - log_joint0))
        return new, new, new, n_ok, ok, mh, 1., not ok, outside
    left, right, prop, n_ok, ok, mh, n_steps, div, out = yield from _subtree(edge, log_u, step, depth - 1, log_joint0,
                                                                             rs, count)
    if ok:
        if step < 0:
            left, _, prop2, n_ok2, ok, mh2, n_steps2, div, out = yield from _subtree(left, log_u, step, depth - 1,
                                                                                     log_joint0, rs, count)
        else:
            _, right, prop2, n_ok2, ok, mh2, n_steps2, div, out = yield from _subtree(right, log_u, step, depth - 1,
                                                                                      log_joint0, rs, count)
        if n_ok2 > 0:
            if float(n_ok2) / (n_ok + n_ok2) > rs.rand():
                prop = prop2
        mh += mh2
        n_steps += n_steps2
        span = right.q - left.q
        ok = ok and (np.inner(span, left.p) >= 0) and (np.inner(span, right.p) >= 0)
        n_ok += n_ok2
    return left, right, prop, n_ok, ok, mh, n_steps, div, out


def nuts_chain(n_iter, params0, n_adapt=None, target_prob=0.6, max_depth=5, seed=0, info_freq=100,
               max_retry_inits=20, stepsize=None):
    """One NUTS chain as a coroutine (arguments as mcmc.nuts, mcmc.py:114-160, without the two
    callables).  Returns the (n_iter, d) samples including those drawn during adaptation."""
    rs = np.random.RandomState(seed)
    params0 = np.asarray(params0, dtype=float)
    n_adapt = n_adapt if n_adapt is not None else n_iter // 2
    logp0, grad0 = yield params0
    if np.isinf(logp0):
        raise ValueError("NUTS: Bad initialization point {}, logpdf -> -inf.".format(params0))

    if stepsize is None:  # trial and error from the start point (mcmc.py:171-219)
        tries = 0
        while tries < max_retry_inits:  # may step into a region the prior excludes
            stepsize = np.exp(-tries)
            tries += 1
            momentum0 = rs.randn(*params0.shape)
            start = _Edge(params0, momentum0, logp0, grad0)
            new = yield from _leapfrog(start, stepsize)
            joint0 = logp0 - 0.5 * np.inner(momentum0, momentum0)
            joint1 = new.logp - 0.5 * np.inner(new.p, new.p)
            if np.isfinite(joint1):
                break
            if tries == max_retry_inits:
                raise ValueError("NUTS: Cannot find acceptable stepsize starting from point {}. All "
                                 "trials ended in region with 0 probability.".format(params0))
        plusminus = 1 if np.exp(joint1 - joint0) > 0.5 else -1
        factor = 2. if plusminus == 1 else 0.5
        while factor * np.exp(plusminus * (joint1 - joint0)) > 1.:
            stepsize *= factor
            if stepsize == 0. or stepsize > 1e7:  # bounds as in STAN
                raise SystemExit("NUTS: Found invalid stepsize {} starting from point {}."
                                 .format(stepsize, params0))
            new = yield from _leapfrog(start, stepsize)
            joint1 = new.logp - 0.5 * np.inner(new.p, new.p)

    # dual averaging (Hoffman & Gelman section 3.2; constants as mcmc.py:223-229)
    target_stepsize = np.log(10. * stepsize)
    log_avg_stepsize = 0.
    accept_ratio = 0.
    shrinkage = 0.05
    ii_offset = 10.
    discount = -0.75

    samples = np.empty((n_iter + 1,) + params0.shape)
    samples[0, :] = params0
    cur_logp, cur_grad = logp0, grad0  # value and gradient at the current sample
    n_diverged = n_outside = n_total = 0
    count = [0]  # leapfrog steps (= evaluations) of this chain

    for ii in range(1, n_iter + 1):
        momentum0 = rs.randn(*params0.shape)
        prev = samples[ii - 1, :]
        log_joint0 = cur_logp - 0.5 * np.inner(momentum0, momentum0)
        log_u = log_joint0 - rs.exponential()
        samples[ii, :] = prev
        left = right = _Edge(prev.copy(), momentum0, cur_logp, cur_grad)
        depth = 0
        n_ok = 1
        all_ok = True
        while all_ok and depth <= max_depth:
            direction = 1 if rs.rand() < 0.5 else -1
            if direction == -1:
                left, _, prop, n_sub, sub_ok, mh_ratio, n_steps, is_div, is_out = yield from _subtree(
                    left, log_u, -stepsize, depth, log_joint0, rs, count)
            else:
                _, right, prop, n_sub, sub_ok, mh_ratio, n_steps, is_div, is_out = yield from _subtree(
                    right, log_u, stepsize, depth, log_joint0, rs, count)
            if sub_ok == 1:
                if rs.rand() < float(n_sub) / n_ok:
                    samples[ii, :] = prop.q
                    cur_logp, cur_grad = prop.logp, prop.grad
            n_ok += n_sub
            if not is_out:
                n_diverged += is_div
            n_outside += is_out
            n_total += n_steps
            span = right.q - left.q
            all_ok = sub_ok and (np.inner(span, left.p) >= 0) and (np.inner(span, right.p) >= 0)
            depth += 1

        if ii <= n_adapt:
            accept_ratio = (1. - 1. / (ii + ii_offset)) * accept_ratio \
                + (target_prob - float(mh_ratio) / n_steps) / (ii + ii_offset)
            log_stepsize = target_stepsize - np.sqrt(ii) / shrinkage * accept_ratio
            log_avg_stepsize = ii ** discount * log_stepsize + (1. - ii ** discount) * log_avg_stepsize
            stepsize = np.exp(log_stepsize)
        elif ii == n_adapt + 1:  # adaptation finished: fix the averaged step size
            stepsize = np.exp(log_avg_stepsize)
            n_diverged = n_outside = n_total = 0
        if ii % info_freq == 0 and ii < n_iter:
            logger.info("NUTS: Iterations performed: {}/{}...".format(ii, n_iter))

    if n_total > 0:
        logger.info("NUTS: Acceptance ratio: {:.3f}".format(float(n_iter - n_adapt) / n_total))
    if n_outside > 0:
        logger.info("NUTS: after warmup {} proposals were outside of the region allowed by priors".format(n_outside))
    if n_diverged > 0:
        logger.warning("NUTS: Diverged proposals after warmup (i.e. n_adapt={} steps): {}".format(n_adapt,
                                                                                                  n_diverged))
    return samples[1:, :]


def metropolis_chain(n_samples, params0, sigma_proposals, warmup=0, seed=0):
    """One random-walk Metropolis chain as a coroutine (arguments as mcmc.metropolis,
    mcmc.py:379-429, without the callable).  Returns the (n_samples, d) samples after warmup."""
    rs = np.random.RandomState(seed)
    params0 = np.asarray(params0, dtype=float)
    samples = np.empty((n_samples + warmup + 1,) + params0.shape)
    samples[0, :] = params0
    current, _ = yield params0
    if np.isinf(current):
        raise ValueError("Metropolis: Bad initialization point {},logpdf -> -inf.".format(params0))
    n_accepted = 0
    for ii in range(1, n_samples + warmup + 1):
        samples[ii, :] = samples[ii - 1, :] + sigma_proposals * rs.randn(*params0.shape)
        previous = current
        current, _ = yield samples[ii, :].copy()
        with np.errstate(over='ignore', invalid='ignore'):
            reject = (np.exp(current - previous) < rs.rand()) or np.isinf(current) or np.isnan(current)
        if reject:
            samples[ii, :] = samples[ii - 1, :]
            current = previous
        else:
            n_accepted += 1
    logger.info("{}: Total acceptance ratio: {:.3f}".format(__name__, float(n_accepted) / (n_samples + warmup)))
    return samples[(1 + warmup):, :]


def nuts(n_iter, initials, evaluate, seeds=None, **kwargs):
    """All chains of a NUTS run in lock-step: initials (C, d), evaluate as in `run_lockstep`.
    Returns (C, n_iter, d)."""
    initials = np.atleast_2d(np.asarray(initials, dtype=float))
    seeds = list(range(len(initials))) if seeds is None else list(seeds)
    chains = [nuts_chain(n_iter, x0, seed=s, **kwargs) for x0, s in zip(initials, seeds)]
    return np.asarray(run_lockstep(chains, evaluate))


def metropolis(n_samples, initials, evaluate, sigma_proposals, warmup=0, seeds=None):
    """All chains of a Metropolis run in lock-step.  Returns (C, n_samples, d)."""
    initials = np.atleast_2d(np.asarray(initials, dtype=float))
    seeds = list(range(len(initials))) if seeds is None else list(seeds)
    chains = [metropolis_chain(n_samples, x0, sigma_proposals, warmup=warmup, seed=s)
              for x0, s in zip(initials, seeds)]
    return np.asarray(run_lockstep(chains, evaluate))
