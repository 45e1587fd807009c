"""Lock-step MCMC chains (elfi_amd/chains.py) against chains recorded from the reference's own samplers.

tests/golden/mcmc_chains.npz holds elfi.methods.mcmc.nuts / metropolis output (oracle/make_golden_posterior.py)
for an analytic bounded target; the coroutine chains must reproduce it bit for bit: same algorithm, same
random draws in the same order, same floating-point operations per point.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from elfi_amd import chains  # noqa: E402  (host-only module: importable without the HIP library)

G = np.load(os.path.join(HERE, 'golden', 'mcmc_chains.npz'))
LO, HI, A = G['lo'], G['hi'], G['A']


def _target(x):
    if np.any(x < LO) or np.any(x > HI):
        return -np.inf
    return float(-0.5 * x @ A @ x - 0.1 * x[0] ** 4)


def _grad(x):
    if np.any(x < LO) or np.any(x > HI):
        return np.zeros_like(x)
    g = -A @ x
    g[0] -= 0.4 * x[0] ** 3
    return g


def evaluate(X):
    evaluate.calls += 1
    evaluate.rows += len(X)
    return np.array([_target(x) for x in X]), np.array([_grad(x) for x in X])


evaluate.calls = evaluate.rows = 0


def test_nuts_chains_reproduce_the_reference_bit_for_bit():
    evaluate.calls = evaluate.rows = 0
    ours = chains.nuts(300, G['inits'], evaluate, seeds=G['seeds'].tolist(), n_adapt=150)
    assert ours.shape == G['nuts'].shape == (4, 300, 2)
    assert np.array_equal(ours, G['nuts'])
    # all chains advance together: far fewer batched calls than point evaluations
    assert evaluate.calls == chains.run_lockstep.n_rounds and evaluate.rows == chains.run_lockstep.n_points
    assert evaluate.calls < 0.4 * evaluate.rows


def test_nuts_with_given_stepsize_depth_and_target_probability():
    ours = chains.nuts(120, G['inits'], evaluate, seeds=G['seeds'].tolist(), n_adapt=40, stepsize=0.3, max_depth=3,
                       target_prob=0.7)
    assert np.array_equal(ours, G['nuts_fixed'])


def test_metropolis_chains_reproduce_the_reference_bit_for_bit():
    ours = chains.metropolis(500, G['inits'], evaluate, G['sigma'], warmup=100, seeds=G['seeds'].tolist())
    assert ours.shape == (4, 500, 2) and np.array_equal(ours, G['metropolis'])


def test_a_single_chain_equals_its_row_of_the_batched_run():
    one = chains.nuts(60, G['inits'][2:3], evaluate, seeds=[int(G['seeds'][2])], n_adapt=30)
    assert np.array_equal(one[0], chains.nuts(60, G['inits'], evaluate, seeds=G['seeds'].tolist(), n_adapt=30)[2])


def test_error_behaviour_matches_the_reference():
    outside = np.array([[10., 10.]])
    with pytest.raises(ValueError, match='Bad initialization point'):
        chains.nuts(10, outside, evaluate)
    with pytest.raises(ValueError, match='Bad initialization point'):
        chains.metropolis(10, outside, evaluate, G['sigma'])

    def nowhere(X):  # finite at the start point only: every first leapfrog lands at zero density
        lp = np.array([0.0 if np.allclose(x, 0.0) else -np.inf for x in X])
        return lp, np.ones_like(X)

    with pytest.raises(ValueError, match='Cannot find acceptable stepsize'):
        chains.nuts(5, np.zeros((1, 2)), nowhere, max_retry_inits=4)


@pytest.mark.skipif(not os.path.isdir('/root/reference/elfi'), reason='reference checkout not present')
def test_against_the_live_reference_on_another_target():
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'oracle'))
    import ref_shim
    ref_shim.install()
    from elfi.methods import mcmc

    def t(x):
        return float(-np.sum(np.abs(x) ** 1.5) if np.all(np.abs(x) < 5) else -np.inf)

    def g(x):
        return -1.5 * np.sign(x) * np.abs(x) ** 0.5 if np.all(np.abs(x) < 5) else np.zeros_like(x)

    ev = lambda X: (np.array([t(x) for x in X]), np.array([g(x) for x in X]))
    inits = np.array([[0.3, -0.2, 1.0], [2.0, 2.0, -3.0]])
    ours = chains.nuts(80, inits, ev, seeds=[5, 6])
    for c in range(2):
        assert np.array_equal(ours[c], mcmc.nuts(80, inits[c], t, g, seed=5 + c))


# ---- the reference's own sampler tests (tests/unit/test_mcmc.py:1-46), on the lock-step chains ---------------------
def _gaussian_problem():
    rs = np.random.RandomState(42)
    n = 5
    true_cov = rs.rand(n, n) * 0.5
    true_cov += true_cov.T
    true_cov += n * np.eye(n)
    prec = np.linalg.inv(true_cov)
    evaluate_batch = lambda X: (-0.5 * np.einsum('si,ij,sj->s', X, prec, X), -X @ prec)
    return n, true_cov, evaluate_batch, rs


def test_metropolis_recovers_a_gaussian_covariance():
    n, true_cov, ev, rs = _gaussian_problem()
    n_samples = 30000          # the reference draws 200000 in one chain; here 8 chains of 30000 advance together
    samples = chains.metropolis(n_samples, rs.rand(8, n), ev, np.ones(n), seeds=range(8))
    assert samples.shape == (8, n_samples, n)
    cov = np.cov(samples[:, n_samples // 2:, :].reshape(-1, n).T)
    assert np.allclose(cov, true_cov, atol=0.3, rtol=0.1)


def test_nuts_recovers_a_gaussian_covariance():
    n, true_cov, ev, rs = _gaussian_problem()
    n_samples, n_adapt = 2500, 500
    samples = chains.nuts(n_samples, rs.rand(8, n), ev, seeds=range(8), n_adapt=n_adapt)
    assert samples.shape == (8, n_samples, n)
    cov = np.cov(samples[:, n_adapt:, :].reshape(-1, n).T)
    assert np.allclose(cov, true_cov, atol=0.35, rtol=0.1)
