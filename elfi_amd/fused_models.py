"""ELFI's two example models with the whole simulation on the GPU: same priors, same node names, same inference calls.

    m = elfi_amd.fused_models.ma2_model(n_obs=100, seed_obs=4)        # elfi.examples.ma2.get_model(...)
    m = elfi_amd.fused_models.gauss_model(n_obs=50, seed_obs=4)       # elfi.examples.gauss.get_model(...)
    elfi_amd.HipRejection(m['d'], batch_size=10**6, seed=1).sample(1000, n_sim=10**8)

The reference's Simulator node draws the series on the host (MA2: random_state.randn(batch, n_obs + 2),
elfi/examples/ma2.py:35; gauss: ss.norm.rvs(size=(batch, n_obs)), elfi/examples/gauss.py:31-33) -- 0.8 KB per
simulation -- and the Summary nodes reduce it again.  Here the Simulator node's operation IS the fused device kernel
(csrc/summaries.hip ma2 / csrc/gauss.hip): the draws (Philox4x32-10, keyed by a seed taken from the node's own
random_state, so ELFI's batch seeding decides them: elfi/loader.py:164-169), the simulator arithmetic and both summaries
in one pass; the series never exists in memory.  What the node returns is the (batch, 2) array of summaries; the
Summary nodes pick their column and the Distance node is elfi.Distance over elfi_amd.HipDistance.  The observed data are
generated exactly as the reference's get_model does (its own simulator on the host with RandomState(seed_obs)) and
reduced to the observed summaries once, at model construction.

The draws are the device generator's, not MT19937's: a run of this model is a different (equally valid) random
realisation than a run of elfi.examples.*.get_model with the same seed; the arithmetic on given draws is bit-identical
to the reference's (tests/test_summaries_gpu.py, tests/test_gauss_gpu.py).
"""
import sys
from functools import partial

import numpy as np

from .distance import HipDistance, randn_rows
from .priors import SIMULATOR_STREAM_BASE
from .summaries import gauss_distance, ma2_draw_distance


def _elfi():
    mod = sys.modules.get('elfi')
    if mod is None:
        raise ImportError('the fused models are built from the running program\'s ELFI: `import elfi` first')
    return mod


def _seed_of(random_state):
    rs = random_state or np.random
    return int(rs.randint(0, 2 ** 31 - 1))


def ma2_summaries(t1, t2, observed_summaries=(0.0, 0.0), n_obs=100, batch_size=1, random_state=None):
    """Simulator operation: (batch, 2) = [autocov(x, 1), autocov(x, 2)] of MA2 series drawn and reduced on the device."""
    S1, S2, _ = ma2_draw_distance(t1, t2, observed_summaries, n_obs=n_obs, seed=_seed_of(random_state),
                                  stream=SIMULATOR_STREAM_BASE)
    return np.column_stack((S1, S2))


def gauss_summaries(mu, sigma, observed_summaries=(0.0, 0.0), n_obs=50, batch_size=1, random_state=None):
    """Simulator operation: (batch, 2) = [mean(y), var(y)] of Gaussian samples drawn and reduced on the device."""
    S1, S2, _ = gauss_distance(mu, sigma, observed_summaries, n_obs=n_obs, seed=_seed_of(random_state),
                               stream=SIMULATOR_STREAM_BASE)
    return np.column_stack((S1, S2))


def first_column(y):
    return np.asarray(y)[:, 0]


def second_column(y):
    return np.asarray(y)[:, 1]


def ma2_model(n_obs=100, true_params=None, seed_obs=None, device_priors=True):
    """elfi.examples.ma2.get_model(n_obs, true_params, seed_obs) with the simulation on the device; nodes t1, t2, MA2,
    S1, S2, d.  device_priors: the two Prior nodes draw on the device as well (elfi_amd.priors: same distributions, the
    library's generator; False keeps the reference's CustomPrior1 / CustomPrior2 and their SciPy draws)."""
    elfi = _elfi()
    from elfi.examples import ma2
    if true_params is None:
        true_params = [.6, .2]
    y = ma2.MA2(*true_params, n_obs=n_obs, random_state=np.random.RandomState(seed_obs))
    obs = np.array([[ma2.autocov(y)[0], ma2.autocov(y, 2)[0]]])
    m = elfi.ElfiModel()
    from . import priors
    elfi.Prior(priors.MA2Prior1 if device_priors else ma2.CustomPrior1, 2, model=m, name='t1')
    elfi.Prior(priors.MA2Prior2 if device_priors else ma2.CustomPrior2, m['t1'], 1, name='t2')
    elfi.Simulator(partial(ma2_summaries, observed_summaries=tuple(obs[0]), n_obs=n_obs), m['t1'], m['t2'], observed=obs,
                   name='MA2')
    elfi.Summary(first_column, m['MA2'], name='S1')
    elfi.Summary(second_column, m['MA2'], name='S2')
    elfi.Distance(HipDistance('euclidean'), m['S1'], m['S2'], name='d')
    return m


def gauss_model(n_obs=50, true_params=None, seed_obs=None, device_priors=True):
    """elfi.examples.gauss.get_model(n_obs, true_params, seed_obs) (1-D mean and standard deviation) with the simulation
    on the device; nodes mu, sigma, gauss, ss_mean, ss_var, d.  device_priors: the uniform prior of mu draws on the device
    (sigma's truncated normal stays SciPy's)."""
    elfi = _elfi()
    from elfi.examples import gauss
    if true_params is None:
        true_params = [4, .4]
    y = gauss.gauss(*true_params, n_obs=n_obs, random_state=np.random.RandomState(seed_obs))
    obs = np.array([[gauss.ss_mean(y)[0], gauss.ss_var(y)[0]]])
    m = elfi.new_model()
    eps_prior = 5
    from . import priors
    mu = elfi.Prior(priors.uniform if device_priors else 'uniform', true_params[0] - eps_prior, 2 * eps_prior, model=m,
                    name='mu')
    sigma = elfi.Prior('truncnorm', np.amax([.01, true_params[1] - eps_prior]), 2 * eps_prior, model=m, name='sigma')
    elfi.Simulator(partial(gauss_summaries, observed_summaries=tuple(obs[0]), n_obs=n_obs), mu, sigma, observed=obs,
                   name='gauss')
    elfi.Summary(first_column, m['gauss'], name='ss_mean')
    elfi.Summary(second_column, m['gauss'], name='ss_var')
    elfi.Distance(HipDistance('euclidean'), m['ss_mean'], m['ss_var'], name='d')
    return m


def gauss_wide_rows(mu, scale=None, batch_size=1, random_state=None):
    """Simulator operation of the wide synthetic Gaussian model: (batch, m) rows mu[:, None] + scale * z, drawn on the device."""
    mu = np.asarray(mu, dtype=np.float64).reshape(-1)
    if mu.shape[0] == 1 and batch_size > 1:
        mu = np.repeat(mu, batch_size)
    return randn_rows(mu, scale, seed=_seed_of(random_state), stream=SIMULATOR_STREAM_BASE)


def gauss_wide_model(m=64, adaptive=True):
    """BASELINE configs[3]'s model: one location parameter mu ~ U(-10, 10), a synthetic Gaussian simulator with m summaries of
    standard deviations 1 .. 20 drawn ON THE DEVICE, observed zeros; the discrepancy node is a HipAdaptiveDistance over the
    simulator's (batch, m) output (adaptive=False: elfi.Distance(HipDistance('euclidean'))).  Nodes mu, sim, d."""
    elfi = _elfi()
    from .adaptive import hip_adaptive_distance_class
    mdl = elfi.new_model()
    from . import priors
    mu = elfi.Prior(priors.uniform, -10, 20, model=mdl, name='mu')      # (ss.uniform with the draws on the device)
    sim = elfi.Simulator(partial(gauss_wide_rows, scale=np.linspace(1.0, 20.0, m)), mu, observed=np.zeros((1, m)), name='sim')
    if adaptive:
        hip_adaptive_distance_class()(sim, name='d')
    else:
        elfi.Distance(HipDistance('euclidean'), sim, name='d')
    return mdl
