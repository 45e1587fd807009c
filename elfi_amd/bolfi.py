"""elfi.BOLFI with the surrogate, the acquisition and the posterior chains on the GPU: a drop-in for the reference's class.

    bolfi = elfi_amd.HipBOLFI(model['d'], batch_size=1, initial_evidence=20, update_interval=10,
                              bounds={'t1': (-2, 2), 't2': (-1, 1)}, acq_noise_var=0.1, seed=1)   # instead of elfi.BOLFI
    post = bolfi.fit(n_evidence=200)            # HipBolfiPosterior
    result = bolfi.sample(1000)                 # the reference's BolfiSample

`HipBOLFI` IS the reference's `elfi.BOLFI` (elfi/methods/inference/bolfi.py:386-580, on top of BayesianOptimization
:24-383) -- a subclass made from the class of the ELFI the running program has imported (as `HipRejection` and the SMC
classes are), so batches, pools, the update / acquire loop, `extract_result`, plotting and the result objects stay the
reference's.  What changes:

  * `__init__` (bolfi.py:28-113): `target_model` defaults to `HipGPRegression` instead of `GPyRegression`, and the default
    acquisition method to `HipLCBSC` over it (same arguments the reference hands its `LCBSC`: prior, noise_var,
    exploration_rate, seed).  Objects the caller passes in are used as they are.
  * `extract_posterior` (bolfi.py:442-462): a `HipBolfiPosterior` (elfi_amd/posterior.py) when the surrogate is a device
    model, the reference's `BolfiPosterior` otherwise.
  * `sample` (bolfi.py:464-580): the reference farms one `mcmc.nuts` / `mcmc.metropolis` call per chain to the client
    (:543-566), each of which calls `posterior.logpdf` / `gradient_logpdf` point by point.  Here the chains advance in
    lock-step (elfi_amd/chains.py): every round is ONE batched device evaluation of value and gradient for all chains.
    Arguments, defaults, checks, error texts, the choice of initial points, the per-chain seeds `get_sub_seed(seed, ii)`,
    the order of random draws inside a chain, the printed diagnostics and the returned `BolfiSample` are the reference's:
    chain ii equals what `mcmc.nuts(..., seed=get_sub_seed(self.seed, ii))` returns on the same surrogate.
"""
import sys

import numpy as np

from .gp import HipGPRegression
from .lcb_acquisition import HipLCBSC
from .posterior import HipBolfiPosterior, sample_posterior

_CLASSES = {}


def _reference_bolfi():
    mod = sys.modules.get('elfi.methods.inference.bolfi')
    if mod is None:
        raise ImportError("HipBOLFI subclasses the running program's elfi.BOLFI: `import elfi` first")
    return mod.BOLFI, mod


def hip_bolfi_class():
    """The subclass of the imported ELFI's BOLFI (made once per reference class)."""
    BOLFI, mod = _reference_bolfi()
    cls = _CLASSES.get(BOLFI)
    if cls is not None:
        return cls
    ModelPrior, BolfiSample, mcmc = mod.ModelPrior, mod.BolfiSample, mod.mcmc

    class HipBOLFI(BOLFI):
        __doc__ = __doc__

        def __init__(self, model, target_name=None, bounds=None, initial_evidence=None, update_interval=10,
                     target_model=None, acquisition_method=None, acq_noise_var=0, exploration_rate=10, batch_size=1,
                     batches_per_acquisition=None, async_acq=False, **kwargs):
            if target_model is None:
                resolved, _ = self._resolve_model(model, target_name)
                target_model = HipGPRegression(resolved.parameter_names, bounds=bounds)
            super(HipBOLFI, self).__init__(model, target_name=target_name, bounds=bounds,
                                           initial_evidence=initial_evidence, update_interval=update_interval,
                                           target_model=target_model, acquisition_method=acquisition_method,
                                           acq_noise_var=acq_noise_var, exploration_rate=exploration_rate,
                                           batch_size=batch_size, batches_per_acquisition=batches_per_acquisition,
                                           async_acq=async_acq, **kwargs)
            if acquisition_method is None and isinstance(self.target_model, HipGPRegression):
                # what bolfi.py:103-107 builds, with the multi-start search on the device
                prior = ModelPrior(self.model, parameter_names=self.target_model.parameter_names)
                self.acquisition_method = HipLCBSC(self.target_model, prior=prior, noise_var=acq_noise_var,
                                                   exploration_rate=exploration_rate, seed=self.seed)

        # -- bolfi.py:442-462 ---------------------------------------------------------------------------------
        def extract_posterior(self, threshold=None):
            if not isinstance(self.target_model, HipGPRegression):
                return super(HipBOLFI, self).extract_posterior(threshold)
            if self.state['n_evidence'] == 0:
                raise ValueError('Model is not fitted yet, please see the `fit` method.')
            prior = ModelPrior(self.model, parameter_names=self.target_model.parameter_names)
            return HipBolfiPosterior(self.target_model, threshold=threshold, prior=prior)

        # -- bolfi.py:464-580 ---------------------------------------------------------------------------------
        def sample(self, n_samples, warmup=None, n_chains=4, threshold=None, initials=None, algorithm='nuts',
                   sigma_proposals=None, n_evidence=None, **kwargs):
            if not isinstance(self.target_model, HipGPRegression):
                return super(HipBOLFI, self).sample(n_samples, warmup=warmup, n_chains=n_chains, threshold=threshold,
                                                    initials=initials, algorithm=algorithm,
                                                    sigma_proposals=sigma_proposals, n_evidence=n_evidence, **kwargs)
            if self.state['n_batches'] == 0:
                self.fit(n_evidence)
            if algorithm not in ['nuts', 'metropolis']:
                raise ValueError("Unknown posterior sampler.")
            prior = ModelPrior(self.model, parameter_names=self.target_model.parameter_names)
            warmup = warmup or n_samples // 2
            chains, posterior = sample_posterior(self.target_model, prior, n_samples, warmup=warmup, n_chains=n_chains,
                                                 threshold=threshold, initials=initials, algorithm=algorithm,
                                                 sigma_proposals=sigma_proposals, seed=self.seed, **kwargs)
            chains = np.asarray(chains)
            print("{} chains of {} iterations acquired. Effective sample size and Rhat for each "
                  "parameter:".format(n_chains, n_samples))
            for ii, node in enumerate(self.target_model.parameter_names):
                print(node, mcmc.eff_sample_size(chains[:, :, ii]), mcmc.gelman_rubin_statistic(chains[:, :, ii]))
            return BolfiSample(method_name='BOLFI', chains=chains, parameter_names=self.target_model.parameter_names,
                               warmup=warmup, threshold=float(posterior.threshold), n_sim=self.state['n_evidence'],
                               seed=self.seed)

    HipBOLFI.__name__ = 'HipBOLFI'
    HipBOLFI.__qualname__ = 'HipBOLFI'
    _CLASSES[BOLFI] = HipBOLFI
    return HipBOLFI


def HipBOLFI(*args, **kwargs):
    """elfi.BOLFI(model, target_name=None, bounds=None, ...) with surrogate, acquisition and posterior chains on the GPU."""
    return hip_bolfi_class()(*args, **kwargs)
