"""The SMC-ABC samplers of the reference with their per-round arithmetic on the GPU: drop-ins for
elfi.SMC / elfi.AdaptiveDistanceSMC / elfi.AdaptiveThresholdSMC.

    smc = elfi_amd.HipAdaptiveDistanceSMC(d, batch_size=10**6, seed=1)      # d: a HipAdaptiveDistance node
    sample = smc.sample(1000, 5, quantile=0.5)                               # elfi.AdaptiveDistanceSMC's own signature

Each class IS the reference's class (elfi/methods/inference/samplers.py:319-549, 562-660, 663-860) -- a subclass made
from the class of the ELFI the running program has imported: rounds, objectives, the proposal draws from the previous
population (GMDistribution.rvs: sequential host random numbers, :441-451), the bookkeeping of populations and the
result objects are the reference's code.  What changes:

  * `_set_rejection_round` (samplers.py:474-487) builds an `elfi_amd.HipRejection` instead of a `Rejection`: the
    round's sample state lives on the device (csrc/reject.hip); with an AdaptiveDistanceSMC the objective's
    `threshold` is the list [inf, threshold of population 1, ...] (:657-660) and the state applies it column by column;
  * `_compute_weights_means_and_cov` (samplers.py:505-534): the proposal density of the new population
    (GMDistribution.logpdf over all N x N component densities, elfi/methods/utils.py:139-198) and the weighted
    variance of the parameters (utils.py:108-139) run on the device (csrc/gmix.hip, csrc/wstats.hip);
  * `_set_threshold` (samplers.py:540-549): weighted_sample_quantile of elfi_amd.weighted (bit-exact order statistics).

Same populations, weights and thresholds as the reference on the same seed: the distances are bit-identical; the
mixture density and the weighted variance agree to ~1e-12 relative (device exp / summation order).
"""
import sys

import numpy as np

from .gmix import GMDistribution
from .sampler import hip_rejection_class
from .weighted import weighted_sample_quantile, weighted_var

_CLASSES = {}


def _samplers_module():
    mod = sys.modules.get('elfi.methods.inference.samplers')
    if mod is None:
        raise ImportError("the Hip SMC samplers subclass the running program's ELFI classes: `import elfi` first")
    return mod


def hip_smc_class(base_name='SMC'):
    """The device subclass of the imported ELFI's `base_name` ('SMC', 'AdaptiveDistanceSMC', 'AdaptiveThresholdSMC')."""
    mod = _samplers_module()
    Base = getattr(mod, base_name)
    cls = _CLASSES.get(Base)
    if cls is not None:
        return cls
    logger = mod.logger
    get_sub_seed = mod.get_sub_seed

    class HipSMCRound(Base):
        __doc__ = __doc__

        # -- samplers.py:474-487 ------------------------------------------------------------------------------
        def _set_rejection_round(self, round):
            self._update_round_info(self.state['round'])
            seed = self.seed if round == 0 else get_sub_seed(self.seed, round)
            self._round_random_state = np.random.RandomState(seed)
            self._rejection = hip_rejection_class()(
                self.model,
                discrepancy_name=self.discrepancy_name,
                output_names=self.output_names,
                batch_size=self.batch_size,
                seed=seed,
                max_parallel_batches=self.max_parallel_batches)

        # -- samplers.py:434-459: the proposals of a batch; device_proposals = True draws them on the device -------
        # (a different random realisation than the reference's MT19937 stream, hence opt-in: with False a run equals the
        # reference class's sample by sample on the same seed)
        device_proposals = False

        def prepare_new_batch(self, batch_index):
            if not self.device_proposals or self.state['round'] == 0:
                return super().prepare_new_batch(batch_index)
            params = GMDistribution.rvs(*self._gm_params, size=self.batch_size, prior_logpdf=self._prior.logpdf,
                                        random_state=self._round_random_state)
            return mod.arr2d_to_batch(params, self.parameter_names)

        # -- samplers.py:505-534 ------------------------------------------------------------------------------
        def _compute_weights_means_and_cov(self, pop):
            params = np.column_stack(tuple([pop.outputs[p] for p in self.parameter_names]))
            if self._populations:
                q_logpdf = GMDistribution.logpdf(params, *self._gm_params)
                p_logpdf = self._prior.logpdf(params)
                w = np.exp(p_logpdf - q_logpdf)
            else:
                w = np.ones(pop.n_samples)
            means = params.copy()
            if np.count_nonzero(w) == 0:
                raise RuntimeError("All sample weights are zero. If you are using a prior "
                                   "with a bounded support, this may be caused by specifying "
                                   "a too small sample size.")
            cov = 2 * np.diag(weighted_var(params, w))
            if not np.all(np.isfinite(cov)):
                logger.warning("Could not estimate the sample covariance. This is often "
                               "caused by majority of the sample weights becoming zero."
                               "Falling back to using unit covariance.")
                cov = np.diag(np.ones(params.shape[1]))
            return means, w, cov

    if base_name == 'SMC':
        # -- samplers.py:540-549 (AdaptiveDistanceSMC / AdaptiveThresholdSMC set their thresholds without it) ----
        def _set_threshold(self):
            previous_population = self._populations[self.state['round'] - 1]
            threshold = weighted_sample_quantile(
                x=previous_population.discrepancies,
                alpha=self._quantiles[self.state['round']],
                weights=previous_population.weights)
            logger.info('ABC-SMC: Selected threshold for next population %.3f' % (threshold))
            self.objective['thresholds'][self.state['round']] = threshold
        HipSMCRound._set_threshold = _set_threshold

    name = 'Hip' + base_name
    HipSMCRound.__name__ = name
    HipSMCRound.__qualname__ = name
    _CLASSES[Base] = HipSMCRound
    return HipSMCRound


def HipSMC(*args, **kwargs):
    """elfi.SMC(model, discrepancy_name=None, output_names=None, **kwargs) with the rounds on the GPU."""
    return hip_smc_class('SMC')(*args, **kwargs)


def HipAdaptiveDistanceSMC(*args, **kwargs):
    """elfi.AdaptiveDistanceSMC(model, discrepancy_name=None, output_names=None, **kwargs) with the rounds on the GPU;
    the discrepancy node may be an elfi.AdaptiveDistance or (one device pass per batch) an elfi_amd.HipAdaptiveDistance."""
    return hip_smc_class('AdaptiveDistanceSMC')(*args, **kwargs)


def HipAdaptiveThresholdSMC(*args, **kwargs):
    """elfi.AdaptiveThresholdSMC(model, ...) with the rounds on the GPU."""
    return hip_smc_class('AdaptiveThresholdSMC')(*args, **kwargs)
