"""elfi_amd -- the MI355X (gfx950) hot path for ELFI (elfi-dev/elfi).

One data-parallel path, built from scratch in HIP behind ELFI's own operation and
surrogate-model interfaces:

  * batched summary -> distance evaluation (elfi.Distance / elfi.AdaptiveDistance
    operations; the reference delegates to scipy.spatial.distance.cdist), and
  * the BOLFI Gaussian-process surrogate loop (the `target_model=` object of
    elfi.BOLFI; the reference delegates to GPy).

Everything else (graph compilation, samplers, storage, plotting ...) stays reference
ELFI.  The compute lives in libelfihip.so (C ABI: include/elfihip.h); this package is
the thin host-side mirror of the reference interfaces.  See DESIGN.md / INTEGRATION.md.
"""
from ._lib import (LIB_PATH, Context, ElfiHipError, default_context, device_count, load_library,  # noqa: F401
                   set_device_handover)
from .distance import (AdaptiveDistanceState, HipDiscrepancy, HipDistance, cdist_cols,  # noqa: F401
                       cdist_rows, nested_weighted_euclidean, welford_update)

from .gmix import GMDistribution  # noqa: F401
from .weighted import weighted_sample_quantile, weighted_var  # noqa: F401
from .gp import GPHandle, HipGPRegression  # noqa: F401
from .selection import RunningBest, merge_batch, smallest_k  # noqa: F401
from .sampler import HipRejection, hip_rejection_class  # noqa: F401
from .distance import adaptive_batch, randn_rows  # noqa: F401
from .adaptive import hip_adaptive_distance_class  # noqa: F401
from .smc import HipAdaptiveDistanceSMC, HipAdaptiveThresholdSMC, HipSMC, hip_smc_class  # noqa: F401
from .summaries import autocov, gauss_distance, ma2_distance, ma2_draw_distance, ss_mean, ss_var  # noqa: F401
from .lcb_acquisition import HipLCBSC  # noqa: F401
from .posterior import HipBolfiPosterior, sample_posterior  # noqa: F401
from .bolfi import HipBOLFI, hip_bolfi_class  # noqa: F401
from . import chains, fused_models, multistart, priors  # noqa: F401
from .maxvar_acquisition import HipExpIntVar, HipMaxVar, HipRandMaxVar  # noqa: F401



def HipAdaptiveDistance(*summaries, **kwargs):
    """elfi.AdaptiveDistance(*summaries, **kwargs) with the node's arithmetic on the GPU (elfi_amd/adaptive.py)."""
    return hip_adaptive_distance_class()(*summaries, **kwargs)


__version__ = "0.1.0"
