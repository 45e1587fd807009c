"""Prior draws on the GPU: scipy-like distribution objects for elfi.Prior whose `rvs` runs on the device.

    elfi.Prior(elfi_amd.priors.uniform, 0, 2, model=m, name='mu')          # ss.uniform
    elfi.Prior(elfi_amd.priors.MA2Prior1, 2, model=m, name='t1')           # elfi.examples.ma2.CustomPrior1
    elfi.Prior(elfi_amd.priors.MA2Prior2, m['t1'], 1, name='t2')           # elfi.examples.ma2.CustomPrior2

A Prior node's operation is rvs_from_distribution (elfi/model/utils.py:6-35): `distribution.rvs(*params, size=(batch,),
random_state=...)` -- scipy.stats on the host.  Once simulator and distance run on the GPU that call is most of what is
left of a batch: 64 % of the reference loop's time at batch_size 10^6 on the MA2 example (72 of 112 ms per batch on the
build container, profile in DESIGN.md section 7).  Here the uniforms come from the library's counter-based generator
(Philox4x32-10, keyed by a seed taken from the node's own random_state, as the fused simulators are:
fused_models._seed_of) and are transformed on the device in the reference's own operation order
(csrc/gauss.hip: prior_transform).  A run is a different, equally valid, random realisation than MT19937's; the
transform of given uniforms is bit-identical to NumPy's (tests/test_priors_gpu.py).  Densities are the reference's own
(pdf / logpdf delegate to scipy.stats / elfi.examples.ma2), so ModelPrior and BOLFI see the same prior.
"""
import ctypes as C

import numpy as np

from . import _lib

UNIFORM, MA2_T1, MA2_T2 = 0, 1, 2

# Counter-stream namespaces of the consumers that key Philox with a 31-bit seed drawn from an ELFI RandomState: a prior
# node, a fused simulator and an SMC proposal that happen to draw the SAME seed (one pair in ~2^31 / N^2 ...: likely over
# the 10^4-10^5 node-batches of a long run) must not read the same counter words.  Callers that pass `stream` themselves
# (tests, device-resident pipelines) keep full control.
PRIOR_STREAM_BASE = 0x5052494F52000000      # "PRIOR"; + the prior kind
SIMULATOR_STREAM_BASE = 0x53494D0000000000  # "SIM"
PROPOSAL_STREAM_BASE = 0x474D525653000000   # "GMRVS"; + 2 x the redraw round


def _seed_of(random_state):
    rs = random_state or np.random
    return int(rs.randint(0, 2 ** 31 - 1))


def _shape(size):
    if size is None:
        return ()
    return tuple(int(v) for v in np.atleast_1d(size))


def prior_draw(kind, params, cond=None, size=1, random_state=None, seed=None, stream=None, ctx=None):
    """n = prod(size) draws of the prior `kind` (UNIFORM: params (loc, scale); MA2_T1: (b,); MA2_T2: (a,), cond = t1,
    broadcast to `size`).  The seed comes from `random_state` (one randint, as ELFI's batch seeding decides it) unless
    given."""
    shape = _shape(size)
    n = int(np.prod(shape)) if shape else 1
    a = np.ascontiguousarray(np.asarray(params, dtype=np.float64).reshape(-1))
    if a.shape[0] != (2 if kind == UNIFORM else 1):
        raise ValueError('prior kind %d takes %d scalar parameter(s)' % (kind, 2 if kind == UNIFORM else 1))
    c = None
    if kind == MA2_T2:
        c = np.ascontiguousarray(np.broadcast_to(np.asarray(cond, dtype=np.float64), shape or (1,)).reshape(-1))
    out = np.empty(n, dtype=np.float64)
    ctx = ctx or _lib.default_context()
    if stream is None:
        stream = PRIOR_STREAM_BASE + int(kind) if seed is None else 0
    ctx.call("elfihip_prior_draw", int(kind), C.c_uint64(_seed_of(random_state) if seed is None else int(seed)),
             C.c_uint64(int(stream)), n, _lib.ptr(a), _lib.ptr(c), _lib.ptr(out))
    return out.reshape(shape) if shape else out[0]


class uniform:
    """scipy.stats.uniform(loc, scale) with scalar parameters: rvs on the device, densities SciPy's."""

    @classmethod
    def rvs(cls, loc=0, scale=1, size=1, random_state=None):
        if np.ndim(loc) or np.ndim(scale):
            raise ValueError('elfi_amd.priors.uniform draws with scalar loc / scale (a prior without parent nodes); give '
                             'elfi.Prior scipy.stats.uniform itself for array parameters')
        return prior_draw(UNIFORM, (loc, scale), None, size, random_state)

    @classmethod
    def pdf(cls, x, loc=0, scale=1):
        import scipy.stats as ss
        return ss.uniform.pdf(x, loc, scale)

    @classmethod
    def logpdf(cls, x, loc=0, scale=1):
        import scipy.stats as ss
        return ss.uniform.logpdf(x, loc, scale)

    @classmethod
    def cdf(cls, x, loc=0, scale=1):
        import scipy.stats as ss
        return ss.uniform.cdf(x, loc, scale)

    @classmethod
    def ppf(cls, q, loc=0, scale=1):
        import scipy.stats as ss
        return ss.uniform.ppf(q, loc, scale)


class _MA2Backed:
    _ref = None

    @classmethod
    def _reference(cls):
        from elfi.examples import ma2
        return getattr(ma2, cls._ref)

    @classmethod
    def pdf(cls, x, *params):
        return cls._reference().pdf(x, *params)

    @classmethod
    def logpdf(cls, x, *params):
        return cls._reference().logpdf(x, *params)


class MA2Prior1(_MA2Backed):
    """elfi.examples.ma2.CustomPrior1 (ma2.py:96-135): t1 on [-b, b], triangular; rvs on the device."""
    _ref = 'CustomPrior1'

    @classmethod
    def rvs(cls, b, size=1, random_state=None):
        return prior_draw(MA2_T1, (b,), None, size, random_state)


class MA2Prior2(_MA2Backed):
    """elfi.examples.ma2.CustomPrior2 (ma2.py:138-190): t2 given t1, uniform on [max(-a - t1, t1 - a), a]; rvs on the
    device."""
    _ref = 'CustomPrior2'

    @classmethod
    def rvs(cls, t1, a, size=1, random_state=None):
        return prior_draw(MA2_T2, (a,), t1, size, random_state)
