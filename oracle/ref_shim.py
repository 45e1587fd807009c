"""TEST INFRASTRUCTURE ONLY -- makes reference ELFI (/root/reference, v0.8.7) importable here.

The reference cannot be imported as-is in this container: GPy, paramz, toolz,
numdifftools, arviz, dask and ipyparallel are not installed (no network) and
elfi/methods/inference/samplers.py:119 uses np.Inf, which NumPy 2 removed.  This module
installs stub modules and the two NumPy aliases, then puts /root/reference on sys.path.
With it the whole distance / summary / Rejection / SMC / AdaptiveDistanceSMC path of
the reference runs unmodified; everything that needs
GPy raises RuntimeError('unavailable').

Only oracle/ scripts that GENERATE golden fixtures and tests that check plumbing
against the real reference may import this.  It is never imported by elfi_amd/.
"""
import os
import sys
import types

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def _find_root():
    """$ELFI_REFERENCE_ROOT, else /root/reference (build container), else oracle/_ref -- the copy oracle/make_ref.sh
    makes so that the GPU box (where /root/reference does not exist) can run the real reference loops as a checker."""
    env = os.environ.get("ELFI_REFERENCE_ROOT")
    if env:
        return env
    for cand in ("/root/reference", os.path.join(_HERE, "_ref")):
        if os.path.isdir(os.path.join(cand, "elfi")):
            return cand
    return "/root/reference"


REFERENCE_ROOT = _find_root()


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "elfi"))


class _Unavailable:
    def __init__(self, *a, **k):
        raise RuntimeError("third-party dependency unavailable in this container")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _compose(*fs):
    # toolz.functoolz.compose: right-to-left composition (elfi/model/augmenter.py:6)
    def composed(*a, **k):
        r = fs[-1](*a, **k)
        for f in reversed(fs[:-1]):
            r = f(r)
        return r
    return composed


def install():
    """Idempotently install the stubs; return the imported reference package."""
    if not available():
        raise ImportError("reference ELFI not found under %s" % REFERENCE_ROOT)
    if not hasattr(np, "Inf"):
        np.Inf = np.inf
    if not hasattr(np, "float_"):
        np.float_ = np.float64
    if "GPy" not in sys.modules:
        g = _stub("GPy")
        g.kern = _stub("GPy.kern", RBF=_Unavailable, Bias=_Unavailable)
        g.models = _stub("GPy.models", GPClassification=_Unavailable, GPRegression=_Unavailable)
        g.priors = _stub("GPy.priors", Gamma=_Unavailable)
    for name in ("arviz", "numdifftools"):
        if name not in sys.modules:
            _stub(name)
    if "toolz" not in sys.modules:
        tz = _stub("toolz")
        tz.functoolz = _stub("toolz.functoolz", compose=_compose)
    if "dask" not in sys.modules:
        d = _stub("dask")
        d.distributed = _stub("dask.distributed", Client=_Unavailable)
    if "ipyparallel" not in sys.modules:
        _stub("ipyparallel", Client=_Unavailable)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import elfi  # noqa: E402
    return elfi
