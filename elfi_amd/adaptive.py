"""elfi.AdaptiveDistance with its arithmetic on the GPU: a drop-in NODE class.

    d = elfi_amd.HipAdaptiveDistance(S1, S2, ...)            # instead of elfi.AdaptiveDistance(S1, S2, ...)
    elfi_amd.HipAdaptiveDistanceSMC(d, batch_size=..., seed=...).sample(n, rounds, quantile=...)
    # ... or the reference's own elfi.AdaptiveDistanceSMC(d, ...) / elfi.Rejection(d, ...)

`HipAdaptiveDistance` IS the reference's node class (elfi/model/elfi_model.py:1047-1151) -- a subclass made from the
class of the ELFI the running program has imported, so `isinstance(node, elfi.AdaptiveDistance)` holds where the
samplers test it (elfi/methods/inference/samplers.py:89,590) and the node's `state` keeps the reference's keys ('w',
'distance_functions', 'store', 'scale') with the reference's sharing rules (the lists are mutated in place: a
sampler works on a COPY of the model whose node dictionaries share them, elfi/model/graphical_model.py:139-144) --
with the three methods that hold arithmetic replaced:

  * the node's operation (elfi_model.py:1082-1083: distance_as_discrepancy over nested_distance, utils.py:37-52): ONE
    device pass over the batch's rows (csrc/adaptive.hip) gives the (n, K) nested distances under the weights of the
    earlier rounds AND the batch's own column statistics (count, mean, M2);
  * `add_data` (elfi_model.py:1104-1125), which Rejection._merge_batch calls with the batch's summaries
    (samplers.py:213-216): if those are the arrays the operation just saw (same objects: the native client hands the
    executor's outputs through), the batch statistics kept from that pass are folded into `state['store']` with Chan's
    pairwise update -- no second look at the data; otherwise (arrays that came back from worker processes, direct
    calls) the statistics are computed on the device from the data handed in;
  * `update_distance` / `nested_distance`: the reference's bookkeeping; the distance functions appended are device
    callables with cdist's (n, 1) result shape.

`state['store']` therefore holds [count, mean, M2] of everything added in the round, as in the reference; its values
agree with the reference's batched Welford update up to rounding (tests: a-priori bound), the distances are
bit-identical to scipy's cdist given the same weights.
"""
import sys
import uuid
import weakref

import numpy as np

from . import _lib
from .distance import _SHAPE_HINT, adaptive_batch, cdist_rows, welford_update
from .sharding import merge_welford

_CLASSES = {}
_PENDING_MAX = 8
# batch statistics waiting for their add_data call, per node state: token (a plain string kept in the node's state, so the
# state stays picklable) -> {id(first summary array): (weak references to the arrays, (count, mean, M2))}
_PENDING = {}


class _Cdist1:
    """What the reference keeps in state['distance_functions']: partial(cdist, metric='euclidean', w=...) -> (n, 1)."""

    def __init__(self, w=None, device=-1):
        self.w = None if w is None else np.array(w, dtype=np.float64)
        self.device = device

    def __call__(self, u, v):
        return cdist_rows(u, v, 'euclidean', w=self.w, ctx=_lib.default_context(self.device)).reshape(-1, 1)


class _AdaptiveOperation:
    """The node's operation: op(*summaries, observed) -> (n,) or (n, K) (elfi/model/utils.py:37-52 over the node's
    nested_distance).  Holds the node reference exactly as the reference's partial(distance_as_discrepancy,
    self.nested_distance) does."""

    def __init__(self, node):
        self.node = node

    def __call__(self, *summaries, observed):
        return self.node._hip_discrepancy(summaries, observed)


def _reference_node_class():
    mod = sys.modules.get('elfi.model.elfi_model')
    if mod is None:
        raise ImportError("HipAdaptiveDistance subclasses the running program's elfi.AdaptiveDistance: `import elfi` first")
    return mod.AdaptiveDistance


def hip_adaptive_distance_class():
    """The subclass of the imported ELFI's AdaptiveDistance (made once per reference class)."""
    Base = _reference_node_class()
    cls = _CLASSES.get(Base)
    if cls is not None:
        return cls

    class HipAdaptiveDistance(Base):
        __doc__ = __doc__

        def __init__(self, *summaries, **kwargs):
            super(HipAdaptiveDistance, self).__init__(*summaries, **kwargs)
            # (elfi_model.py:1082-1086 has built partial(distance_as_discrepancy, self.nested_distance) and the cdist
            # partial; the operation that runs is the fused device pass)
            self.state['attr_dict']['_operation'] = _AdaptiveOperation(self)

        # -- elfi_model.py:1088-1102 ------------------------------------------------------------------------
        def init_state(self):
            self.state['w'] = [None]
            self.state['distance_functions'] = [_Cdist1(None)]
            self.state['store'] = 3 * [None]
            self.state['_hip_token'] = uuid.uuid4().hex      # names this state's entry of _PENDING
            self.init_adaptation_round()

        # -- the operation: utils.py:37-52 + elfi_model.py:1135-1151, with the batch statistics on the way ----
        def _weight_matrix(self, m):
            rows = [np.ones(m) if w is None else np.asarray(w, dtype=np.float64) ** 2 for w in self.state['w']]
            return np.vstack(rows)

        def _hip_discrepancy(self, summaries, observed):
            if not summaries:
                raise ValueError("This node requires that at least one parent is specified.")
            obs = np.concatenate([np.atleast_2d(o) for o in observed], axis=1)
            try:
                if len(summaries) == 1 and np.ndim(summaries[0]) == 2:
                    X = np.asarray(summaries[0])
                else:
                    X = np.column_stack(summaries)
                if 'w' not in self.state:
                    self.init_state()
                d, (cnt, mean, M2) = adaptive_batch(X, obs, self._weight_matrix(X.shape[1]), store=(0, 0.0, 0.0))
            except ValueError as e:
                raise ValueError(_SHAPE_HINT.format(e))
            self._remember(summaries, (cnt, mean, M2))
            if d.ndim == 2 and d.shape[1] == 1:
                d = _lib.alias_kept(d.reshape(-1), d)   # (the view is what the sampler will see)
            return d

        def _remember(self, summaries, stats):
            if len(_PENDING) > 64:
                for tok in [t for t, p in _PENDING.items() if not p]:
                    del _PENDING[tok]
            pend = _PENDING.setdefault(self.state.setdefault('_hip_token', uuid.uuid4().hex), {})
            for key in [k for k, (refs, _) in pend.items() if any(r() is None for r in refs)]:
                del pend[key]
            while len(pend) >= _PENDING_MAX:
                del pend[next(iter(pend))]
            try:
                refs = tuple(weakref.ref(s) for s in summaries)
            except TypeError:
                return          # (not weakly referenceable: nothing to recognise the arrays by later)
            pend[id(summaries[0])] = (refs, stats)

        def _recall(self, data):
            pend = _PENDING.get(self.state.get('_hip_token'))
            if not pend or not data:
                return None
            ent = pend.get(id(data[0]))
            if ent is None:
                return None
            refs, stats = ent
            if len(refs) != len(data) or any(r() is not a for r, a in zip(refs, data)):
                return None
            del pend[id(data[0])]
            return stats

        # -- elfi_model.py:1104-1125 ------------------------------------------------------------------------
        def add_data(self, *data):
            st = self.state['store']
            stats = self._recall(data)
            if stats is not None:
                st[0], st[1], st[2] = merge_welford([(st[0], st[1], st[2]), stats])
            else:
                X = data[0] if len(data) == 1 and np.ndim(data[0]) == 2 else np.column_stack(data)
                st[0], st[1], st[2] = welford_update(X, st[0], st[1], st[2])
            self.state['scale'] = np.sqrt(st[2] / st[0])

        # -- elfi_model.py:1127-1133 ------------------------------------------------------------------------
        def update_distance(self):
            weis = 1 / self.state['scale']
            self.state['w'].append(weis)
            self.init_adaptation_round()
            self.state['distance_functions'].append(_Cdist1(weis ** 2))

        # -- elfi_model.py:1135-1151 ------------------------------------------------------------------------
        def nested_distance(self, u, v):
            u = np.asarray(u)
            if u.ndim != 2:
                raise ValueError('XA must be a 2-dimensional array.')
            d, _ = adaptive_batch(u, v, self._weight_matrix(u.shape[1]))
            return d

    HipAdaptiveDistance.__name__ = 'HipAdaptiveDistance'
    HipAdaptiveDistance.__qualname__ = 'HipAdaptiveDistance'
    HipAdaptiveDistance.__module__ = __name__
    _CLASSES[Base] = HipAdaptiveDistance
    return HipAdaptiveDistance


def __getattr__(name):
    # `elfi_amd.adaptive.HipAdaptiveDistance` resolves to the class itself (a saved model's node state names its class:
    # elfi_model.py:512, pickled by reference)
    if name == 'HipAdaptiveDistance':
        return hip_adaptive_distance_class()
    raise AttributeError(name)
