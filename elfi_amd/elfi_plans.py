"""Replay plans for ELFI's ModelPrior: its compiled nets loaded ONCE, then run without the per-call graph work.

ModelPrior.rvs / pdf / logpdf (elfi/model/extensions.py:156-210) build, for every call, a ComputationContext, a loaded copy
of a compiled net (networkx graph copies), its execution order (a topological sort) and run the nodes through
Executor._run (elfi/executor.py:44-160) -- 0.25-0.9 ms per call whatever the batch holds.  With the GP on the GPU these
calls are what an acquisition's start points, a sampling round's prior term or a proposal check cost.  A prior's nets do
not change between calls: a plan keeps, per (net, batch size), the node operations in execution order with the wiring of
their arguments -- taken from the reference's own load_data / get_execution_order -- and a call runs them on the caller's
inputs: the same operations on the same values in the same order, hence the same results (and, for rvs, the same generator
state afterwards).  Every plan is VERIFIED against the public call the first time it is used; a prior whose plan does not
reproduce it keeps the public call for good.
"""
import importlib
import weakref

import numpy as np


class NetPlan:
    def __init__(self, prior, net, batch_size, seed, feeds):
        ext = importlib.import_module(type(prior).__module__)
        executor = importlib.import_module('elfi.executor').Executor
        context = ext.ComputationContext(batch_size, seed=seed)
        loaded = prior.client.load_data(net, context, batch_index=0)
        for k in feeds:                      # nodes the caller supplies: outputs, as the reference overrides them
            loaded.nodes[k].update({'output': None})
            loaded.nodes[k].pop('operation', None)
        self.steps = []
        for node in executor.get_execution_order(loaded):
            attr = loaded.nodes[node]
            if 'operation' not in attr:
                continue
            pos, kw = [], {}
            for parent in loaded.predecessors(node):
                param = loaded[parent][node]['param']
                if isinstance(param, int):
                    pos.append((param, parent))
                else:
                    kw[param] = parent
            self.steps.append((node, attr['operation'], [p for _, p in sorted(pos, key=lambda t: t[0])], kw))
        self.const = {k: a['output'] for k, a in loaded.nodes.items() if 'output' in a and k not in feeds}

    def run(self, feeds):
        out = dict(self.const)
        out.update(feeds)
        for node, op, pos, kw in self.steps:
            out[node] = op(*[out[p] for p in pos], **{k: out[p] for k, p in kw.items()})
        return out


_PLANS = weakref.WeakKeyDictionary()   # prior -> {key: NetPlan (verified) | False (public call only)}


def _is_model_prior(prior):
    return (hasattr(prior, '_rvs_net') and hasattr(prior, '_logpdf_net') and hasattr(prior, 'client')
            and hasattr(prior, 'parameter_names') and hasattr(prior, 'dim') and type(prior).__module__.startswith('elfi.'))


_MAX_PLANS = 16   # per prior: posterior sampling asks for a handful of row counts; anything beyond is an unusual caller


class _PlanCache(dict):
    """key -> NetPlan | False, at most _MAX_PLANS entries (the oldest goes first: every plan holds a loaded net's constants)."""

    def __setitem__(self, key, value):
        if key not in self and len(self) >= _MAX_PLANS:
            del self[next(iter(self))]
        dict.__setitem__(self, key, value)


def _plans_of(prior):
    try:
        return _PLANS.setdefault(prior, _PlanCache())
    except TypeError:            # (not weakly referenceable)
        return None


def _same_state(a, b):
    return a[0] == b[0] and np.array_equal(a[1], b[1]) and tuple(a[2:]) == tuple(b[2:])


def prior_rvs(prior, n, random_state):
    """prior.rvs(n, random_state=random_state) -- ELFI's ModelPrior with a RandomState through a plan; else the public call."""
    if not (_is_model_prior(prior) and hasattr(random_state, 'get_state') and isinstance(n, (int, np.integer)) and n >= 1):
        return prior.rvs(n, random_state=random_state)
    plans = _plans_of(prior)
    if plans is None:
        return prior.rvs(n, random_state=random_state)
    key = ('rvs', int(n))
    plan = plans.get(key)
    if plan is False:
        return prior.rvs(n, random_state=random_state)

    def run(p):
        out = p.run({'_random_state': random_state})
        rvs = np.column_stack([out[q] for q in prior.parameter_names])
        return rvs.reshape(int(n)) if prior.dim == 1 else rvs
    if plan is not None:
        return run(plan)
    before = random_state.get_state()
    ref = prior.rvs(n, random_state=random_state)
    after = random_state.get_state()
    plans[key] = False
    try:
        cand = NetPlan(prior, prior._rvs_net, int(n), 'global', ['_random_state'])
        random_state.set_state(before)
        got = run(cand)
        if np.shape(got) == np.shape(ref) and np.array_equal(got, ref) and _same_state(random_state.get_state(), after):
            plans[key] = cand
    except Exception:
        pass
    random_state.set_state(after)
    return ref


def prior_logpdf(prior, x, log=True):
    """prior.logpdf(x) / prior.pdf(x) for x (n, dim) -- ELFI's ModelPrior through a plan; else the public call."""
    public = prior.logpdf if log else prior.pdf
    x = np.asanyarray(x)
    if not (_is_model_prior(prior) and x.ndim == 2 and x.shape[1] == prior.dim and len(x) >= 1):
        return public(x)
    plans = _plans_of(prior)
    if plans is None:
        return public(x)
    key = ('logpdf' if log else 'pdf', len(x))
    plan = plans.get(key)
    if plan is False:
        return public(x)
    node = prior._logpdf_node if log else prior._pdf_node
    names = list(prior.parameter_names)

    def run(p):
        return p.run({q: x[:, i] for i, q in enumerate(names)})[node]
    if plan is not None:
        return run(plan)
    ref = public(x)
    plans[key] = False
    try:
        cand = NetPlan(prior, prior._logpdf_net if log else prior._pdf_net, len(x), 0, names)
        # a density net that consumed randomness could not be replayed from a plan (the reference makes a fresh RandomState
        # per call): without the generator in the plan's constants such a net fails here and keeps the public call
        cand.const.pop('_random_state', None)
        got = run(cand)
        if np.shape(got) == np.shape(ref) and np.array_equal(got, ref, equal_nan=True):
            plans[key] = cand
    except Exception:
        pass
    return ref
