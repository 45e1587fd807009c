"""One-process-per-GPU sharding of the distance path.

ABC batches are independent units keyed by their batch index (elfi/loader.py:164-169 derives
each batch's sub-seed from it; elfi/methods/parameter_inference.py:283-292 already farms batches
out to workers), so the path shards with NO collective inside the data path:

  * batch b is owned by rank  b % world_size  (`batch_owner`);
  * every rank computes the distances of its own batches on its own GPU;
  * ONE exchange per round: the per-rank results go to rank 0 with a gather
    (`gather_rows`; RCCL over xGMI on GPUs, gloo on CPU in the tests) and are put back into
    batch-index order, which is the order the reference consumes batches in
    (elfi/client.py:172-182) and the one that decides `argsort` ties (samplers.py:234-237);
  * for the adaptive distance the running column statistics (count, mean, M2 --
    elfi/model/elfi_model.py:1104-1125) of the ranks are all-gathered (1 + 2m doubles per rank)
    and merged with Chan's pairwise formula in FIXED rank order, so every rank ends up with
    bit-identical scales and no floating-point all-reduce is involved.

The arithmetic itself is the C ABI's (`elfi_amd.distance`); `backend` exists so the
world_size-2 CPU tests can drive exactly this host logic with a stand-in for the GPU calls.
"""
import numpy as np


def batch_owner(batch_index, world_size):
    return int(batch_index) % int(world_size)


def owned_batches(n_batches, rank, world_size):
    return list(range(int(rank), int(n_batches), int(world_size)))


def strong_partition(total, world_size):
    """Rows [row0, row0 + rows) of a job of `total` rows for every rank: equal blocks of ceil(total / world), the last
    ranks take what is left (possibly nothing).  Global row numbers are what the ranks' sampler states carry, so the
    merged result names rows of the whole job."""
    total, world_size = int(total), int(world_size)
    per = -(-total // world_size)
    return [(min(r * per, total), max(0, min(per, total - r * per))) for r in range(world_size)]


def merge_best(states, k):
    """The k smallest (distance, global row) pairs of the union of the ranks' sampler states, ascending, ties to the
    lower row -- what Rejection._merge_batch (samplers.py:209-237) would hold had it seen every rank's rows.
    states: iterable of (values, rows) per rank; unfilled entries are +inf."""
    vals = np.concatenate([np.asarray(v, dtype=np.float64) for v, _ in states])
    rows = np.concatenate([np.asarray(r, dtype=np.int64) for _, r in states])
    keep = np.isfinite(vals)
    vals, rows = vals[keep], rows[keep]
    order = np.lexsort((rows, vals))[:int(k)]
    return vals[order], rows[order]


def merge_welford(states):
    """Chan et al. pairwise merge of (count, mean, M2) triples, left to right (fixed order)."""
    N, mean, M2 = 0, 0.0, 0.0
    for n_b, mean_b, M2_b in states:
        n_b = int(n_b)
        if n_b == 0:
            continue
        mean_b = np.asarray(mean_b, dtype=np.float64)
        M2_b = np.asarray(M2_b, dtype=np.float64)
        if N == 0:
            N, mean, M2 = n_b, mean_b.copy(), M2_b.copy()
            continue
        tot = N + n_b
        delta = mean_b - mean
        M2 = M2 + M2_b + delta * delta * (N * (n_b / tot))
        mean = mean + delta * (n_b / tot)
        N = tot
    return N, mean, M2


class HipBackend:
    """The product backend: libelfihip.so through elfi_amd.distance (no CPU fallback)."""

    def __init__(self, device=-1):
        from . import _lib
        self.ctx = _lib.default_context(device)

    def welford(self, X, count, mean, M2):
        from .distance import welford_update
        return welford_update(X, count, mean, M2, ctx=self.ctx)

    def nested(self, X, y, W):
        from .distance import nested_weighted_euclidean
        return nested_weighted_euclidean(X, y, W, ctx=self.ctx)

    def distance(self, X, y, metric='euclidean', **kw):
        from .distance import cdist_rows
        return cdist_rows(X, y, metric, ctx=self.ctx, **kw)


def _dist():
    import torch.distributed as dist
    return dist


def collective_device(device=None):
    """Where the buffers of a collective live: what the caller says, else what the process group's backend needs --
    the rank's GPU under nccl (RCCL moves device memory over xGMI), host memory under gloo."""
    if device is not None:
        return device
    dist = _dist()
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl':
        import torch
        return torch.device('cuda', torch.cuda.current_device())
    return 'cpu'


def _tensor(a, device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(collective_device(device))


def all_gather_small(vec, device=None):
    """All-gather a small float64 vector; returns the list in rank order (identical everywhere)."""
    import torch
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [np.asarray(vec, dtype=np.float64)]
    t = _tensor(np.asarray(vec, dtype=np.float64), device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.cpu().numpy() for o in out]


def gather_rows(local, dst=0, device=None):
    """Gather per-rank row blocks (possibly of different lengths) to `dst`.

    Returns the list of arrays in rank order on `dst`, None elsewhere.  Lengths travel first (one
    tiny all-gather), then one gather of blocks padded to the longest."""
    import torch
    dist = _dist()
    local = np.ascontiguousarray(local, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local]
    world, rank = dist.get_world_size(), dist.get_rank()
    lens = [int(v[0]) for v in all_gather_small(np.array([local.shape[0]], dtype=np.float64), device)]
    width = local.shape[1:] if local.ndim > 1 else ()
    pad = np.zeros((max(lens),) + tuple(width))
    pad[:local.shape[0]] = local
    t = _tensor(pad, device)
    bufs = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
    dist.gather(t, bufs, dst=dst)
    if rank != dst:
        return None
    return [b.cpu().numpy()[:lens[r]] for r, b in enumerate(bufs)]


def interleave_batches(per_rank, n_batches, world_size):
    """Put gathered per-rank batch lists back into batch-index order.

    per_rank[r] is the list of results of rank r's batches in the order of owned_batches()."""
    out = [None] * n_batches
    for r in range(world_size):
        for k, b in enumerate(owned_batches(n_batches, r, world_size)):
            out[b] = per_rank[r][k]
    return out


class ShardedAdaptiveDistance:
    """AdaptiveDistance state machine (elfi_model.py:1088-1151) over rank-local shards.

    add_data() folds the rank's rows into its local Welford state; sync_scale() merges the
    ranks' states (all-gather + fixed-order Chan merge) so that update_distance() appends the
    same weights on every rank; nested_distance() is purely local."""

    def __init__(self, m, backend=None, device=None):
        self.m = int(m)
        self.backend = backend or HipBackend()
        self.device = device
        self.w = [None]
        self.scale = None
        self.init_adaptation_round()

    def init_adaptation_round(self):
        self.local = (0, np.zeros(self.m), np.zeros(self.m))

    def add_data(self, X):
        X = np.asarray(X, dtype=np.float64)
        if X.ndim != 2 or X.shape[1] != self.m:
            raise ValueError('data must be (rows, %d)' % self.m)
        if X.shape[0]:
            self.local = self.backend.welford(X, *self.local)

    def sync_scale(self):
        cnt, mean, M2 = self.local
        vec = np.concatenate([[float(cnt)], mean, M2])
        states = [(v[0], v[1:1 + self.m], v[1 + self.m:]) for v in all_gather_small(vec, self.device)]
        N, mean, M2 = merge_welford(states)
        self.global_state = (N, mean, M2)
        self.scale = np.sqrt(M2 / N)
        return self.scale

    def update_distance(self):
        self.w.append(1.0 / self.scale)
        self.init_adaptation_round()

    def weight_matrix(self):
        return np.vstack([np.ones(self.m) if w is None else np.asarray(w) ** 2 for w in self.w])

    def nested_distance(self, X, y):
        X = np.asarray(X, dtype=np.float64)
        return self.backend.nested(X, y, self.weight_matrix())
