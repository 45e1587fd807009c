"""DRY-RUN stand-in for the device side of bench.py (`python bench.py --gpus N --fake-device`).

NOT the product and NOT a measurement: an 8-GPU node has never been available to the builder, so the N > 1 control flow of
bench.py (claim / publish / gather of the packed sampler states, the all-gather + fixed-order merge of the column
statistics, the strong-scaling partition, max-over-ranks timing, the nested configs[3] leg) had never executed.  With
--fake-device the same script runs end to end on CPU tensors under torch.distributed's gloo backend: every library entry
point bench.py calls in those paths is answered here by NumPy on the host buffers (the arithmetic of the checker,
oracle/distance_oracle.py), streams and events become no-ops, and the JSON line is marked `"data": "fake (CPU dry run)"`
with `value` kept only so that the line is well formed.  tests/test_bench_fake_gloo.py runs it with two ranks.

Only bench.py imports this, and only under --fake-device.
"""
import ctypes as C
import time

import numpy as np


def _f64(ptr, n):
    return np.ctypeslib.as_array((C.c_double * int(n)).from_address(int(ptr)))


def _i64(ptr, n):
    return np.ctypeslib.as_array((C.c_int64 * int(n)).from_address(int(ptr)))


class _State:
    """The sampler state of elfihip_reject_*: the k best (distance, global row) pairs, ascending by (distance, row)."""

    def __init__(self, k):
        self.k = int(k)
        self.vals = np.empty(0)
        self.rows = np.empty(0, dtype=np.int64)
        self.export = None      # address of the (2k) doubles a merge leaves the packed state in

    def push(self, d, row_base):
        v = np.concatenate([self.vals, d])
        r = np.concatenate([self.rows, row_base + np.arange(len(d), dtype=np.int64)])
        order = np.lexsort((r, v))[:self.k]
        self.vals, self.rows = v[order], r[order]
        if self.export:
            kk = len(self.vals)
            _f64(self.export, self.k)[:kk] = self.vals
            _i64(self.export + 8 * self.k, self.k)[:kk] = self.rows

    def reset(self):
        self.vals = np.empty(0)
        self.rows = np.empty(0, dtype=np.int64)


class FakeLib:
    """The entry points bench.py's scaling legs call, on host memory (status 0 = ELFIHIP_OK)."""

    def __init__(self):
        import distance_oracle as O       # the checker's arithmetic: this is a dry run, not the product
        self.O = O

    def _dist(self, X_ptr, n, m, ldx, y_ptr, aux_ptr):
        X = _f64(X_ptr, n * ldx).reshape(n, ldx)[:, :m]
        y = _f64(y_ptr, m).reshape(1, m)
        w = _f64(aux_ptr, m) if aux_ptr else None
        return self.O.cdist_rows(X, y, 'euclidean', w=w)

    def elfihip_dist_rows_dev(self, h, metric, X_ptr, n, m, ldx, y_ptr, aux_ptr, p, out_ptr):
        _f64(out_ptr, n)[:] = self._dist(X_ptr, n, m, ldx, y_ptr, aux_ptr)
        return 0

    def elfihip_reject_push_rows_dev(self, st, metric, X_ptr, n, m, ldx, y_ptr, aux_ptr, p, out_ptr, row_base):
        d = self._dist(X_ptr, n, m, ldx, y_ptr, aux_ptr)
        _f64(out_ptr, n)[:] = d
        st.push(d, int(row_base))
        return 0

    def elfihip_reject_export_dev(self, st, best_ptr):
        st.export = int(best_ptr)
        return 0

    def elfihip_reject_flush(self, st):
        return 0

    def elfihip_reject_reset(self, st):
        st.reset()
        return 0

    def elfihip_adaptive_push_dev(self, h, st, X_ptr, n, m, ldx, y_ptr, W_ptr, K, out_ptr, state_ptr, row_base):
        X = _f64(X_ptr, n * ldx).reshape(n, ldx)[:, :m]
        y = _f64(y_ptr, m).reshape(1, m)
        W = _f64(W_ptr, K * m).reshape(K, m)
        out = _f64(out_ptr, n * K).reshape(n, K)
        for k in range(K):
            out[:, k] = self.O.cdist_rows(X, y, 'euclidean', w=W[k])
        s = _f64(state_ptr, 1 + 2 * m)          # (count, mean (m), M2 (m)): elfi_model.py:1104-1125 on this shard
        cnt = s[0] + n
        d1 = X - s[1:1 + m]
        mean = s[1:1 + m] + d1.sum(axis=0) / cnt
        s[1 + m:] = s[1 + m:] + (d1 * (X - mean)).sum(axis=0)
        s[1:1 + m] = mean
        s[0] = cnt
        if st is not None:
            st.push(out[:, K - 1].copy(), int(row_base))
        return 0

    def elfihip_welford_merge_dev(self, h, states_ptr, world, m, merged_ptr, w_ptr):
        from elfi_amd import sharding
        sh = _f64(states_ptr, world * (1 + 2 * m)).reshape(world, 1 + 2 * m)
        N, mean, M2 = sharding.merge_welford([(v[0], v[1:1 + m].copy(), v[1 + m:].copy()) for v in sh])
        out = _f64(merged_ptr, 1 + 2 * m)
        out[0], out[1:1 + m], out[1 + m:] = N, mean, M2
        _f64(w_ptr, m)[:] = 1.0 / (M2 / N)
        return 0


class FakeCtx:
    handle = None

    def __init__(self):
        self.lib = FakeLib()
        self._t0 = 0.0

    def call(self, name, *args):
        rc = getattr(self.lib, name)(None, *args)
        if rc != 0:
            raise RuntimeError('%s failed in the dry run' % name)

    def set_stream(self, s):
        pass

    def timer_start(self):
        self._t0 = time.perf_counter()

    def timer_stop(self):
        return 1e3 * (time.perf_counter() - self._t0)


class _Null:
    """A stream / event that orders nothing (CPU calls are synchronous)."""

    def wait_event(self, e):
        pass

    def record(self, s=None):
        pass


def make_job(JobBase, K_BEST):
    """FakeJob: bench.py's Job with CPU tensors, gloo collectives and the stand-in library."""

    class FakeJob(JobBase):
        fake = True

        def __init__(self, dev, local_rank, world, rank, comm="torch"):
            import torch
            self.torch, self.dev, self.world, self.rank = torch, dev, world, rank
            self.comm_kind, self.comm = "torch", None
            self.main, self.side = _Null(), _Null()
            self.ctx = FakeCtx()
            self.best = [torch.zeros(2 * K_BEST, dtype=torch.float64) for _ in range(2)]
            self.gath = [torch.zeros(2 * K_BEST, dtype=torch.float64) for _ in range(world)] \
                if (world > 1 and rank == 0) else None
            self.ev_done = [_Null(), _Null()]
            self.ev_free = [_Null(), _Null()]
            self.gathers = 0

        def sync(self):
            pass

        def randn(self, shape, seed, stream=0, device_gen=True):
            gen = self.torch.Generator()
            gen.manual_seed(int(seed) * 1000 + int(stream))
            return self.torch.randn(*shape, dtype=self.torch.float64, generator=gen)

        def new_state(self, k):
            self.state = _State(k)
            return self.state

        def check(self, rc):
            if rc != 0:
                raise RuntimeError('library call failed in the dry run')

        def claim(self, b):
            self.check(self.ctx.lib.elfihip_reject_export_dev(self.state, self.best[b].data_ptr()))

        def publish(self, b):
            if self.world > 1:
                import torch.distributed as dist
                dist.gather(self.best[b], self.gath, dst=0)
                self.gathers += 1

        def barrier(self):
            if self.world > 1:
                import torch.distributed as dist
                dist.barrier()

        def all_gather_small(self, out, inp):
            if self.world > 1:
                import torch.distributed as dist
                # (gloo wants a flat output for all_gather_into_tensor; nccl / RCCL take the (world, ns) tensor as it is)
                dist.all_gather_into_tensor(out.view(-1), inp)
            else:
                out[0].copy_(inp)

    return FakeJob
