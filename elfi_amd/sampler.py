"""elfi.Rejection with its sample state on the GPU: a drop-in for the reference's sampler class.

    rej = elfi_amd.HipRejection(model['d'], batch_size=10000, seed=1)      # instead of elfi.Rejection(...)
    res = rej.sample(1000, quantile=0.01)

`HipRejection` IS the reference's `elfi.Rejection` (elfi/methods/inference/samplers.py:24-299) -- a subclass made from
the class of the ELFI the running program has imported, so the whole control flow (batch submission, pools, objectives,
`extract_result`, SMC's use of a Rejection per round) stays the reference's -- with the three methods that hold the
arithmetic replaced:

  * `_init_samples_lazy` (samplers.py:180-207): the same checks of the batch's outputs; the sample arrays hold n_samples
    rows (the reference's n_samples + batch_size rows are workspace of its argsort);
  * `_merge_batch` (samplers.py:209-237: threshold mask + copy of the batch behind the sample + argsort of
    n_samples + batch_size distances, every batch): the batch's distances go through the device state of
    csrc/reject.hip (`RunningBest`: candidate pass against the running k-th distance with the acceptance threshold of a
    threshold objective applied on the device, :219-225; a merge of the few candidates), then only the n_samples (distance,
    row) pairs come back and the other outputs are GATHERED for the rows that are new -- O(n_samples) host work per batch;
  * `_update_distances` (samplers.py:279-299, adaptive distance): the re-rank of the n_samples rows under the new
    distance by the device selection, with the reference's assignments.

`_update_objective_n_batches` (samplers.py:245-277) counts the acceptable rows of the reference's LARGER array; that
count is reproduced exactly from the per-batch accepted counts the device keeps (see `_n_acceptable`), so a threshold
objective stops after the same number of batches as the reference.

Same sample as the reference for continuous distances (ties between equal distances go to the earlier row here; NumPy's
default argsort leaves their order unspecified).
"""
import sys

import numpy as np

from .selection import RunningBest, smallest_k

_CLASSES = {}


def _reference_rejection():
    mod = sys.modules.get('elfi.methods.inference.samplers')
    if mod is None:
        raise ImportError("HipRejection subclasses the running program's elfi.Rejection: `import elfi` first")
    return mod.Rejection, mod


def hip_rejection_class():
    """The subclass of the imported ELFI's Rejection (made once per reference class)."""
    Rejection, mod = _reference_rejection()
    cls = _CLASSES.get(Rejection)
    if cls is not None:
        return cls
    is_array = mod.is_array

    class HipRejection(Rejection):
        __doc__ = __doc__

        # -- device state -----------------------------------------------------------------------------------
        def set_objective(self, *args, **kwargs):
            super(HipRejection, self).set_objective(*args, **kwargs)
            self._hip_best = None          # made with the first batch (needs n_samples and the threshold)
            self._hip_rows = None          # global row numbers of the sample rows in use
            self._hip_pushed = 0
            self._hip_held = 0             # rows of the reference's array that are acceptable (see _n_acceptable)

        def _hip_state(self):
            if self._hip_best is None:
                self._hip_best = RunningBest(self.objective['n_samples'], accept=self.objective.get('threshold'))
            return self._hip_best

        # -- samplers.py:180-207 ------------------------------------------------------------------------------
        def _init_samples_lazy(self, batch):
            samples = {}
            e_noarr = "Node {} output must be in a numpy array of length {} (batch_size)."
            e_len = "Node {} output has array length {}. It should be equal to the batch size {}."
            for node in self.output_names:
                if node not in batch:
                    raise KeyError("Did not receive outputs for node {}".format(node))
                nbatch = batch[node]
                if not is_array(nbatch):
                    raise ValueError(e_noarr.format(node, self.batch_size))
                elif len(nbatch) != self.batch_size:
                    raise ValueError(e_len.format(node, len(nbatch), self.batch_size))
                shape = (self.objective['n_samples'],) + nbatch.shape[1:]
                if node == self.discrepancy_name:
                    samples[node] = np.ones(shape, dtype=nbatch.dtype) * np.inf
                else:
                    samples[node] = np.empty(shape, dtype=nbatch.dtype)
            self.state['samples'] = samples
            self._hip_rows = np.empty(0, dtype=np.int64)

        # -- samplers.py:209-237 ------------------------------------------------------------------------------
        def _merge_batch(self, batch):
            samples = self.state['samples']
            if self.adaptive:
                observed_sums = [batch[s] for s in self.sums]
                self.model[self.discrepancy_name].add_data(*observed_sums)
            best = self._hip_state()
            base = self._hip_pushed
            best.push_distances(batch[self.discrepancy_name], row_base=base)
            self._hip_pushed += self.batch_size
            vals, rows = best.result()
            # which sample rows are new, where the others were
            old_rows = self._hip_rows
            new = rows >= base
            keep = ~new
            src_old = np.empty(0, dtype=np.int64)
            if np.any(keep):
                sorter = np.argsort(old_rows, kind='stable')
                src_old = sorter[np.searchsorted(old_rows, rows[keep], sorter=sorter)]
            src_new = rows[new] - base
            c = len(rows)
            for node, v in samples.items():
                merged = np.empty((c,) + v.shape[1:], dtype=v.dtype)
                if len(src_old):
                    merged[keep] = v[src_old]
                if len(src_new):
                    merged[new] = np.asarray(batch[node])[src_new]
                v[:c] = merged
            self._hip_rows = rows
            # rows the reference's (n_samples + batch_size)-row array would hold as acceptable after this batch
            if self.objective.get('threshold') is not None:
                _, accepted, _ = best.meta()
                cap = self.objective['n_samples'] + self.batch_size
                self._hip_held = min(self._hip_held, cap - accepted) + accepted

        def _n_acceptable(self):
            """What samplers.py:255-257 counts: rows of the reference's sample array with every column <= threshold.  That
            array holds accepted rows only (others are never copied in), the best `cap - accepted` of the old ones survive
            a batch: held_j = min(held_{j-1}, cap - A_j) + A_j."""
            return self._hip_held

        # -- samplers.py:245-277, with the count above ----------------------------------------------------------
        def _update_objective_n_batches(self):
            if self.objective.get('threshold') is None:
                return
            from math import ceil
            s = self.state
            n_samples = self.objective.get('n_samples')
            n_acceptable = self._n_acceptable() if s['samples'] else 0
            if n_acceptable == 0:
                n_batches = self.objective['n_batches'] + 1
            else:
                accept_rate_t = n_acceptable / s['n_sim']
                margin = .2 * self.batch_size * int(n_acceptable < n_samples)
                n_batches = (n_samples / accept_rate_t + margin) / self.batch_size
                n_batches = ceil(n_batches)
            self.objective['n_batches'] = n_batches

        # -- samplers.py:279-299 ------------------------------------------------------------------------------
        def _update_distances(self):
            self.model[self.discrepancy_name].update_distance()
            nums = self.objective['n_samples']
            data = {s: self.state['samples'][s][:nums] for s in self.sums}
            ds = self.model[self.discrepancy_name].generate(with_values=data)
            sort_distance = np.atleast_2d(np.transpose(ds))[-1]
            # the re-rank: order of the rows under the new distance, by the device selection (ascending by
            # (distance, row)); rows with a NaN distance last, as np.argsort lists them
            _, sort_mask = smallest_k(sort_distance, nums)
            self.state['samples'][self.discrepancy_name] = sort_distance
            for k in self.state['samples'].keys():
                if k != self.discrepancy_name:
                    self.state['samples'][k][:nums] = self.state['samples'][k][sort_mask]
            self._update_state_meta()

    HipRejection.__name__ = 'HipRejection'
    HipRejection.__qualname__ = 'HipRejection'
    _CLASSES[Rejection] = HipRejection
    return HipRejection


def HipRejection(*args, **kwargs):
    """elfi.Rejection(model, discrepancy_name=None, output_names=None, **kwargs) with the sample state on the GPU."""
    return hip_rejection_class()(*args, **kwargs)
