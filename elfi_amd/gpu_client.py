"""ELFI client that runs batches on several GPUs of one node: one worker process per GPU.

Plug-in surface 4 of SURVEY.md section 8b -- the reference's client API (elfi/client.py:195-347,
registration pattern of elfi/clients/multiprocessing.py:11-98):

    import elfi_amd.gpu_client as gc
    gc.set_as_default()                       # or elfi.set_client(gc.Client(num_gpus=8))
    elfi.Rejection(d, batch_size=10**6).sample(...)

ELFI submits batches in batch-index order (elfi/client.py:118-148) and each submission is an
independent unit (its random stream is derived from the batch index, elfi/loader.py:164-169), so
batch b is simply sent to GPU b % num_gpus.  A worker is a spawned process (never forked: a HIP
context must not cross a fork) whose `ELFI_AMD_DEVICE` is set before anything touches the GPU, so
every `elfi_amd` operation inside the batch (HipDistance, HipDiscrepancy, summaries ...) uses
that worker's GPU through its lazily created per-process context.  Results return as ordinary
pickled NumPy arrays, exactly as with the reference's multiprocessing client; no collective is
involved (the "gather" is the pool's result pipe).

TWO workers per GPU by default (`workers_per_gpu`): ELFI keeps `num_cores` batches in flight
(`max_parallel_batches`, elfi/methods/parameter_inference.py:93), and with one batch per GPU the GPU idles
while its worker simulates, unpickles and uploads the next batch.  Two processes on one device have their
own contexts and streams: the upload and the host-side simulator of batch b + N overlap the kernels of
batch b (copy engines and compute run side by side; compute of the two is time-sliced).

This module imports `elfi`; it is not imported by `elfi_amd` itself.
"""
import itertools
import logging
import multiprocessing

import elfi.client

from . import _worker

logger = logging.getLogger(__name__)


def set_as_default():
    """Make this the default client class (pattern of elfi/clients/multiprocessing.py:11-14)."""
    elfi.client.set_client()
    elfi.client.set_default_class(Client)


class Client(elfi.client.ClientBase):
    """One pool of `workers_per_gpu` processes per GPU; submissions are dealt round-robin (batch b -> GPU b % N)."""

    def __init__(self, num_gpus=None, worker_setup=None, start_method='spawn', workers_per_gpu=2):
        if num_gpus is None:
            import elfi_amd
            num_gpus = elfi_amd.device_count() or 1
        self.num_gpus = int(num_gpus)
        self.workers_per_gpu = max(1, int(workers_per_gpu))
        ctx = multiprocessing.get_context(start_method)
        self.pools = [ctx.Pool(processes=self.workers_per_gpu, initializer=_worker.init, initargs=(g, worker_setup))
                      for g in range(self.num_gpus)]
        self.tasks = {}
        self._id_counter = itertools.count()
        self._sync_counter = itertools.count()

    def _pool_for(self, task_id):
        return self.pools[task_id % self.num_gpus]

    def apply(self, kallable, *args, **kwargs):
        id = self._id_counter.__next__()
        self.tasks[id] = self._pool_for(id).apply_async(kallable, args, kwargs)
        return id

    def apply_sync(self, kallable, *args, **kwargs):
        # round-robin as well: a loop of synchronous calls (e.g. model.generate) uses every GPU, not GPU 0 alone
        return self._pool_for(self._sync_counter.__next__()).apply(kallable, args, kwargs)

    def get_result(self, task_id):
        return self.tasks.pop(task_id).get()

    def is_ready(self, task_id):
        return self.tasks[task_id].ready()

    def remove_task(self, task_id):
        if task_id in self.tasks:
            del self.tasks[task_id]

    def reset(self):
        for p in self.pools:
            p.terminate()
            p.join()
        self.tasks.clear()

    @property
    def num_cores(self):
        """Batches ELFI keeps in flight (max_parallel_batches): workers_per_gpu per GPU."""
        return self.num_gpus * self.workers_per_gpu
