"""Where the wall time of a BOLFI run goes: timers around the three calls the reference's loop makes into the device objects.

elfi.BOLFI.fit (elfi/methods/inference/bolfi.py:201-254,289-292) calls, per acquired point,
`target_model.update(x, y, optimize)` and `acquisition_method.acquire(n, t)`; `update` calls `optimize()` (the MAP
search of the hyper-parameters, gpy_regression.py:317-323) every `update_interval` points.  `instrument(gp, acq)` wraps
exactly these three bound methods on the two INSTANCES handed to elfi.BOLFI and returns the accumulator; everything
else of the run's wall time is the reference's own Python loop (graph execution, pools, batch bookkeeping).
"""
import time


class LoopTimes:
    """Seconds and call counts per phase; `searches` keeps one record per MAP search."""

    def __init__(self):
        self.fit_s = 0.0        # update() without the search it may trigger: append + bordering update or rebuild
        self.search_s = 0.0     # optimize(): SCG, every evaluation a device rebuild + gradient
        self.acquire_s = 0.0    # acquire(): multi-start L-BFGS-B in lock-step on the device + jitter
        self.updates = 0
        self.acquires = 0
        self.searches = []      # (n_evidence, seconds, device fits, SCG status)
        self.t_first_acquire = None   # seconds from instrument() to the first acquire(): the initial-evidence phase

    def summary(self, wall_s, n_points):
        n_fits = sum(s[2] for s in self.searches)
        host = wall_s - self.fit_s - self.search_s - self.acquire_s
        per = lambda s, k: 1e3 * s / k if k else None
        return {
            "wall_s": wall_s, "iters_per_s": n_points / wall_s if wall_s > 0 else None, "points": n_points,
            "ms_device_fit": per(self.fit_s, self.updates), "ms_device_acquire": per(self.acquire_s, self.acquires),
            "ms_map_search": per(self.search_s, len(self.searches)), "ms_host_loop": per(host, n_points),
            "share": {"fit": self.fit_s / wall_s, "map_search": self.search_s / wall_s,
                      "acquire": self.acquire_s / wall_s, "host_loop": host / wall_s},
            "updates": self.updates, "acquisitions": self.acquires, "map_searches": len(self.searches),
            "rebuilds_in_searches": n_fits,
            "ms_per_search_rebuild": per(self.search_s, n_fits),
        }


def instrument(gp, acq):
    """Wrap gp.update / gp.optimize / acq.acquire (instance attributes; the classes stay untouched)."""
    T = LoopTimes()
    update, optimize, acquire = gp.update, gp.optimize, acq.acquire
    inside = {"search": 0.0}
    t_begin = time.perf_counter()

    def timed_optimize():
        t0 = time.perf_counter()
        try:
            return optimize()
        finally:
            dt = time.perf_counter() - t0
            inside["search"] += dt
            info = getattr(gp, '_opt_info', None) or {}
            T.searches.append((gp.n_evidence, dt, int(info.get('n_fits', 0)), info.get('status')))
            T.search_s += dt

    def timed_update(x, y, optimize=False):
        inside["search"] = 0.0
        t0 = time.perf_counter()
        try:
            return update(x, y, optimize)
        finally:
            T.fit_s += time.perf_counter() - t0 - inside["search"]
            T.updates += 1

    def timed_acquire(n, t=None):
        t0 = time.perf_counter()
        if T.t_first_acquire is None:
            T.t_first_acquire = t0 - t_begin
        try:
            return acquire(n, t)
        finally:
            T.acquire_s += time.perf_counter() - t0
            T.acquires += 1

    gp.optimize = timed_optimize
    gp.update = timed_update
    acq.acquire = timed_acquire

    def restore():
        # the wrappers live in the instances' __dict__: take them out again (copies / pickles of the model would
        # otherwise carry closures over THIS object)
        for obj, name in ((gp, 'optimize'), (gp, 'update'), (acq, 'acquire')):
            obj.__dict__.pop(name, None)
    T.restore = restore
    return T
