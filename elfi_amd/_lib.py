"""ctypes binding of libelfihip.so (the C ABI declared in include/elfihip.h).

The product path has no CPU fallback: if the shared library is missing or a GPU entry
point fails, the error is raised to the caller.  Mapping of status codes to Python
exceptions follows the reference's error conventions:
shape/argument errors -> ValueError (what scipy.cdist raises and
elfi/model/utils.py:42-49 re-raises), non-PD Cholesky -> numpy.linalg.LinAlgError
(what GPy raises and elfi/methods/bo/gpy_regression.py:320-323 catches), HIP failures
-> RuntimeError.
"""
import ctypes as C
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libelfihip.so")

OK, ERR_ARG, ERR_HIP, ERR_NOT_PD, ERR_STATE, ERR_NOMEM = range(6)

METRICS = {
    "euclidean": 0,
    "sqeuclidean": 1,
    "cityblock": 2,
    "chebyshev": 3,
    "minkowski": 4,
    "seuclidean": 5,
    "mahalanobis": 6,
}

c_double_p = C.POINTER(C.c_double)
c_void_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes).  tests/test_abi.py checks this table against the header.
PROTOTYPES = {
    "elfihip_version": (C.c_int, []),
    "elfihip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "elfihip_ctx_create": (C.c_int, [C.c_int, c_void_pp]),
    "elfihip_ctx_destroy": (C.c_int, [C.c_void_p]),
    "elfihip_last_error": (C.c_char_p, [C.c_void_p]),
    "elfihip_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "elfihip_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "elfihip_device_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int64), C.c_char_p, C.c_int]),
    "elfihip_timer_start": (C.c_int, [C.c_void_p]),
    "elfihip_timer_stop": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "elfihip_dist_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    "elfihip_dist_rows_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int,
                                        C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    "elfihip_dist_cols": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    "elfihip_dist_cols_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int,
                                        C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    "elfihip_dist_multiw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64,
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "elfihip_dist_multiw_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64,
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "elfihip_welford_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64,
                                         C.POINTER(C.c_int64), C.c_void_p, C.c_void_p]),
    "elfihip_welford_update_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64,
                                             C.c_void_p]),
    "elfihip_welford_merge_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "elfihip_topk_smallest": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                        C.c_void_p]),
    "elfihip_topk_smallest_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                            C.c_void_p]),
    "elfihip_reject_create": (C.c_int, [C.c_void_p, C.c_int64, c_void_pp]),
    "elfihip_reject_free": (C.c_int, [C.c_void_p]),
    "elfihip_reject_reset": (C.c_int, [C.c_void_p]),
    "elfihip_reject_push_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p,
                                          C.c_void_p, C.c_double, C.c_void_p, C.c_int64]),
    "elfihip_reject_push_rows_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p,
                                              C.c_void_p, C.c_double, C.c_void_p, C.c_int64]),
    "elfihip_reject_push_multiw_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p,
                                                C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    "elfihip_reject_push_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]),
    "elfihip_reject_push": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64]),
    "elfihip_reject_set_accept": (C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    "elfihip_reject_set_accept_cols": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "elfihip_adaptive_push_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p,
                                           C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64]),
    "elfihip_adaptive_push": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p,
                                       C.c_int64]),
    "elfihip_host_alloc": (C.c_int, [C.c_size_t, c_void_pp]),
    "elfihip_host_free": (C.c_int, [C.c_void_p]),
    "elfihip_kept_rows": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "elfihip_adaptive_push_kept": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                            C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_int64]),
    "elfihip_randn_rows": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "elfihip_prior_draw": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "elfihip_prior_draw_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "elfihip_kept_distances": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "elfihip_reject_push_kept": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64]),
    "elfihip_ma2_draw_distance": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "elfihip_reject_meta": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "elfihip_reject_state_dev": (C.c_int, [C.c_void_p, c_void_pp, c_void_pp]),
    "elfihip_reject_export_dev": (C.c_int, [C.c_void_p, C.c_void_p]),
    "elfihip_reject_flush": (C.c_int, [C.c_void_p]),
    "elfihip_reject_result": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "elfihip_gm_rvs": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                C.c_void_p, C.c_void_p]),
    "elfihip_gm_pdf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                 C.c_void_p, C.c_double, C.c_void_p]),
    "elfihip_weighted_var": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "elfihip_weighted_var_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p,
                                           C.c_void_p]),
    "elfihip_row_summary": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int,
                                      C.c_void_p]),
    "elfihip_row_summary_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int,
                                          C.c_void_p]),
    "elfihip_ma2_distance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_double,
                                       C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "elfihip_ma2_distance_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p,
                                           C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "elfihip_ma2_draw_distance_dev": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "elfihip_gp_kernel_matrix": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "elfihip_comm_unique_id": (C.c_int, [C.c_void_p, C.c_void_p]),
    "elfihip_comm_init_rank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, c_void_pp]),
    "elfihip_comm_free": (C.c_int, [C.c_void_p]),
    "elfihip_comm_allgather_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "elfihip_comm_gather_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]),
    "elfihip_comm_bcast_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
    "elfihip_comm_bcast_factor": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "elfihip_topk_set_form": (C.c_int, [C.c_void_p, C.c_int]),
    "elfihip_dist_set_form": (C.c_int, [C.c_void_p, C.c_int]),
    "elfihip_gauss_distance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    "elfihip_gauss_distance_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
    "elfihip_randn_dev": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64, C.c_double, C.c_double, C.c_void_p]),
    "elfihip_random_bits_dev": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64, C.c_void_p]),
    "elfihip_gp_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, c_void_pp]),
    "elfihip_gp_free": (C.c_int, [C.c_void_p]),
    "elfihip_gp_set_hyper": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]),
    "elfihip_gp_set_data": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "elfihip_gp_append": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "elfihip_gp_factorize": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "elfihip_gp_jitchol": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "elfihip_gp_profile": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "elfihip_gp_set_schedule": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "elfihip_gp_set_dense_threshold": (C.c_int, [C.c_void_p, C.c_int64, C.c_int]),
    "elfihip_gp_set_lockstep_form": (C.c_int, [C.c_void_p, C.c_int]),
    "elfihip_gp_lockstep_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "elfihip_gp_nlml_grad": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_void_p]),
    "elfihip_gp_form_kinv": (C.c_int, [C.c_void_p]),
    "elfihip_gp_extend": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_double)]),
    "elfihip_gp_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                  C.POINTER(C.c_int)]),
    "elfihip_gp_get": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "elfihip_gp_predict": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "elfihip_gp_predict_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    "elfihip_gp_lcb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p]),
    "elfihip_gp_lcb_minimize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_double,
                                          C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "elfihip_gp_set_acq_options": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "elfihip_gp_set_integration_points": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "elfihip_gp_cross_cov": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "elfihip_gp_maxvar": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "elfihip_gp_expintvar": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "elfihip_lbfgsb_create": (C.c_int, [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.POINTER(C.c_void_p)]),
    "elfihip_lbfgsb_pending": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "elfihip_lbfgsb_feed": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "elfihip_lbfgsb_result": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "elfihip_lbfgsb_free": (C.c_int, [C.c_void_p]),
}

_lib = None
_lib_lock = threading.Lock()


class ElfiHipError(RuntimeError):
    """HIP runtime / device failure inside libelfihip.so."""


def load_library():
    """dlopen libelfihip.so (once per process) and attach the prototypes.

    If torch is already imported its bundled HIP runtime (same soname,
    libamdhip64.so.7) is the one the dynamic loader binds to, so device pointers and
    streams can be shared with torch; otherwise /opt/rocm's runtime is used.
    """
    global _lib
    if _lib is not None:
        return _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libelfihip.so is not built ({}). Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C elfi_amd/csrc`; there is no CPU fallback.".format(LIB_PATH))
        try:
            import torch  # noqa: F401  (binds torch's HIP runtime first when available)
        except Exception:  # pragma: no cover - torch is optional for the C ABI itself
            pass
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (restype, argtypes) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
        return _lib


def _raise(lib, ctx, rc):
    msg = lib.elfihip_last_error(ctx)
    msg = msg.decode("utf-8", "replace") if msg else "status %d" % rc
    if rc == ERR_ARG:
        raise ValueError(msg)
    if rc == ERR_NOT_PD:
        raise np.linalg.LinAlgError(msg)
    if rc == ERR_NOMEM:
        raise MemoryError(msg)
    raise ElfiHipError(msg)


def check(ctx, rc):
    if rc != OK:
        _raise(load_library(), ctx, rc)


def ptr(a):
    """Address of a numpy array's data (None -> NULL)."""
    if a is None:
        return None
    return a.ctypes.data


class Context:
    """One GPU + one HIP stream.  Created lazily per process (never before a fork).

    Mirrors the per-process laziness the reference's picklable operations need
    (elfi/clients/multiprocessing.py:50 pickles operations into pool workers).
    """

    def __init__(self, device=-1):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.elfihip_ctx_create(int(device), C.byref(h))
        if rc != OK:
            _raise(self.lib, None, rc)
        self.handle = h
        self.pid = os.getpid()

    def close(self):
        if getattr(self, "handle", None) is not None and self.pid == os.getpid():
            self.lib.elfihip_ctx_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def call(self, name, *args):
        rc = getattr(self.lib, name)(self.handle, *args)
        if rc != OK:
            _raise(self.lib, self.handle, rc)

    def synchronize(self):
        self.call("elfihip_ctx_synchronize")

    def set_stream(self, stream_handle):
        self.call("elfihip_ctx_set_stream", C.c_void_p(stream_handle or 0))

    def device_info(self):
        cu, clk, mclk, bus = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        mem = C.c_int64()
        name = C.create_string_buffer(256)
        self.call("elfihip_device_info", C.byref(cu), C.byref(clk), C.byref(mclk), C.byref(bus),
                  C.byref(mem), name, 256)
        return dict(cu_count=cu.value, clock_khz=clk.value, mem_clock_khz=mclk.value,
                    mem_bus_bits=bus.value, total_mem=mem.value, name=name.value.decode())

    def kept_epoch(self):
        """Counter of the host-form distance calls on this context: names the device copy of what the last one returned."""
        ep = C.c_uint64()
        self.call("elfihip_kept_distances", C.byref(ep), None, None)
        return ep.value

    def kept_shape(self):
        """(epoch, rows, columns) of the device copy the latest host-form distance call left."""
        ep, n, k = C.c_uint64(), C.c_int64(), C.c_int()
        self.call("elfihip_kept_distances", C.byref(ep), C.byref(n), C.byref(k))
        return ep.value, n.value, k.value

    def timer_start(self):
        self.call("elfihip_timer_start")

    def timer_stop(self):
        ms = C.c_float()
        self.call("elfihip_timer_stop", C.byref(ms))
        return ms.value


# Distances that are still on the device: id(array a distance call returned) -> (weak reference, context, epoch).  The
# sampler state asks for the entry of the array it is handed (selection.RunningBest.push_distances) and skips the upload
# when the context still keeps that call's copy (include/elfihip.h: elfihip_kept_distances).
_KEPT = {}

# The hand-over is keyed by the IDENTITY of the host array, so nobody may write into the array between the call that
# returned it and the call that consumes it (ELFI's graph never does; a user operation might: `d[mask] = inf`, `X *= s` on
# randn_rows output, an `out=` argument).  That is enforced, not assumed: an array that names a device copy is handed out
# READ-ONLY (`flags.writeable = False`: an in-place edit raises ValueError -- edit a copy, which takes the upload path, or
# switch the hand-over off with set_device_handover(False)), and a consumer that finds the array writeable again drops the
# entry and reads the host array.
_HANDOVER = [True]


def set_device_handover(enabled=True):
    """Switch the device-resident hand-over of distances / simulator rows between node operations on or off (default on).
    Off: consumers always read the host arrays they are given.  Returns the previous setting."""
    prev = _HANDOVER[0]
    _HANDOVER[0] = bool(enabled)
    if not enabled:
        _KEPT.clear()
        _ROWS.clear()
    return prev


def _freeze(arr):
    """Mark `arr` read-only; False when the object cannot be (not an ndarray)."""
    try:
        arr.flags.writeable = False
        return True
    except (AttributeError, ValueError):
        return False


def _frozen(arr):
    try:
        return not arr.flags.writeable
    except AttributeError:
        return False


def remember_kept(arr, ctx):
    if len(_KEPT) > 32:
        for key in [k for k, ent in _KEPT.items() if ent[0]() is None]:
            del _KEPT[key]
        while len(_KEPT) > 32:
            del _KEPT[next(iter(_KEPT))]
    if not _HANDOVER[0]:
        return arr
    try:
        if _freeze(arr):
            _KEPT[id(arr)] = (weakref.ref(arr), ctx, ctx.kept_epoch())
    except TypeError:
        pass
    return arr


class _PinnedPool:
    """Page-locked result buffers, recycled: pinning 512 MB costs ~0.1 s, so a buffer goes back to the pool when the last
    NumPy view of it dies (the finalizer sits on the ctypes object every view's base chain ends in)."""

    def __init__(self, keep=4):
        self.free, self.keep = {}, keep        # nbytes -> [addresses]

    def array(self, shape):
        lib = load_library()
        count = int(np.prod(shape))
        nbytes = max(8, count * 8)
        lst = self.free.get(nbytes)
        if lst:
            addr = lst.pop()
        else:
            p = C.c_void_p()
            rc = lib.elfihip_host_alloc(nbytes, C.byref(p))
            if rc != OK or not p.value:
                return np.empty(shape, dtype=np.float64)       # (no pinned memory left: an ordinary array)
            addr = p.value
        buf = (C.c_double * max(1, count)).from_address(addr)
        weakref.finalize(buf, self._release, addr, nbytes)
        return np.frombuffer(buf, dtype=np.float64, count=count).reshape(shape)

    def _release(self, addr, nbytes):
        lst = self.free.setdefault(nbytes, [])
        if len(lst) < self.keep:
            lst.append(addr)
        else:
            try:
                load_library().elfihip_host_free(C.c_void_p(addr))
            except Exception:
                pass


pinned = _PinnedPool()

_ROWS = {}      # id(array a device-side simulator returned) -> (weak reference, context, epoch of its device copy)


def remember_rows(arr, ctx):
    for key in [k for k, ent in _ROWS.items() if ent[0]() is None]:
        del _ROWS[key]
    while len(_ROWS) > 8:
        del _ROWS[next(iter(_ROWS))]
    if not _HANDOVER[0]:
        return arr
    ep = C.c_uint64()
    ctx.call("elfihip_kept_rows", C.byref(ep), None, None)
    if _freeze(arr):
        _ROWS[id(arr)] = (weakref.ref(arr), ctx, ep.value)
    return arr


def rows_epoch_of(arr, ctx):
    ent = _ROWS.get(id(arr))
    if ent is None or ent[0]() is not arr or ent[1] is not ctx:
        return None
    if not _frozen(arr):      # made writeable again since it was handed out: the device copy may not be this array any more
        del _ROWS[id(arr)]
        return None
    return ent[2]


def alias_kept(new, old):
    """`new` (a reshaped view of `old`) is what the caller will hand on: it names the same device copy."""
    ent = _KEPT.get(id(old))
    if ent is not None and ent[0]() is old:
        try:
            if _frozen(old) and _freeze(new):       # (a view of a read-only array is read-only already)
                _KEPT[id(new)] = (weakref.ref(new), ent[1], ent[2])
        except TypeError:
            pass
    return new


def kept_epoch_of(arr, ctx):
    """Epoch of the device copy of `arr` on `ctx`, or None."""
    ent = _KEPT.get(id(arr))
    if ent is None or ent[0]() is not arr or ent[1] is not ctx:
        return None
    if not _frozen(arr):      # made writeable again since it was handed out: the device copy may not be this array any more
        del _KEPT[id(arr)]
        return None
    return ent[2]


_ctx_by_device = {}
_ctx_lock = threading.Lock()


def default_context(device=-1):
    """Process-local context for `device` (-1: current HIP device, or $ELFI_AMD_DEVICE)."""
    if device < 0:
        env = os.environ.get("ELFI_AMD_DEVICE")
        if env is not None:
            device = int(env)
    key = (os.getpid(), device)
    ctx = _ctx_by_device.get(key)
    if ctx is None:
        with _ctx_lock:
            ctx = _ctx_by_device.get(key)
            if ctx is None:
                ctx = Context(device)
                _ctx_by_device[key] = ctx
    return ctx


def device_count():
    lib = load_library()
    n = C.c_int()
    rc = lib.elfihip_device_count(C.byref(n))
    return n.value if rc == OK else 0
