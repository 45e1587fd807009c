"""Developer tool: the fused adaptive-distance pass alone (csrc/adaptive.hip), for rocprofv3 runs and timing sweeps.
usage: python scripts/adaptive_once.py [n] [m] [K] [reps] [mode]     mode: fused | nostats | noout | state | multiw | all"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elfi_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10 ** 7
m = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K = int(sys.argv[3]) if len(sys.argv) > 3 else 3
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
mode = sys.argv[5] if len(sys.argv) > 5 else 'fused'
dev = torch.device('cuda', 0)
ctx = elfi_amd.Context(0)
X = torch.empty(n, m, dtype=torch.float64, device=dev)
ctx.call("elfihip_randn_dev", C.c_uint64(1), C.c_uint64(0), X.numel(), C.c_double(0.0), C.c_double(1.0), X.data_ptr())
y = torch.zeros(1, m, dtype=torch.float64, device=dev)
W = torch.ones(K, m, dtype=torch.float64, device=dev)
out = torch.empty(n, K, dtype=torch.float64, device=dev)
st = torch.zeros(1 + 2 * m, dtype=torch.float64, device=dev)
rb = elfi_amd.RunningBest(1000, ctx=ctx)
torch.cuda.synchronize()


def one(i):
    if mode == 'multiw':
        ctx.call("elfihip_dist_multiw_dev", X.data_ptr(), n, m, m, y.data_ptr(), W.data_ptr(), K, out.data_ptr())
    elif mode == 'state':
        rb.reset()
        ctx.call("elfihip_adaptive_push_dev", rb.h, X.data_ptr(), n, m, m, y.data_ptr(), W.data_ptr(), K, out.data_ptr(),
                 st.data_ptr(), 0)
    elif mode == 'nostats':
        ctx.call("elfihip_adaptive_push_dev", None, X.data_ptr(), n, m, m, y.data_ptr(), W.data_ptr(), K, out.data_ptr(),
                 None, 0)
    elif mode == 'noout':
        ctx.call("elfihip_adaptive_push_dev", None, X.data_ptr(), n, m, m, y.data_ptr(), W.data_ptr(), K, None,
                 st.data_ptr(), 0)
    else:
        ctx.call("elfihip_adaptive_push_dev", None, X.data_ptr(), n, m, m, y.data_ptr(), W.data_ptr(), K, out.data_ptr(),
                 st.data_ptr(), 0)


modes = ['fused', 'nostats', 'noout', 'state', 'multiw'] if mode == 'all' else [mode]
for form in (2, 1):
    ctx.call('elfihip_dist_set_form', form)
    for mode in modes:
        one(0)
        ctx.synchronize()
        ctx.timer_start()
        for i in range(reps):
            one(i)
        ms = ctx.timer_stop() / reps
        by = (8.0 * m + 8.0 * K) * n
        print("form %d (%s) %s n=%d m=%d K=%d: %.4f ms  %.0f GB/s  %.3f of 8 TB/s" % (
            form, 'LDS-DMA' if form == 2 else 'register stage', mode, n, m, K, ms, by / ms / 1e6, by / ms / 1e6 / 8000), flush=True)
