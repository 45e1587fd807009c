"""Developer tool: the mahalanobis row kernel alone, for rocprofv3 counter passes.
usage: python scripts/maha_once.py [n] [m] [reps]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elfi_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1250000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device('cuda', 0)
ctx = elfi_amd.Context(0)
X = torch.empty(n, m, dtype=torch.float64, device=dev)
ctx.call("elfihip_randn_dev", C.c_uint64(1), C.c_uint64(0), X.numel(), C.c_double(0.0), C.c_double(1.0), X.data_ptr())
y = torch.zeros(1, m, dtype=torch.float64, device=dev)
A_ = torch.randn(m, m, dtype=torch.float64, device=dev)
VI = (A_ @ A_.t() / m + torch.eye(m, dtype=torch.float64, device=dev)).contiguous()
out = torch.empty(n, dtype=torch.float64, device=dev)
torch.cuda.synchronize()


def one():
    ctx.call('elfihip_dist_rows_dev', 6, X.data_ptr(), n, m, m, y.data_ptr(), VI.data_ptr(), C.c_double(2.0), out.data_ptr())


one()
ctx.synchronize()
ctx.timer_start()
for _ in range(reps):
    one()
ms = ctx.timer_stop() / reps
by, fl = (8.0 * m + 8.0) * n, 2.0 * m * m * n
print("mahalanobis n=%d m=%d: %.4f ms  %.0f GB/s (%.3f of 8 TB/s)  %.1f TFLOP/s (%.3f of 78.6)"
      % (n, m, ms, by / ms / 1e6, by / ms / 1e6 / 8000, fl / ms / 1e9, fl / ms / 1e9 / 78.6))
