"""Developer timing of the row-distance kernel alone (configs[1] shape unless given): HIP events over many launches on
rotating buffers, both forms of the stream.   python scripts/dist_once.py [n m reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import elfi_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10**6
m = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device('cuda', 0)
ctx = elfi_amd.Context(0)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
ctx.set_stream(stream.cuda_stream)
NB = max(3, -(-640 * 2**20 // (8 * n * m)))
Xs = [torch.randn(n, m, dtype=torch.float64, device=dev) for _ in range(NB)]
y = torch.randn(1, m, dtype=torch.float64, device=dev)
w = torch.rand(m, dtype=torch.float64, device=dev) + 0.5
out = torch.empty(n, dtype=torch.float64, device=dev)
for form in (0, 1, 0, 1):
    ctx.call('elfihip_dist_set_form', form)
    for aux in (None, w):
        for i in range(5):
            ctx.call('elfihip_dist_rows_dev', 0, Xs[i % NB].data_ptr(), n, m, m, y.data_ptr(),
                     aux.data_ptr() if aux is not None else None, 2.0, out.data_ptr())
        torch.cuda.synchronize()
        ctx.timer_start()
        for i in range(reps):
            ctx.call('elfihip_dist_rows_dev', 0, Xs[i % NB].data_ptr(), n, m, m, y.data_ptr(),
                     aux.data_ptr() if aux is not None else None, 2.0, out.data_ptr())
        ms = ctx.timer_stop() / reps
        by = (8 * m + 8) * n
        print('form %d (%s) %s: %.2f us  %.0f GB/s  %.3f of 8 TB/s' % (form, 'LDS-DMA' if form == 0 else 'register pipeline',
              'weighted' if aux is not None else 'plain', ms * 1e3, by / ms / 1e6, by / ms / 1e6 / 8000), flush=True)
# the same batch as m separate columns (what ELFI holds before column_stack: HipDiscrepancy's path)
XT = [x.t().contiguous() for x in Xs]
del Xs
for form in (0, 1, 0, 1):
    ctx.call('elfihip_dist_set_form', form)
    for i in range(5):
        ctx.call('elfihip_dist_cols_dev', 0, XT[i % NB].data_ptr(), n, m, n, y.data_ptr(), None, 2.0, out.data_ptr())
    torch.cuda.synchronize()
    ctx.timer_start()
    for i in range(reps):
        ctx.call('elfihip_dist_cols_dev', 0, XT[i % NB].data_ptr(), n, m, n, y.data_ptr(), None, 2.0, out.data_ptr())
    ms = ctx.timer_stop() / reps
    by = (8 * m + 8) * n
    print('cols form %d (%s loads): %.2f us  %.0f GB/s  %.3f of 8 TB/s' % (form, 'non-temporal' if form == 0 else 'default',
          ms * 1e3, by / ms / 1e6, by / ms / 1e6 / 8000), flush=True)
