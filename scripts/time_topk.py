"""Developer tool: time elfihip_topk_smallest_dev.  usage: python scripts/time_topk.py [form]  (1 = nine-launch form)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import elfi_amd
ctx = elfi_amd.Context(0)
ctx.call('elfihip_topk_set_form', int(sys.argv[1]) if len(sys.argv) > 1 else 0)
dev = torch.device('cuda', 0)
for n, k in [(10**6, 1000), (10**6, 100000), (10**7, 1000), (65536, 100), (4096, 10)]:
    d = torch.rand(n, dtype=torch.float64, device=dev)
    v = torch.empty(k, dtype=torch.float64, device=dev)
    i = torch.empty(k, dtype=torch.int64, device=dev)
    f = lambda: ctx.call('elfihip_topk_smallest_dev', d.data_ptr(), n, 1, k, v.data_ptr(), i.data_ptr())
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 50
    for _ in range(R):
        f()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / R * 1e3
    ref = torch.sort(d).values[:k]
    ok = torch.equal(torch.sort(v).values, ref) and torch.equal(d[i], v)
    print('n=%-9d k=%-7d %.3f ms  %s' % (n, k, ms, 'ok' if ok else 'MISMATCH'))
