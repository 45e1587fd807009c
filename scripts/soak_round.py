"""Developer tool: soak of the SMC round's selection path (csrc/reject.hip: provisional threshold from a prefix, resident
selection, early merge + mail_wait) -- fresh sampler state per round, random / sorted / tied inputs, shapes around the chunk
and slice boundaries; every round's k best against torch.  usage: python scripts/soak_round.py [rounds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elfi_amd  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device('cuda', 0)
ctx = elfi_amd.Context(0)
g = torch.Generator(device='cuda')
g.manual_seed(seed)
rs = np.random.RandomState(seed)
m, K = 16, 2
y = torch.zeros(1, m, dtype=torch.float64, device=dev)
W = torch.ones(K, m, dtype=torch.float64, device=dev)
W[1] = 0.5
bad = 0
kinds = {'random': 0, 'ascending': 0, 'descending': 0, 'ties': 0}
t0 = time.time()
rb = {}
for r in range(rounds):
    n = int(rs.choice([1 << 20, (1 << 20) + 4099, 1250000, 1500000 + rs.randint(0, 5000), 2 * 10 ** 6 + 17]))
    k = int(rs.choice([1000, 1000, 1000, 257, 2048]))
    kind = rs.choice(['random'] * 12 + ['ascending', 'descending', 'ties'])
    X = torch.randn(n, m, dtype=torch.float64, device=dev, generator=g)
    if kind == 'ties':
        X = torch.round(X * 2.0) / 2.0
    elif kind != 'random':
        key = torch.argsort((X * X).sum(1), descending=(kind == 'descending'))
        X = X[key].contiguous()
    kinds[kind] += 1
    if os.environ.get('SOAK_VERBOSE'):
        print('round %d n=%d k=%d kind=%s' % (r, n, k, kind), flush=True)
    out = torch.empty(n, K, dtype=torch.float64, device=dev)
    wel = torch.zeros(1 + 2 * m, dtype=torch.float64, device=dev)
    if k not in rb:
        rb[k] = elfi_amd.RunningBest(k, ctx=ctx)
    rb[k].reset()
    base = int(rs.randint(0, 10 ** 9))
    if not os.environ.get('SOAK_NO_SYNC'):
        torch.cuda.synchronize()   # torch's stream and the context's own stream are not ordered against each other
    ctx.call("elfihip_adaptive_push_dev", rb[k].h, X.data_ptr(), n, m, m, y.data_ptr(), W.data_ptr(), K, out.data_ptr(),
             wel.data_ptr(), base)
    vals, rows = rb[k].result()
    d = out[:, K - 1]
    rv = torch.topk(d, k, largest=False, sorted=True).values.cpu().numpy()
    ok = np.array_equal(vals, rv)
    rr = np.asarray(rows) - base
    ok = ok and rr.min() >= 0 and rr.max() < n and len(set(rr.tolist())) == k
    ok = ok and np.array_equal(d[torch.from_numpy(rr).to(dev)].cpu().numpy(), vals)
    # ties to the earlier row: (distance, row) ascending
    ok = ok and np.array_equal(np.lexsort((rr, vals)), np.arange(k))
    ok = ok and int(wel[0].item()) == n
    if not ok:
        bad += 1
        print('MISMATCH round %d n=%d k=%d kind=%s' % (r, n, k, kind), flush=True)
print('%d rounds (%s) in %.1f s: %d mismatches' % (rounds, kinds, time.time() - t0, bad))
sys.exit(1 if bad else 0)
