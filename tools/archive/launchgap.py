"""Per-launch GPU cost of back-to-back kernels inside one hipGraph (slope over launches per graph)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import _lib, ops
from tools_common import mk

def graph_time(fn, n, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3)
    return best

q = mk(1, 8, 0)
for C in (50, 250, 500, 1000, 2000, 4000):
    c = mk(C, 8, 1)
    out = torch.empty(C, device='cuda')
    f_ot = lambda: ops.ot_sinkhorn(q, c, out=out)
    f_l2 = lambda: ops.l2max_scores(q, c)
    t1, t20 = graph_time(f_ot, 1), graph_time(f_ot, 21)
    u1, u20 = graph_time(f_l2, 1), graph_time(f_l2, 21)
    print(f'C={C:5d}  ot: 1 launch {t1:7.1f} us, per extra launch {(t20-t1)/20:7.1f} us | l2max: 1 launch {u1:7.1f} us, per extra {(u20-u1)/20:7.1f} us')
sc = torch.randn(1, 1000, device='cuda')
f_tk = lambda: ops.topk_desc(sc, 100)
t1, t20 = graph_time(f_tk, 1), graph_time(f_tk, 21)
print(f'topk 1x1000 k=100: 1 launch {t1:.1f} us, per extra {(t20-t1)/20:.1f} us')
x = torch.zeros(64, device='cuda')
f_e = lambda: x.add_(1.0)
t1, t20 = graph_time(f_e, 1), graph_time(f_e, 21)
print(f'tiny torch add: 1 launch {t1:.1f} us, per extra {(t20-t1)/20:.1f} us')
