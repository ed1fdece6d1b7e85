import sys, time; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
from aspire_amd import ops
from kbench import timeit, mk
q, c = mk(32, 8, 0), mk(50000, 8, 1)
c.prepare_planes()
q.prepare_planes(like=c)
def ev(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print('planes kept      ', ev(lambda: ops.l2max_scores(q, c)))
def call():
    q.drop_planes(); return ops.l2max_scores(q, c)
print('drop each call   ', ev(call))
def call2():
    q.drop_planes(); q.prepare_planes(like=c); return ops.l2max_scores(q, c)
print('explicit prepare ', ev(call2))
print('planes kept      ', ev(lambda: ops.l2max_scores(q, c)))
t0 = time.perf_counter()
for _ in range(200): call()
print('host per call us ', (time.perf_counter() - t0) / 200 * 1e6, '(GPU queue backs up: upper bound)')
torch.cuda.synchronize()
