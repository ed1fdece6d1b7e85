"""otAspire over batched jobs (aspire_ot_rank_batch_f32), documents of <= 8 rows, small and medium batches: the fused streaming
kernel (OT_FORM=fused; <= 64 jobs: no tables launch) against the small-pool kernels (OT_FORM=small) and the two-kernel form
(OT_FORM=tile), for the crossover in ot_rank_batch.   python tools/otbatchcross.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import _lib, ops

dev = torch.device('cuda')
g = torch.Generator().manual_seed(4)


def repset(lens):
    start = torch.cumsum(lens, 0) - lens
    rows = torch.randn(int(lens.sum()), 768, generator=g).to(dev)
    return ops.DeviceRepSet(rows, start.to(torch.int32).to(dev), lens.to(torch.int32).to(dev), ext=0, max_len=int(lens.max()))


def timed(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


for J, NC in ((1, 125), (2, 20), (5, 20), (5, 50), (10, 50), (20, 100), (50, 125), (100, 125), (20, 400), (100, 200)):
    c = repset(torch.randint(3, 9, (J * NC,), generator=g))
    q = repset(torch.randint(1, 9, (J,), generator=g))
    job_off = (torch.arange(J + 1, dtype=torch.int32) * NC).to(dev)
    k = min(NC, 100)
    out = ops.ot_rank_batch(q, c, job_off, NC, k)
    res = {}
    for rnd in range(2):                   # twice: the first pass also warms the clocks
        for name, pins in (('default', {}), ('small', dict(OT_FORM='small')), ('tile', dict(OT_FORM='tile')), ('fused', dict(OT_FORM='fused'))):
            with _lib.pinned(**pins):
                res[name] = timed(lambda: ops.ot_rank_batch(q, c, job_off, NC, k, out=out))
    print(f'{J:4d} x {NC:5d} (groups of four {J * ((NC + 3) // 4):6d}): default {res["default"]:7.1f} us   small-pool kernels {res["small"]:7.1f}   '
          f'two kernels {res["tile"]:7.1f}   fused {res["fused"]:7.1f}', flush=True)

print('one query x one pool (aspire_ot_rank_f32):')
for NC in (500, 1000, 2000, 3000, 4000, 6000, 8000):
    c = repset(torch.randint(3, 9, (NC,), generator=g))
    q = repset(torch.randint(5, 9, (1,), generator=g))
    res = {}
    for rnd in range(2):
        for name, pins in (('default', {}), ('small', dict(OT_FORM='small')), ('tile', dict(OT_FORM='tile')), ('fused', dict(OT_FORM='fused'))):
            with _lib.pinned(**pins):
                res[name] = timed(lambda: ops.ot_rank(q, c, 100))
    print(f'   1 x {NC:5d} (groups of four {(NC + 3) // 4:6d}): default {res["default"]:7.1f} us   small-pool kernels {res["small"]:7.1f}   '
          f'two kernels {res["tile"]:7.1f}   fused {res["fused"]:7.1f}', flush=True)
