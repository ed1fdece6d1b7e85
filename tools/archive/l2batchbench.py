"""tsAspire over batched jobs (aspire_l2max_rank_batch_f32) at small and medium batch sizes: the streaming kernels against the
one-workgroup-per-candidate kernel (OT_FORM=small), for the crossover constant in score.hip.   python tools/l2batchbench.py"""
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
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


for smax in (8, 16, 20):
    for J, NC in ((2, 20), (5, 20), (5, 50), (10, 50), (20, 100), (50, 125), (100, 200), (20, 1000), (50, 1000)):
        c = repset(torch.randint(3, smax + 1, (J * NC,), generator=g))
        q = repset(torch.randint(1, min(smax, 8) + 1, (J,), generator=g))
        job_off = (torch.arange(J + 1, dtype=torch.int32) * NC).to(dev)
        k = min(NC, 100)
        out = ops.l2max_rank_batch(q, c, job_off, NC, k)
        res = {}
        for name, pins in (('default', {}), ('small', dict(OT_FORM='small')), ('stream', dict(OT_FORM='tile' if smax > 8 else 'fused'))):
            with _lib.pinned(**pins):
                res[name] = timed(lambda: ops.l2max_rank_batch(q, c, job_off, NC, k, out=out))
        print(f'rows <= {smax:2d}  {J:4d} x {NC:5d} (groups of four {J * ((NC + 3) // 4):6d}): default {res["default"]:7.1f} us   one workgroup per candidate '
              f'{res["small"]:7.1f}   streaming {res["stream"]:7.1f}', flush=True)
