"""ONE short query against a big pool of ragged abstracts (the single-pool entry point, aspire_ot_rank_f32): the CHUNK form against
the forms it replaces (pinned OT_FORM=tile: 16-row streaming / tile-loop kernels + block Sinkhorn).  python tools/chunk1bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import _lib, ops

dev = torch.device('cuda')
g = torch.Generator().manual_seed(7)


def repset(lens):
    start = torch.cumsum(lens, 0) - lens
    rows = torch.randn(int(lens.sum()), 768, generator=g).to(dev)
    return ops.DeviceRepSet(rows, start.to(torch.int32).to(dev), lens.to(torch.int32).to(dev), ext=0, max_len=int(lens.max()))


def timed(fn, n=30):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:          # clocks settle on the workload itself
        fn()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


for name, C, lens_fn in (('3..20 rows', 20000, lambda n: torch.randint(3, 21, (n,), generator=g)),
                         ('3..16 rows', 20000, lambda n: torch.randint(3, 17, (n,), generator=g)),
                         ('mostly <= 8, 1 % of 12', 20000, lambda n: torch.where(torch.rand(n, generator=g) < 0.01, 12, torch.randint(3, 9, (n,), generator=g))),
                         ('3..20 rows', 2000, lambda n: torch.randint(3, 21, (n,), generator=g)),
                         ('3..20 rows', 500, lambda n: torch.randint(3, 21, (n,), generator=g))):
    c = repset(lens_fn(C))
    q = repset(torch.tensor([5]))
    res = {}
    for form in ('', 'tile', 'chunk'):
        try:
            with _lib.pinned(OT_FORM=form):
                res[form or 'default'] = timed(lambda: ops.ot_rank(q, c, 100, want=_lib.OT_SIMILARITY))
        except Exception as e:
            res[form or 'default'] = type(e).__name__
    print(f'1 x {C} ({name}), query of 5 rows: ' + ', '.join(f'{k} {v:.1f} us' if isinstance(v, float) else f'{k} {v}' for k, v in res.items()), flush=True)
