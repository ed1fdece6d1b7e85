"""Small single pools on rows with a common component: which kernel forms still redo most entries (GPU box).
  python tools/clusteredsmall.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import ops, _lib
dev = torch.device('cuda')
g = torch.Generator().manual_seed(3)


def timed(fn, n=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


for NQ, NC, S in ((1, 125, 8), (1, 125, 20), (1, 1000, 8), (1, 1000, 12), (1, 4000, 12), (4, 1000, 8), (1, 300, 40)):
    row = []
    for mu in (0.0, 2.0):
        base = torch.randn(768, generator=g)
        rows = (torch.randn(NC * S, 768, generator=g) + mu * base).to(dev)
        qrows = (torch.randn(NQ * S, 768, generator=g) + mu * base).to(dev)
        ar = torch.arange(NC, dtype=torch.int32, device=dev)
        c = ops.DeviceRepSet(rows, (ar * S).contiguous(), torch.full((NC,), S, dtype=torch.int32, device=dev), ext=0, max_len=S)
        q = ops.DeviceRepSet(qrows, (ar[:NQ] * S).contiguous(), torch.full((NQ,), S, dtype=torch.int32, device=dev), ext=0, max_len=S)
        k = min(100, NC)
        t_ot = timed(lambda: ops.ot_rank(q, c, k, want=_lib.OT_SIMILARITY))
        t_l2 = timed(lambda: ops.l2max_scores(q, c))
        row.append(f'common x {mu:.0f}: otAspire + rank {t_ot:7.1f} us, tsAspire {t_l2:7.1f} us')
    print(f'{NQ} x {NC} x {S}:  ' + '   |   '.join(row), flush=True)
