"""Batches of mostly short documents with a share of 9 .. 16-row ones: the device-side hybrid of aspire_ot_rank_batch_f32
(fused kernel + per-pair repair, or the 16-row streaming kernel + block Sinkhorn, picked by a census on the device)
against the streaming path pinned.   python tools/hybridbench.py [J NC]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aspire_amd import _lib, ops

J, NC = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (20, 1000)
dev = torch.device('cuda')
g = torch.Generator().manual_seed(0)
for p_long in (0.0, 0.002, 0.01, 0.03, 0.06, 0.12, 0.3, 1.0):
    lens = torch.where(torch.rand(J * NC, generator=g) < p_long, torch.randint(9, 17, (J * NC,), generator=g), torch.randint(3, 9, (J * NC,), generator=g))
    if p_long == 0.0:
        lens[7] = 12                                  # one long document in 20 000
    start = torch.cumsum(lens, 0) - lens
    rows = torch.randn(int(lens.sum()), 768, device=dev)
    c = ops.DeviceRepSet(rows, start.to(torch.int32).to(dev), lens.to(torch.int32).to(dev), ext=0, max_len=int(lens.max()))
    qrows = torch.randn(J * 8, 768, device=dev)
    q = ops.DeviceRepSet(qrows, (torch.arange(J, dtype=torch.int32) * 8).to(dev), torch.full((J,), 8, dtype=torch.int32, device=dev), ext=0, max_len=8)
    job_off = (torch.arange(J + 1, dtype=torch.int32) * NC).to(dev)
    res = {}
    for name, pins in (('hybrid', {}), ('stream16', dict(OT_FORM='tile'))):
        with _lib.pinned(**pins):
            out = None
            for _ in range(3):
                out = ops.ot_rank_batch(q, c, job_off, NC, 100)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 30
            for _ in range(n):
                ops.ot_rank_batch(q, c, job_off, NC, 100, out=out)
            torch.cuda.synchronize()
            res[name] = 1e6 * (time.perf_counter() - t0) / n
    print(f'J={J} NC={NC} long documents {p_long:5.3f}: hybrid {res["hybrid"]:7.1f} us   16-row streaming path {res["stream16"]:7.1f} us', flush=True)
