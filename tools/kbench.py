"""Kernel micro-benchmarks (GPU box): time individual C-ABI calls with HIP events.
usage: python tools/kbench.py [Q C S]"""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import _lib, ops

def timeit(fn, n=int(os.environ.get("KBENCH_N", "200")), warm=int(os.environ.get("KBENCH_WARM", "20"))):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us

def mk(n, s, seed):
    g = torch.Generator().manual_seed(seed)
    rows = torch.randn(n * s, 768, generator=g)
    if os.environ.get('KBENCH_CLUSTER'):
        # sentence vectors around one common direction with row-specific spread: a few percent of the sentence pairs
        # are close enough for the matmul expansion to cancel (exercises the direct-formula work list of gram.hip)
        base = torch.randn(768, generator=torch.Generator().manual_seed(123)) * (15.0 / 768 ** 0.5)
        spread = 0.05 + 0.25 * torch.rand(n * s, 1, generator=g)
        rows = base[None, :] + spread * rows
    rows = rows.cuda()
    ar = torch.arange(n, device='cuda', dtype=torch.int32)
    return ops.DeviceRepSet(rows, (ar * s).contiguous(), torch.full((n,), s, device='cuda', dtype=torch.int32), ext=0, max_len=s)

def main():
    Q, C, S = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (1, 1000, 8)
    q, c = mk(Q, S, 0), mk(C, S, 1)
    out = torch.empty(Q * C, device='cuda')
    nbytes = 4 * 768 * (C * S + Q * S) + 4 * Q * C
    for name, kw in [('ot scaling=0.9 (75 eps steps)', dict(scaling=0.9)),
                     ('ot scaling=0.5', dict(scaling=0.5)),
                     ('ot scaling=0.001 (~4 eps steps)', dict(scaling=0.001))]:
        us = timeit(lambda: ops.ot_sinkhorn(q, c, out=out, **kw))
        print(f'{name:36s} {us:9.1f} us  {Q*C/us:8.2f} Mpairs/s  {nbytes/us/1e3:8.1f} GB/s algorithmic')
    us = timeit(lambda: ops.ot_sinkhorn(q, c, out=out, want=_lib.OT_PLAN_SIM))
    print(f'{"ot plan-sim":36s} {us:9.1f} us')
    us = timeit(lambda: ops.l2max_scores(q, c))
    print(f'{"l2max":36s} {us:9.1f} us  {Q*C/us:8.2f} Mpairs/s  {nbytes/us/1e3:8.1f} GB/s algorithmic')
    sc = out.view(Q, C)
    us = timeit(lambda: ops.topk_desc(sc, min(100, C)))
    print(f'{"topk k=100":36s} {us:9.1f} us')
    x = torch.empty(64 << 20, device='cuda'); y = torch.empty_like(x)
    us = timeit(lambda: y.copy_(x), n=20)
    print(f'{"torch copy 256MB (r+w 512MB)":36s} {us:9.1f} us  {2*x.numel()*4/us/1e3:8.1f} GB/s')
    us = timeit(lambda: None)
    print(f'{"empty loop":36s} {us:9.1f} us')

if __name__ == "__main__":
    main()
