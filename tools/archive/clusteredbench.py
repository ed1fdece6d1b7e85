import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from aspire_amd import ops, _lib
dev = torch.device('cuda')
g = torch.Generator().manual_seed(3)
def timed(fn, n=10):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n
J, NC, S = 20, 1000, 8
only = float(sys.argv[1]) if len(sys.argv) > 1 else None
for name, mu in (('iid N(0,1)', 0.0), ('common vector x 2 + N(0,1)  (cos ~0.8)', 2.0), ('common vector x 6 + N(0,1)  (cos ~0.97)', 6.0)):
    if only is not None and mu != only:
        continue
    base = torch.randn(768, generator=g)
    rows = (torch.randn(J * NC * S, 768, generator=g) + mu * base).to(dev)
    qrows = (torch.randn(J * S, 768, generator=g) + mu * base).to(dev)
    ar = torch.arange(J * NC, dtype=torch.int32, device=dev)
    c = ops.DeviceRepSet(rows, (ar * S).contiguous(), torch.full((J * NC,), S, dtype=torch.int32, device=dev), ext=0, max_len=S)
    q = ops.DeviceRepSet(qrows, (ar[:J] * S).contiguous(), torch.full((J,), S, dtype=torch.int32, device=dev), ext=0, max_len=S)
    job_off = (torch.arange(J + 1, dtype=torch.int32) * NC).to(dev)
    out = ops.ot_rank_batch(q, c, job_off, NC, 100)
    t = timed(lambda: ops.ot_rank_batch(q, c, job_off, NC, 100, out=out))
    cos = torch.nn.functional.normalize(rows[:1024], dim=1); mc = float((cos @ cos.T).mean())
    print(f'{name}: mean cosine {mc:.2f}: 20 x 1000 x 8 batched otAspire {t:.1f} us', flush=True)
