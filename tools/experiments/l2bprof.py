import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from aspire_amd import _lib, ops
dev = torch.device('cuda'); g = torch.Generator().manual_seed(4)
def repset(lens):
    start = torch.cumsum(lens, 0) - lens
    rows = torch.randn(int(lens.sum()), 768, generator=g).to(dev)
    return ops.DeviceRepSet(rows, start.to(torch.int32).to(dev), lens.to(torch.int32).to(dev), ext=0, max_len=int(lens.max()))
J, NC, smax = 100, 200, 16
c = repset(torch.randint(3, smax + 1, (J * NC,), generator=g)); q = repset(torch.randint(1, 9, (J,), generator=g))
job_off = (torch.arange(J + 1, dtype=torch.int32) * NC).to(dev)
out = ops.l2max_rank_batch(q, c, job_off, NC, 100)
for _ in range(20): ops.l2max_rank_batch(q, c, job_off, NC, 100, out=out)
torch.cuda.synchronize()
