import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aspire_amd import ops
J, NC, S = 20, 1000, 8
dev = 'cuda'
g = torch.Generator().manual_seed(4)
sets = []
for rep in range(6):
    crows = torch.randn(J * NC * S, 768, generator=g).to(dev)
    qrows = torch.randn(J * S, 768, generator=g).to(dev)
    ar = torch.arange(J * NC, device=dev, dtype=torch.int32)
    c = ops.DeviceRepSet(crows, (ar * S).contiguous(), torch.full((J * NC,), S, device=dev, dtype=torch.int32), ext=0, max_len=S)
    q = ops.DeviceRepSet(qrows, (ar[:J] * S).contiguous(), torch.full((J,), S, device=dev, dtype=torch.int32), ext=0, max_len=S)
    sets.append((q, c))
job_off = (torch.arange(J + 1, dtype=torch.int32) * NC).to(dev)
out = ops.ot_rank_batch(sets[0][0], sets[0][1], job_off, NC, 100)
i = [0]
def call():
    i[0] += 1
    q, c = sets[i[0] % 6]
    ops.ot_rank_batch(q, c, job_off, NC, 100, out=out)
print('clock under the bench call: %.2f GHz' % ops.clock_under(call, wall_us=6000))
from aspire_amd._lib import pinned
with pinned(FUSED_NOSOLVE='1'):
    print('clock under the cost phase alone: %.2f GHz' % ops.clock_under(call, wall_us=6000))
