"""Run a few aspire_ot_rank_batch_f32 calls at one shape with the scoring form pinned (for rocprofv3 passes).
usage: python tools/fusedprof.py J NC FORM [reps] [NOSOLVE]      FORM = fused | tile | small"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import _lib, ops

J, NC = int(sys.argv[1]), int(sys.argv[2])
form = sys.argv[3]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
nosolve = len(sys.argv) > 5 and sys.argv[5] == '1'
S, D = 8, 768
dev = torch.device('cuda')
g = torch.Generator().manual_seed(0)
ar = torch.arange(J * NC, device=dev, dtype=torch.int32)
sets = []
for _ in range(3):
    qrows = torch.randn(J * S, D, generator=g).to(dev)
    crows = torch.randn(J * NC * S, D, device=dev)
    q = ops.DeviceRepSet(qrows, (ar[:J] * S).contiguous(), torch.full((J,), S, device=dev, dtype=torch.int32), 0, S)
    c = ops.DeviceRepSet(crows, (ar * S).contiguous(), torch.full((J * NC,), S, device=dev, dtype=torch.int32), 0, S)
    sets.append((q, c))
job_off = (torch.arange(J + 1, dtype=torch.int32) * NC).to(dev)
pins = dict(OT_FORM=form)
if nosolve:
    pins['FUSED_NOSOLVE'] = 1
with _lib.pinned(**pins):
    for r in range(reps):
        q, c = sets[r % 3]
        ops.ot_rank_batch(q, c, job_off, NC, 100)
    torch.cuda.synchronize()
