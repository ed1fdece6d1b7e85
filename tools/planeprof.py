"""A few max-sim (or otAspire) calls on the fp16-plane tiles for rocprofv3: python tools/planeprof.py [Q C S reps form ot]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import ops
from aspire_amd._lib import pinned
from kbench import mk

Q, C, S = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 50000, 8)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
form = sys.argv[5] if len(sys.argv) > 5 else ''
ot = len(sys.argv) > 6 and sys.argv[6] == 'ot'
q, c = mk(Q, S, 0), mk(C, S, 1)
if form in ('', 'planes'):
    form = ''
    c.prepare_planes()
    q.prepare_planes(like=c)
out = torch.empty(Q * C, device='cuda')
with pinned(COST_PATH='mfma', GEMM=form):
    for _ in range(reps):
        if ot:
            ops.ot_sinkhorn(q, c, out=out)
        else:
            ops.l2max_scores(q, c)
torch.cuda.synchronize()
