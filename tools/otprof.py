"""Run a few otAspire + tsAspire calls at one shape (for rocprofv3 --kernel-trace --stats).
usage: python tools/otprof.py Q C S [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import ops
from tools.kbench import mk  # noqa

Q, C, S = (int(x) for x in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
q, c = mk(Q, S, 0), mk(C, S, 1)
out = torch.empty(Q * C, device='cuda')
for _ in range(reps):
    ops.ot_sinkhorn(q, c, out=out)
    ops.l2max_scores(q, c)
torch.cuda.synchronize()
