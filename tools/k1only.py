import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import ops
from tools_common import mk
C = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
q, c = mk(1, 8, 0), mk(C, 8, 1)
out = torch.empty(C, device='cuda')
for _ in range(5):
    ops.ot_sinkhorn(q, c, out=out)
    ops.l2max_scores(q, c)
torch.cuda.synchronize()
