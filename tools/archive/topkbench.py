import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import ops
for C, k in ((1000, 100), (4000, 100), (100, 10)):
    sc = torch.randn(1, C, device='cuda')
    for _ in range(20): ops.topk_desc(sc, k)
    torch.cuda.synchronize()
