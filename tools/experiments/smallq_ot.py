"""otAspire with few queries on a plane pool with cached boxes against the default paths: python tools/experiments/smallq_ot.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import ops
from kbench import timeit, mk
for Q, C, S in ((1, 20000, 12), (2, 20000, 8), (4, 20000, 8), (8, 20000, 8), (2, 10000, 12), (1, 20000, 8), (3, 20000, 16)):
    q, c = mk(Q, S, 0), mk(C, S, 1)
    out = torch.empty(Q * C, device='cuda')
    base = timeit(lambda: ops.ot_sinkhorn(q, c, out=out), n=30, warm=4)
    ref = out.clone()
    c.prepare_planes(); c.prepare_boxes()
    us = timeit(lambda: ops.ot_sinkhorn(q, c, out=out), n=30, warm=4)
    print(f'{Q} x {C} x {S}: default {base:7.1f} us  plane pool + boxes {us:7.1f} us   maxdiff {(out-ref).abs().max().item():.2e}')
    c.drop_planes(); q.drop_planes()
