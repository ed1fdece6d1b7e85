import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aspire_amd import ops
from aspire_amd._lib import pinned
from kbench import timeit, mk
for Q, C, S in ((1, 20000, 12), (1, 20000, 8), (4, 20000, 8), (8, 20000, 8), (2, 10000, 12)):
    q, c = mk(Q, S, 0), mk(C, S, 1)
    base = timeit(lambda: ops.l2max_scores(q, c), n=50, warm=5)
    ref = ops.l2max_scores(q, c).clone()
    c.prepare_planes(); q.prepare_planes(like=c)
    with pinned(GRAM_PP='2', COST_PATH='mfma'):
        us = timeit(lambda: ops.l2max_scores(q, c), n=50, warm=5)
        got = ops.l2max_scores(q, c)
    nbytes = 4 * 768 * (C * S + Q * S)
    print(f'{Q} x {C} x {S}: default {base:7.1f} us  planes(128-col tile) {us:7.1f} us = {nbytes/us/1e3:.0f} GB/s  maxdiff {(got-ref).abs().max().item():.2e}')
    c.drop_planes(); q.drop_planes()
