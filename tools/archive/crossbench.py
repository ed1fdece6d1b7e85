"""Few queries x a big pool of short documents: the Gram (matrix-core) forms against the fused kernel, tsAspire and otAspire.
usage: python tools/crossbench.py [C S]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import _lib, ops
from tools.kbench import mk, timeit

C, S = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (20000, 8)
c = mk(C, S, 1)
for Q in (1, 2, 3, 4, 6, 8, 16, 32):
    q = mk(Q, S, 0)
    out = torch.empty(Q * C, device='cuda')
    res = {}
    for name, pins in (('gram', dict(COST_PATH='mfma')), ('fused', dict(COST_PATH='valu', OT_FORM='fused') if S <= 8 else dict(COST_PATH='valu', OT_FORM='tile')), ('default', {})):
        with _lib.pinned(**pins):
            res[name] = (timeit(lambda: ops.l2max_scores(q, c), n=30, warm=5), timeit(lambda: ops.ot_sinkhorn(q, c, out=out), n=30, warm=5))
    print(f'Q={Q:3d} C={C} S={S}:  l2max us ' + ' '.join(f'{k}={v[0]:8.1f}' for k, v in res.items()) + '   ot us ' +
          ' '.join(f'{k}={v[1]:8.1f}' for k, v in res.items()), flush=True)
