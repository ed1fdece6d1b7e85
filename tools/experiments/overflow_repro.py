"""A candidate sharing a sentence with the query at 1 - 3 x the N(0, 1) vector scale, per kernel form, against the oracle: the case the
fuzz sweep found (fused kernel: NaN before the repair launch became unconditional).   python tools/experiments/overflow_repro.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from aspire_amd import scorer, _lib
from oracle import aspire_oracle as orc
for scale in (1.0, 2.0, 3.0):
    for seed in range(6):
        g = torch.Generator().manual_seed(seed)
        q = [scale * torch.randn(8, 768, generator=g), scale * torch.randn(5, 768, generator=g)]
        c = [scale * torch.randn(int(torch.randint(1, 9, (1,), generator=g)), 768, generator=g) for _ in range(4097)]
        c[1] = torch.cat([q[0][:1], scale * torch.randn(1, 768, generator=g)])
        c[2] = torch.cat([q[0][:1], scale * torch.randn(7, 768, generator=g)])
        c[3] = q[0][:1].clone()
        res = {}
        for form in ('fused', 'tile', 'small'):
            with _lib.pinned(OT_FORM=form):
                res[form] = scorer.score_pool(q, c, method='ot', schedule='pair').cpu().numpy()[0, 1:4]
        want = [orc.get_similarity(q[0], c[j]) for j in (1, 2, 3)]
        print(scale, seed, {k: np.round(v, 3).tolist() for k, v in res.items()}, 'oracle', np.round(want, 3).tolist(), flush=True)
