import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from aspire_amd import scorer, _lib, ops
from oracle import aspire_oracle as orc
junk = [torch.full((1 << 24,), float('nan'), device='cuda') for _ in range(8)]   # poison the allocator's pool
del junk
for scale, seed in ((2.0, 20), (3.0, 30)):
    g = torch.Generator().manual_seed(seed)
    mk = lambda n: scale * torch.randn(n, 768, generator=g)
    q = [mk(8), mk(5)]
    c = [mk(int(torch.randint(1, 9, (1,), generator=g))) for _ in range(4100)]
    c[1] = torch.cat([q[0][:1], mk(1)]); c[2] = torch.cat([q[0][:1], mk(7)]); c[3] = q[0][:1].clone()
    for rep in range(3):
        got = scorer.score_pool(q, c, method='ot', schedule='pair').cpu().numpy()
        print(scale, 'default', got[0, 1:4].tolist(), 'nonfinite', int((~np.isfinite(got)).sum()), 'positive', int((got > 1e-3).sum()), flush=True)
    with _lib.pinned(OT_FORM='small'):
        got = scorer.score_pool(q, c, method='ot', schedule='pair').cpu().numpy()
        print(scale, 'small  ', got[0, 1:4].tolist(), 'positive', int((got > 1e-3).sum()), flush=True)
