import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import aspire_oracle as orc
from aspire_amd import scorer, _lib, ops
scale, seed = 2.0, 20
g = torch.Generator().manual_seed(seed)
mk = lambda n: scale * torch.randn(n, 768, generator=g)
q = [mk(8), mk(5)]
c = [mk(int(torch.randint(1, 9, (1,), generator=g))) for _ in range(4100)]
c[1] = torch.cat([q[0][:1], mk(1)]); c[2] = torch.cat([q[0][:1], mk(7)]); c[3] = q[0][:1].clone()
if len(sys.argv) > 1:
    want = [orc.get_similarity(q[0], c[j]) for j in (1, 2, 3)]
for rep in range(3):
    got = scorer.score_pool(q, c, method='ot', schedule='pair').cpu().numpy()[0, 1:4]
    print('call', rep, got.tolist(), flush=True)
with _lib.pinned(OT_FORM='small'):
    print('small', scorer.score_pool(q, c, method='ot', schedule='pair').cpu().numpy()[0, 1:4].tolist())
