import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from aspire_amd import scorer, _lib, ops
from oracle import aspire_oracle as orc
scale, seed = 2.0, 20
g = torch.Generator().manual_seed(seed)
mk = lambda n: scale * torch.randn(n, 768, generator=g)
q = [mk(8), mk(5)]
c = [mk(int(torch.randint(1, 9, (1,), generator=g))) for _ in range(4100)]
c[1] = torch.cat([q[0][:1], mk(1)]); c[2] = torch.cat([q[0][:1], mk(7)]); c[3] = q[0][:1].clone()
keep = [t.clone() for t in (q[0], c[1], c[2], c[3])]
got0 = scorer.score_pool(q, c, method='ot', schedule='pair').cpu().numpy()[0, 1:4]
want = [orc.get_similarity(q[0], c[j]) for j in (1, 2, 3)]
print('inputs unchanged by the oracle:', [torch.equal(a, b) for a, b in zip(keep, (q[0], c[1], c[2], c[3]))], 'default dtype', torch.get_default_dtype(), 'threads', torch.get_num_threads())
got1 = scorer.score_pool(q, c, method='ot', schedule='pair').cpu().numpy()[0, 1:4]
print('before oracle', got0.tolist(), 'after oracle', got1.tolist(), 'oracle', want)
