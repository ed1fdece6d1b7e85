import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from aspire_amd import scorer, _lib, ops
from oracle import aspire_oracle as orc
for scale, seed in ((3.0, 30), (3.0, 31), (3.0, 32), (4.0, 33), (2.5, 34)):
    g = torch.Generator().manual_seed(seed)
    mk = lambda n: scale * torch.randn(n, 768, generator=g)
    q = [mk(8), mk(5)]
    c = [mk(int(torch.randint(1, 9, (1,), generator=g))) for _ in range(4100)]
    c[1] = torch.cat([q[0][:1], mk(1)]); c[2] = torch.cat([q[0][:1], mk(7)]); c[3] = q[0][:1].clone()
    res = {}
    for form in ('fused', 'tile', 'small'):
        with _lib.pinned(OT_FORM=form):
            res[form] = scorer.score_pool(q, c, method='ot', schedule='pair').cpu().numpy()[0, 1:4]
    with _lib.pinned(OT_FORM='fused'):
        qs, cs = ops.DeviceRepSet.from_list(q), ops.DeviceRepSet.from_list(c)
        d = ops.ot_sinkhorn(qs, cs, want=_lib.OT_DISTANCE).cpu().numpy().reshape(2, -1)[0, 1:4]
    want = [orc.get_similarity(q[0], c[j]) for j in (1, 2, 3)]
    print(scale, seed, {k: np.round(v, 4).tolist() for k, v in res.items()}, 'dist(fused)', np.round(d, 4).tolist(), 'oracle', np.round(want, 4).tolist(), flush=True)
