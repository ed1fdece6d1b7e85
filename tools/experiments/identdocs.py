"""Adversarial parity check (round 6): the QUERY ITSELF, its sentences in another order, and half of them, inside its own candidate pool -- documents of 3 .. 40 rows, small /
medium / big pools and a plane store, isotropic rows and rows with a common vector: otAspire against the float64 oracle at 1e-4, tsAspire at 1e-4 (5e-2 beyond 25 rows).
Prints only failures; round 6: none, worst 2.1e-5.   python tools/experiments/identdocs.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from aspire_amd import scorer, ops
from aspire_amd._lib import pinned
from oracle import aspire_oracle as orc
g = torch.Generator().manual_seed(5)
worst = 0
for qlen in (3, 8, 12, 20, 26, 32, 40):
    for nc, pins, planes in ((60, {}, False), (900, {}, False), (4200, {}, False), (2300, dict(COST_PATH='mfma'), True)):
        if qlen > 32 and nc > 100: continue
        for sigma in (0.0, 2.0):
            common = sigma * torch.randn(768, generator=g)
            q = [torch.randn(qlen, 768, generator=g) + common, torch.randn(max(1, qlen - 2), 768, generator=g) + common]
            c = [torch.randn(int(n), 768, generator=g) + common for n in torch.randint(max(1, qlen - 3), qlen + 1, (nc,), generator=g)]
            c[1] = q[0].clone()                       # the query itself
            c[2] = q[0][torch.randperm(qlen, generator=g)].clone()      # its sentences in another order
            c[3] = torch.cat([q[0][:qlen // 2], c[3]])[:qlen]            # half of them
            pool = scorer.CandidatePool(c)
            if planes: pool.prepare_planes()
            with pinned(**pins):
                ot = scorer.score_pool(q, pool, method='ot', schedule='pair').cpu().numpy()
                l2 = scorer.score_pool(q, pool, method='l2max').cpu().numpy()
            ranked = scorer.rank_pools([q[0]], [pool], k=5)[0]
            for j in (1, 2, 3, 7):
                w = orc.get_similarity(q[0].double(), c[j].double())
                e = abs(float(ot[0, j]) - w)
                worst = max(worst, e)
                flag = '' if e < 1e-4 else '  <<<<<< FAIL'
                if flag: print(f'qlen {qlen} nc {nc} planes {planes} sigma {sigma} cand {j}: hip {ot[0,j]:.6f} f64 {w:.6f} err {e:.2e}{flag}')
                l64 = -torch.cdist(q[0].double(), c[j].double()).min().item()
                tol = 5e-2 if (j < 4 and qlen > 25) else 1e-4
                if abs(float(l2[0, j]) - l64) > tol: print(f'   l2max qlen {qlen} nc {nc} planes {planes} cand {j}: hip {l2[0,j]:.6f} f64 {l64:.6f}  <<<<<< FAIL')
            top = [i for i, _ in ranked[:3]]
            if sorted(top[:2]) != [1, 2]: print('   rank: top', ranked[:4])
    print('qlen', qlen, 'done; worst so far', worst, flush=True)
