"""Adversarial parity check (round 6): otAspire under OTHER hyper-parameters than the defaults every other sweep uses -- geoml_blur, geoml_scaling, sent_sm_temp drawn at random
(pair_distances.py:24-30: the reference reads them from model_hparams) -- across the kernel families (small / one-wave / fused / tile16 / plane tiles / batched CHUNK), sampled pairs
against the fp32 oracle.   python tools/experiments/hparamfuzz.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from aspire_amd import scorer
from oracle import aspire_oracle as orc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(n_cases):
    hp = {'geoml_blur': float(rng.choice([0.01, 0.05, 0.05, 0.2, 0.5])), 'geoml_scaling': float(rng.choice([0.5, 0.8, 0.9, 0.9, 0.95])),
          'sent_sm_temp': float(rng.choice([0.1, 1.0, 1.0, 5.0]))}
    smax = int(rng.choice([8, 8, 14, 20, 32]))
    nq = int(rng.choice([1, 2, 24]))
    nc = int(rng.choice([40, 900, 2600, 4200]))
    planes = nq == 24 and nc >= 2300 and rng.random() < 0.7
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    q = [torch.randn(int(rng.integers(1, smax + 1)), 768, generator=g) for _ in range(nq)]
    c = [torch.randn(int(rng.integers(1, smax + 1)), 768, generator=g) for _ in range(nc)]
    pool = scorer.CandidatePool(c)
    if planes:
        pool.prepare_planes()
    ot = scorer.score_pool(q, pool, method='ot', schedule='pair', hparams=hp).cpu().numpy()
    assert np.isfinite(ot).all(), (case, hp)
    errs = []
    for _ in range(8):
        i, j = int(rng.integers(nq)), int(rng.integers(nc))
        w = orc.get_similarity(q[i], c[j], hp)
        errs.append(abs(float(ot[i, j]) - w))
    # batched: the first query against a pool of its own (CHUNK / fused / small forms by size)
    ranked = scorer.rank_pools([q[0][:min(8, len(q[0]))]], [c[:min(nc, 600)]], k=3, hparams=hp)[0]
    for i, s in ranked:
        errs.append(abs(s - orc.get_similarity(q[0][:min(8, len(q[0]))], c[i], hp)))
    e = max(errs)
    worst = max(worst, e)
    print(f'case {case}: {hp} Q={nq} C={nc} S<={smax} planes={planes}: max err {e:.2e}' + ('' if e < 1e-4 else '   <<<<<< FAIL'), flush=True)
print('worst', worst)
