"""Error of the GPU OT distance against the float64 and float32 CPU oracle on config-2 style data."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aspire_amd import scorer
from oracle import aspire_oracle as orc
g = torch.Generator().manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
query = torch.randn(8, 768, generator=g)
cands = [torch.randn(8, 768, generator=g) for _ in range(n)]
got = scorer.score_pool([query], cands, method='ot', schedule='pair').cpu().numpy()[0].astype(np.float64)
w32 = np.array([orc.get_similarity(query, c) for c in cands])
w64 = np.array([orc.get_similarity(query.double(), c.double()) for c in cands])
print(f'n={n}: |gpu-f64| max {np.abs(got-w64).max():.2e} mean {np.abs(got-w64).mean():.2e} | |cpu32-f64| max {np.abs(w32-w64).max():.2e} mean {np.abs(w32-w64).mean():.2e} | |gpu-cpu32| max {np.abs(got-w32).max():.2e}')
# ragged
cands = [torch.randn(int(torch.randint(1, 9, (1,), generator=g)), 768, generator=g) for _ in range(n)]
q5 = query[:5]
got = scorer.score_pool([q5], cands, method='ot', schedule='pair').cpu().numpy()[0].astype(np.float64)
w64 = np.array([orc.get_similarity(q5.double(), c.double()) for c in cands])
w32 = np.array([orc.get_similarity(q5, c) for c in cands])
print(f'ragged: |gpu-f64| max {np.abs(got-w64).max():.2e} | |cpu32-f64| max {np.abs(w32-w64).max():.2e}')
