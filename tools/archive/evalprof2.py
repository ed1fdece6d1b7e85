import os, sys, tempfile, time, cProfile, pstats
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from aspire_amd import evaluate as ev
from aspire_amd.repstore import RepStore
rng = np.random.default_rng(0)
pids = [f'p{i}' for i in range(800)]
reps = {p: rng.standard_normal((int(rng.integers(3, 21)), 768)).astype(np.float32) for p in pids}
test_pool = {pids[j]: {'cands': [pids[i] for i in rng.choice(800, 125, replace=False)]} for j in range(50)}
tmp = tempfile.mkdtemp()
store = RepStore(reps)
for _ in range(3): ev.score(tmp, test_pool, store, method='ot')
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): ev.score(tmp, test_pool, store, method='ot')
torch.cuda.synchronize()
print('ms per step', 1e3 * (time.perf_counter() - t0) / 20)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): ev.score(tmp, test_pool, store, method='ot')
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
