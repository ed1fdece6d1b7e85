"""evaluate.score end to end (host included) on the config-4 shape: 50 queries x 125-candidate pools drawn from 800 papers of
3 .. 20 sentences; per-query rank_pool calls with per-pool uploads (round-1 form) against the batched call over pools that
index one resident matrix.   python tools/evalbench.py"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aspire_amd import evaluate as ev
from aspire_amd.repstore import RepStore

rng = np.random.default_rng(0)
pids = [f'p{i}' for i in range(800)]
reps = {p: rng.standard_normal((int(rng.integers(3, 21)), 768)).astype(np.float32) for p in pids}
test_pool = {pids[j]: {'cands': [pids[i] for i in rng.choice(800, 125, replace=False)]} for j in range(50)}
tmp = tempfile.mkdtemp()
for name, kw in (('one rank_pool call per query, pools uploaded per query', dict(queries_per_call=1, resident=False)),
                 ('rank_pools 32 queries per call, pools uploaded per call', dict(resident=False)),
                 ('rank_pools 32 queries per call, resident store', dict())):
    store = RepStore(reps)
    for method in ('ot', 'l2max'):
        ev.score(tmp, test_pool, store, method=method, **kw)          # warm-up (and, resident: the one upload)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ev.score(tmp, test_pool, store, method=method, **kw)
        torch.cuda.synchronize()
        print(f'{method:5s} {name}: {1e3 * (time.perf_counter() - t0) / 3:8.1f} ms per score step (6250 pairs, json written)', flush=True)
