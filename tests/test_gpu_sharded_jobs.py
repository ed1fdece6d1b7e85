"""GPU: BASELINE config 4 sharded by JOB.  CSFCube gives every query its OWN pool of ~125 candidates
(src/evaluation/evaluate.py:58-76), so candidate blocks on multiples of 64 would leave six of eight ranks empty; the jobs are dealt out
instead (parallel.rank_pools_sharded, evaluate.score(..., sharded)): world = 8 (and 3) ranks on cuda:0 over gloo run the real code --
each rank one aspire_ot_rank_batch_f32 call on its block, one all-gather of the ranked lists -- and must reproduce the un-sharded step."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LABELS = ['background_label', 'method_label', 'result_label']


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _dataset(n_jobs=50, pool_size=125, seed=91):
    """A CSFCube-shaped test pool: abstracts of 3 .. 20 sentences (pp_settings.py:2-3), every query with its own pool (sizes around
    125, papers shared between pools, one pool short, one empty), per-sentence facet labels, a duplicate paper (an exact tie)."""
    g = torch.Generator().manual_seed(seed)
    n_papers = 900
    pids = [f'p{i}' for i in range(n_papers)]
    lens = torch.randint(3, 21, (n_papers,), generator=g).tolist()
    reps = {p: torch.randn(n, 768, generator=g).numpy() for p, n in zip(pids, lens)}
    reps['p401'] = reps['p400'].copy()
    labels = {p: [LABELS[(i + j) % 3] for j in range(reps[p].shape[0])] for i, p in enumerate(pids)}
    test_pool = {}
    for j in range(n_jobs):
        size = pool_size + int(torch.randint(-8, 9, (1,), generator=g)) if j not in (11, 29) else (0 if j == 11 else 5)
        cands = [pids[int(c)] for c in torch.randperm(n_papers - 60, generator=g)[:size] + 60]
        if j == 3:
            cands[10], cands[40] = 'p400', 'p401'
        test_pool[pids[j]] = {'cands': cands}
    return reps, labels, test_pool


def _score_worker(rank, world, port, out_dir, deterministic, method, backend='gloo'):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    if backend == 'nccl':              # one process per GPU over RCCL / xGMI: the real thing
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(rank)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
    else:
        torch.cuda.set_device(0)
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from aspire_amd import evaluate as ev
    from aspire_amd.parallel import job_bounds
    from aspire_amd.repstore import RepStore
    reps, labels, test_pool = _dataset()
    store = RepStore(reps)
    res = ev.score(os.path.join(out_dir, 'sharded'), test_pool, store, facet='method', pred_labels=labels, method=method,
                   deterministic=deterministic)
    lo, hi = job_bounds(len(test_pool), world, rank)
    mine = set().union(*[set(test_pool[q]['cands']) for q in list(test_pool)[lo:hi]]) if hi > lo else set()
    with open(os.path.join(out_dir, f'r{rank}.json'), 'w') as fp:
        json.dump({'res': res, 'resident_papers': len(getattr(store, '_dev_index', {}) or {}), 'papers_of_my_pools': len(mine)}, fp)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('world,deterministic,method', [(8, True, 'ot'), (8, False, 'ot'), (3, False, 'l2max')])
def test_score_step_sharded_by_job_equals_the_single_process_step(tmp_path, world, deterministic, method):
    import torch.multiprocessing as mp
    from aspire_amd import evaluate as ev
    from aspire_amd.repstore import RepStore
    reps, labels, test_pool = _dataset()
    one = ev.score(str(tmp_path / 'one'), test_pool, RepStore(reps), facet='method', pred_labels=labels, method=method,
                   deterministic=deterministic, sharded=False)
    mp.spawn(_score_worker, args=(world, _free_port(), str(tmp_path), deterministic, method), nprocs=world, join=True)
    outs = [json.load(open(tmp_path / f'r{r}.json')) for r in range(world)]
    written = json.load(open(ev.get_scores_filename(str(tmp_path / 'sharded'), 'method')))
    want = json.loads(json.dumps(one))
    for o in outs:
        assert o['res'] == outs[0]['res']                                   # every rank returns the whole result
        assert 0 < o['resident_papers'] == o['papers_of_my_pools'] < 900    # ... having uploaded only its own pools' papers
    assert written == outs[0]['res']
    assert list(written) == list(want)
    if deterministic:
        assert written == want                                              # one kernel form: the same bits whatever the split
    else:
        n_moved = 0
        for q in want:
            a, b = written[q], want[q]
            assert len(a) == len(b) == len(test_pool[q]['cands'])
            assert sorted(c for c, _ in a) == sorted(c for c, _ in b)
            np.testing.assert_allclose([s for _, s in a], [s for _, s in b], atol=1e-4, rtol=0)
            sb = dict(b)
            for (ca, _), (cb, _) in zip(a, b):                              # positions may differ only between near-ties
                if ca != cb:
                    n_moved += 1
                    assert abs(sb[ca] - sb[cb]) < 1e-4
        assert n_moved <= 4
    # the exact tie keeps pool order (evaluate.py:76: stable sort) on whichever rank ranked that job
    q3 = [c for c, _ in written['p3']]
    assert q3.index('p400') + 1 == q3.index('p401')
    assert written['p11'] == [] and len(written['p29']) == 5


@pytest.mark.timeout(900)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='RCCL needs at least two GPUs (the build boxes have one)')
def test_score_step_sharded_by_job_over_rccl(tmp_path):
    """the same step with one process per GPU over RCCL (torch.distributed backend 'nccl'): the ranked lists' all-gather moves GPU tensors
    over xGMI.  Runs wherever a driver offers two or more GPUs."""
    import torch.multiprocessing as mp
    from aspire_amd import evaluate as ev
    from aspire_amd.repstore import RepStore
    world = min(torch.cuda.device_count(), 8)
    reps, labels, test_pool = _dataset()
    one = ev.score(str(tmp_path / 'one'), test_pool, RepStore(reps), facet='method', pred_labels=labels, deterministic=True, sharded=False)
    mp.spawn(_score_worker, args=(world, _free_port(), str(tmp_path), True, 'ot', 'nccl'), nprocs=world, join=True)
    written = json.load(open(ev.get_scores_filename(str(tmp_path / 'sharded'), 'method')))
    assert written == json.loads(json.dumps(one))
    outs = [json.load(open(tmp_path / f'r{r}.json')) for r in range(world)]
    assert all(o['res'] == written for o in outs)


def _rank_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from aspire_amd.parallel import job_bounds, rank_pools_sharded
    reps, labels, test_pool = _dataset(n_jobs=20, pool_size=60, seed=93)
    qids = list(test_pool)
    lo, hi = job_bounds(len(qids), world, rank)
    # a rank holds ONLY its block's queries and pools; the others are None (their sizes are known everywhere)
    queries = [reps[q][:6] if lo <= j < hi else None for j, q in enumerate(qids)]
    pools = [[reps[c] for c in test_pool[q]['cands']] if lo <= j < hi else None for j, q in enumerate(qids)]
    ts, ti = rank_pools_sharded(queries, pools, k=25, pool_sizes=[len(test_pool[q]['cands']) for q in qids], deterministic=True)
    torch.save({'ts': ts.cpu(), 'ti': ti.cpu()}, os.path.join(out_dir, f'k{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_rank_pools_sharded_top_k(tmp_path):
    """rank_pools_sharded with k < the pools' sizes on 8 ranks (20 jobs: 3 3 3 3 2 2 2 2), every rank holding only its own block:
    the un-sharded rank_pools' top-25, bit for bit (deterministic form)."""
    import torch.multiprocessing as mp
    from aspire_amd import scorer
    mp.spawn(_rank_worker, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    reps, labels, test_pool = _dataset(n_jobs=20, pool_size=60, seed=93)
    qids = list(test_pool)
    ranked = scorer.rank_pools([reps[q][:6] for q in qids], [[reps[c] for c in test_pool[q]['cands']] for q in qids], k=25,
                               deterministic=True)
    outs = [torch.load(tmp_path / f'k{r}.pt') for r in range(8)]
    for o in outs[1:]:
        assert torch.equal(o['ts'], outs[0]['ts']) and torch.equal(o['ti'], outs[0]['ti'])
    for j, r in enumerate(ranked):
        n = len(r)
        assert outs[0]['ti'][j, :n].tolist() == [i for i, _ in r]
        assert outs[0]['ts'][j, :n].tolist() == [s for _, s in r]
        assert (outs[0]['ti'][j, n:] == -1).all()
