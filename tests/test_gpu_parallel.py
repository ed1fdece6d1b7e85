"""SURVEY.md 8(e) on the ONE GPU this box has: 2 and 3 ranks spawned on cuda:0 over gloo run the real sharded code path --
ShardedPoolRanker.rank_queries (HIP scoring of the rank's block, key-form local top-k, all-gather, merge kernel) and
rank_queries_full (all-gather of all scores + full stable sort) -- and must reproduce the un-sharded rank_pool bit for
bit: ties across shard edges, an empty shard, a shard shorter than k, and the world * k > 4096 fallback.  Also bench.py's
N = 2 control flow (ASPIRE_BENCH_ONE_GPU=1).  RCCL itself needs more than one GPU: no 1 -> 8 curve exists yet."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _pool(n, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(1, 9, (n,), generator=g).tolist()
    docs = [torch.randn(l, 768, generator=g) for l in lens]
    for i in range(3, n, 11):          # duplicates: exact score ties, some across shard edges (64, 128, ...)
        if i + 61 < n:
            docs[i + 61] = docs[i].clone()
    return docs


def _worker(rank, world, port, n_pool, k, method, out_dir, backend='gloo'):
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
    from aspire_amd.parallel import ShardedPoolRanker
    pool = _pool(n_pool, 5)
    g = torch.Generator().manual_seed(6)
    queries = [torch.randn(8, 768, generator=g), torch.randn(3, 768, generator=g), pool[min(7, n_pool - 1)].clone()]
    ranker = ShardedPoolRanker(pool)
    ts, ti = ranker.rank_queries(queries, k, method=method)
    fs, fi = ranker.rank_queries_full(queries, method=method)
    torch.save({'ts': ts.cpu(), 'ti': ti.cpu(), 'fs': fs.cpu(), 'fi': fi.cpu(), 'lo': ranker.lo, 'n_local': len(ranker.pool)},
               os.path.join(out_dir, f'r{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,n_pool,k,method', [(2, 300, 50, 'ot'), (3, 130, 100, 'ot'), (2, 100, 30, 'l2max'),
                                                   (3, 9000, 1500, 'l2max')])
def test_sharded_ranker_equals_unsharded(tmp_path, world, n_pool, k, method):
    """(3, 130, 100): blocks of 64 / 64 / 2 -> a shard shorter than k; (2, 100, ...): blocks 64 / 36;
    (3, 9000, 1500): world * k > 4096 -> the torch-op merge fallback, and local full sorts beyond one chunk"""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_pool, k, method, str(tmp_path)), nprocs=world, join=True)
    from aspire_amd import scorer
    pool = _pool(n_pool, 5)
    g = torch.Generator().manual_seed(6)
    queries = [torch.randn(8, 768, generator=g), torch.randn(3, 768, generator=g), pool[min(7, n_pool - 1)].clone()]
    scores = scorer.score_pool(queries, pool, method=method).cpu()
    outs = [torch.load(os.path.join(str(tmp_path), f'r{r}.pt')) for r in range(world)]
    assert sum(o['n_local'] for o in outs) == n_pool
    for qi in range(3):
        order = np.argsort(-scores[qi].numpy().astype(np.float64), kind='stable')
        for o in outs:
            kk = min(k, n_pool)
            assert o['ti'][qi, :kk].tolist() == order[:kk].tolist(), (qi, 'topk')
            assert torch.equal(o['ts'][qi, :kk], scores[qi][order[:kk]])
            assert o['fi'][qi].tolist() == order.tolist(), (qi, 'full')
            assert torch.equal(o['fs'][qi], scores[qi][order])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='RCCL needs at least two GPUs (the build boxes have one)')
@pytest.mark.parametrize('n_pool,k,method', [(300, 50, 'ot'), (9000, 1500, 'l2max')])
def test_sharded_ranker_over_rccl(tmp_path, n_pool, k, method):
    """The same comparison with one process per GPU over RCCL (torch.distributed backend 'nccl'): all_gather_into_tensor of GPU
    tensors on the xGMI links instead of gloo's host round trip.  Runs wherever a driver offers two or more GPUs."""
    import torch.multiprocessing as mp
    world = min(torch.cuda.device_count(), 8)
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_pool, k, method, str(tmp_path), 'nccl'), nprocs=world, join=True)
    from aspire_amd import scorer
    pool = _pool(n_pool, 5)
    g = torch.Generator().manual_seed(6)
    queries = [torch.randn(8, 768, generator=g), torch.randn(3, 768, generator=g), pool[min(7, n_pool - 1)].clone()]
    scores = scorer.score_pool(queries, pool, method=method).cpu()
    outs = [torch.load(os.path.join(str(tmp_path), f'r{r}.pt')) for r in range(world)]
    assert sum(o['n_local'] for o in outs) == n_pool
    for qi in range(3):
        order = np.argsort(-scores[qi].numpy().astype(np.float64), kind='stable')
        kk = min(k, n_pool)
        for o in outs:
            assert o['ti'][qi, :kk].tolist() == order[:kk].tolist() and o['fi'][qi].tolist() == order.tolist()
            # (scores of a shard can differ in the last bits from the un-sharded launch: another grid, another kernel form)
            assert torch.allclose(o['ts'][qi, :kk], scores[qi][order[:kk]], atol=1e-4, rtol=0)


def test_empty_shard(tmp_path):
    """4 candidates over 3 ranks with block edges on multiples of 64: ranks 1 and 2 hold nothing"""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(3, port, 4, 3, 'ot', str(tmp_path)), nprocs=3, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), f'r{r}.pt')) for r in range(3)]
    assert [o['n_local'] for o in outs] == [4, 0, 0]
    for o in outs[1:]:
        assert torch.equal(o['ti'], outs[0]['ti']) and torch.equal(o['fi'], outs[0]['fi'])
    assert sorted(outs[0]['fi'][0].tolist()) == [0, 1, 2, 3]


def test_bench_two_ranks_on_one_gpu():
    """bench.py --gpus 2 with both ranks on cuda:0 over gloo: sharded global indices, all-gather layout, merge kernel,
    and the checks bench.py itself makes on the merged ranking (every rank identical, candidates of both shards)"""
    env = dict(os.environ, ASPIRE_BENCH_ONE_GPU='1', MASTER_ADDR='127.0.0.1', ASPIRE_BENCH_E2E_DOCS='256')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '20', '--warmup', '5',
           '--repeats', '6']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    j = json.loads(line)
    assert j['n_gpus'] == 2 and j['steps'] == 20 and j['value'] > 0
    # what the collective backend saw, gathered from every rank (gloo here; 'nccl' and distinct device uuids on a real node)
    r = j['rccl']
    assert r['ranks_seen'] == [0, 1] and r['world_size'] == 2 and r['backend'] == 'gloo'
    assert r['merged_ranking_agrees_on_all_ranks'] is True and r['shards_in_merged_top_k'] == 2
    assert r['all_gather_us']['median'] > 0 and len(r['devices']) == 2
    # config 5's flow on every rank: each encodes its own block, ranks the replicated queries, merges
    e = j['e2e']
    assert len(e['ranks']) == 2 and e['merged_top1_agrees'] and e['docs_per_s'] > 0 and e['pairs_per_s'] > 0
    assert all(r['shards_in_top_k'] == 2 for r in e['ranks'])
    # config 4 sharded by JOB (bench.py: config4_sharded_probe): 25 + 25 jobs, one all-gather, the one-GPU ranking
    c4 = j['config4']
    assert [r['jobs'] for r in c4['ranks']] == [[0, 25], [25, 50]]
    assert c4['all_ranks_hold_the_same_result'] is True and c4['order_equals_one_gpu'] is True
    assert c4['max_abs_score_diff_vs_one_gpu'] < 1e-4 and c4['step_us'] > 0
    assert j['config']['config4_sharded_step_us'] == c4['step_us'] and j['config']['e2e_docs_per_s'] == e['docs_per_s']
