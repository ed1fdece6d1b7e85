"""Config 5's multi-GPU path end to end, rehearsed on the ONE GPU a build box has (2 ranks on cuda:0 over gloo): every rank
encodes ITS block of the corpus straight into HBM (AspireConSent.encode_to_pool), wraps it as its shard
(ShardedPoolRanker.from_resident), ranks the replicated queries against it and merges over the collective -- and gets the
un-sharded ranking.  Reference flow: src/pre_process/pp_gen_nearest.py:141-202 (encode uncached documents, then score every
candidate of the pool, then sort); SURVEY.md section 8(e)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_DOCS, N_Q, K = 200, 9, 40


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _setup():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_gpu_pipeline import _bert, _doc_batches
    from aspire_amd import AspireConSent
    model = AspireConSent(bert_model=_bert(2, seed=3))
    # batches of 8 documents: block edges (multiples of 64) fall on batch edges
    return model, _doc_batches(21, N_DOCS, 8, 3000, 8), _doc_batches(22, N_Q, N_Q, 3000, 8)


def _query_reps(model, qbatches):
    qpool = model.encode_to_pool(qbatches)
    return [qpool.repset.rows[s:s + n].clone() for s, n in zip(qpool.repset.start.tolist(), qpool.repset.len.tolist())]


def _worker(rank, world, port, out_dir, method, planes):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    model, batches, qbatches = _setup()
    from aspire_amd.parallel import ShardedPoolRanker, shard_bounds
    lo, hi = shard_bounds(N_DOCS, world, rank, 64)
    mine = batches[lo // 8:(hi + 7) // 8]                      # this rank's documents only
    block = model.encode_to_pool(mine)
    assert len(block) == hi - lo
    ranker = ShardedPoolRanker.from_resident(block, lo, N_DOCS, planes=planes)
    queries = _query_reps(model, qbatches)
    ts, ti = ranker.rank_queries(queries, K, method=method)
    fs, fi = ranker.rank_queries_full(queries, method=method)
    torch.save({'ts': ts.cpu(), 'ti': ti.cpu(), 'fs': fs.cpu(), 'fi': fi.cpu(), 'n_local': len(block),
                'has_planes': block.repset.planes is not None, 'mu': None if ranker.__dict__.get('mu') is None else ranker.mu.cpu()},
               os.path.join(out_dir, f'r{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('method,planes', [('ot', False), ('l2max', False), ('l2max', True)])
def test_every_rank_encodes_its_block_ranks_and_merges(tmp_path, method, planes):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), method, planes), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), f'r{r}.pt')) for r in range(world)]
    assert [o['n_local'] for o in outs] == [128, 72]
    # un-sharded: the whole corpus encoded by one process into one resident store, ranked by one call
    model, batches, qbatches = _setup()
    from aspire_amd import scorer
    pool = model.encode_to_pool(batches)
    queries = _query_reps(model, qbatches)
    if planes:
        assert all(o['has_planes'] for o in outs) and torch.equal(outs[0]['mu'], outs[1]['mu'])
        pool.prepare_planes(mu=outs[0]['mu'].cuda())             # the shards' centre: the same roundings
    scores = scorer.score_pool(queries, pool, method=method).cpu()
    for qi in range(N_Q):
        order = np.argsort(-scores[qi].numpy().astype(np.float64), kind='stable')
        for o in outs:
            # (the blocks' encoder calls hold other documents than the un-sharded store's: reps agree to rounding, scores to 1e-4,
            # the order wherever two scores are further apart than that)
            assert torch.allclose(o['fs'][qi], scores[qi][order], atol=1e-4, rtol=0)
            for a, b in zip(o['fi'][qi].tolist(), order.tolist()):
                assert a == b or abs(scores[qi][a] - scores[qi][b]) < 2e-4
            assert o['ti'][qi].tolist() == o['fi'][qi][:K].tolist() and torch.equal(o['ts'][qi], o['fs'][qi][:K])
    assert torch.equal(outs[0]['fi'], outs[1]['fi']) and torch.equal(outs[0]['fs'], outs[1]['fs'])


def test_many_queries_on_plane_shards_equal_the_plane_store_bit_for_bit(tmp_path):
    """shards that hold copies of the SAME rows (no re-encoding) around one broadcast centre: 40 queries (more than 64 query rows:
    the fp16-plane cost tiles run) give the same bits sharded and un-sharded"""
    import torch.multiprocessing as mp
    world = 2
    g = torch.Generator().manual_seed(9)
    docs = [torch.randn(int(n), 768, generator=g) + 0.7 for n in torch.randint(1, 9, (9000,), generator=g)]          # each shard well beyond 128 candidate tiles
    queries = [torch.randn(8, 768, generator=g) + 0.7 for _ in range(40)]
    torch.save({'docs': docs, 'queries': queries}, os.path.join(str(tmp_path), 'in.pt'))
    mp.spawn(_plane_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), f'p{r}.pt')) for r in range(world)]
    from aspire_amd import scorer
    from aspire_amd._lib import pinned
    pool = scorer.CandidatePool(docs).prepare_planes(mu=outs[0]['mu'].cuda())
    with pinned(COST_PATH='mfma'):
        scores = scorer.score_pool(queries, pool, method='l2max').cpu()
    for qi in range(len(queries)):
        order = np.argsort(-scores[qi].numpy().astype(np.float64), kind='stable')
        for o in outs:
            assert o['ti'][qi].tolist() == order[:50].tolist()
            assert torch.equal(o['ts'][qi], scores[qi][order[:50]])


def _plane_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from aspire_amd.parallel import ShardedPoolRanker
    from aspire_amd._lib import pinned
    z = torch.load(os.path.join(out_dir, 'in.pt'))
    ranker = ShardedPoolRanker(z['docs']).prepare_planes()
    with pinned(COST_PATH='mfma'):
        ts, ti = ranker.rank_queries(z['queries'], 50, method='l2max')
    torch.save({'ts': ts.cpu(), 'ti': ti.cpu(), 'mu': ranker.mu.cpu()}, os.path.join(out_dir, f'p{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()
