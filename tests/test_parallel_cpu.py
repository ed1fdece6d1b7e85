"""CPU: the sharded ranker's exchange step with world_size 2 over gloo (the GPU path uses the same code over
RCCL), and its host logic: contiguous shard bounds on multiples of 64, merge with the stable tie rule."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import aspire_oracle as orc


def test_shard_bounds_cover_pool_on_multiples():
    from aspire_amd.parallel import shard_bounds
    for n in (0, 1, 63, 64, 65, 1000, 4207, 1_000_000):
        for w in (1, 2, 3, 8):
            edges = [shard_bounds(n, w, r, 64) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            for (lo, hi), (lo2, _) in zip(edges, edges[1:]):
                assert hi == lo2 and lo <= hi
            for lo, hi in edges:
                assert lo % 64 == 0 or lo == n
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) < 128   # one unit of 64 plus the ragged tail


def test_merge_topk_tie_rule_matches_stable_sort():
    from aspire_amd.parallel import merge_topk
    g = torch.Generator().manual_seed(0)
    scores = torch.randn(3, 40, generator=g)
    scores[:, ::5] = 0.5                                  # ties
    idx = torch.stack([torch.randperm(40, generator=g) for _ in range(3)])   # shards arrive in any order
    s, i = merge_topk(scores, idx, 12)
    for q in range(3):
        by_global = [0.0] * 40
        for sc, gi in zip(scores[q].tolist(), idx[q].tolist()):
            by_global[gi] = sc
        want = orc.rank_descending(by_global)[:12]
        assert i[q].tolist() == want
        assert s[q].tolist() == [by_global[j] for j in want]
    # padding entries (idx -1) never win
    idx2 = idx.clone()
    idx2[:, :30] = -1
    s, i = merge_topk(scores, idx2, 12)
    assert (i[:, :10] >= 0).all() and (i[:, 10:] == -1).all()


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_cand, k, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from aspire_amd.parallel import shard_bounds, all_gather_topk, merge_topk
        g = torch.Generator().manual_seed(123)
        full = torch.randn(4, n_cand, generator=g)       # the scores an un-sharded ranker would see
        full[:, ::9] = 1.25
        lo, hi = shard_bounds(n_cand, world, rank, 64)
        local = full[:, lo:hi]
        # local top-k as the HIP kernel produces it: stable descending, global indices
        kk = min(k, hi - lo)
        order = torch.argsort(local, dim=1, descending=True, stable=True)[:, :kk]
        ls = torch.full((4, k), float('-inf'))
        li = torch.full((4, k), -1, dtype=torch.int64)
        ls[:, :kk] = torch.gather(local, 1, order)
        li[:, :kk] = order + lo
        s, i = all_gather_topk(ls, li, k)
        want = [orc.rank_descending(full[q].tolist())[:k] for q in range(4)]
        ok = all(i[q].tolist() == want[q] for q in range(4))
        ok = ok and all(torch.equal(s[q], full[q][want[q]]) for q in range(4))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_cand,k', [(1000, 100), (130, 100)])
def test_all_gather_topk_world2_gloo(n_cand, k):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, n_cand, k, ret), nprocs=2, join=True)
    assert ret[0] is True and ret[1] is True


# ---- config 4: per-query pools sharded by JOB (parallel.rank_pools_sharded's exchange) ------------------------------------------------
def test_job_bounds_and_packing():
    from aspire_amd.parallel import job_bounds, pack_ranked, unpack_ranked
    assert [job_bounds(50, 8, r) for r in range(8)] == [(0, 7), (7, 14), (14, 20), (20, 26), (26, 32), (32, 38), (38, 44), (44, 50)]
    assert [job_bounds(3, 8, r) for r in range(8)] == [(0, 1), (1, 2), (2, 3)] + [(3, 3)] * 5
    s = torch.tensor([[1.5, -2.25, float('-inf'), 0.0, -0.0, float('nan')]])
    i = torch.tensor([[0, 124, -1, 7, 2 ** 31 - 1, 3]])
    s2, i2 = unpack_ranked(pack_ranked(s, i))
    assert torch.equal(s2.view(torch.int32), s.view(torch.int32)) and torch.equal(i2, i)


def _jobs_worker(rank, world, port, n_jobs, k, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from aspire_amd.parallel import job_bounds, all_gather_ranked_jobs
        g = torch.Generator().manual_seed(7)
        sizes = torch.randint(1, k + 1, (n_jobs,), generator=g).tolist()          # ragged pools, the longest <= k
        full = torch.randn(n_jobs, k, generator=g)
        full[:, ::4] = 0.75                                                       # ties: pool order
        want_s = torch.full((n_jobs, k), float('-inf'))
        want_i = torch.full((n_jobs, k), -1, dtype=torch.int64)
        for j, n in enumerate(sizes):
            order = orc.rank_descending(full[j, :n].tolist())
            want_s[j, :n] = full[j, :n][order]
            want_i[j, :n] = torch.tensor(order)
        lo, hi = job_bounds(n_jobs, world, rank)
        # what this rank's aspire_ot_rank_batch_f32 call hands back: its jobs, columns up to ITS longest pool
        if hi > lo:
            kl = max(sizes[lo:hi])
            ls, li = want_s[lo:hi, :kl].clone(), want_i[lo:hi, :kl].clone()
        else:
            ls = li = None
        s, i = all_gather_ranked_jobs(ls, li, n_jobs, k, device=torch.device('cpu'))
        ret[rank] = bool(torch.equal(s, want_s) and torch.equal(i, want_i))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,n_jobs,k', [(2, 50, 125), (8, 50, 125), (8, 5, 40)])
def test_all_gather_ranked_jobs_gloo(world, n_jobs, k):
    """50 CSFCube-sized jobs over 2 and 8 ranks (7 7 6 6 6 6 6 6), and fewer jobs than ranks: every rank ends with every job's
    ranking in the caller's order, (-inf, -1) beyond a pool's size."""
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_jobs_worker, args=(world, port, n_jobs, k, ret), nprocs=world, join=True)
    assert all(ret[r] is True for r in range(world)), dict(ret)
