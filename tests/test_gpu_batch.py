"""aspire_ot_rank_batch_f32: J independent (query, pool) re-ranks in ONE call -- the per-query loop of evaluate.py:58-76
batched over queries -- against J separate aspire_ot_rank_f32 calls, the oracle, and Python's stable sort.

Bit-for-bit equality holds whenever both sides run the same kernel forms (the arithmetic of a pair never depends on the
grid): the small-pool forms for small batches, the throughput forms when the one-job calls are pinned to them as well.  Between DIFFERENT forms (another summation order) scores agree to a few 1e-5."""
import numpy as np
import pytest
import torch

from oracle import aspire_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope='module')
def amd():
    from aspire_amd import ops, scorer, _lib
    assert torch.cuda.is_available()
    return type('NS', (), dict(ops=ops, scorer=scorer, lib=_lib, pinned=_lib.pinned))


def _jobs(seed, sizes, smax, smin=1, d=768):
    """queries [J], pools [J][n_j] of ragged documents"""
    g = torch.Generator().manual_seed(seed)
    queries = [torch.randn(int(torch.randint(smin, smax + 1, (1,), generator=g)), d, generator=g) for _ in sizes]
    pools = [[torch.randn(int(n), d, generator=g) for n in torch.randint(smin, smax + 1, (int(sz),), generator=g)] for sz in sizes]
    return queries, pools


def _batch(amd, queries, pools, k, **kw):
    q = amd.ops.DeviceRepSet.from_list(queries)
    c = amd.ops.DeviceRepSet.from_list([d for p in pools for d in p])
    sizes = [len(p) for p in pools]
    job_off = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32).cuda()
    s, ts, ti = amd.ops.ot_rank_batch(q, c, job_off, max(sizes), k, **kw)
    torch.cuda.synchronize()
    off = job_off.cpu().numpy()
    return [s[off[j]:off[j + 1]].cpu() for j in range(len(pools))], ts.cpu(), ti.cpu()


def _check_rank(scores_j, ts, ti, k):
    """every job's list = the stable descending sort of ITS OWN scores (evaluate.py:76), padded with (-inf, -1)"""
    for j, s in enumerate(scores_j):
        n = len(s)
        order = np.argsort(-s.numpy().astype(np.float64), kind='stable')[:k]
        kk = min(k, n)
        assert ti[j, :kk].tolist() == order.tolist(), j
        assert torch.equal(ts[j, :kk], s[order]), j
        assert torch.all(ti[j, kk:] == -1) and torch.all(ts[j, kk:] == float('-inf'))


def test_small_batch_equals_separate_calls_bit_for_bit(amd):
    """ragged pools (one of them empty, one with a single candidate), ragged documents of 1..8 rows"""
    queries, pools = _jobs(11, [37, 0, 1, 64, 5, 120], 8)
    k = 50
    sc, ts, ti = _batch(amd, queries, pools, k)
    _check_rank(sc, ts, ti, k)
    for j, (qd, pool) in enumerate(zip(queries, pools)):
        if not pool:
            continue
        kk = min(k, len(pool))
        s1, t1, i1 = amd.ops.ot_rank(amd.ops.DeviceRepSet.from_list([qd]), amd.ops.DeviceRepSet.from_list(pool), kk,
                                     want=amd.lib.OT_SIMILARITY)
        assert torch.equal(s1[0].cpu(), sc[j]), j
        assert torch.equal(i1[0].cpu(), ti[j, :kk]) and torch.equal(t1[0].cpu(), ts[j, :kk])
        want = np.array([orc.get_similarity(qd, c) for c in pool], dtype=np.float32)
        np.testing.assert_allclose(sc[j].numpy(), want, atol=TOL, rtol=0)


@pytest.mark.parametrize('smax', [12, 20, 32])
def test_long_document_batches(amd, smax):
    """documents of more than 8 rows: sub-tile items (few pairs) and the per-pair tile-loop kernel (many pairs)"""
    for sizes in ([30, 45, 7], [300, 420, 100]):
        queries, pools = _jobs(20 + smax, sizes, smax, smin=max(1, smax - 9))
        sc, ts, ti = _batch(amd, queries, pools, 25)
        _check_rank(sc, ts, ti, 25)
        for j in (0, 2):
            s1 = amd.scorer.score_pool([queries[j]], pools[j], method='ot', schedule='pair')[0].cpu()
            if sum(sizes) < 512:       # both sides on the sub-tile kernel
                assert torch.equal(s1, sc[j]), (sizes, j)
            else:                      # per-pair tile-loop kernel here, sub-tile items there
                np.testing.assert_allclose(sc[j].numpy(), s1.numpy(), atol=3e-5, rtol=0)
        want = np.array([orc.get_similarity(queries[1], c) for c in pools[1][:12]], dtype=np.float32)
        np.testing.assert_allclose(sc[1][:12].numpy(), want, atol=TOL, rtol=0)


def test_throughput_forms_and_single_jobs(amd):
    """the throughput kernels (four candidates of ONE job per wave; costs + solves fused in one launch, or the tile cost
    kernel + the block Sinkhorn kernel): job sizes that are not multiples of four, jobs of fewer than four candidates.
    A job scored alone on the same form gives the same bits; the forms agree with each other and the oracle."""
    sizes = [1503, 2, 997, 1250, 3, 2048, 1, 1100, 777]
    queries, pools = _jobs(31, sizes, 8)
    k = 100
    out = {}
    for form in ('fused', 'tile'):
        with amd.pinned(OT_FORM=form):
            out[form] = _batch(amd, queries, pools, k)
            _check_rank(*out[form], k)
            for j in (0, 1, 4, 6, 8):      # J separate one-job calls on the same kernel form: bit for bit
                one = _batch(amd, [queries[j]], [pools[j]], k)
                assert torch.equal(one[0][0], out[form][0][j]), (form, j)
                assert torch.equal(one[2][0], out[form][2][j]) and torch.equal(one[1][0], out[form][1][j]), (form, j)
    dflt = _batch(amd, queries, pools, k)          # C = 8681: the default is the fused form
    for a, b, t in zip(dflt[0], out['fused'][0], out['tile'][0]):
        assert torch.equal(a, b)
        np.testing.assert_allclose(b.numpy(), t.numpy(), atol=5e-5, rtol=0)
    # ... and the default single-pool entry point (small-pool kernels, another summation order) to a few 1e-5
    for j in (0, 2, 6):
        s1 = amd.scorer.score_pool([queries[j]], pools[j], method='ot', schedule='pair')[0].cpu()
        np.testing.assert_allclose(dflt[0][j].numpy(), s1.numpy(), atol=5e-5, rtol=0)
    rng = np.random.default_rng(0)
    for j in (0, 3, 5):
        pick = rng.choice(sizes[j], 6, replace=False)
        want = np.array([orc.get_similarity(queries[j], pools[j][i]) for i in pick], dtype=np.float32)
        np.testing.assert_allclose(dflt[0][j].numpy()[pick], want, atol=TOL, rtol=0)


def test_fused_in_wave_tables_equal_the_tables_launch(amd):
    """batches of <= 64 jobs: the fused kernel derives an item's documents and the query's box itself (no tables launch);
    pinned back to the tables launch + table-driven kernel it gives the same bits -- ragged jobs, empty jobs in front,
    in the middle and at the end, exactly 64 jobs; 65 jobs take the tables launch by themselves"""
    for seed, sizes in ((51, [0, 801, 2, 0, 0, 1203, 5, 640, 0]), (52, [37] * 63 + [41]), (53, [33] * 65)):
        queries, pools = _jobs(seed, sizes, 8)
        with amd.pinned(OT_FORM='fused'):
            a = _batch(amd, queries, pools, 50)
        with amd.pinned(OT_FORM='fused', FUSED_NOSELF=1):
            b = _batch(amd, queries, pools, 50)
        _check_rank(*a, 50)
        for x, y in zip(a[0], b[0]):
            assert torch.equal(x, y)
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        j = 1 if seed == 51 else 7
        want = np.array([orc.get_similarity(queries[j], c) for c in pools[j][:8]], dtype=np.float32)
        np.testing.assert_allclose(a[0][j].numpy()[:8], want, atol=TOL, rtol=0)


def test_batch_hparams_and_wants(amd):
    queries, pools = _jobs(41, [40, 25], 8)
    for want, sign in ((amd.lib.OT_DISTANCE, -1.0), (amd.lib.OT_SIMILARITY, 1.0)):
        sc, ts, ti = _batch(amd, queries, pools, 10, want=want, blur=0.1, scaling=0.8, sent_sm_temp=5.0)
        _check_rank(sc, ts, ti, 10)
        hp = dict(geoml_blur=0.1, geoml_scaling=0.8, sent_sm_temp=5.0)
        ref = np.array([orc.get_similarity(queries[1], c, hp) for c in pools[1]], dtype=np.float32)
        np.testing.assert_allclose(sign * sc[1].numpy(), ref, atol=TOL, rtol=0)


@pytest.mark.parametrize('long_doc', [False, True])
def test_batch_under_graph_capture(amd, long_doc):
    """the call is capturable into a hipGraph (everything runs on the caller's stream) and replays to the same bits -- also
    the device-side hybrid (a 13-row document among the pools: counter reset, census, gated kernels are all stream work)"""
    sizes = [1200] * 8
    queries, pools = _jobs(51, sizes, 8, smin=8)
    if long_doc:
        pools[3][17] = torch.randn(13, 768, generator=torch.Generator().manual_seed(1))
    q = amd.ops.DeviceRepSet.from_list(queries)
    c = amd.ops.DeviceRepSet.from_list([d for p in pools for d in p])
    job_off = torch.arange(0, 9601, 1200, dtype=torch.int32).cuda()
    eager = amd.ops.ot_rank_batch(q, c, job_off, 1200, 100)
    torch.cuda.synchronize()
    out = tuple(torch.empty_like(t) for t in eager)
    qs, cs = q.struct(), c.struct()
    import ctypes
    ws = torch.empty(amd.lib.lib.aspire_ot_rank_batch_workspace_bytes(ctypes.byref(qs), ctypes.byref(cs), 1200, 100),
                     dtype=torch.uint8, device='cuda')
    amd.ops.ot_rank_batch(q, c, job_off, 1200, 100, out=out, workspace=ws)      # warm up outside the capture
    torch.cuda.synchronize()
    for t in out:
        t.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        amd.ops.ot_rank_batch(q, c, job_off, 1200, 100, out=out, workspace=ws)
    g.replay()
    torch.cuda.synchronize()
    for a, b in zip(eager, out):
        assert torch.equal(a, b)


def test_back_to_back_batches_share_a_workspace(amd):
    """consecutive calls on one stream reuse the same workspace (item counter, tables, rank scratch)"""
    import ctypes
    sizes = [1000] * 10
    runs = []
    ws = None
    for seed in (61, 62, 63):
        queries, pools = _jobs(seed, sizes, 8, smin=6)
        q = amd.ops.DeviceRepSet.from_list(queries)
        c = amd.ops.DeviceRepSet.from_list([d for p in pools for d in p])
        runs.append((q, c))
    job_off = torch.arange(0, 10001, 1000, dtype=torch.int32).cuda()
    qs, cs = runs[0][0].struct(), runs[0][1].struct()
    ws = torch.empty(amd.lib.lib.aspire_ot_rank_batch_workspace_bytes(ctypes.byref(qs), ctypes.byref(cs), 1000, 100) + 4096,
                     dtype=torch.uint8, device='cuda')
    alone = []
    for q, c in runs:
        alone.append(amd.ops.ot_rank_batch(q, c, job_off, 1000, 100))
        torch.cuda.synchronize()
    outs = [tuple(torch.empty_like(t) for t in alone[0]) for _ in runs]
    for rep in range(3):
        for (q, c), out in zip(runs, outs):
            amd.ops.ot_rank_batch(q, c, job_off, 1000, 100, out=out, workspace=ws)
    torch.cuda.synchronize()
    for a, b in zip(alone, outs):
        for x, y in zip(a, b):
            assert torch.equal(x, y)


def test_rank_pools_is_the_evaluate_loop(amd):
    """scorer.rank_pools(queries, pools) = [rank_pool(q, pool) for q, pool in ...], pids and all"""
    queries, pools = _jobs(71, [33, 18, 60], 8)
    cps = [amd.scorer.CandidatePool(p, pids=[f'j{j}c{i}' for i in range(len(p))]) for j, p in enumerate(pools)]
    got = amd.scorer.rank_pools(queries, cps)
    for j, (qd, cp) in enumerate(zip(queries, cps)):
        want = amd.scorer.rank_pool([qd], cp)[0]
        assert got[j] == want
    assert amd.scorer.rank_pools([], []) == []
    assert amd.scorer.rank_pools(queries[:1], [[]]) == [[]]


def test_config2_batch_of_twenty_full_size(amd):
    """BASELINE config 2 as the bench runs it: 20 steps = 20 (query, 1000-candidate pool) jobs, 8 sentences x 768-d.
    Size-independent properties: each job's list is the stable sort of its scores, a job scored alone gives the same
    bits (pinned to the same forms), permuting the jobs permutes the outputs, oracle spot checks."""
    J, n, S = 20, 1000, 8
    g = torch.Generator().manual_seed(0)
    qrows = torch.randn(J * S, 768, generator=g).cuda()
    crows = torch.randn(J * n * S, 768, generator=g).cuda()
    ar = torch.arange(J * n, dtype=torch.int32).cuda()
    q = amd.ops.DeviceRepSet(qrows, (ar[:J] * S).contiguous(), torch.full((J,), S, dtype=torch.int32).cuda(), 0, S)
    c = amd.ops.DeviceRepSet(crows, (ar * S).contiguous(), torch.full((J * n,), S, dtype=torch.int32).cuda(), 0, S)
    job_off = (torch.arange(J + 1, dtype=torch.int32) * n).cuda()
    s, ts, ti = amd.ops.ot_rank_batch(q, c, job_off, n, 100)
    torch.cuda.synchronize()
    sj = [s[j * n:(j + 1) * n].cpu() for j in range(J)]
    _check_rank(sj, ts.cpu(), ti.cpu(), 100)
    assert torch.isfinite(s).all() and (s < 0).all()
    # reversed job order
    perm = torch.arange(J - 1, -1, -1)
    q2 = amd.ops.DeviceRepSet(qrows.view(J, S, 768)[perm.cuda()].reshape(J * S, 768).contiguous(), q.start, q.len, 0, S)
    c2 = amd.ops.DeviceRepSet(crows.view(J, n * S, 768)[perm.cuda()].reshape(J * n * S, 768).contiguous(), c.start, c.len, 0, S)
    s2, ts2, ti2 = amd.ops.ot_rank_batch(q2, c2, job_off, n, 100)
    assert torch.equal(s2.view(J, n)[perm.cuda()], s.view(J, n)) and torch.equal(ti2[perm.cuda()], ti)
    for j in (0, 7, 19):
        want = np.array([orc.get_similarity(qrows[j * S:(j + 1) * S].cpu(), crows[(j * n + i) * S:(j * n + i + 1) * S].cpu())
                         for i in (0, 499, 999)], dtype=np.float32)
        np.testing.assert_allclose(sj[j].numpy()[[0, 499, 999]], want, atol=TOL, rtol=0)


def test_in_flight_ranker_equals_rank_pools(amd):
    """independent rank_pools calls in flight on three streams (own buffers each): the same ranked lists, whatever the
    order in which the results are collected; an empty request and a request of empty pools among them"""
    reqs = []
    for seed, sizes in ((71, [300, 5, 0, 1200]), (72, [2500] * 3), (73, [17]), (74, [0, 0]), (75, [900, 901, 902, 903, 904])):
        reqs.append(_jobs(seed, sizes, 8))
    reqs.insert(2, ([], []))
    want = [amd.scorer.rank_pools(q, p, k=50) for q, p in reqs]
    ranker = amd.scorer.InFlightRanker(n_lanes=3, k=50)
    for rounds in range(2):
        tickets = [ranker.submit(q, p) for q, p in reqs]
        order = range(len(tickets)) if rounds == 0 else reversed(range(len(tickets)))
        for i in order:
            assert ranker.result(tickets[i]) == want[i], i


@pytest.mark.parametrize('smax,sizes', [(8, [1503, 2, 0, 997, 1250, 3, 2048, 1, 1100]),      # fused max-sim form
                                        (14, [1203, 1, 998, 2, 1501, 700, 0, 1600]),       # 16-row streaming kernel (most pairs long)
                                        (8, [40, 0, 25, 3]),                                # tiny (<= 64 jobs: still the streaming kernel, in-wave tables)
                                        (8, [340, 0, 325, 3, 297, 310]), (15, [330, 2, 0, 325, 301]),  # >= 384 groups of four: the streaming kernels again
                                        (23, [300, 1, 77]), (32, [150, 0, 61])])            # 17 .. 32 rows: one workgroup per candidate
def test_l2max_rank_batch(amd, smax, sizes):
    """tsAspire over batched jobs (aspire_l2max_rank_batch_f32, rank_pools(method='l2max')): every job's scores = the
    per-pool call's, = -min cdist in torch on a sample; every list = the stable descending sort of its own scores"""
    queries, pools = _jobs(90 + smax, sizes, smax)
    q = amd.ops.DeviceRepSet.from_list(queries)
    c = amd.ops.DeviceRepSet.from_list([d for p in pools for d in p])
    job_off = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32).cuda()
    k = 30
    s, ts, ti = amd.ops.l2max_rank_batch(q, c, job_off, max(sizes), k)
    torch.cuda.synchronize()
    off = job_off.cpu().numpy()
    sc = [s[off[j]:off[j + 1]].cpu() for j in range(len(sizes))]
    _check_rank(sc, ts.cpu(), ti.cpu(), k)
    for j, n in enumerate(sizes):
        if n == 0:
            continue
        one = amd.scorer.score_pool([queries[j]], pools[j], method='l2max')[0].cpu().numpy()
        np.testing.assert_allclose(sc[j].numpy(), one, atol=4e-5, rtol=0)
        for i in (0, n - 1):
            want = -torch.cdist(queries[j], pools[j][i]).min().item()
            assert abs(sc[j][i].item() - want) < 4e-5, (j, i)
    ranked = amd.scorer.rank_pools(queries, pools, k=k, method='l2max')
    for j, n in enumerate(sizes):
        assert [i for i, _ in ranked[j]] == ti.cpu()[j, :min(k, n)].tolist()
    # the other form of the same batch (streaming kernels <-> one workgroup per candidate), pinned
    for form in (['small'] + (['fused'] if smax <= 8 else ['tile'] if smax <= 16 else [])):
        with amd.pinned(OT_FORM=form):
            s2 = amd.ops.l2max_rank_batch(q, c, job_off, max(sizes), k)[0]
        np.testing.assert_allclose(s2.cpu().numpy(), s.cpu().numpy(), atol=4e-5, rtol=0)
    if smax <= 8:
        # <= 64 jobs: the streaming kernel derives the job tables itself (no tables launch); pinned back to the tables launch +
        # table-driven kernel: the same bits
        with amd.pinned(FUSED_NOSELF=1, OT_FORM='fused'):
            s3 = amd.ops.l2max_rank_batch(q, c, job_off, max(sizes), k)[0]
        assert torch.equal(s3, s)


def test_l2max_rank_batch_hybrid(amd):
    """tsAspire batch with a handful of 9 .. 16-row documents among short ones: the fused max-sim kernel scores the short pairs,
    the 16-row kernel only the long ones (census on the device) -- against the 16-row path pinned and torch"""
    sizes = [1500, 1203, 2, 998, 1777, 1501, 3, 700]
    queries, pools = _jobs(123, sizes, 8)
    g = torch.Generator().manual_seed(9)
    for j, i, n in ((0, 5, 16), (4, 1776, 9), (5, 0, 12), (7, 350, 11)):
        pools[j][i] = torch.randn(n, 768, generator=g)
    q = amd.ops.DeviceRepSet.from_list(queries)
    c = amd.ops.DeviceRepSet.from_list([d for p in pools for d in p])
    job_off = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32).cuda()
    hyb = amd.ops.l2max_rank_batch(q, c, job_off, max(sizes), 20)
    with amd.pinned(OT_FORM='tile'):
        ref = amd.ops.l2max_rank_batch(q, c, job_off, max(sizes), 20)
    torch.cuda.synchronize()
    assert torch.isfinite(hyb[0]).all()
    np.testing.assert_allclose(hyb[0].cpu().numpy(), ref[0].cpu().numpy(), atol=4e-5, rtol=0)
    off = np.concatenate([[0], np.cumsum(sizes)])
    for j, i in ((0, 5), (4, 1776), (5, 0), (7, 350), (0, 0), (3, 997)):
        want = -torch.cdist(queries[j], pools[j][i]).min().item()
        assert abs(hyb[0][off[j] + i].item() - want) < 4e-5, (j, i)
    _check_rank([hyb[0][off[j]:off[j + 1]].cpu() for j in range(len(sizes))], hyb[1].cpu(), hyb[2].cpu(), 20)
