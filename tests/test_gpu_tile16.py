"""The streaming cost kernel for documents of 9 .. 16 sentence rows (tile16.hip: two candidates of one query per wave, dot
products on the matrix pipe) against the per-pair tile-loop kernel it replaces at scale, the oracle, and torch.cdist --
single pools (CROSS) and batched jobs (MAPPED)."""
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


def _pool(seed, n, smin, smax):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(smin, smax + 1, (n,), generator=g).tolist()
    return [torch.randn(l, 768, generator=g) for l in lens]


@pytest.mark.parametrize('nq,nc,smax', [(1, 4501, 16), (2, 2300, 12), (1, 4400, 9)])
def test_tile16_single_pool_matches_small_form_and_oracle(amd, nq, nc, smax):
    cands = _pool(500 + nc, nc, 1, smax)
    queries = _pool(13, nq, max(2, smax - 6), smax)
    cands[7] = torch.cat([queries[0][3:5], cands[7][:9]])               # shares two sentences with query 0: direct-formula redo
    cands[nc - 1] = queries[nq - 1].clone()                              # a copy of the last query (odd pool size: lone last item)
    q, c = amd.ops.DeviceRepSet.from_list(queries), amd.ops.DeviceRepSet.from_list(cands)
    with amd.pinned(COST_PATH='valu'):
        new = amd.ops.ot_sinkhorn(q, c).view(nq, nc).cpu().numpy()
    with amd.pinned(COST_PATH='valu', OT_FORM='small'):
        old = amd.ops.ot_sinkhorn(q, c).view(nq, nc).cpu().numpy()
    assert np.isfinite(new).all()
    # duplicate sentences: the solver side carries geomloss's cancellation noise
    noisy = np.zeros_like(new, dtype=bool)
    noisy[0, 7] = noisy[nq - 1, nc - 1] = True
    np.testing.assert_allclose(new[~noisy], old[~noisy], atol=5e-5, rtol=0)
    np.testing.assert_allclose(new[noisy], old[noisy], atol=5e-2, rtol=0)
    idx = [0, 1, 2, nc // 2, nc - 2]
    ref = np.array([[-orc.get_similarity(x, cands[i]) for i in idx] for x in queries], dtype=np.float32)
    np.testing.assert_allclose(new[:, idx], ref, atol=TOL, rtol=0)


def test_tile16_marginals_see_the_direct_formula(amd):
    """pair outputs come from the same workspace slots: -cdist of a shared sentence is exactly 0 (torch.cdist's direct
    formula), the marginals match the oracle"""
    g = torch.Generator().manual_seed(3)
    query = torch.randn(11, 768, generator=g)
    cands = _pool(9, 4200, 3, 14)
    cands[0] = torch.cat([cands[0][:4], query[2:3], cands[0][4:9]])
    qd = {'sent_reps': query.numpy()}
    with amd.pinned(COST_PATH='valu'):
        ret = amd.scorer.caching_score(qd, [{'sent_reps': x.numpy()} for x in cands[:64]])
        q, c = amd.ops.DeviceRepSet.from_list([query]), amd.ops.DeviceRepSet.from_list(cands)
        big = amd.ops.ot_sinkhorn(q, c).cpu().numpy()
    want = np.array([-orc.get_similarity(query, x) for x in cands[:6]], dtype=np.float32)
    np.testing.assert_allclose(big[1:6], want[1:6], atol=TOL, rtol=0)
    np.testing.assert_allclose(big[0], want[0], atol=5e-2, rtol=0)
    assert ret['batch_scores'].shape == (64,)


def test_tile16_batched_jobs(amd):
    """aspire_ot_rank_batch_f32 on jobs of long documents: the mapped form of the kernel (items = halves of the job tables'
    groups of four; job sizes that are not multiples of two or four, single-candidate and empty jobs) against the small form
    and the oracle"""
    g = torch.Generator().manual_seed(21)
    sizes = [1203, 1, 0, 998, 2, 1501, 3, 700]
    queries = [torch.randn(int(torch.randint(5, 15, (1,), generator=g)), 768, generator=g) for _ in sizes]
    pools = [[torch.randn(int(n), 768, generator=g) for n in torch.randint(1, 15, (sz,), generator=g)] for sz in sizes]
    q = amd.ops.DeviceRepSet.from_list(queries)
    c = amd.ops.DeviceRepSet.from_list([d for p in pools for d in p])
    job_off = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32).cuda()
    out = {}
    for form in ('', 'small'):
        with amd.pinned(OT_FORM=form):
            s, ts, ti = amd.ops.ot_rank_batch(q, c, job_off, max(sizes), 20)
            torch.cuda.synchronize()
            out[form] = (s.cpu().numpy(), ts.cpu().numpy(), ti.cpu().numpy())
    np.testing.assert_allclose(out[''][0], out['small'][0], atol=5e-5, rtol=0)
    off = np.concatenate([[0], np.cumsum(sizes)])
    for j in (0, 1, 3, 6):
        pick = list(range(min(4, sizes[j])))
        want = np.array([orc.get_similarity(queries[j], pools[j][i]) for i in pick], dtype=np.float32)
        np.testing.assert_allclose(out[''][0][off[j]:off[j + 1]][pick], want, atol=TOL, rtol=0)
        sc = out[''][0][off[j]:off[j + 1]]
        order = np.argsort(-sc.astype(np.float64), kind='stable')[:20]
        kk = min(20, sizes[j])
        assert out[''][2][j, :kk].tolist() == order.tolist()


def test_tile16_l2max_matches_the_tile_loop_kernel_and_torch(amd):
    """tsAspire for one query against a big pool of 9 .. 16-row documents on the streaming kernel: against l2max_kernel<2>
    (pinned OT_FORM=small), the Gram form, and -min cdist in torch; shared sentences give exactly 0"""
    nc = 4601
    cands = _pool(900, nc, 1, 16)
    query = _pool(17, 1, 13, 13)[0]
    cands[3] = torch.cat([cands[3][:5], query[12:13]])
    cands[nc - 1] = query.clone()
    q, c = amd.ops.DeviceRepSet.from_list([query]), amd.ops.DeviceRepSet.from_list(cands)
    new = amd.ops.l2max_scores(q, c).cpu().numpy()
    with amd.pinned(OT_FORM='small', COST_PATH='valu'):
        old = amd.ops.l2max_scores(q, c).cpu().numpy()
    with amd.pinned(COST_PATH='mfma'):
        gram = amd.ops.l2max_scores(q, c).cpu().numpy()
    np.testing.assert_allclose(new, old, atol=4e-5, rtol=0)
    np.testing.assert_allclose(new, gram, atol=4e-5, rtol=0)
    assert new[3] == 0.0 and new[nc - 1] == 0.0
    idx = [0, 1, 2, 3, nc // 2, nc - 2, nc - 1]
    ref = np.array([-torch.cdist(query, cands[i]).min().item() for i in idx], dtype=np.float32)
    np.testing.assert_allclose(new[idx], ref, atol=4e-5, rtol=0)


def test_tile16_other_entry_modes(amd):
    """the same kernel behind the other ways of calling the path: plan-weighted similarity (want = PLAN_SIM), caller-supplied
    group diameters (schedule='batch': caching_score's one schedule per group of 64 -- the kernel's own box term unused),
    a workspace smaller than the pool's slots (candidate chunks: slots relative to the chunk)"""
    nc = 9001
    cands = _pool(700, nc, 2, 15)
    query = _pool(19, 1, 12, 12)[0]
    q, c = amd.ops.DeviceRepSet.from_list([query]), amd.ops.DeviceRepSet.from_list(cands)
    res = {}
    for form in ('', 'small'):
        with amd.pinned(COST_PATH='valu', OT_FORM=form):
            plan = amd.ops.ot_sinkhorn(q, c, want=amd.lib.OT_PLAN_SIM).cpu().numpy()
            batch = amd.scorer.score_pool([query], cands, method='ot', schedule='batch').cpu().numpy()[0]
            small_ws = torch.empty(4600 * (2 * 256 + 1) * 4 + 2 * 768 * 4 + 64, dtype=torch.uint8, device='cuda')      # two chunks, both big enough for the streaming kernel
            chunked = amd.ops.ot_sinkhorn(q, c, workspace=small_ws).cpu().numpy()
            full = amd.ops.ot_sinkhorn(q, c).cpu().numpy()
        res[form] = (plan, batch, chunked, full)
    np.testing.assert_array_equal(res[''][2], res[''][3])                       # chunks do not change a bit
    np.testing.assert_allclose(res[''][3], res['small'][3], atol=5e-5, rtol=0)
    # plan-weighted similarity: fp32 conditioning of exp((f + g - d) / 0.05)
    np.testing.assert_allclose(res[''][0], res['small'][0], atol=2e-2, rtol=0)
    np.testing.assert_allclose(res[''][1], res['small'][1], atol=2e-2, rtol=0)
    want = np.array(orc.rank_pool_caching(query.numpy(), [x.numpy() for x in cands[:64]]), dtype=np.float32)
    np.testing.assert_allclose(res[''][1][:64], want, atol=2e-2, rtol=0)


@pytest.mark.parametrize('mode', ['CDIST_DIRECT', 'CDIST_MM'])
def test_streaming_kernels_follow_the_cdist_mode(amd, mode):
    """torch.cdist's two formulas as the caller pins them (scorer._cdist_runs does, per padded group): the direct formula
    gives exactly 0 for a shared sentence (the kernels redo cancelled entries), the matmul expansion keeps its cancellation
    noise -- fused kernel (<= 8 rows) and tile16 (9 .. 16 rows) against the small forms, tsAspire"""
    cm = getattr(amd.lib, mode)
    for smax, nc in ((8, 8300), (14, 4400)):
        cands = _pool(40 + smax, nc, 2, smax)
        query = _pool(41, 1, smax, smax)[0]
        cands[2] = torch.cat([query[1:2], cands[2][:smax - 1]])
        q, c = amd.ops.DeviceRepSet.from_list([query]), amd.ops.DeviceRepSet.from_list(cands)
        new = amd.ops.l2max_scores(q, c, cdist_mode=cm).cpu().numpy()
        with amd.pinned(OT_FORM='small', COST_PATH='valu'):
            old = amd.ops.l2max_scores(q, c, cdist_mode=cm).cpu().numpy()
        keep = np.ones(nc, dtype=bool)
        keep[2] = False
        np.testing.assert_allclose(new[keep], old[keep], atol=4e-5, rtol=0)
        if mode == 'CDIST_DIRECT':
            assert new[2] == 0.0 and old[2] == 0.0
        else:
            assert abs(new[2]) < 5e-2 and abs(old[2]) < 5e-2          # sqrt(clamp(cancellation noise)): 0 or ~1e-2 by rounding luck
        ot_new = amd.ops.ot_sinkhorn(q, c, cdist_mode=cm).cpu().numpy()
        with amd.pinned(OT_FORM='small', COST_PATH='valu'):
            ot_old = amd.ops.ot_sinkhorn(q, c, cdist_mode=cm).cpu().numpy()
        np.testing.assert_allclose(ot_new[keep], ot_old[keep], atol=5e-5, rtol=0)


def _batch16(amd, queries, pools, k, **pins):
    q = amd.ops.DeviceRepSet.from_list(queries)
    c = amd.ops.DeviceRepSet.from_list([d for p in pools for d in p])
    sizes = [len(p) for p in pools]
    job_off = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32).cuda()
    with amd.pinned(**pins):
        s, ts, ti = amd.ops.ot_rank_batch(q, c, job_off, max(sizes), k)
        torch.cuda.synchronize()
    return s.cpu().numpy(), ts.cpu().numpy(), ti.cpu().numpy()


@pytest.mark.parametrize('p_long', [0.004, 0.6])
def test_batch_hybrid_short_pools_with_a_few_long_documents(amd, p_long):
    """batches whose documents reach 9 .. 16 rows: a census on the device picks the kernel family.  Few long pairs (0.4 %):
    the fused kernel scores the short pairs, the long-form kernel exactly the poisoned ones -- every score against the
    16-row streaming path (pinned OT_FORM=tile) and the oracle.  Many long pairs (60 %): the census sends the whole batch
    to the streaming path -- bit for bit the pinned result (the short-document kernels returned at once, wrote nothing)."""
    g = torch.Generator().manual_seed(int(p_long * 1000) + 5)
    sizes = [1500, 1203, 2, 998, 1777, 1501, 3, 700]
    def doc():
        long = torch.rand(1, generator=g).item() < p_long
        n = int(torch.randint(9, 17, (1,), generator=g)) if long else int(torch.randint(1, 9, (1,), generator=g))
        return torch.randn(n, 768, generator=g)
    queries = [torch.randn(int(torch.randint(3, 9, (1,), generator=g)), 768, generator=g) for _ in sizes]
    if p_long > 0.5:
        queries[3] = torch.randn(13, 768, generator=g)           # a long query: every pair of its job is long
    pools = [[doc() for _ in range(sz)] for sz in sizes]
    pools[0][5] = torch.randn(16, 768, generator=g)              # at least one long document in any case
    pools[4][1776] = torch.randn(9, 768, generator=g)
    hyb = _batch16(amd, queries, pools, 25)
    ref = _batch16(amd, queries, pools, 25, OT_FORM='tile')
    assert np.isfinite(hyb[0]).all()
    if p_long > 0.5:
        assert np.array_equal(hyb[0], ref[0]) and np.array_equal(hyb[2], ref[2])
    else:
        np.testing.assert_allclose(hyb[0], ref[0], atol=5e-5, rtol=0)
    off = np.concatenate([[0], np.cumsum(sizes)])
    for j, i in ((0, 5), (4, 1776), (0, 0), (3, 10), (7, 699)):
        want = orc.get_similarity(queries[j], pools[j][i])
        assert abs(hyb[0][off[j] + i] - want) < TOL, (j, i)
    for j in range(len(sizes)):                                   # every job's list = the stable sort of its own scores
        sc = hyb[0][off[j]:off[j + 1]]
        kk = min(25, sizes[j])
        assert hyb[2][j, :kk].tolist() == np.argsort(-sc.astype(np.float64), kind='stable')[:kk].tolist()


@pytest.mark.parametrize('p_long', [0.003, 0.5])
def test_single_pool_hybrid(amd, p_long):
    """one short query against a big pool with a share of 9 .. 16-row documents (aspire_ot_sinkhorn_f32 / aspire_ot_rank_f32):
    the fused kernel's CHUNK form (round 3; before: a device-side hybrid of the 8-row fused kernel and the 16-row kernels),
    against the 16-row streaming path pinned and the oracle"""
    g = torch.Generator().manual_seed(77)
    nc = 9100
    lens = torch.where(torch.rand(nc, generator=g) < p_long, torch.randint(9, 17, (nc,), generator=g), torch.randint(1, 9, (nc,), generator=g))
    lens[11] = 15
    cands = [torch.randn(int(n), 768, generator=g) for n in lens]
    query = torch.randn(7, 768, generator=g)
    q, c = amd.ops.DeviceRepSet.from_list([query]), amd.ops.DeviceRepSet.from_list(cands)
    hyb = amd.ops.ot_sinkhorn(q, c).cpu().numpy()
    with amd.pinned(OT_FORM='tile'):
        ref = amd.ops.ot_sinkhorn(q, c).cpu().numpy()
    assert np.isfinite(hyb).all()
    np.testing.assert_allclose(hyb, ref, atol=5e-5, rtol=0)
    idx = [0, 11, 12, nc - 1] + [int(i) for i in np.nonzero(lens.numpy() > 8)[0][:3]]
    want = np.array([-orc.get_similarity(query, cands[i]) for i in idx], dtype=np.float32)
    np.testing.assert_allclose(hyb[idx], want, atol=TOL, rtol=0)
    sc, ts, ti = amd.ops.ot_rank(q, c, 50, want=amd.lib.OT_SIMILARITY)
    order = np.argsort(-sc.cpu().numpy()[0].astype(np.float64), kind='stable')[:50]
    assert ti.cpu().numpy()[0].tolist() == order.tolist()
