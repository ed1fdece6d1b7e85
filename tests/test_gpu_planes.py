"""The many-query cost tiles on pre-split fp16 planes (aspire_amd/csrc/gramp.hip; include/aspire_hip.h: aspire_rep_planes)
against the oracle, against float64, and against the forms that read the fp32 rows.  Reference arithmetic:
src/learning/facetid_models/pair_distances.py:48-55, 138-186.  Every call goes through the C ABI."""
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
    return type('NS', (), dict(ops=ops, scorer=scorer, lib=_lib))


def _set(amd, docs):
    return amd.ops.DeviceRepSet.from_list(docs)


def _docs(seed, lens, scale=1.0, shift=0.0):
    g = torch.Generator().manual_seed(seed)
    return [scale * torch.randn(int(n), 768, generator=g) + shift for n in lens]


def _l2max_oracle(q, c):
    return -orc.allpair_masked_dist_l2max(orc.RepLen(q[None].permute(0, 2, 1), [len(q)]),
                                          orc.RepLen(c[None].permute(0, 2, 1), [len(c)])).item()


def _with_planes(amd, qdocs, cdocs):
    c = _set(amd, cdocs).prepare_planes()
    q = _set(amd, qdocs).prepare_planes(like=c)
    return q, c


def test_planes_blob_is_the_rows(amd):
    """h + l planes, per-row scale, norms and the store's mean as the header describes them"""
    g = torch.Generator().manual_seed(3)
    rows = (torch.randn(37, 768, generator=g) * torch.logspace(-3, 2, 37)[:, None] + 0.5).cuda()
    rows[5] = 0
    rp = amd.ops.RowPlanes(rows)
    pr, n = rp.c.plane_rows, 37
    assert pr % 16 == 0 and pr > n and rp.c.total_rows == n
    mu = rp.mu.cpu()
    np.testing.assert_allclose(mu.numpy(), rows.cpu().mean(0).numpy(), atol=1e-5)       # fewer than 4096 rows: all of them
    blob = rp.blob.cpu()
    nrm = blob[4096:4096 + 4 * pr].view(torch.float32)
    isc = blob[4096 + 4 * pr:4096 + 8 * pr].view(torch.float32)
    planes = blob[4096 + 8 * pr:].view(torch.float16).view(48, pr, 2, 2, 8)           # [kb][row][plane][k half][8]
    v = rows.cpu() - mu
    np.testing.assert_allclose(nrm[:n].numpy(), (v.double() ** 2).sum(1).numpy(), rtol=1e-5)
    assert (nrm[n:] == 0).all() and (isc[n:] == 0).all() and (planes[:, n:] == 0).all()
    rec = (planes[:, :, 0].double() + planes[:, :, 1].double()).permute(1, 0, 2, 3).reshape(pr, 768)[:n] * isc[:n, None].double()
    err = (rec - v.double()).abs().max(1).values / v.abs().max(1).values.clamp(min=1e-30).double()
    assert err.max() < 2.0 ** -21, err.max()
    s = 1.0 / isc[:n].double()
    top = (v.double().abs().max(1).values * s)
    live = v.abs().max(1).values > 0
    assert ((top[live] >= 2 ** 14) & (top[live] < 2 ** 15)).all()
    assert (np.log2(s.numpy()) % 1 == 0).all()


@pytest.mark.parametrize('qlens,clens', [
    ([8] * 12, [8] * 40),                                     # 96 query rows: one 128-column tile
    ([5, 8, 1, 7, 3] * 4, [8, 3, 1, 6, 7, 2, 8, 5] * 5),      # ragged
    ([12] * 11, [12] * 23),                                   # 12-row slots, 10 per tile, two query tiles
    ([9, 16, 13] * 3, [11, 1, 16, 4] * 6),                    # T = 2 ragged
    ([20, 3, 24, 17], [17, 24, 2, 9] * 3),                    # T = 3
    ([32, 30, 27], [32, 1, 30, 26, 25] * 2),                  # T = 4, crosses the cdist 25/26 switch
])
def test_planes_ot_and_l2max_match_oracle(amd, qlens, clens):
    from aspire_amd._lib import pinned
    qd, cd = _docs(11, qlens), _docs(12, clens)
    q, c = _with_planes(amd, qd, cd)
    with pinned(COST_PATH='mfma'):
        l2 = amd.ops.l2max_scores(q, c).view(len(qd), len(cd)).cpu().numpy()
        ot = -amd.ops.ot_sinkhorn(q, c).view(len(qd), len(cd)).cpu().numpy()
    want_l2 = np.array([[_l2max_oracle(x, y) for y in cd] for x in qd], dtype=np.float32)
    np.testing.assert_allclose(l2, want_l2, atol=TOL, rtol=0)
    # the OT oracle is ~40 ms per pair: every pair of the small cases, 160 sampled pairs (every query and every candidate among them)
    # of the big ones
    pairs = [(i, j) for i in range(len(qd)) for j in range(len(cd))]
    if len(pairs) > 160:
        rs = np.random.RandomState(7)
        keep = {(i, int(rs.randint(len(cd)))) for i in range(len(qd))} | {(int(rs.randint(len(qd))), j) for j in range(len(cd))}
        rest = [p for p in pairs if p not in keep]
        keep |= {rest[k] for k in rs.choice(len(rest), size=160 - len(keep), replace=False)}
        pairs = sorted(keep)
    for i, j in pairs:
        assert ot[i, j] == pytest.approx(orc.get_similarity(qd[i], cd[j]), abs=TOL), (i, j)


def test_planes_near_duplicates_and_scales(amd):
    """a candidate that repeats a query sentence exactly / nearly (direct-formula work list from the fp32 rows); rows of very
    different magnitudes (per-row scales); an all-zero row"""
    from aspire_amd._lib import pinned
    g = torch.Generator().manual_seed(5)
    qd = _docs(31, [8] * 10)
    cd = _docs(32, [8] * 24)
    cd[3][2] = qd[1][5]
    cd[7][0] = qd[2][0] + 1e-3 * torch.randn(768, generator=g)
    cd[9][7] = qd[0][1] + 1e-2 * torch.randn(768, generator=g)
    cd[11] = cd[11] * 1e-3
    cd[12] = cd[12] * 300.0
    cd[13][4] = 0
    qd[4] = qd[4] * 50.0
    q, c = _with_planes(amd, qd, cd)
    with pinned(COST_PATH='mfma'):
        l2 = amd.ops.l2max_scores(q, c).view(len(qd), len(cd)).cpu().numpy()
    want = np.array([[-torch.cdist(x.double(), y.double()).min().item() for y in cd] for x in qd])
    np.testing.assert_allclose(l2, want, atol=0, rtol=3e-6)
    assert l2[1, 3] == 0.0


@pytest.mark.parametrize('shift', [0.0, 2.0])
def test_planes_agree_with_fp32_row_forms_at_size(amd, shift):
    """bench-sized grid with tail tiles, i.i.d. rows and rows with a large common component (mean cosine 0.8): the plane
    tiles against the bf16x3 tiles and against float64 on a sample"""
    from aspire_amd._lib import pinned
    nq, nc, s = 32, 4001, 8
    g = torch.Generator().manual_seed(77)
    qrows = (torch.randn(nq * s, 768, generator=g) * torch.linspace(0.3, 2.0, 768) + shift).cuda()
    crows = (torch.randn(nc * s, 768, generator=g) + shift).cuda()
    mk = lambda rows, n: amd.ops.DeviceRepSet(rows, (torch.arange(n, device='cuda', dtype=torch.int32) * s).contiguous(),
                                              torch.full((n,), s, device='cuda', dtype=torch.int32), ext=0, max_len=s)
    q, c = mk(qrows, nq), mk(crows, nc)
    c.prepare_planes()
    q.prepare_planes(like=c)
    out = {}
    for form in ('', 'bf16x3'):
        with pinned(COST_PATH='mfma', GEMM=form):
            out[form] = (amd.ops.l2max_scores(q, c).view(nq, nc).cpu().numpy(), amd.ops.ot_sinkhorn(q, c).view(nq, nc).cpu().numpy())
    assert not np.array_equal(out[''][0], out['bf16x3'][0])          # two different kernels ran
    # both against float64 (the bf16x3 tiles centre on the tile's first query row, the planes on the store's mean: on rows with a
    # common component the planes are the closer of the two -- tools/planeerr.py)
    d = torch.cdist(qrows.double(), crows.double())
    want = -d.view(nq, s, nc, s).permute(0, 2, 1, 3).reshape(nq, nc, s * s).min(-1).values.cpu().numpy()
    assert np.abs(out[''][0] - want).max() < 1e-5
    assert np.abs(out['bf16x3'][0] - want).max() < 6e-5
    np.testing.assert_allclose(out[''][1], out['bf16x3'][1], atol=5e-5, rtol=0)


def test_without_planes_or_with_another_centre_the_fp32_forms_run(amd):
    """the fallback: a rep set without planes, or one prepared around another vector, takes the kernels that read the fp32
    rows -- same scores to rounding, never an error"""
    from aspire_amd._lib import pinned
    qd, cd = _docs(41, [8] * 16), _docs(42, [8] * 2100)          # 131 candidate tiles: a pool the plane tiles are used on
    q, c = _with_planes(amd, qd, cd)
    with pinned(COST_PATH='mfma'):
        a = amd.ops.l2max_scores(q, c).cpu().numpy()
        q2 = _set(amd, qd)                                   # no planes on the query side: ops prepares them around the pool's centre
        a2 = amd.ops.l2max_scores(q2, c).cpu().numpy()
        assert q2.planes is not None and q2.planes.c.mu == c.planes.c.mu and np.array_equal(a, a2)
        q3 = _set(amd, qd).prepare_planes()                  # its own centre: left alone
        b = amd.ops.l2max_scores(q3, c).cpu().numpy()
        with pinned(GEMM='bf16x3'):
            b4 = amd.ops.l2max_scores(q, c).cpu().numpy()
        b5 = amd.ops.l2max_scores(_set(amd, qd), _set(amd, cd)).cpu().numpy()      # no planes anywhere
    assert np.array_equal(b, b4) and np.array_equal(b, b5)
    assert not np.array_equal(a, b)
    np.testing.assert_allclose(a, b, atol=2e-5, rtol=0)


def test_pools_as_index_lists_share_the_store_planes(amd):
    """a pool = an index list into the resident matrix (RepStore.pool): the tiles gather its rows from the store's planes"""
    from aspire_amd._lib import pinned
    docs = _docs(51, np.random.RandomState(1).randint(1, 9, size=400))
    store = _set(amd, docs).prepare_planes()
    pick = np.random.RandomState(2).permutation(400)[:300]
    idx = torch.as_tensor(pick, device='cuda')
    pool = amd.ops.DeviceRepSet(store.rows, store.start[idx].contiguous(), store.len[idx].contiguous(), ext=0, max_len=8)
    qsel = torch.as_tensor(np.arange(20), device='cuda')
    q = amd.ops.DeviceRepSet(store.rows, store.start[qsel].contiguous(), store.len[qsel].contiguous(), ext=0, max_len=8)
    assert pool.planes is store.planes and q.planes is store.planes
    with pinned(COST_PATH='mfma'):
        got = amd.ops.l2max_scores(q, pool).view(20, 300).cpu().numpy()
    want = np.array([[_l2max_oracle(docs[i], docs[j]) for j in pick] for i in range(20)], dtype=np.float32)
    np.testing.assert_allclose(got, want, atol=TOL, rtol=0)


@pytest.mark.parametrize('tile,pp', [('128256', ''), ('256128', ''), ('256256', ''), ('256256', '1')])
def test_wider_tile_forms_give_the_same_scores(amd, tile, pp):
    """GRAM_TILE pins the 128 x 256 / 256 x 128 / 256 x 256 tiles (wider wave tiles, one or two workgroups per CU), GRAM_PP the ping-pong
    schedule of the 256 x 256 form -- kept for A/B runs, none beats the 128 x 128 default (NOTES.md, round 4): the same
    products in the same order per entry, so the same bits; ragged documents and tail tiles included"""
    from aspire_amd._lib import pinned
    qd = _docs(61, np.random.RandomState(3).randint(1, 13, size=50))
    cd = _docs(62, np.random.RandomState(4).randint(1, 13, size=700))
    q, c = _with_planes(amd, qd, cd)
    with pinned(COST_PATH='mfma'):
        a = amd.ops.l2max_scores(q, c).cpu().numpy()
        ot_a = amd.ops.ot_sinkhorn(q, c).cpu().numpy()
        with pinned(GRAM_TILE=tile, GRAM_PP=pp):
            b = amd.ops.l2max_scores(q, c).cpu().numpy()
            ot_b = amd.ops.ot_sinkhorn(q, c).cpu().numpy()
    assert np.array_equal(a, b)
    assert np.array_equal(ot_a, ot_b)


@pytest.mark.parametrize('nq,s', [(1, 12), (4, 8), (2, 20), (1, 32)])
def test_few_queries_on_a_big_plane_pool_take_the_plane_tiles(amd, nq, s):
    """max-sim with both sides on planes runs on the matrix-pipe tiles whatever the number of queries (most query columns of a
    tile are then zero rows): ragged documents, against float64 and against the streaming kernels that read the fp32 rows"""
    g = torch.Generator().manual_seed(100 * nq + s)
    nc = 128 * 128 // (((s + 3) // 4) * 4) + 37
    cd = [torch.randn(int(n), 768, generator=g) for n in torch.randint(1, s + 1, (nc,), generator=g)]
    qd = [torch.randn(int(n), 768, generator=g) for n in torch.randint(max(1, s - 3), s + 1, (nq,), generator=g)]
    c = _set(amd, cd).prepare_planes()
    q = _set(amd, qd)
    got = amd.ops.l2max_scores(q, c).view(nq, nc).cpu().numpy()
    assert q.planes is not None                                   # ops gave the queries the pool's centre
    ref = amd.ops.l2max_scores(_set(amd, qd), _set(amd, cd)).view(nq, nc).cpu().numpy()
    assert not np.array_equal(got, ref)                           # another kernel ran
    np.testing.assert_allclose(got, ref, atol=5e-5, rtol=0)
    for ci in (0, 17, nc - 1):
        want = -torch.cdist(qd[0].double(), cd[ci].double()).min().item()
        assert abs(got[0, ci] - want) < 2e-5


def test_cached_document_boxes_give_the_same_ot_scores(amd):
    """aspire_repset.doc_box: a resident pool's per-document boxes, formed once (CandidatePool.prepare_planes does) instead of per
    call -- the same values, so the many-query otAspire scores are the same bits; the boxes themselves against torch"""
    from aspire_amd._lib import pinned
    g = torch.Generator().manual_seed(21)
    cd = [torch.randn(int(n), 768, generator=g) for n in torch.randint(1, 13, (2000,), generator=g)]
    qd = [torch.randn(int(n), 768, generator=g) for n in torch.randint(1, 13, (30,), generator=g)]
    c = _set(amd, cd).prepare_planes()
    q = _set(amd, qd)
    with pinned(COST_PATH='mfma'):
        a = amd.ops.ot_sinkhorn(q, c).cpu()
        c.prepare_boxes()
        box = c.doc_box.cpu()
        b = amd.ops.ot_sinkhorn(q, c).cpu()
        tail = amd.ops.ot_sinkhorn(q, c.slice(1500, 2000)).cpu()
    for k in (0, 7, 1999):
        assert torch.equal(box[k, 0], cd[k].min(0).values) and torch.equal(box[k, 1], cd[k].max(0).values)
    assert torch.equal(a, b)
    # a slice of the pool takes its slice of the boxes (another grid size: another Sinkhorn layout, so to rounding)
    np.testing.assert_allclose(tail.view(30, 500).numpy(), b.view(30, 2000)[:, 1500:].numpy(), atol=5e-5, rtol=0)
    want = np.array([[orc.get_similarity(qd[i], cd[j]) for j in (0, 1999)] for i in (0, 29)], dtype=np.float32)
    np.testing.assert_allclose(-b.view(30, 2000)[[0, 29]][:, [0, 1999]].numpy(), want, atol=TOL, rtol=0)


def test_batch_schedule_on_a_plane_pool(amd):
    """caching_score's grouping (one epsilon schedule per 64 candidates, plan-weighted similarity: pp_gen_nearest.py:182-202) with
    the costs from the plane tiles: against the float64 oracle, no further from it than the reference's own fp32 path
    (tests/plan_sim_floor.py), and the ranking step on top of it"""
    import plan_sim_floor
    from aspire_amd._lib import pinned
    g = torch.Generator().manual_seed(31)
    cd = [torch.randn(int(n), 768, generator=g) for n in torch.randint(1, 9, (2200,), generator=g)]
    qd = [torch.randn(int(n), 768, generator=g) for n in (8, 6, 8, 7, 5, 8, 8, 3, 8)]
    pool = amd.scorer.CandidatePool(cd).prepare_planes()
    with pinned(COST_PATH='mfma'):
        got = amd.scorer.score_pool(qd, pool, method='ot', schedule='batch').cpu().numpy()
        ranked = amd.scorer.rank_pool(qd[:2], pool, k=30, method='ot', schedule='batch')
    for qi in (0, 7):
        sub = cd[:192]
        want = np.array(orc.rank_pool_caching(qd[qi].numpy(), [c.numpy() for c in sub]), dtype=np.float32)
        truth = np.array(orc.rank_pool_caching(qd[qi].numpy(), [c.numpy() for c in sub], dtype=torch.float64))
        plan_sim_floor.check(got[qi, :192], want, truth, 'plane tiles, batch schedule')
    # the ranking step (a 2-query call: another kernel family forms its costs, so the scores agree to the plan-similarity's
    # rounding and near-ties may swap): descending, and every listed candidate's score is the 9-query call's to that rounding
    sc = [v for _, v in ranked[0]]
    assert sc == sorted(sc, reverse=True) and len(sc) == 30
    for i, v in ranked[0]:
        assert abs(v - got[0, i]) < 5e-3
    assert ranked[0][0][0] == int(np.argmax(got[0]))


@pytest.mark.parametrize('nq,s', [(1, 12), (3, 8), (8, 8), (2, 20)])
def test_few_query_otaspire_on_a_plane_pool_with_cached_boxes(amd, nq, s):
    """otAspire's cost stage on the plane tiles with FEW queries (gram_planes_wanted_ot): the pool carries planes and its
    documents' boxes, the pairs' diameters come from pair_box_few_kernel -- against the oracle, ragged documents"""
    g = torch.Generator().manual_seed(300 + 10 * nq + s)
    nc = 128 * 128 // (((s + 3) // 4) * 4) + 21
    cd = [torch.randn(int(n), 768, generator=g) for n in torch.randint(1, s + 1, (nc,), generator=g)]
    qd = [torch.randn(int(n), 768, generator=g) for n in torch.randint(max(1, s - 3), s + 1, (nq,), generator=g)]
    if nq == 1:
        qd[0] = torch.randn(s, 768, generator=g)            # (one query of <= 8 rows would stay on the fused kernel)
    pool = amd.scorer.CandidatePool(cd).prepare_planes()
    assert pool.repset.doc_box is not None
    got = amd.scorer.score_pool(qd, pool, method='ot', schedule='pair').cpu().numpy()
    ref = amd.scorer.score_pool(qd, cd, method='ot', schedule='pair').cpu().numpy()       # no planes: the default kernels
    assert not np.array_equal(got, ref)
    np.testing.assert_allclose(got, ref, atol=1e-4, rtol=0)
    for qi in (0, nq - 1):
        for ci in (0, 5, nc - 1):
            assert abs(got[qi, ci] - orc.get_similarity(qd[qi], cd[ci])) < 1e-4
    ranked = amd.scorer.rank_pool(qd, pool, k=10, method='ot')
    assert [i for i, _ in ranked[0]] == np.argsort(-got[0].astype(np.float64), kind='stable')[:10].tolist()


def test_planes_and_boxes_follow_the_rows_when_a_store_is_refilled_in_place(amd):
    """the fp16 planes and the cached boxes are a CACHE of the row matrix: refill the rows in place -- through torch, or
    through the C ABI (span_mean_pool_rows reports its write) -- and the next call scores the NEW rows; queries whose planes ops
    prepared for another pool's centre get new ones"""
    from aspire_amd._lib import pinned
    g = torch.Generator().manual_seed(77)
    cd = [torch.randn(8, 768, generator=g) for _ in range(2100)]
    qd = [torch.randn(8, 768, generator=g) for _ in range(6)]
    pool = amd.scorer.CandidatePool(cd).prepare_planes()
    q = _set(amd, qd)
    with pinned(COST_PATH='mfma'):
        a = amd.ops.l2max_scores(q, pool.repset).view(6, -1).cpu()
        ot_a = amd.ops.ot_sinkhorn(q, pool.repset).view(6, -1).cpu()
        old_planes = pool.repset.planes
        # the store refilled in place (a torch write): planes and boxes must not describe the old rows
        fresh = torch.randn(2100 * 8, 768, generator=g)
        pool.repset.rows.copy_(fresh.cuda())
        b = amd.ops.l2max_scores(q, pool.repset).view(6, -1).cpu()
        ot_b = amd.ops.ot_sinkhorn(q, pool.repset).view(6, -1).cpu()
    assert pool.repset.planes is not old_planes
    want = -torch.cdist(torch.cat(qd).double(), fresh.double()).view(6, 8, 2100, 8).permute(0, 2, 1, 3).reshape(6, 2100, 64).min(-1).values
    assert (b.double() - want).abs().max().item() < 1e-5
    assert (a - b).abs().max().item() > 1.0                               # other rows, other scores
    for j in (0, 1000, 2099):
        assert -ot_b[0, j].item() == pytest.approx(orc.get_similarity(qd[0], fresh[8 * j:8 * j + 8]), abs=TOL)
    assert not torch.equal(ot_a, ot_b)
    # the query rows rewritten in place: their auto-prepared planes are stale too
    q.rows.mul_(0.5)
    with pinned(COST_PATH='mfma'):
        c2 = amd.ops.l2max_scores(q, pool.repset).view(6, -1).cpu()
    want2 = -torch.cdist(0.5 * torch.cat(qd).double(), fresh.double()).view(6, 8, 2100, 8).permute(0, 2, 1, 3).reshape(6, 2100, 64).min(-1).values
    assert (c2.double() - want2).abs().max().item() < 1e-5
    # the same queries against ANOTHER plane pool (another centre): planes made for the first pool are not reused
    other = amd.scorer.CandidatePool([torch.randn(8, 768, generator=g) + 2.0 for _ in range(2100)]).prepare_planes()
    with pinned(COST_PATH='mfma'):
        d = amd.ops.l2max_scores(q, other.repset).view(6, -1)
    assert q.planes.mu.data_ptr() == other.repset.planes.mu.data_ptr()
    assert torch.isfinite(d).all()
