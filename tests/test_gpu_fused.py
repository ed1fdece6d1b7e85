"""The fused otAspire kernel (fused.hip: costs + Sinkhorn solves of four pairs per wave in one launch) against the oracle
and against the two-kernel forms, for single big pools (CROSS) and batched jobs; its max-sim (tsAspire) form."""
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


def _pool(seed, n, smin=1, smax=8):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(smin, smax + 1, (n,), generator=g).tolist()
    return [torch.randn(l, 768, generator=g) for l in lens]


@pytest.mark.parametrize('nq,nc', [(1, 8203), (2, 4101)])
@pytest.mark.parametrize('want', ['distance', 'similarity', 'plan'])
def test_fused_single_pool_matches_oracle_and_tile_form(amd, nq, nc, want):
    cands = _pool(100 + nc, nc)
    queries = _pool(7, nq, 3, 8)
    q, c = amd.ops.DeviceRepSet.from_list(queries), amd.ops.DeviceRepSet.from_list(cands)
    w = dict(distance=amd.lib.OT_DISTANCE, similarity=amd.lib.OT_SIMILARITY, plan=amd.lib.OT_PLAN_SIM)[want]
    dflt = amd.ops.ot_sinkhorn(q, c, want=w).view(nq, nc).cpu().numpy()
    with amd.pinned(OT_FORM='fused'):
        fused = amd.ops.ot_sinkhorn(q, c, want=w).view(nq, nc).cpu().numpy()
    with amd.pinned(OT_FORM='tile'):
        tile = amd.ops.ot_sinkhorn(q, c, want=w).view(nq, nc).cpu().numpy()
    assert np.array_equal(dflt, fused) and np.isfinite(fused).all()
    if nq == 1:      # one query: the kernel forms the query's box itself; with the box launch in front (pinned) the same bits
        with amd.pinned(OT_FORM='fused', FUSED_NOSELF=1):
            boxed = amd.ops.ot_sinkhorn(q, c, want=w).view(nq, nc).cpu().numpy()
        assert np.array_equal(boxed, fused)
    # plan-weighted similarity: exp((f + g - d) / 0.05) with |f|, |g|, |d| ~ 38 -- the reference's own fp32 value is off by up
    # to 1.6e-2 from a float64 evaluation
    np.testing.assert_allclose(fused, tile, atol=5e-5 if want != 'plan' else 2e-2, rtol=0)
    idx = [0, 1, 2, 3, nc // 2, nc - 3, nc - 2, nc - 1]
    if want == 'plan':
        ref = np.array([[orc.AllPairMaskedWasserstein({}).compute_distance(
            orc.RepLen(x[None].permute(0, 2, 1), [len(x)]), orc.RepLen(cands[i][None].permute(0, 2, 1), [len(cands[i])]),
            return_pair_sims=True)[0].item() for i in idx] for x in queries], dtype=np.float32)
        np.testing.assert_allclose(fused[:, idx], ref, atol=2e-2, rtol=0)       # fp32 conditioning of exp((f + g - d) / 0.05)
    else:
        sign = 1.0 if want == 'similarity' else -1.0
        ref = np.array([[orc.get_similarity(x, cands[i]) for i in idx] for x in queries], dtype=np.float32)
        np.testing.assert_allclose(sign * fused[:, idx], ref, atol=TOL, rtol=0)


@pytest.mark.parametrize('hp', [dict(geoml_scaling=0.5), dict(geoml_blur=0.5, geoml_scaling=0.99), dict(sent_sm_temp=5000.0),
                                dict(geoml_scaling=0.01), dict(geoml_blur=1e-3)])
def test_fused_hparams_and_exact_fallback(amd, hp):
    """scaling = 0.01 overflows the shifted sums: those pairs are solved again in the kernel with max-shifted log-sum-exps"""
    cands = _pool(5, 40)
    query = _pool(6, 1, 8, 8)
    with amd.pinned(OT_FORM='fused'):
        got = amd.scorer.score_pool(query, cands, method='ot', schedule='pair', hparams=hp).cpu().numpy()[0]
    want = np.array([orc.get_similarity(query[0], y, hp) for y in cands], dtype=np.float32)
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want, atol=TOL, rtol=0)


def test_fused_duplicate_sentences_and_group_schedule(amd):
    """candidates sharing sentences with the query (the direct-formula redo path) and caching_score's one schedule per
    group of 64 (caller-supplied diameters): fused = two-kernel form to a few 1e-5"""
    g = torch.Generator().manual_seed(9)
    query = torch.randn(8, 768, generator=g)
    cands = _pool(10, 300)
    for i in range(0, 300, 7):
        cands[i][0] = query[i % 8]
    res = {}
    for form in ('fused', 'tile', 'small'):
        with amd.pinned(OT_FORM=form):
            res[form] = (amd.scorer.score_pool([query], cands, method='ot', schedule='pair').cpu().numpy()[0],
                         amd.scorer.score_pool([query], cands, method='ot', schedule='batch').cpu().numpy()[0])
    for k in range(2):
        # a shared sentence: the expansion formula of geomloss's cost cancels there, 1e-4 .. 3e-2 depending on the summation
        # order (test_gpu_scoring.test_duplicate_sentence_pair); every other candidate agrees closely
        np.testing.assert_allclose(res['fused'][k], res['small'][k], atol=5e-2, rtol=0)
        clean = np.array([i % 7 != 0 for i in range(300)])
        np.testing.assert_allclose(res['fused'][k][clean], res['tile'][k][clean], atol=1e-2 if k else 5e-5, rtol=0)
        np.testing.assert_allclose(res['fused'][k][clean], res['small'][k][clean], atol=1e-2 if k else 5e-5, rtol=0)
    want = orc.caching_score(query.numpy(), [c.numpy() for c in cands[64:128]])[0]
    np.testing.assert_allclose(res['fused'][1][64:128][clean[64:128]], want[clean[64:128]], atol=1e-2, rtol=0)


@pytest.mark.parametrize('form', ['fused'])
def test_fused_schedule_length_at_its_discontinuities(amd, form):
    """as test_gpu_edges.test_schedule_length_at_its_discontinuities, for the fused kernel's copy of the schedule"""
    blur, scaling = 0.05, 0.9
    g = torch.Generator().manual_seed(77)
    q = torch.randn(6, 768, generator=g)
    c = torch.randn(7, 768, generator=g)
    diams = []
    for k in (60, 68, 72, 80):
        d0 = blur * scaling ** (-k)
        for rel in (0.0, 1e-7, -1e-7, 3e-7, -3e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-3, -1e-3):
            diams.append(np.float32(d0 * (1.0 + rel)))
    diams = np.array(diams, dtype=np.float32)
    n = len(diams)
    qs = amd.ops.DeviceRepSet.from_list([q])
    cs = amd.ops.DeviceRepSet.from_list([c] * n)
    with amd.pinned(OT_FORM=form):
        got = amd.ops.ot_sinkhorn(qs, cs, blur=blur, scaling=scaling, diameter=torch.from_numpy(diams).cuda(), diam_group=1).cpu().numpy()
    w = orc.AllPairMaskedWasserstein({'geoml_blur': blur, 'geoml_scaling': scaling})
    qt = orc.RepLen(q[None].permute(0, 2, 1), [6])
    ct = orc.RepLen(c[None].permute(0, 2, 1), [7])
    want = np.array([w.compute_distance(qt, ct, diameter=float(d)).item() for d in diams], dtype=np.float32)
    np.testing.assert_allclose(got, want, atol=1e-4, rtol=0)


def test_fused_matrix_pipe_and_valu_forms_agree(amd):
    """the dot products on v_mfma_f32_4x4x1 (default) and as VALU FMAs: exact fp32 multiply-adds either way, another
    summation order"""
    cands = _pool(31, 8300)
    queries = _pool(32, 2, 2, 8)
    q, c = amd.ops.DeviceRepSet.from_list(queries), amd.ops.DeviceRepSet.from_list(cands)
    with amd.pinned(OT_FORM='fused'):
        mfma = amd.ops.ot_sinkhorn(q, c).cpu().numpy()
    with amd.pinned(OT_FORM='fused', FUSED_VALU=1):
        valu = amd.ops.ot_sinkhorn(q, c).cpu().numpy()
    assert np.isfinite(mfma).all()
    np.testing.assert_allclose(mfma, valu, atol=5e-5, rtol=0)


def test_fused_repeated_launches_on_one_workspace(amd):
    """back-to-back launches on one workspace score every pair"""
    cands = _pool(21, 9000, 8, 8)
    query = _pool(22, 1, 8, 8)
    q, c = amd.ops.DeviceRepSet.from_list(query), amd.ops.DeviceRepSet.from_list(cands)
    import ctypes
    qs, cs = q.struct(), c.struct()
    ws = torch.empty(amd.lib.lib.aspire_ot_workspace_bytes(ctypes.byref(qs), ctypes.byref(cs), 0), dtype=torch.uint8, device='cuda')
    first = amd.ops.ot_sinkhorn(q, c, workspace=ws).clone()
    for _ in range(5):
        out = torch.full((9000,), float('nan'), device='cuda')
        amd.ops.ot_sinkhorn(q, c, out=out, workspace=ws)
        assert torch.equal(out, first)


@pytest.mark.parametrize('nq,nc', [(1, 8203), (3, 2735)])
def test_fused_l2max_matches_the_per_candidate_kernel_and_torch(amd, nq, nc):
    """tsAspire on the fused kernel's streaming phase (few queries x a big CSR pool of short documents): against
    l2max_kernel<1> (pinned: OT_FORM=small) and against -min cdist of the valid block in torch; candidates that repeat a query
    sentence give exactly 0 (torch.cdist's direct formula), ragged documents, a tail group of fewer than four candidates"""
    cands = _pool(300 + nc, nc)
    queries = _pool(11, nq, 2, 8)
    cands[5] = torch.cat([queries[0][1:2], cands[5][:3]])                 # shares a sentence with query 0
    cands[nc - 1] = queries[nq - 1].clone()                              # a copy of the last query
    q, c = amd.ops.DeviceRepSet.from_list(queries), amd.ops.DeviceRepSet.from_list(cands)
    dflt = amd.ops.l2max_scores(q, c).view(nq, nc).cpu().numpy()
    with amd.pinned(OT_FORM='fused'):
        fused = amd.ops.l2max_scores(q, c).view(nq, nc).cpu().numpy()
    with amd.pinned(OT_FORM='small'):
        small = amd.ops.l2max_scores(q, c).view(nq, nc).cpu().numpy()
    assert np.array_equal(dflt, fused) and np.isfinite(fused).all()
    np.testing.assert_allclose(fused, small, atol=4e-5, rtol=0)
    assert fused[0, 5] == 0.0 and fused[nq - 1, nc - 1] == 0.0
    idx = [0, 1, 5, nc // 2, nc - 2, nc - 1]
    ref = np.array([[-torch.cdist(x, cands[i]).min().item() for i in idx] for x in queries], dtype=np.float32)
    np.testing.assert_allclose(fused[:, idx], ref, atol=4e-5, rtol=0)


@pytest.mark.parametrize('scale', [2.0, 3.0])
def test_fused_kernel_overflowed_pairs_are_resolved(amd, scale):
    """A candidate that shares a sentence with the query, on vectors 2 - 3 x N(0, 1): a zero cost next to costs of ~80 - 120
    takes the fused kernel's shifted sums out of fp32 range.  The poisoned pairs are re-solved in the max-shifted form (found by
    the fuzz sweep: they came back NaN) -- one pool (two queries), batched jobs without tables (SELF: the repair searches
    job_off), and the hybrid with a long document in the pool."""
    g = torch.Generator().manual_seed(int(scale * 10))
    mk = lambda n: scale * torch.randn(n, 768, generator=g)
    q = [mk(8), mk(5)]
    c = [mk(int(torch.randint(1, 9, (1,), generator=g))) for _ in range(4100)]
    c[1] = torch.cat([q[0][:1], mk(1)])
    c[2] = torch.cat([q[0][:1], mk(7)])
    c[3] = q[0][:1].clone()
    want = [orc.get_similarity(q[0], c[j]) for j in (1, 2, 3)]
    tol = 5e-2 * scale                      # coincident sentences: geomloss's own cancellation noise (test_gpu_scoring)
    got = amd.scorer.score_pool(q, c, method='ot', schedule='pair').cpu().numpy()
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got[0, 1:4], want, atol=tol, rtol=0)
    # batched jobs, <= 64 of them: no candidate -> job table for the repair
    pools = [c[:1500], c[1500:2900], [c[3], c[1]] + c[2900:4100]]
    queries = [q[1], q[1], q[0]]
    ranked = amd.scorer.rank_pools(queries, pools, k=None)
    scores = dict(ranked[2])
    assert all(np.isfinite(s) for r in ranked for _, s in r)
    np.testing.assert_allclose([scores[1], scores[0]], [want[0], want[2]], atol=tol, rtol=0)
    # one long document in the pool: the hybrid (fused kernel on the short pairs, 16-row kernels on the long ones)
    c_long = list(c)
    c_long[10] = mk(13)
    got = amd.scorer.score_pool([q[0]], c_long, method='ot', schedule='pair').cpu().numpy()
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got[0, 1:4], want, atol=tol, rtol=0)
    assert abs(got[0, 10] - orc.get_similarity(q[0], c_long[10])) < 1e-4 * scale
