"""Matrix-core cost stage (aspire_amd/csrc/gram.hip) against the oracle and against the VALU kernels.

The library picks the form by shape; aspire_debug_set("COST_PATH", mfma|valu) pins it so both can be run on the same
inputs.  Every comparison goes through the C ABI (ops -> libaspire_hip.so)."""
import os

import numpy as np
import pytest
import torch

from oracle import aspire_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope='module')
def amd():
    from aspire_amd import ops, scorer, pair_distances, _lib
    assert torch.cuda.is_available()
    return type('NS', (), dict(ops=ops, scorer=scorer, pd=pair_distances, lib=_lib))


def cost_path(which):
    from aspire_amd._lib import pinned
    return pinned(COST_PATH=which)


def _docs(seed, lens, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return [scale * torch.randn(int(n), 768, generator=g) for n in lens]


def _l2max_oracle(q, c):
    return -orc.allpair_masked_dist_l2max(orc.RepLen(q[None].permute(0, 2, 1), [len(q)]),
                                          orc.RepLen(c[None].permute(0, 2, 1), [len(c)])).item()


@pytest.mark.parametrize('qlens,clens', [
    ([8, 8, 8], [8] * 40),                                    # aligned 8-row documents, 32-row query tile
    ([5, 8, 1, 7, 3], [8, 3, 1, 6, 7, 2, 8, 5] * 5),          # ragged
    ([12] * 11, [12] * 23),                                   # 12-row slots, 10 per tile, two query tiles
    ([9, 16, 13], [11, 1, 16, 4] * 6),                        # T = 2 ragged
    ([20, 3], [17, 24, 2, 9] * 3),                            # T = 3
    ([32], [32, 1, 30, 26, 25] * 2),                          # T = 4, crosses the cdist 25/26 switch
])
def test_gram_ot_and_l2max_match_oracle(amd, qlens, clens):
    q, c = _docs(11, qlens), _docs(12, clens)
    with cost_path('mfma'):
        ot = amd.scorer.score_pool(q, c, method='ot', schedule='pair').cpu().numpy()
        l2 = amd.scorer.score_pool(q, c, method='l2max').cpu().numpy()
    want_ot = np.array([[orc.get_similarity(x, y) for y in c] for x in q], dtype=np.float32)
    want_l2 = np.array([[_l2max_oracle(x, y) for y in c] for x in q], dtype=np.float32)
    np.testing.assert_allclose(ot, want_ot, atol=TOL, rtol=0)
    np.testing.assert_allclose(l2, want_l2, atol=TOL, rtol=0)


def test_gram_batch_schedule_matches_valu(amd):
    """caching_score's grouping (one epsilon schedule per 64 candidates, plan-weighted similarity)."""
    q, c = _docs(21, [8, 6, 8, 7]), _docs(22, np.random.RandomState(0).randint(1, 9, size=200))
    with cost_path('mfma'):
        a = amd.scorer.score_pool(q, c, method='ot', schedule='batch').cpu().numpy()
    with cost_path('valu'):
        b = amd.scorer.score_pool(q, c, method='ot', schedule='batch').cpu().numpy()
    # the plan-weighted similarity amplifies fp32 rounding of the costs: both forms against the float64 oracle, each no further
    # from it than the reference's own fp32 path (tests/plan_sim_floor.py)
    import plan_sim_floor
    for qi, qd in enumerate(q):
        want = np.array(orc.rank_pool_caching(qd.numpy(), [x.numpy() for x in c]), dtype=np.float32)
        truth = np.array(orc.rank_pool_caching(qd.numpy(), [x.numpy() for x in c], dtype=torch.float64))
        plan_sim_floor.check(a[qi], want, truth, 'matrix-pipe tiles')
        plan_sim_floor.check(b[qi], want, truth, 'valu tiles')
    assert np.median(np.abs(a - b)) < 1e-3


def test_gram_near_duplicate_sentences(amd):
    """x ~ y: the expansion |x|^2 - 2 x.y + |y|^2 cancels; those entries are recomputed directly, so a candidate
    that repeats a query sentence (exactly, or up to a 1e-3 perturbation) scores like the reference's direct
    torch.cdist."""
    g = torch.Generator().manual_seed(5)
    q = _docs(31, [8, 8, 8, 8])
    c = _docs(32, [8] * 24)
    c[3][2] = q[1][5]                                                # exact copy -> distance 0
    c[7][0] = q[2][0] + 1e-3 * torch.randn(768, generator=g)         # distance ~0.028
    c[9][7] = q[0][1] + 1e-2 * torch.randn(768, generator=g)         # distance ~0.28
    with cost_path('mfma'):
        l2 = amd.scorer.score_pool(q, c, method='l2max').cpu().numpy()
        ot = amd.scorer.score_pool(q, c, method='ot', schedule='pair').cpu().numpy()
    want_l2 = np.array([[_l2max_oracle(x, y) for y in c] for x in q], dtype=np.float32)
    np.testing.assert_allclose(l2, want_l2, atol=2e-5, rtol=0)
    assert l2[1, 3] == 0.0
    # The OT COST is the expansion in the reference too (geomloss squared_distances), so for the three coincident
    # pairs the reference's own value is rounding noise of its summation order (sqrt of a cancelled ~1e-4): only the
    # other pairs can be held to 1e-4; the coincident ones agree to the size of that noise.
    want_ot = np.array([[orc.get_similarity(x, y) for y in c] for x in q], dtype=np.float32)
    noisy = np.zeros_like(want_ot, dtype=bool)
    noisy[1, 3] = noisy[2, 7] = noisy[0, 9] = True
    np.testing.assert_allclose(ot[~noisy], want_ot[~noisy], atol=TOL, rtol=0)
    np.testing.assert_allclose(ot[noisy], want_ot[noisy], atol=5e-2, rtol=0)


@pytest.mark.parametrize('nq,nc,s', [(32, 3000, 8), (16, 1500, 12), (3, 2500, 8), (1, 1500, 20)])
def test_gram_matches_valu_at_size(amd, nq, nc, s):
    """Bench-sized grids (several tiles per XCD, tail tiles): both forms of the cost stage agree; the default
    dispatch picks one of them."""
    g = torch.Generator().manual_seed(nq * 1000 + s)
    qrows = torch.randn(nq * s, 768, generator=g).cuda()
    crows = torch.randn(nc * s, 768, generator=g).cuda()
    mk = lambda rows, n: amd.ops.DeviceRepSet(rows, (torch.arange(n, device='cuda', dtype=torch.int32) * s).contiguous(),
                                              torch.full((n,), s, device='cuda', dtype=torch.int32), ext=0, max_len=s)
    q, c = mk(qrows, nq), mk(crows, nc)
    out = {}
    for path in ('mfma', 'valu'):
        with cost_path(path):
            out[path] = (amd.ops.ot_sinkhorn(q, c).cpu().numpy(), amd.ops.l2max_scores(q, c).cpu().numpy())
    dflt = (amd.ops.ot_sinkhorn(q, c).cpu().numpy(), amd.ops.l2max_scores(q, c).cpu().numpy())
    for k in range(2):
        assert np.isfinite(out['mfma'][k]).all()
        np.testing.assert_allclose(out['mfma'][k], out['valu'][k], atol=5e-5, rtol=0)
        assert np.array_equal(dflt[k], out['mfma'][k]) or np.array_equal(dflt[k], out['valu'][k])


def test_gram_bf16x3_and_fp32_input_forms_agree(amd):
    """the 128-column Gram tiles on the bf16 matrix pipe (three-way operand split, six products: the default) against the
    same tiles on the fp32-input MFMA (pinned GEMM=f32): max-sim to 2e-5 (both are fp32-accurate expansions), otAspire to
    5e-5, and both against -min cdist in float64 on a sample"""
    from aspire_amd._lib import pinned
    nq, nc, s = 32, 4000, 8
    g = torch.Generator().manual_seed(77)
    qrows = (torch.randn(nq * s, 768, generator=g) * torch.linspace(0.3, 2.0, 768)).cuda()
    crows = (torch.randn(nc * s, 768, generator=g) + 0.25).cuda()
    mk = lambda rows, n: amd.ops.DeviceRepSet(rows, (torch.arange(n, device='cuda', dtype=torch.int32) * s).contiguous(),
                                              torch.full((n,), s, device='cuda', dtype=torch.int32), ext=0, max_len=s)
    q, c = mk(qrows, nq), mk(crows, nc)
    out = {}
    for form in ('bf16x3', 'f32'):
        with pinned(COST_PATH='mfma', GEMM=form):
            out[form] = (amd.ops.l2max_scores(q, c).view(nq, nc).cpu().numpy(), amd.ops.ot_sinkhorn(q, c).view(nq, nc).cpu().numpy())
    np.testing.assert_allclose(out['bf16x3'][0], out['f32'][0], atol=2e-5, rtol=0)
    np.testing.assert_allclose(out['bf16x3'][1], out['f32'][1], atol=5e-5, rtol=0)
    qd, cd = qrows.double().cpu().view(nq, s, 768), crows.double().cpu().view(nc, s, 768)
    for qi, ci in ((0, 0), (5, 123), (31, 3999), (17, 2048)):
        want = -torch.cdist(qd[qi], cd[ci]).min().item()
        for form in out:
            assert abs(out[form][0][qi, ci] - want) < 3e-5, (form, qi, ci)


def test_gram_workspace_chunking(amd):
    """A workspace smaller than the pool's slots + boxes: candidates run in chunks, results unchanged."""
    q, c = _docs(41, [8] * 6), _docs(42, [8] * 300)
    qs, cs = amd.ops.DeviceRepSet.from_list(q), amd.ops.DeviceRepSet.from_list(c)
    with cost_path('mfma'):
        full = amd.ops.ot_sinkhorn(qs, cs).cpu().numpy()
        small = torch.empty(6 * 70 * 516 + 70 * 6144 + 6 * 6144 + 64, dtype=torch.uint8, device='cuda')
        chunked = amd.ops.ot_sinkhorn(qs, cs, workspace=small).cpu().numpy()
    np.testing.assert_array_equal(full, chunked)


def test_gram_clustered_sentence_vectors(amd):
    """Vectors around one common direction (like real encoder outputs): a sizeable share of the sentence pairs is close
    enough for the expansion to cancel and goes through the direct-formula work list; results still meet the oracle."""
    g = torch.Generator().manual_seed(8)
    base = torch.randn(768, generator=g) * (15.0 / 768 ** 0.5)

    def doc(n):
        return base[None, :] + (0.05 + 0.25 * torch.rand(n, 1, generator=g)) * torch.randn(n, 768, generator=g)
    q = [doc(8), doc(5), doc(12)]
    c = [doc(int(n)) for n in torch.randint(1, 13, (45,), generator=g)]
    with cost_path('mfma'):
        l2 = amd.scorer.score_pool(q, c, method='l2max').cpu().numpy()
        ot = amd.scorer.score_pool(q, c, method='ot', schedule='pair').cpu().numpy()
    with cost_path('valu'):
        l2v = amd.scorer.score_pool(q, c, method='l2max').cpu().numpy()
    want_l2 = np.array([[_l2max_oracle(x, y) for y in c] for x in q], dtype=np.float32)
    want_ot = np.array([[orc.get_similarity(x, y) for y in c] for x in q], dtype=np.float32)
    # entries just above the work-list threshold keep the expansion: its error there is ~5e-7 (|x|^2+|y|^2) / 2d,
    # i.e. up to ~4e-5 on this data -- inside the 1e-4 bar; the direct-formula kernels sit at 1e-6
    np.testing.assert_allclose(l2, want_l2, atol=6e-5, rtol=0)
    np.testing.assert_allclose(l2v, want_l2, atol=1e-5, rtol=0)
    np.testing.assert_allclose(ot, want_ot, atol=TOL, rtol=0)


def test_baseline_config3_full_size_properties(amd):
    """BASELINE config 3 at full size (32 queries x 50 000 candidates, 8 x 768, tsAspire max-sim): the oracle cannot
    score 1.6 M pairs in seconds, so parity is pinned through size-independent properties -- any subset of the pool
    scored on its own gives the same numbers (each pair is independent), the two kernel families agree on a sample,
    and the oracle agrees on a handful of pairs."""
    g = torch.Generator().manual_seed(1)
    nq, nc, s = 32, 50_000, 8
    qrows = torch.randn(nq * s, 768, generator=g).cuda()
    crows = torch.randn(nc * s, 768, generator=g).cuda()
    mk = lambda rows, n: amd.ops.DeviceRepSet(rows, (torch.arange(n, device='cuda', dtype=torch.int32) * s).contiguous(),
                                              torch.full((n,), s, device='cuda', dtype=torch.int32), ext=0, max_len=s)
    q, c = mk(qrows, nq), mk(crows, nc)
    full = amd.ops.l2max_scores(q, c).view(nq, nc)
    assert torch.isfinite(full).all() and (full < 0).all()
    pick = torch.randperm(nc, generator=g)[:3000].sort().values
    sub_rows = crows.view(nc, s, 768)[pick.cuda()].reshape(-1, 768).contiguous()
    sub = amd.ops.l2max_scores(q, mk(sub_rows, len(pick))).view(nq, len(pick))
    np.testing.assert_allclose(full[:, pick.cuda()].cpu().numpy(), sub.cpu().numpy(), atol=1e-6, rtol=0)
    with cost_path('valu'):
        valu = amd.ops.l2max_scores(q, mk(sub_rows, len(pick))).view(nq, len(pick))
    np.testing.assert_allclose(sub.cpu().numpy(), valu.cpu().numpy(), atol=2e-5, rtol=0)
    for qi, ci in [(0, 0), (31, 49_999), (7, 12_345), (19, 33_333)]:
        want = _l2max_oracle(qrows[qi * s:(qi + 1) * s].cpu(), crows[ci * s:(ci + 1) * s].cpu())
        assert full[qi, ci].item() == pytest.approx(want, abs=TOL)
    # per-query top-100 of the full matrix == stable descending sort of that row
    top_s, top_i = amd.ops.topk_desc(full.contiguous(), 100)
    for qi in (0, 17):
        order = orc.rank_descending(full[qi].cpu().tolist())[:100]
        assert top_i[qi].cpu().tolist() == order


def test_baseline_config3_full_size_on_the_plane_tiles(amd):
    """The kernel the bench's `config3` key times, at the size it times it: pair_gram_p_kernel on a 50 000-document store with fp16
    planes (gramp.hip), ALL 1.6 M max-sim scores against torch.cdist in float64 on the GPU (pair_distances.py:138-186), five
    calls with fresh query planes each -- the race of round 4 (NOTES.md round 4, item 4: a stale k block, 1.3e-4 in ONE tile of
    ~3000, run-dependent) was invisible below this size.  Beside it the tiles that read the fp32 rows (bf16x3), and the otAspire
    cost slots of the same tiles through the Sinkhorn stage against the fp32-row tiles and the oracle."""
    from aspire_amd._lib import pinned
    g = torch.Generator().manual_seed(1)
    nq, nc, s = 32, 50_000, 8
    qrows = torch.randn(nq * s, 768, generator=g).cuda()
    crows = torch.randn(nc * s, 768, generator=g).cuda()
    mk = lambda rows, n: amd.ops.DeviceRepSet(rows, (torch.arange(n, device='cuda', dtype=torch.int32) * s).contiguous(),
                                              torch.full((n,), s, device='cuda', dtype=torch.int32), ext=0, max_len=s)
    q, c = mk(qrows, nq), mk(crows, nc)
    want = torch.empty(nq, nc, device='cuda', dtype=torch.float64)
    q64 = qrows.double()
    for lo in range(0, nc, 10_000):                     # [256, 80 000] float64 distances per block
        d = torch.cdist(q64, crows[lo * s:(lo + 10_000) * s].double())
        want[:, lo:lo + 10_000] = -d.view(nq, s, 10_000, s).permute(0, 2, 1, 3).reshape(nq, 10_000, s * s).min(-1).values
    rows_form = amd.ops.l2max_scores(q, c).view(nq, nc).double()                  # no planes: the bf16x3 tiles
    assert (rows_form - want).abs().max().item() < 6e-5
    c.prepare_planes()
    worst = 0.0
    for rep in range(5):
        q.drop_planes()
        got = amd.ops.l2max_scores(q, c).view(nq, nc)
        assert q.planes is not None                                              # the plane tiles ran
        err = (got.double() - want).abs()
        worst = max(worst, err.max().item())
        assert err.max().item() < 1e-5, (rep, err.max().item(), int((err > 1e-5).sum()), torch.nonzero(err > 1e-5)[:4].tolist())
    print(f'plane tiles 32 x 50 000 x 8, five calls: max |score - float64| {worst:.2e}')
    # otAspire: the same tiles write the pairs' cost / -cdist slots for the Sinkhorn stage
    ot_p = amd.ops.ot_sinkhorn(q, c, want=amd.lib.OT_SIMILARITY).view(nq, nc)
    assert torch.isfinite(ot_p).all()
    q.drop_planes()
    c.drop_planes()
    ot_r = amd.ops.ot_sinkhorn(q, c, want=amd.lib.OT_SIMILARITY).view(nq, nc)
    diff = (ot_p - ot_r).abs()
    # (a pair whose diameter sits on a step of geomloss's schedule length flips between summation orders: see the config-5 test below)
    assert diff.median().item() < 1e-5 and int((diff > 5e-5).sum()) <= 40 and diff.max().item() < 2e-3, \
        (diff.median().item(), int((diff > 5e-5).sum()), diff.max().item())
    for qi, ci in [(0, 0), (31, 49_999), (7, 12_345), (19, 33_333), (3, 25_000)]:
        w = orc.get_similarity(qrows[qi * s:(qi + 1) * s].cpu(), crows[ci * s:(ci + 1) * s].cpu())
        assert ot_p[qi, ci].item() == pytest.approx(w, abs=TOL)


def test_baseline_config5_slice_properties(amd):
    """BASELINE config 5 shape (128 queries, 12 x 768 sentences) on a slice of one GPU's shard (8192 candidates): subset
    consistency of the otAspire similarities, agreement of the kernel families, and the oracle on a few pairs."""
    g = torch.Generator().manual_seed(2)
    nq, nc, s = 128, 8192, 12
    qrows = torch.randn(nq * s, 768, generator=g).cuda()
    crows = torch.randn(nc * s, 768, generator=g).cuda()
    mk = lambda rows, n: amd.ops.DeviceRepSet(rows, (torch.arange(n, device='cuda', dtype=torch.int32) * s).contiguous(),
                                              torch.full((n,), s, device='cuda', dtype=torch.int32), ext=0, max_len=s)
    q, c = mk(qrows, nq), mk(crows, nc)
    full = amd.ops.ot_sinkhorn(q, c, want=amd.lib.OT_SIMILARITY).view(nq, nc)
    assert torch.isfinite(full).all()
    pick = torch.arange(100, 1124)                      # 1024 candidates: several Gram tiles, block Sinkhorn
    sub_rows = crows.view(nc, s, 768)[pick.cuda()].reshape(-1, 768).contiguous()
    with cost_path('mfma'):        # the same kernel family as the full pool (1024 candidates alone would take the VALU one)
        sub = amd.ops.ot_sinkhorn(q, mk(sub_rows, len(pick)), want=amd.lib.OT_SIMILARITY).view(nq, len(pick))
    assert torch.equal(full[:, pick.cuda()], sub)
    # The two kernel families agree to a few 1e-6 -- except where geomloss's schedule length, ceil(log(diam/blur) /
    # -log(scaling)), flips: the diameter is summed in a different order, and a pair whose diameter sits on a
    # boundary gains or loses one annealing step (5e-4 on the result; the reference flips the same way under any
    # change of its own rounding).  1 pair in 131 072 here.
    with cost_path('valu'):
        with_valu = amd.ops.ot_sinkhorn(q, mk(sub_rows, len(pick)), want=amd.lib.OT_SIMILARITY).view(nq, len(pick))
    diff = (sub - with_valu).abs()
    assert diff.median().item() < 1e-5
    assert int((diff > 5e-5).sum()) <= 8 and diff.max().item() < 2e-3
    for qi, ci in [(0, 100), (127, 1123), (64, 600)]:
        want = orc.get_similarity(qrows[qi * s:(qi + 1) * s].cpu(), crows[ci * s:(ci + 1) * s].cpu())
        assert full[qi, ci].item() == pytest.approx(want, abs=TOL)
