"""The forms of the Sinkhorn kernel (one solve per wave / block forms) agree with each other and with the oracle.
aspire_debug_set("SINKHORN", wave|block|block16) pins the form; the default picks by grid size."""

import numpy as np
import pytest
import torch

from oracle import aspire_oracle as orc

pytestmark = pytest.mark.gpu


def _plan_sim(x, y, dtype):
    """AllPairMaskedWasserstein.compute_distance(..., return_pair_sims=True)[0] of one pair (pair_distances.py:61-86), in `dtype`"""
    from oracle import aspire_oracle as orc
    xt = orc.RepLen(x[None].to(dtype).permute(0, 2, 1), [len(x)])
    yt = orc.RepLen(y[None].to(dtype).permute(0, 2, 1), [len(y)])
    return orc.AllPairMaskedWasserstein({}).compute_distance(xt, yt, return_pair_sims=True)[0][0]
TOL = 1e-4


@pytest.fixture(scope='module')
def amd():
    from aspire_amd import ops, scorer, _lib
    assert torch.cuda.is_available()
    return type('NS', (), dict(ops=ops, scorer=scorer, lib=_lib))


def pinned(**kv):
    from aspire_amd._lib import pinned as _pinned
    return _pinned(**kv)


def _docs(seed, lens, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return [scale * torch.randn(int(n), 768, generator=g) for n in lens]


@pytest.mark.parametrize('qlens,clens', [
    ([8], [8, 5, 1, 3, 8, 7, 2, 6] * 3),              # T = 1: 16 solves per wave, tail wave
    ([4, 7], [8, 1] * 9),
    ([12], [12, 9, 16, 1, 13] * 3),                   # T = 2: 4 solves per wave
    ([20, 17], [24, 3, 18] * 2),                      # T = 3 (R = 6)
    ([32], [32, 26, 1, 30, 25]),                      # T = 4 (R = 8)
])
@pytest.mark.parametrize('cost', ['valu', 'mfma'])
def test_block_form_matches_oracle(amd, qlens, clens, cost):
    q, c = _docs(51, qlens), _docs(52, clens)
    with pinned(SINKHORN='block', COST_PATH=cost):
        got = amd.scorer.score_pool(q, c, method='ot', schedule='pair').cpu().numpy()
    want = np.array([[orc.get_similarity(x, y) for y in c] for x in q], dtype=np.float32)
    np.testing.assert_allclose(got, want, atol=TOL, rtol=0)


@pytest.mark.parametrize('hp', [dict(geoml_scaling=0.5), dict(geoml_blur=0.5, geoml_scaling=0.99),
                                dict(sent_sm_temp=10.0), dict(geoml_scaling=0.01), dict(geoml_blur=1e-3)])
def test_block_form_hparams_and_repair(amd, hp):
    """Small scaling makes the shifted sums overflow in the block form: those pairs come back through the
    repair kernel (max-shifted solver) -- nothing is left NaN."""
    q, c = _docs(61, [8, 6]), _docs(62, [8, 2, 7, 5] * 4)
    with pinned(SINKHORN='block'):
        got = amd.scorer.score_pool(q, c, method='ot', schedule='pair', hparams=hp).cpu().numpy()
    with pinned(SINKHORN='wave'):
        ref = amd.scorer.score_pool(q, c, method='ot', schedule='pair', hparams=hp).cpu().numpy()
    assert np.isfinite(got).all()
    want = np.array([[orc.get_similarity(x, y, hp) for y in c] for x in q], dtype=np.float32)
    np.testing.assert_allclose(ref, want, atol=TOL, rtol=0)
    np.testing.assert_allclose(got, want, atol=TOL, rtol=0)


@pytest.mark.parametrize('scale', [1e-3, 30.0])
def test_block_form_extreme_diameters(amd, scale):
    q, c = _docs(71, [8], scale), _docs(72, [8, 4, 6, 1] * 2, scale)
    with pinned(SINKHORN='block'):
        got = amd.scorer.score_pool(q, c, method='ot', schedule='pair').cpu().numpy()
    want = np.array([[orc.get_similarity(x, y) for y in c] for x in q], dtype=np.float32)
    np.testing.assert_allclose(got, want, atol=TOL * max(1.0, scale), rtol=0)


@pytest.mark.parametrize('nq,nc,s', [(2, 5000, 8), (1, 4500, 12), (2, 2100, 16), (3, 1700, 20), (2, 2100, 23), (1, 4100, 27), (1, 4100, 32)])
@pytest.mark.parametrize('want', ['distance', 'plan'])
def test_forms_agree_at_size(amd, nq, nc, s, want):
    g = torch.Generator().manual_seed(nq * 100 + s)
    lens_q = torch.randint(1, s + 1, (nq,), generator=g)
    lens_c = torch.randint(1, s + 1, (nc,), generator=g)
    lens_q[0] = s
    q = amd.ops.DeviceRepSet.from_list([torch.randn(int(n), 768, generator=g) for n in lens_q])
    c = amd.ops.DeviceRepSet.from_list([torch.randn(int(n), 768, generator=g) for n in lens_c])
    w = amd.lib.OT_DISTANCE if want == 'distance' else amd.lib.OT_PLAN_SIM
    out = {}
    # block-dense / block-wide pin the lanes-per-pair layout of the block form (by default the grid size picks it)
    forms = ['wave', 'block'] + (['block16'] if s <= 8 else []) + (['block-dense', 'block-wide'] if s <= 16 else [])
    # documents of <= 8 rows at this size run costs + solves fused in one launch by default: OT_FORM='tile' is the two-kernel
    # form whose Sinkhorn kernel SINKHORN= pins
    for form in forms:
        with pinned(SINKHORN=form, OT_FORM='tile'):
            out[form] = amd.ops.ot_sinkhorn(q, c, want=w).cpu().numpy()
    dflt = amd.ops.ot_sinkhorn(q, c, want=w).cpu().numpy()
    if want == 'distance':
        tol = 5e-5
    else:
        # plan-weighted similarity: the bound of this case from a sample of its pairs -- the reference's fp32 path against the
        # same arithmetic in float64 (tests/plan_sim_floor.py); every form must sit inside it, two forms inside twice that
        import plan_sim_floor
        rows_q, rows_c = q.rows.cpu(), c.rows.cpu()
        sq, sc = q.start.cpu().tolist(), c.start.cpu().tolist()
        lq, lc = q.len.cpu().tolist(), c.len.cpu().tolist()
        pick = np.random.RandomState(nq + s).choice(nc, 24, replace=False)
        w32, w64, idx = [], [], []
        for qi in range(nq):
            x = rows_q[sq[qi]:sq[qi] + lq[qi]]
            for ci in pick:
                y = rows_c[sc[ci]:sc[ci] + lc[ci]]
                w32.append(float(_plan_sim(x, y, torch.float32)))
                w64.append(float(_plan_sim(x, y, torch.float64)))
                idx.append(qi * nc + int(ci))
        b = plan_sim_floor.bound(w32, w64)
        for form in forms:
            plan_sim_floor.check(out[form].reshape(-1)[idx], w32, w64, form)
        tol = 2 * b
    for form in forms[1:]:
        assert np.isfinite(out[form]).all()
        np.testing.assert_allclose(out[form], out['wave'], atol=tol, rtol=0)
    if s <= 8:
        with pinned(OT_FORM='fused'):
            fused = amd.ops.ot_sinkhorn(q, c, want=w).cpu().numpy()
        assert np.array_equal(dflt, fused)          # few queries x a big pool of short documents: the fused kernel
        np.testing.assert_allclose(fused, out['wave'], atol=tol, rtol=0)
    else:
        assert np.array_equal(dflt, out['block'])   # >= 2500 pairs of long documents: the block form is the default
    # at the reference's hyper-parameters the block form needs no repairs (its sums stay in fp32 range)
    with pinned(SINKHORN='block-norepair', OT_FORM='tile'):
        raw = amd.ops.ot_sinkhorn(q, c, want=w).cpu().numpy()
    assert np.array_equal(raw, out['block'])


@pytest.mark.parametrize('qlens,clens', [
    ([8], [8, 5, 1, 3, 8, 7, 2, 6] * 5),          # ragged single-tile pool
    ([3, 8], [1, 8, 4] * 7),                      # two queries: items alternate between them
    ([5], [8, 2, 7] * 5),                         # odd pool
])
@pytest.mark.parametrize('blocks', [0, 7])
def test_small_pool_cost_kernel_matches_oracle(amd, qlens, clens, blocks):
    """The small-pool cost kernel against the oracle, one pair per workgroup (default) and with 7 persistent workgroups
    walking several pairs each (the software-pipelined loop: next item's rows in flight, the redo mask's own LDS word)
    -- bit for bit the same either way."""
    q, c = _docs(61, qlens), _docs(62, clens)
    want = np.array([[orc.get_similarity(x, y) for y in c] for x in q], dtype=np.float32)
    with pinned(OT_FORM='small'):           # (the default for such a pool is the one-launch kernel, round 5: checked against the oracle too)
        ref = amd.scorer.score_pool(q, c, method='ot', schedule='pair').cpu().numpy()
        with pinned(COST1_BLOCKS=blocks):
            got = amd.scorer.score_pool(q, c, method='ot', schedule='pair').cpu().numpy()
    np.testing.assert_allclose(got, want, atol=TOL, rtol=0)
    np.testing.assert_array_equal(got, ref)
    np.testing.assert_allclose(amd.scorer.score_pool(q, c, method='ot', schedule='pair').cpu().numpy(), want, atol=TOL, rtol=0)


def test_persistent_cost_kernel_with_duplicate_sentences(amd):
    """Several items per workgroup AND entries that take the direct-formula redo path (a candidate sharing sentences with
    the query): the redo mask of one item must not be clobbered by the next item's reduction scratch."""
    g = torch.Generator().manual_seed(5)
    query = torch.randn(8, 768, generator=g)
    cands = []
    for i in range(40):
        c = torch.randn(int(torch.randint(1, 9, (1,), generator=g)), 768, generator=g)
        if i % 3 == 0:
            c[0] = query[i % 8]
        cands.append(c)
    want = np.array([orc.get_similarity(query, c) for c in cands], dtype=np.float32)
    with pinned(OT_FORM='small'):
        ref = amd.scorer.score_pool([query], cands, method='ot', schedule='pair').cpu().numpy()[0]
        with pinned(COST1_BLOCKS=3):
            got = amd.scorer.score_pool([query], cands, method='ot', schedule='pair').cpu().numpy()[0]
    np.testing.assert_allclose(ref, want, atol=2e-2, rtol=0)      # duplicate sentences: the expansion formula cancels, 1e-4 .. 1.6e-2 either way (see test_gpu_scoring.test_duplicate_sentence_pair)
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize('qlens,clens', [([8], [8] * 70), ([3, 8, 1], [1, 8, 4, 7, 2] * 12), ([5], [6])])
def test_one_wave_per_pair_kernel_matches_oracle_and_the_two_launch_form(amd, qlens, clens):
    """pair_one_kernel (round 5: costs + solve of a pair on ONE wave, one launch -- the default of every small grid of short documents)
    against the oracle and against pair_cost1_kernel + sinkhorn_kernel<1>: ragged documents, several queries, every output of
    compute_distance, rows with a common component (the centred form), PAIRED pairing through caching_score's padded call"""
    q, c = _docs(71, qlens), _docs(72, clens)
    want = np.array([[orc.get_similarity(x, y) for y in c] for x in q], dtype=np.float32)
    with pinned(OT_FORM='one'):
        one = amd.scorer.score_pool(q, c, method='ot', schedule='pair').cpu().numpy()
    with pinned(OT_FORM='small'):
        two = amd.scorer.score_pool(q, c, method='ot', schedule='pair').cpu().numpy()
    dflt = amd.scorer.score_pool(q, c, method='ot', schedule='pair').cpu().numpy()
    np.testing.assert_allclose(one, want, atol=TOL, rtol=0)
    np.testing.assert_allclose(one, two, atol=5e-5, rtol=0)
    np.testing.assert_array_equal(dflt, one)                       # the default IS this kernel
    # rows that share a large common component: ops sets ASPIRE_OT_FLAG_CENTER from a sample of the pool
    g = torch.Generator().manual_seed(73)
    common = 3.0 * torch.randn(768, generator=g)
    qc, cc = [x + common for x in q], [y + common for y in c]
    wantc = np.array([[orc.get_similarity(x.double(), y.double()) for y in cc] for x in qc])
    with pinned(OT_FORM='one'):
        onec = amd.scorer.score_pool(qc, cc, method='ot', schedule='pair').cpu().numpy()
    np.testing.assert_allclose(onec, wantc, atol=3e-4, rtol=0)


def test_one_wave_per_pair_kernel_transport_plan_outputs(amd):
    """the five return_pair_sims outputs through the one-launch kernel (sinkhorn_pair<1> writes them) = the two-launch form's"""
    from aspire_amd import AllPairMaskedWasserstein, rep_len_tup
    g = torch.Generator().manual_seed(74)
    lens_q, lens_c = [7, 3, 8, 5], [6, 8, 2, 4]
    qp = torch.zeros(4, 8, 768)
    cp = torch.zeros(4, 8, 768)
    for i, (a_, b_) in enumerate(zip(lens_q, lens_c)):
        qp[i, :a_] = torch.randn(a_, 768, generator=g)
        cp[i, :b_] = torch.randn(b_, 768, generator=g)
    qt = rep_len_tup(embed=qp.permute(0, 2, 1), abs_lens=lens_q)
    ct = rep_len_tup(embed=cp.permute(0, 2, 1), abs_lens=lens_c)
    res = {}
    for form in ('one', 'small'):
        with pinned(OT_FORM=form):
            wd, inter = AllPairMaskedWasserstein({}).compute_distance(qt, ct, return_pair_sims=True)
        res[form] = [wd] + list(inter)
    for a_, b_ in zip(res['one'], res['small']):
        np.testing.assert_allclose(a_.numpy(), b_.numpy(), atol=5e-4, rtol=0)
