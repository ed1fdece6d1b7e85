"""The fused kernel's CHUNK form (fused.hip + chunk_prep_kernel): batched jobs of short (facet-selected) queries against
abstracts of up to 32 sentences -- BASELINE config 4's shape (pp_settings.py:2-3, evaluate.py:58-76, models.py:127-163) -- with
costs and Sinkhorn solves of every pair in one launch.  Against the oracle, and against the small-batch kernels on the same jobs."""
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


def _batch(amd, seed, sizes, cmax=20, qmax=8, scale=1.0, cmin=1):
    g = torch.Generator().manual_seed(seed)
    c_lens = torch.randint(cmin, cmax + 1, (sum(sizes),), generator=g).tolist()
    q_lens = torch.randint(1, qmax + 1, (len(sizes),), generator=g).tolist()
    cands = [scale * torch.randn(n, 768, generator=g) for n in c_lens]
    queries = [scale * torch.randn(n, 768, generator=g) for n in q_lens]
    off = np.concatenate([[0], np.cumsum(sizes)])
    return queries, cands, off


def _run(amd, queries, cands, off, k=None, **kw):
    q = amd.ops.DeviceRepSet.from_list(queries)
    c = amd.ops.DeviceRepSet.from_list(cands)
    job_off = torch.tensor(off, dtype=torch.int32).cuda()
    max_job = int(np.diff(off).max())
    k = max_job if k is None else k
    scores, top_s, top_i = amd.ops.ot_rank_batch(q, c, job_off, max_job, k, **kw)
    torch.cuda.synchronize()
    return scores.cpu().numpy(), top_s.cpu().numpy(), top_i.cpu().numpy()


@pytest.mark.parametrize('seed,sizes,cmax', [(11, [60] * 12, 20), (12, [125, 0, 1, 37, 200, 3, 90], 32), (13, [700], 16),
                                             (14, [5, 6, 7, 300], 9)])
def test_chunk_form_against_the_oracle_and_the_small_batch_kernels(amd, seed, sizes, cmax):
    queries, cands, off = _batch(amd, seed, sizes, cmax=cmax)
    with amd.lib.pinned(OT_FORM='chunk'):
        got, top_s, top_i = _run(amd, queries, cands, off)
    with amd.lib.pinned(OT_FORM='small'):
        other, _, _ = _run(amd, queries, cands, off)
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, other, atol=5e-5, rtol=0)
    # a sample of pairs, the first and last candidates of every job among them, against the oracle
    rng = np.random.default_rng(seed)
    for j, n in enumerate(sizes):
        if n == 0:
            assert (top_i[j] == -1).all()
            continue
        pick = sorted(set([0, n - 1] + rng.integers(0, n, size=min(n, 8)).tolist()))
        want = np.array([orc.get_similarity(queries[j], cands[off[j] + i]) for i in pick], dtype=np.float32)
        np.testing.assert_allclose(got[off[j] + np.array(pick)], want, atol=TOL, rtol=0)
        # the job's ranking is the stable descending order of its own scores
        order = orc.rank_descending(got[off[j]:off[j + 1]].tolist())
        assert top_i[j, :n].tolist() == order
        assert (top_i[j, n:] == -1).all()


def test_chunk_form_is_the_default_for_the_config4_shape_and_deterministic(amd):
    """50 jobs x 125 abstracts of 3 .. 20 sentences, facet-selected queries of 1 .. 8 rows: the default dispatch takes the CHUNK
    form (same bits as the pinned form), twice in a row (the prep kernel's item order is not deterministic -- atomics -- but a
    pair's arithmetic does not depend on which item it lands in)."""
    queries, cands, off = _batch(amd, 21, [125] * 50, cmax=20, cmin=3)
    a, _, ia = _run(amd, queries, cands, off)
    with amd.lib.pinned(OT_FORM='chunk'):
        b, _, ib = _run(amd, queries, cands, off)
        c, _, ic = _run(amd, queries, cands, off)
    assert np.array_equal(a, b) and np.array_equal(b, c) and np.array_equal(ia, ib) and np.array_equal(ib, ic)
    idx = [0, 1, 124, 125, 3000, 6249]
    job = [i // 125 for i in idx]
    want = np.array([orc.get_similarity(queries[j], cands[i]) for i, j in zip(idx, job)], dtype=np.float32)
    np.testing.assert_allclose(a[idx], want, atol=TOL, rtol=0)


def test_chunk_form_shared_sentences_hparams_and_plan_similarity(amd):
    """Candidates that contain one of their query's sentences (cdist's exact zero: the direct-formula redo, in every chunk of a
    long candidate), large vectors (the overflow repair), other hyper-parameters, and the plan-weighted similarity output."""
    queries, cands, off = _batch(amd, 31, [80, 80, 80, 80], cmax=28, scale=2.0)
    for j, (ci, row_c, row_q) in enumerate([(3, 0, 0), (17, 9, 2), (40, 20, 1), (79, 27, 0)]):
        c = cands[off[j] + ci]
        if row_c < len(c):
            c[row_c] = queries[j][min(row_q, len(queries[j]) - 1)]
        else:
            c[len(c) - 1] = queries[j][0]
    with amd.lib.pinned(OT_FORM='chunk'):
        got, _, _ = _run(amd, queries, cands, off)
        sims, _, _ = _run(amd, queries, cands, off, want=amd.lib.OT_PLAN_SIM)
        hp, _, _ = _run(amd, queries, cands, off, blur=0.1, scaling=0.5, sent_sm_temp=5.0)
    assert np.isfinite(got).all() and np.isfinite(sims).all() and np.isfinite(hp).all()
    idx = [off[0] + 3, off[1] + 17, off[2] + 40, off[3] + 79, 0, 100, 318]
    job = [int(np.searchsorted(off, i, side='right') - 1) for i in idx]
    want = np.array([orc.get_similarity(queries[j], cands[i]) for i, j in zip(idx, job)], dtype=np.float32)
    # the four pairs that share a sentence: geomloss's own cancellation noise on the zero-cost entry (test_gpu_scoring,
    # test_fused_kernel_overflowed_pairs_are_resolved); the others: 2 x N(0,1) vectors, distances of ~80 (relative 5e-6)
    tol = np.array([5e-2 * 2.0] * 4 + [4e-4] * 3)
    assert (np.abs(got[idx] - want) <= tol).all(), (got[idx], want)
    hparams = {'geoml_blur': 0.1, 'geoml_scaling': 0.5, 'sent_sm_temp': 5.0}
    want_hp = np.array([orc.get_similarity(queries[j], cands[i], hparams) for i, j in zip(idx, job)], dtype=np.float32)
    assert (np.abs(hp[idx] - want_hp) <= tol).all(), (hp[idx], want_hp)
    with amd.lib.pinned(OT_FORM='small'):
        sims_other, _, _ = _run(amd, queries, cands, off, want=amd.lib.OT_PLAN_SIM)
    np.testing.assert_allclose(sims, sims_other, atol=2e-2 * 2.0, rtol=0)     # plan-similarity noise floor (test_gpu_scoring) x the vectors' scale


@pytest.mark.parametrize('seed,sizes,cmax', [(41, [125] * 50, 20), (42, [300, 0, 7, 90, 1], 32)])
def test_chunk_form_max_sim(amd, seed, sizes, cmax):
    """tsAspire on the same items (the streaming phase with the max epilogue, fused.hip L2MAX + CHUNK): against the oracle's
    allpair_masked_dist_l2max and against the one-workgroup-per-candidate kernels on the same jobs."""
    queries, cands, off = _batch(amd, seed, sizes, cmax=cmax)
    q = amd.ops.DeviceRepSet.from_list(queries)
    c = amd.ops.DeviceRepSet.from_list(cands)
    job_off = torch.tensor(off, dtype=torch.int32).cuda()
    max_job = int(np.diff(off).max())
    got, _, top_i = amd.ops.l2max_rank_batch(q, c, job_off, max_job, max_job)
    with amd.lib.pinned(OT_FORM='small'):
        other, _, _ = amd.ops.l2max_rank_batch(q, c, job_off, max_job, max_job)
    got, other, top_i = got.cpu().numpy(), other.cpu().numpy(), top_i.cpu().numpy()
    np.testing.assert_allclose(got, other, atol=4e-5, rtol=0)
    rng = np.random.default_rng(seed)
    for j, n in enumerate(sizes):
        if n == 0:
            continue
        pick = sorted(set([0, n - 1] + rng.integers(0, n, size=min(n, 6)).tolist()))
        want = [-orc.allpair_masked_dist_l2max(orc.RepLen(queries[j][None].permute(0, 2, 1), [len(queries[j])]),
                                               orc.RepLen(cands[off[j] + i][None].permute(0, 2, 1), [len(cands[off[j] + i])])).item()
                for i in pick]
        np.testing.assert_allclose(got[off[j] + np.array(pick)], want, atol=TOL, rtol=0)
        assert top_i[j, :n].tolist() == orc.rank_descending(got[off[j]:off[j + 1]].tolist())


@pytest.mark.parametrize('seed,sizes,cmax,qmax', [(51, [90] * 8, 20, 20), (52, [300, 0, 1, 45, 130], 32, 32), (53, [400], 24, 12)])
def test_record_items_for_whole_abstracts_on_both_sides(amd, seed, sizes, cmax, qmax):
    """Un-faceted queries: whole abstracts of up to 32 sentences on BOTH sides (pp_settings.py:2-3).  The 16-row streaming kernel on
    record items (tile16.hip REC: a 16-row half of the query against two candidate halves) + the block Sinkhorn kernel on 24- /
    32-row workspace slots, against the oracle and against the per-candidate tile-loop kernels; tsAspire for queries of <= 16."""
    queries, cands, off = _batch(amd, seed, sizes, cmax=cmax, qmax=qmax)
    cands[0] = torch.randn(cmax, 768, generator=torch.Generator().manual_seed(seed))       # the longest document is present
    got, _, top_i = _run(amd, queries, cands, off)
    with amd.lib.pinned(OT_FORM='small'):
        other, _, _ = _run(amd, queries, cands, off)
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, other, atol=5e-5, rtol=0)
    rng = np.random.default_rng(seed)
    for j, n in enumerate(sizes):
        if n == 0:
            continue
        pick = sorted(set([0, n - 1] + rng.integers(0, n, size=min(n, 5)).tolist()))
        want = np.array([orc.get_similarity(queries[j], cands[off[j] + i]) for i in pick], dtype=np.float32)
        np.testing.assert_allclose(got[off[j] + np.array(pick)], want, atol=TOL, rtol=0)
        assert top_i[j, :n].tolist() == orc.rank_descending(got[off[j]:off[j + 1]].tolist())
    if qmax <= 16:
        q = amd.ops.DeviceRepSet.from_list(queries)
        c = amd.ops.DeviceRepSet.from_list(cands)
        job_off = torch.tensor(off, dtype=torch.int32).cuda()
        max_job = int(np.diff(off).max())
        l2, _, _ = amd.ops.l2max_rank_batch(q, c, job_off, max_job, max_job)
        with amd.lib.pinned(OT_FORM='small'):
            l2o, _, _ = amd.ops.l2max_rank_batch(q, c, job_off, max_job, max_job)
        np.testing.assert_allclose(l2.cpu().numpy(), l2o.cpu().numpy(), atol=4e-5, rtol=0)
        want = -orc.allpair_masked_dist_l2max(orc.RepLen(queries[0][None].permute(0, 2, 1), [len(queries[0])]),
                                              orc.RepLen(cands[0][None].permute(0, 2, 1), [len(cands[0])])).item()
        assert abs(float(l2[0]) - want) < TOL


def test_a_few_facet_queries_against_one_pool(amd):
    """score_pool with the facets of one paper (three short queries) against one pool of whole abstracts: scored as batched jobs
    over the same index list (CHUNK form) -- the same numbers as one query per call and as the oracle."""
    g = torch.Generator().manual_seed(61)
    lens = torch.randint(3, 21, (700,), generator=g).tolist()
    cands = [torch.randn(n, 768, generator=g) for n in lens]
    queries = [torch.randn(n, 768, generator=g) for n in (2, 5, 8)]
    got = amd.scorer.score_pool(queries, cands, method='ot', schedule='pair').cpu().numpy()
    assert got.shape == (3, 700) and np.isfinite(got).all()
    for i in range(3):
        one = amd.scorer.score_pool([queries[i]], cands, method='ot', schedule='pair').cpu().numpy()[0]
        np.testing.assert_allclose(got[i], one, atol=5e-5, rtol=0)
    idx = [(0, 0), (1, 350), (2, 699)]
    want = np.array([orc.get_similarity(queries[i], cands[j]) for i, j in idx], dtype=np.float32)
    np.testing.assert_allclose([got[i, j] for i, j in idx], want, atol=TOL, rtol=0)
