"""Round-3 parity / edge items: batched jobs with a document beyond the tile kernels' 32 rows, the ONE_FORM (deterministic)
request -- batched and per-query rankings bit-equal --, documents without sentences, and the hybrid forms under a pinned
one-solve-per-wave Sinkhorn kernel."""
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


def _docs(seed, lens, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return [scale * torch.randn(int(n), 768, generator=g) for n in lens]


def test_batched_jobs_with_a_document_beyond_32_rows(amd, tmp_path):
    """AspireNER's appended entity rows (models.py:224-233) make documents of 33 .. 128 rows; one of them in a pool used to make
    the default evaluate.score (rank_pools -> aspire_ot_rank_batch_f32) raise NotImplementedError while queries_per_call=1
    scored it.  Now the batch entry follows its tile kernels with the long-document kernel, as the single-pool entry does."""
    from aspire_amd import evaluate as ev
    from aspire_amd.repstore import RepStore
    g = torch.Generator().manual_seed(91)
    pids = [f'p{i}' for i in range(40)]
    lens = torch.randint(2, 14, (40,), generator=g).tolist()
    lens[17], lens[3] = 40, 33                     # p17 sits in two pools; p3 is a query
    store = RepStore({p: torch.randn(int(n), 768, generator=g).numpy() for p, n in zip(pids, lens)})
    test_pool = {'p0': {'cands': pids[10:30]}, 'p1': {'cands': pids[15:40]}, 'p3': {'cands': pids[20:28]}}
    got = ev.score(str(tmp_path / 'b'), test_pool, store, method='ot', schedule='pair')        # default: 32 queries per call
    one = ev.score(str(tmp_path / 'o'), test_pool, store, method='ot', schedule='pair', queries_per_call=1)
    for q, d in test_pool.items():
        want = [orc.get_similarity(torch.from_numpy(store.get(q)), torch.from_numpy(store.get(c))) for c in d['cands']]
        order = [d['cands'][i] for i in orc.rank_descending(want)]
        assert [c for c, _ in got[q]] == order and [c for c, _ in one[q]] == order
        np.testing.assert_allclose([-s for _, s in got[q]], sorted(want, reverse=True), atol=TOL, rtol=0)
    # the library call itself, with a 100-row query and 128-row candidates among short ones, scores + full rank
    qs = _docs(92, [100, 5])
    pools = [_docs(93, [4, 128, 9, 33, 8]), _docs(94, [12, 3, 70])]
    ranked = amd.scorer.rank_pools(qs, pools)
    for qd, pool, r in zip(qs, pools, ranked):
        w = [orc.get_similarity(qd, c) for c in pool]
        assert [i for i, _ in r] == orc.rank_descending(w)
        np.testing.assert_allclose([s for _, s in r], sorted(w, reverse=True), atol=TOL, rtol=0)


@pytest.mark.parametrize('method', ['ot', 'l2max'])
def test_deterministic_batched_and_per_query_rankings_are_bit_equal(amd, method):
    """The config-4 shape (tools/csfbench.py: 50 facet-selected queries x their own pools of 125 ragged abstracts of 3 .. 20
    sentences).  By default the batched call and the per-query calls take different kernel families (scores a few 1e-5 apart,
    near-ties may swap); with deterministic=True every pair goes through one kernel form: the same bits, the same order."""
    J, NC = 50, 125
    g = torch.Generator().manual_seed(4)
    c_lens = torch.randint(3, 21, (J * NC,), generator=g).tolist()
    q_lens = torch.randint(1, 9, (J,), generator=g).tolist()
    docs = _docs(5, c_lens)
    queries = _docs(6, q_lens)
    pools = [docs[j * NC:(j + 1) * NC] for j in range(J)]
    batched = amd.scorer.rank_pools(queries, pools, method=method, deterministic=True)
    for j in (0, 7, 23, 49):
        alone = amd.scorer.rank_pool([queries[j]], pools[j], method=method, deterministic=True)[0]
        assert alone == batched[j]                     # (pid, score) lists: same order, same floats
    # ... and it is the reference's number
    j = 7
    if method == 'ot':
        want = [orc.get_similarity(queries[j], c) for c in pools[j]]
    else:
        want = [-orc.allpair_masked_dist_l2max(orc.RepLen(queries[j][None].permute(0, 2, 1), [q_lens[j]]),
                                               orc.RepLen(c[None].permute(0, 2, 1), [len(c)])).item() for c in pools[j]]
    np.testing.assert_allclose([s for _, s in batched[j]], sorted(want, reverse=True), atol=TOL, rtol=0)
    # score_pool takes the flag too, and rejects it where no single form is built
    one = amd.scorer.score_pool([queries[j]], pools[j], method=method, deterministic=True).cpu().numpy()[0]
    assert sorted(one.tolist(), reverse=True) == [s for _, s in batched[j]]
    with pytest.raises(ValueError):
        amd.scorer.score_pool([queries[j]], pools[j], method='l2top2', deterministic=True)


def test_documents_without_sentences_are_rejected(amd):
    """pair_distances.py:57 via models.py:190-197: torch.max over an empty dimension raises for a document of zero sentences; here
    such a candidate used to score OT distance 0 -- the best possible similarity."""
    q = _docs(95, [4])
    cands = _docs(96, [5, 0, 3])
    with pytest.raises(ValueError):
        amd.scorer.score_pool(q, cands, method='ot')
    with pytest.raises(ValueError):
        amd.scorer.rank_pools(q, [cands])
    with pytest.raises(ValueError):
        amd.ops.DeviceRepSet.from_padded(torch.zeros(2, 4, 768), [3, 0])
    from aspire_amd.repstore import RepStore
    store = RepStore({'a': np.zeros((0, 768), np.float32), 'b': np.ones((2, 768), np.float32)})
    store.to_device()
    with pytest.raises(ValueError):
        store.pool(['a', 'b'])


def test_hybrid_forms_leave_short_pairs_alone_under_a_pinned_wave_solver(amd):
    """with SINKHORN=wave pinned the hybrid (fused kernel in front of the 16-row kernels on the long pairs only) let the
    one-solve-per-wave kernel overwrite the short pairs' scores with solves of uninitialised workspace slots.  The hybrid is now
    taken only with a solve stage that honours the gate."""
    g = torch.Generator().manual_seed(97)
    lens = torch.randint(3, 9, (6400,), generator=g).tolist()
    for i in (5, 777, 4000):
        lens[i] = 14
    cands = _docs(98, lens)
    q = _docs(99, [6])
    idx = [0, 5, 6, 777, 3999, 4000, 6399]
    want = np.array([orc.get_similarity(q[0], cands[i]) for i in idx], dtype=np.float32)
    for pin in ({}, {'SINKHORN': 'wave'}, {'SINKHORN': 'block'}):
        with amd.lib.pinned(**pin):
            got = amd.scorer.score_pool(q, cands, method='ot').cpu().numpy()[0]
        assert np.isfinite(got).all(), pin
        np.testing.assert_allclose(got[idx], want, atol=TOL, rtol=0, err_msg=str(pin))
    # batched jobs: 8 x 800 with the same few long documents
    pools = [cands[j * 800:(j + 1) * 800] for j in range(8)]
    queries = _docs(100, [6] * 8)
    for pin in ({}, {'SINKHORN': 'wave'}):
        with amd.lib.pinned(**pin):
            ranked = amd.scorer.rank_pools(queries, pools, k=5)
        w = [orc.get_similarity(queries[0], c) for c in pools[0][:12]]
        top = dict(ranked[0])
        for i, s in enumerate(w):
            if i in top:
                assert abs(top[i] - s) < TOL


def test_rows_with_a_common_component_are_centred(amd):
    """Anisotropic embeddings (every row = a common vector + its own part, mean cosine ~0.8): the expansion |x|^2 - 2 x.y + |y|^2
    cancels for nearly every entry; with ASPIRE_OT_FLAG_CENTER (set by aspire_amd.ops from a sample of the pool) the fused kernels
    subtract the query's mean row first.  Same numbers as the oracle -- one pool, batched jobs (8-row and chunked), max-sim."""
    g = torch.Generator().manual_seed(111)
    base = 2.0 * torch.randn(768, generator=g)
    mk = lambda n: torch.randn(int(n), 768, generator=g) + base
    lens = torch.randint(1, 9, (4200,), generator=g).tolist()
    cands = [mk(n) for n in lens]
    queries = [mk(7), mk(3)]
    c = amd.ops.DeviceRepSet.from_list(cands)
    assert c.center_hint() and not amd.ops.DeviceRepSet.from_list(_docs(3, [8] * 64)).center_hint()
    got = amd.scorer.score_pool(queries[:1], cands, method='ot').cpu().numpy()[0]
    idx = [0, 1, 2099, 4199]
    want = np.array([orc.get_similarity(queries[0], cands[i]) for i in idx], dtype=np.float32)
    np.testing.assert_allclose(got[idx], want, atol=TOL, rtol=0)
    l2 = amd.scorer.score_pool(queries[:1], cands, method='l2max').cpu().numpy()[0]
    want_l2 = [-orc.allpair_masked_dist_l2max(orc.RepLen(queries[0][None].permute(0, 2, 1), [7]),
                                              orc.RepLen(cands[i][None].permute(0, 2, 1), [len(cands[i])])).item() for i in idx]
    np.testing.assert_allclose(l2[idx], want_l2, atol=TOL, rtol=0)
    # one pool of 9 .. 32-row documents against a 14-row query: the 16-row tile kernels
    long_c = [mk(n) for n in torch.randint(9, 33, (600,), generator=g).tolist()]
    long_q = mk(14)
    for method, ref in (('ot', lambda c_: orc.get_similarity(long_q, c_)),
                        ('l2max', lambda c_: -orc.allpair_masked_dist_l2max(orc.RepLen(long_q[None].permute(0, 2, 1), [14]),
                                                                            orc.RepLen(c_[None].permute(0, 2, 1), [len(c_)])).item())):
        got = amd.scorer.score_pool([long_q], long_c, method=method).cpu().numpy()[0]
        np.testing.assert_allclose(got[[0, 299, 599]], [ref(long_c[i]) for i in (0, 299, 599)], atol=TOL, rtol=0)
    # many queries against one pool: the matrix-pipe cost tiles (gram.hip) centre every tile on its first query row
    many_q = [mk(n) for n in (8, 3, 12, 5, 8, 1, 7, 6)]
    for method in ('ot', 'l2max'):
        got = amd.scorer.score_pool(many_q, cands, method=method).cpu().numpy()
        for qi, ci in ((0, 0), (2, 1), (5, 2099), (7, 4199)):
            if method == 'ot':
                want_v = orc.get_similarity(many_q[qi], cands[ci])
            else:
                want_v = -orc.allpair_masked_dist_l2max(orc.RepLen(many_q[qi][None].permute(0, 2, 1), [len(many_q[qi])]),
                                                        orc.RepLen(cands[ci][None].permute(0, 2, 1), [len(cands[ci])])).item()
            assert abs(got[qi, ci] - want_v) < TOL, (method, qi, ci, got[qi, ci], want_v)
    # batched jobs: 8-row documents (fused kernel) and abstracts of up to 20 rows (CHUNK items)
    for cmax in (8, 20, 32):                               # 32: the 16-row tile kernels' record items
        lens2 = torch.randint(1, cmax + 1, (1600,), generator=g).tolist()
        docs = [mk(n) for n in lens2]
        pools = [docs[:800], docs[800:]]
        ranked = amd.scorer.rank_pools(queries, pools, k=5)
        for j in range(2):
            for pid, s in ranked[j][:3]:
                assert abs(s - orc.get_similarity(queries[j], pools[j][pid])) < TOL, (cmax, j, pid)


def test_pool_batch_reused_with_longer_queries_grows_its_workspace():
    """RepStore.pool_batch advertises re-use of the cached batch for another facet: a later call whose queries are LONGER than
    the candidates (crossing an 8-row boundary of the pair slots) needs more workspace than the first call sized
    ('workspace too small'); the size is asked for on every call"""
    from aspire_amd import scorer
    from aspire_amd.repstore import RepStore
    from oracle import aspire_oracle as orc
    g = torch.Generator().manual_seed(3)
    store = RepStore({f'p{i}': torch.randn(int(n), 768, generator=g).numpy() for i, n in enumerate(torch.randint(1, 8, (60,), generator=g))})
    store.to_device()
    pools = [[f'p{i}' for i in range(0, 30)], [f'p{i}' for i in range(20, 60)]]
    batch = store.pool_batch(pools)
    short = [torch.randn(3, 768, generator=g).numpy(), torch.randn(5, 768, generator=g).numpy()]
    long_ = [torch.randn(20, 768, generator=g).numpy(), torch.randn(11, 768, generator=g).numpy()]
    for queries in (short, long_, short):
        for method in ('ot', 'l2max'):
            ranked = scorer.rank_pool_batch(queries, batch, method=method)
            assert store.pool_batch(pools) is batch
            for q, pids, r in zip(queries, pools, ranked):
                if method == 'ot':
                    want = [orc.get_similarity(torch.as_tensor(q), torch.as_tensor(store.get(p))) for p in pids]
                    got = dict(r)
                    np.testing.assert_allclose([got[p] for p in pids], want, atol=1e-4, rtol=0)
                assert sorted(p for p, _ in r) == sorted(pids)


def test_repstore_chunked_upload_and_planes():
    """to_device through the pinned staging buffer (chunks smaller than the store, a document never split) gives the rows of the
    one-shot upload; planes=True prepares the fp16 planes next to them"""
    from aspire_amd.repstore import RepStore
    g = torch.Generator().manual_seed(4)
    store = RepStore({f'p{i}': torch.randn(int(n), 768, generator=g).numpy() for i, n in enumerate(torch.randint(1, 13, (300,), generator=g))})
    store.to_device(chunk_rows=100, planes=True)
    want = np.concatenate([store.get(p) for p in store.pid2reps], 0)
    assert np.array_equal(store._dev_rows.cpu().numpy(), want)
    pool = store.pool([f'p{i}' for i in (5, 250, 17)])
    assert pool.repset.planes is not None and pool.repset.planes.c.total_rows == want.shape[0]
    s, n = store._dev_index['p250']
    assert np.array_equal(store._dev_rows[s:s + n].cpu().numpy(), store.get('p250'))
