"""A12 beyond one 4096-key chunk: the reference sorts the WHOLE pool (evaluate.py:76, pp_gen_nearest.py:266,339), so
topk_desc(k = C) must work for any pool size -- sorted 4096-key chunks + merge passes -- and give exactly Python's stable
sorted(..., reverse=True): ties by ascending candidate index, -0.0 == +0.0, +-inf in place."""
import numpy as np
import pytest
import torch

from oracle import aspire_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    from aspire_amd import ops, scorer, _lib
    assert torch.cuda.is_available()
    return type('NS', (), dict(ops=ops, scorer=scorer, lib=_lib))


def _scores(qn, cn, seed):
    g = torch.Generator().manual_seed(seed)
    s = torch.randn(qn, cn, generator=g)
    s[:, ::7] = 0.25                       # many exact ties
    s[:, 5::1001] = -0.0
    s[:, 6::1001] = 0.0
    if cn > 10:
        s[0, 3] = float('inf')
        s[0, cn - 2] = float('-inf')
    s[-1, :] = torch.round(s[-1, :] * 4) / 4     # a query whose scores take ~30 distinct values
    return s


def _expect(row, k):
    return np.argsort(-row.numpy().astype(np.float64), kind='stable')[:k]


@pytest.mark.parametrize('qn,cn,k', [(2, 4097, 4097), (3, 8192, 8192), (2, 50000, 50000), (1, 1000000, 1000000),
                                     (3, 9001, 5000), (2, 70000, 1024), (1, 300000, 10000), (2, 12289, 20000)])
def test_full_sort_matches_python_stable_sort(amd, qn, cn, k):
    s = _scores(qn, cn, cn % 97)
    ts, ti = amd.ops.topk_desc(s.cuda(), k, idx_base=7)
    ts, ti = ts.cpu(), ti.cpu()
    kk = min(k, cn)
    for q in range(qn):
        order = _expect(s[q], kk)
        assert np.array_equal(ti[q, :kk].numpy(), order + 7), q
        assert torch.equal(ts[q, :kk], s[q][order]), q
        assert torch.all(ti[q, kk:] == -1) and torch.all(ts[q, kk:] == float('-inf'))
    if cn <= 50000:        # the oracle's own Python sort on the first query
        assert ti[0, :kk].tolist() == [7 + i for i in orc.rank_descending(s[0].tolist())[:kk]]


def test_full_sort_key_form_and_shard_merge_of_long_lists(amd):
    """keys of a full sort carry global indices: two shards' sorted key lists, concatenated and sorted, are the un-sharded order"""
    s = _scores(2, 12000, 5)
    k = 6000
    k0 = amd.ops.topk_keys(s[:, :6000].contiguous().cuda(), k, idx_base=0)
    k1 = amd.ops.topk_keys(s[:, 6000:].contiguous().cuda(), k, idx_base=6000)
    both = torch.cat([k0, k1], 1).cpu().numpy().astype(np.uint64)
    for q in range(2):
        merged = np.sort(both[q])[::-1]
        idx = (0xFFFFFFFF - (merged & np.uint64(0xFFFFFFFF))).astype(np.int64)
        assert np.array_equal(idx, _expect(s[q], 12000))


def test_evaluate_score_on_a_pool_beyond_4096(amd, tmp_path):
    """evaluate.py's score step ranks the whole pool (k = pool size): a 5000-candidate pool (TRECCOVID-sized pools hold
    ~9k) was refused in round 1"""
    from aspire_amd import evaluate as ev
    from aspire_amd.repstore import RepStore
    g = torch.Generator().manual_seed(9)
    n = 5000
    pids = [f'p{i}' for i in range(n + 1)]
    reps = torch.randn(n + 1, 3, 768, generator=g)
    reps[7] = reps[5]                      # two identical candidates: an exact tie, pool order decides
    store = RepStore({p: reps[i].numpy() for i, p in enumerate(pids)})
    test_pool = {'p0': {'cands': pids[1:]}}
    res = ev.score(str(tmp_path), test_pool, store, facet=None, method='ot', schedule='pair')
    ranked = [c for c, _ in res['p0']]
    assert sorted(ranked) == sorted(pids[1:]) and len(ranked) == n
    dists = np.array([s for _, s in res['p0']])          # evaluate.py:77 stores -similarity
    assert np.all(dists[:-1] <= dists[1:])
    assert ranked.index('p5') + 1 == ranked.index('p7')
    scores = amd.scorer.score_pool([reps[0]], [reps[i] for i in range(1, n + 1)], method='ot', schedule='pair')[0].cpu()
    assert ranked == [pids[1 + i] for i in _expect(scores, n)]
    # ... and the other scoring methods take the generic score-then-sort path
    r2 = amd.scorer.rank_pool([reps[0]], [reps[i] for i in range(1, n + 1)], method='l2max')[0]
    s2 = amd.scorer.score_pool([reps[0]], [reps[i] for i in range(1, n + 1)], method='l2max')[0].cpu()
    assert [i for i, _ in r2] == _expect(s2, n).tolist()
