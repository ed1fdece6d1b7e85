"""A resident store beyond 2^31 floats (config 5 at N <= 2 GPUs keeps 500 k - 1 M documents x 12 rows = 4.6 - 9.2 G floats on one
GPU; 36.9 GB fit an MI355X seven times): 3 M rows x 768 = 2.3 G floats = 9.2 GB of fp32 rows (+ the same again as fp16 planes).
Every kernel family that indexes the row matrix is driven at the HIGHEST row indices and spot-checked against the oracle:
the fused / streaming kernels (few queries), the plane tiles and the bf16x3 tiles (many queries), pools given as index lists
into the store.  Reference arithmetic: src/learning/facetid_models/pair_distances.py:21-92, 138-186."""
import numpy as np
import pytest
import torch

from oracle import aspire_oracle as orc

pytestmark = pytest.mark.gpu
N_DOCS, S = 375_000, 8          # 3 000 000 rows


def _l2max_oracle(q, c):
    return -orc.allpair_masked_dist_l2max(orc.RepLen(q[None].permute(0, 2, 1), [len(q)]),
                                          orc.RepLen(c[None].permute(0, 2, 1), [len(c)])).item()


@pytest.fixture(scope='module')
def store():
    from aspire_amd import ops
    if torch.cuda.get_device_properties(0).total_memory < 40 * 2 ** 30:
        pytest.skip('needs ~20 GB of HBM')
    rows = torch.empty(N_DOCS * S, 768, device='cuda')
    assert rows.numel() > 2 ** 31
    g = torch.Generator(device='cuda').manual_seed(7)
    for lo in range(0, N_DOCS * S, 1 << 18):
        rows[lo:lo + (1 << 18)].normal_(generator=g)
    start = (torch.arange(N_DOCS, device='cuda', dtype=torch.int32) * S).contiguous()
    lens = torch.full((N_DOCS,), S, device='cuda', dtype=torch.int32)
    c = ops.DeviceRepSet(rows, start, lens, ext=0, max_len=S, lens_host=None)
    yield c
    del rows
    torch.cuda.empty_cache()


def _docs(store, idx):
    s = store.start[torch.as_tensor(idx, device='cuda')].tolist()
    return [store.rows[a:a + S].cpu() for a in s]


def test_scores_at_the_highest_rows(store):
    from aspire_amd import ops
    from aspire_amd._lib import pinned
    g = torch.Generator().manual_seed(8)
    spots = [0, 1, N_DOCS // 2, 2 ** 31 // (768 * S) + 3, N_DOCS - 2, N_DOCS - 1]      # incl. the first document past 2^31 floats
    cand = _docs(store, spots)
    # one query (the streaming kernels) and 40 queries (the matrix-pipe tiles)
    for nq in (1, 40):
        queries = [torch.randn(S, 768, generator=g) for _ in range(nq)]
        q = ops.DeviceRepSet.from_list(queries)
        ot = -ops.ot_sinkhorn(q, store).view(nq, N_DOCS)[:, spots].cpu().numpy()
        l2 = ops.l2max_scores(q, store).view(nq, N_DOCS)[:, spots].cpu().numpy()
        for qi in (0, nq - 1):
            want_ot = [orc.get_similarity(queries[qi], c) for c in cand]
            want_l2 = [_l2max_oracle(queries[qi], c) for c in cand]
            np.testing.assert_allclose(ot[qi], want_ot, atol=1e-4, rtol=0)
            np.testing.assert_allclose(l2[qi], want_l2, atol=1e-4, rtol=0)
    # the same 40 queries on the fp16 planes of the whole store (+ 9.2 GB) against the tiles that read the fp32 rows
    store.prepare_planes()
    assert store.planes.c.total_rows == N_DOCS * S
    q = ops.DeviceRepSet.from_list(queries)
    with pinned(COST_PATH='mfma'):
        lp = ops.l2max_scores(q, store).view(nq, N_DOCS)
        assert q.planes is not None
        np.testing.assert_allclose(lp[:, spots].cpu().numpy(), l2, atol=2e-5, rtol=0)
        top = lp.argmax(1).cpu()
    with pinned(COST_PATH='mfma', GEMM='bf16x3'):
        assert torch.equal(ops.l2max_scores(q, store).view(nq, N_DOCS).argmax(1).cpu(), top)
    store.drop_planes()


def test_pool_of_index_lists_into_the_big_store(store):
    """a pool = documents picked all over the store, the last ones included (RepStore.pool's index lists), ranked in one call"""
    from aspire_amd import ops, _lib
    g = torch.Generator().manual_seed(9)
    pick = torch.cat([torch.randint(0, N_DOCS, (997,), generator=g), torch.tensor([N_DOCS - 1, N_DOCS - 2, 0])])
    idx = pick.cuda()
    pool = ops.DeviceRepSet(store.rows, store.start[idx].contiguous(), store.len[idx].contiguous(), ext=0, max_len=S)
    queries = [torch.randn(S, 768, generator=g)]
    q = ops.DeviceRepSet.from_list(queries)
    scores, top_s, top_i = ops.ot_rank(q, pool, 100, want=_lib.OT_SIMILARITY)
    s = scores.view(-1).cpu()
    assert top_i[0].tolist() == np.argsort(-s.numpy().astype(np.float64), kind='stable')[:100].tolist()
    for j in (997, 998, 999, 5):
        want = orc.get_similarity(queries[0], store.rows[int(pick[j]) * S:int(pick[j]) * S + S].cpu())
        assert abs(float(s[j]) - want) < 1e-4
