"""Sibling aggregations l2top2 / l2attention through the C ABI, against the golden vectors
the reference's own functions produced (tests/golden/siblings.npz) and against the oracle on CSR pools."""
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


@pytest.fixture(scope='module')
def z(golden_dir):
    return np.load(os.path.join(golden_dir, 'siblings.npz'))


def _tuples(amd, z, name):
    q, c = torch.from_numpy(z[f'{name}_q']), torch.from_numpy(z[f'{name}_c'])
    qt = amd.pd.rep_len_tup(embed=q.permute(0, 2, 1), abs_lens=z[f'{name}_qlens'].tolist())
    ct = amd.pd.rep_len_tup(embed=c.permute(0, 2, 1), abs_lens=z[f'{name}_clens'].tolist())
    return qt, ct


@pytest.mark.parametrize('name', ['s8', 'rag', 'big', 'dup'])
def test_l2top2_matches_reference_vectors(amd, z, name):
    qt, ct = _tuples(amd, z, name)
    np.testing.assert_allclose(amd.pd.allpair_masked_dist_l2topk(qt, ct).numpy(), z[f'{name}_top2_dist'], atol=TOL, rtol=0)
    sims, pair = amd.pd.allpair_masked_dist_l2topk(qt, ct, return_pair_sims=True)
    np.testing.assert_allclose(sims.numpy(), z[f'{name}_top2_sims'], atol=TOL, rtol=0)
    # masked entries are -cdist - 1e9, where one fp32 ulp is 64: compare them to that grain
    np.testing.assert_allclose(pair.numpy(), z[f'{name}_top2_pair'], atol=TOL, rtol=1e-7)


@pytest.mark.parametrize('name', ['s8', 'rag', 'big', 'dup'])
@pytest.mark.parametrize('temp,t', [(1.0, 't1'), (0.2, 't02')])
def test_l2attention_matches_reference_vectors(amd, z, name, temp, t):
    qt, ct = _tuples(amd, z, name)
    att = amd.pd.AllPairMaskedAttention({'cdatt_sm_temp': temp})
    np.testing.assert_allclose(att.compute_distance(qt, ct).numpy(), z[f'{name}_att_{t}_dist'], atol=TOL, rtol=0)
    ds, (ps, sm, ms) = att.compute_distance(qt, ct, return_pair_sims=True)
    np.testing.assert_allclose(ds.numpy(), z[f'{name}_att_{t}_sims'], atol=TOL, rtol=0)
    np.testing.assert_allclose(ps.numpy(), z[f'{name}_att_{t}_pair'], atol=TOL, rtol=0)
    np.testing.assert_allclose(sm.numpy(), z[f'{name}_att_{t}_softmax'], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(ms.numpy(), z[f'{name}_att_{t}_masked'], atol=TOL, rtol=1e-4)


def test_one_valid_entry_picks_a_masked_one(amd):
    """1 x 1 valid block inside a padded 3 x 4 one: torch.topk's second pick is a masked entry (~ -1e9)."""
    g = torch.Generator().manual_seed(3)
    q, c = torch.randn(1, 3, 768, generator=g), torch.randn(1, 4, 768, generator=g)
    q[0, 1:] = 0
    c[0, 1:] = 0
    qt = amd.pd.rep_len_tup(embed=q.permute(0, 2, 1), abs_lens=[1])
    ct = amd.pd.rep_len_tup(embed=c.permute(0, 2, 1), abs_lens=[1])
    want = orc.allpair_masked_dist_l2topk(orc.RepLen(q.permute(0, 2, 1), [1]), orc.RepLen(c.permute(0, 2, 1), [1]))
    got = amd.pd.allpair_masked_dist_l2topk(qt, ct)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-7)


def test_pool_scoring_and_caching_score(amd):
    """CSR pool (no padded extents), all-against-all; and the caching_score branches (disent_models.py:238-245)."""
    g = torch.Generator().manual_seed(9)
    queries = [torch.randn(n, 768, generator=g) for n in (8, 3)]
    cands = [torch.randn(int(n), 768, generator=g) for n in (8, 5, 2, 12, 7, 20, 3)]
    for method, fn in (('l2top2', lambda qt, ct: -orc.allpair_masked_dist_l2topk(qt, ct)),
                       ('l2attention', lambda qt, ct: -orc.AllPairMaskedAttention({'cdatt_sm_temp': 0.5}).compute_distance(qt, ct))):
        got = amd.scorer.score_pool(queries, cands, method=method, hparams={'cdatt_sm_temp': 0.5}).cpu().numpy()
        want = np.array([[fn(orc.RepLen(x[None].permute(0, 2, 1), [len(x)]), orc.RepLen(y[None].permute(0, 2, 1), [len(y)])).item()
                          for y in cands] for x in queries], dtype=np.float32)
        np.testing.assert_allclose(got, want, atol=TOL, rtol=0)
    qd = {'sent_reps': queries[0].numpy()}
    cds = [{'sent_reps': y.numpy()} for y in cands]
    for agg in ('l2top2', 'l2attention'):
        ret = amd.scorer.caching_score(qd, cds, score_agg_type=agg)
        assert ret['batch_scores'].shape == (len(cands),)
        assert len(ret['pair_scores']) == len(cands)
        first = ret['pair_scores'][3]
        if agg == 'l2attention':
            assert len(first) == 3 and first[1].shape == (8, 12)
            np.testing.assert_allclose(first[1].sum(), 1.0, atol=1e-5)
        else:
            assert first.shape == (8, 12)
    with pytest.raises(AssertionError):
        amd.ops.l2agg_scores(amd.ops.DeviceRepSet.from_list(queries), amd.ops.DeviceRepSet.from_list(cands), 7)


def test_a_single_padded_entry_raises_like_torch_topk(amd):
    """pair_distances.py:333: torch.topk(k=2) over the [batch, q_max_sents * c_max_sents] view raises RuntimeError when that view has
    one column (both sides padded to one sentence)."""
    g = torch.Generator().manual_seed(4)
    q, c = torch.randn(2, 1, 768, generator=g), torch.randn(2, 1, 768, generator=g)
    qt = amd.pd.rep_len_tup(embed=q.permute(0, 2, 1), abs_lens=[1, 1])
    ct = amd.pd.rep_len_tup(embed=c.permute(0, 2, 1), abs_lens=[1, 1])
    with pytest.raises(RuntimeError):
        amd.pd.allpair_masked_dist_l2topk(qt, ct)
    with pytest.raises(RuntimeError):
        from oracle import aspire_oracle as orc
        orc.allpair_masked_dist_l2topk(orc.RepLen(q.permute(0, 2, 1), [1, 1]), orc.RepLen(c.permute(0, 2, 1), [1, 1]))


def test_siblings_on_documents_of_33_to_128_rows(amd):
    """beyond the tile kernels' 32 rows (the one-workgroup-per-pair kernel, generic.hip): padded reference tensors with the pair
    outputs, and a CSR pool that mixes short and long documents -- as otAspire / l2max have accepted since round 2"""
    g = torch.Generator().manual_seed(13)
    # padded [B, D, S] tensors, 40 x 70 rows, ragged lengths (torch.cdist's matmul formula: both sides beyond 25 rows)
    q, c = torch.randn(3, 40, 768, generator=g), torch.randn(3, 70, 768, generator=g)
    qlens, clens = [40, 33, 7], [70, 1, 64]
    for b in range(3):
        q[b, qlens[b]:] = 0
        c[b, clens[b]:] = 0
    qt, ct = amd.pd.rep_len_tup(q.permute(0, 2, 1), qlens), amd.pd.rep_len_tup(c.permute(0, 2, 1), clens)
    oq, oc = orc.RepLen(q.permute(0, 2, 1), qlens), orc.RepLen(c.permute(0, 2, 1), clens)
    sims, pair = amd.pd.allpair_masked_dist_l2topk(qt, ct, return_pair_sims=True)
    wsims, wpair = orc.allpair_masked_dist_l2topk(oq, oc, return_pair_sims=True)
    np.testing.assert_allclose(sims.numpy(), wsims.numpy(), atol=TOL, rtol=0)
    np.testing.assert_allclose(pair.numpy(), wpair.numpy(), atol=TOL, rtol=1e-7)
    att, oatt = amd.pd.AllPairMaskedAttention({'cdatt_sm_temp': 0.7}), orc.AllPairMaskedAttention({'cdatt_sm_temp': 0.7})
    ds, (ps, sm, ms) = att.compute_distance(qt, ct, return_pair_sims=True)
    wds, (wps, wsm, wms) = oatt.compute_distance(oq, oc, return_pair_sims=True)
    np.testing.assert_allclose(ds.numpy(), wds.numpy(), atol=TOL, rtol=0)
    np.testing.assert_allclose(ps.numpy(), wps.numpy(), atol=TOL, rtol=0)
    np.testing.assert_allclose(sm.numpy(), wsm.numpy(), atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(ms.numpy(), wms.numpy(), atol=TOL, rtol=1e-4)
    # CSR pool: 128-row and 33-row documents among short ones
    queries = [torch.randn(n, 768, generator=g) for n in (8, 50)]
    cands = [torch.randn(int(n), 768, generator=g) for n in (8, 128, 2, 33, 20, 90)]
    for method, fn in (('l2top2', lambda a, b: -orc.allpair_masked_dist_l2topk(a, b)),
                       ('l2attention', lambda a, b: -orc.AllPairMaskedAttention({'cdatt_sm_temp': 0.5}).compute_distance(a, b))):
        got = amd.scorer.score_pool(queries, cands, method=method, hparams={'cdatt_sm_temp': 0.5}).cpu().numpy()
        want = np.array([[fn(orc.RepLen(x[None].permute(0, 2, 1), [len(x)]), orc.RepLen(y[None].permute(0, 2, 1), [len(y)])).item()
                          for y in cands] for x in queries], dtype=np.float32)
        np.testing.assert_allclose(got, want, atol=TOL, rtol=0)
