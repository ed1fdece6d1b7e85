"""Small parity gaps closed in round 2:
  (a) torch.cdist's formula is chosen from the extents of the padded GROUP caching_score hands it: a group of 64 that holds a
      document of more than 25 rows scores ALL its pairs with the matmul expansion (pair_distances.py:49, 167 on
      disent_models.py:269-297's padded tensors)
  (b) caching_score's document-level CLS term (disent_models.py:305-307, abs_loss_prop > 0)."""
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


def _pool_with_long_docs(seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(1, 13, (200,), generator=g).tolist()
    lens[70] = 27          # group 1 (candidates 64..127) holds a 27-row document
    lens[130] = 26         # group 2 as well; groups 0 and 3 stay on the direct formula
    return [torch.randn(n, 768, generator=g) for n in lens]


def test_cdist_formula_follows_the_group_extents_l2max(amd):
    cands = _pool_with_long_docs(1)
    g = torch.Generator().manual_seed(2)
    query = torch.randn(9, 768, generator=g)
    got = amd.scorer.score_pool([query], cands, method='l2max', schedule='batch').cpu().numpy()[0]
    want = np.array(orc.rank_pool_caching(query.numpy(), [c.numpy() for c in cands], score_agg_type='l2max'), dtype=np.float32)
    # two fp32 evaluations of the SAME expansion differ by up to ~2e-5 (it cancels |x|^2 + |y|^2 ~ 1500 against 2 x.y)
    np.testing.assert_allclose(got, want, atol=4e-5, rtol=0)
    per_pair = amd.scorer.score_pool([query], cands, method='l2max', schedule='pair').cpu().numpy()[0]
    np.testing.assert_allclose(per_pair[:64], want[:64], atol=4e-5, rtol=0)                    # groups without a long document: the direct formula either way
    ranked = amd.scorer.rank_pool([query], cands, method='l2max', schedule='batch')[0]
    assert [i for i, _ in ranked] == np.argsort(-got.astype(np.float64), kind='stable').tolist()


def test_cdist_formula_follows_the_group_extents_ot(amd):
    cands = _pool_with_long_docs(3)
    g = torch.Generator().manual_seed(4)
    query = torch.randn(8, 768, generator=g)
    got = amd.scorer.score_pool([query], cands, method='ot', schedule='batch').cpu().numpy()[0]
    want = np.array(orc.rank_pool_caching(query.numpy(), [c.numpy() for c in cands]), dtype=np.float32)
    import plan_sim_floor
    truth = np.array(orc.rank_pool_caching(query.numpy(), [c.numpy() for c in cands], dtype=torch.float64))
    plan_sim_floor.check(got, want, truth, 'long-document groups')       # plan-weighted similarity: against float64, per case
    # the marginals are where the cdist formula enters: compare them through the drop-in caching_score on the long group
    qd = {'sent_reps': query.numpy()}
    cds = [{'sent_reps': c.numpy()} for c in cands[64:128]]
    ret = amd.scorer.caching_score(qd, cds)
    _, wextra = orc.caching_score(query.numpy(), [c.numpy() for c in cands[64:128]])
    for i in range(64):
        np.testing.assert_allclose(ret['pair_scores'][i][2], wextra[i][2], atol=4e-5, rtol=0)     # masked -cdist, mm formula
        np.testing.assert_allclose(ret['pair_scores'][i][0], wextra[i][0], atol=2e-5, rtol=0)     # query marginals
    ranked = amd.scorer.rank_pool([query], cands, method='ot', schedule='batch')[0]
    assert [i for i, _ in ranked] == np.argsort(-got.astype(np.float64), kind='stable').tolist()
    # a long QUERY puts every group on the expansion
    long_q = torch.randn(26, 768, generator=g)
    got = amd.scorer.score_pool([long_q, query], cands[:64], method='l2max', schedule='batch').cpu().numpy()
    for r, qq in enumerate((long_q, query)):
        want = np.array(orc.rank_pool_caching(qq.numpy(), [c.numpy() for c in cands[:64]], score_agg_type='l2max'), dtype=np.float32)
        np.testing.assert_allclose(got[r], want, atol=4e-5, rtol=0)


def test_caching_score_cls_term(amd):
    g = torch.Generator().manual_seed(5)
    q = {'sent_reps': torch.randn(6, 768, generator=g).numpy(), 'doc_cls_reps': torch.randn(768, generator=g).numpy()}
    cds = [{'sent_reps': torch.randn(int(n), 768, generator=g).numpy(), 'doc_cls_reps': torch.randn(768, generator=g).numpy()}
           for n in (8, 3, 5, 1, 7)]
    cds[2]['doc_cls_reps'] = q['doc_cls_reps'].copy()          # identical CLS reps: the distance is sqrt(768) * 1e-6, not 0
    for agg in ('l2max', 'l2wasserstein'):
        ret = amd.scorer.caching_score(q, cds, score_agg_type=agg, sent_loss_prop=0.7, abs_loss_prop=0.3)
        want, _ = orc.caching_score(q['sent_reps'], [d['sent_reps'] for d in cds], score_agg_type=agg, sent_loss_prop=0.7,
                                    abs_loss_prop=0.3, query_cls_rep=q['doc_cls_reps'], cand_cls_reps=[d['doc_cls_reps'] for d in cds])
        np.testing.assert_allclose(ret['batch_scores'], want, atol=1e-4 if agg == 'l2max' else 1e-2, rtol=0)
        base = amd.scorer.caching_score(q, cds, score_agg_type=agg, sent_loss_prop=0.7)['batch_scores']
        cls_part = ret['batch_scores'] - base
        ref = -0.3 * torch.nn.functional.pairwise_distance(torch.from_numpy(np.vstack([q['doc_cls_reps']] * 5)),
                                                           torch.from_numpy(np.vstack([d['doc_cls_reps'] for d in cds]))).numpy()
        np.testing.assert_allclose(cls_part, ref, atol=2e-5, rtol=0)
    d = amd.ops.cls_l2(torch.from_numpy(q['doc_cls_reps'])[None].cuda(), torch.from_numpy(np.vstack([c['doc_cls_reps'] for c in cds])).cuda(),
                       pairing=amd.lib.PAIR_CROSS).cpu().numpy()
    assert d[2] == pytest.approx(np.sqrt(768) * 1e-6, rel=1e-3)
