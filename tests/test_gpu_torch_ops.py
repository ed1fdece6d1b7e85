"""GPU: the torch.library ops give what the ctypes front end gives -- they ARE the same C-ABI calls --
pass torch.library.opcheck, and agree with the oracle on a small case."""
import numpy as np
import pytest
import torch

from oracle import aspire_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    import aspire_amd.torch_ops as to
    from aspire_amd import ops, _lib
    assert torch.cuda.is_available()
    return type('NS', (), dict(ops=ops, lib=_lib, to=to))


def _padded(seed, lens, s):
    g = torch.Generator().manual_seed(seed)
    t = torch.zeros(len(lens), s, 768)
    for i, n in enumerate(lens):
        t[i, :n] = torch.randn(n, 768, generator=g)
    return t.cuda(), torch.tensor(lens, dtype=torch.int32).cuda()


def test_scoring_ops_match_oracle_and_opcheck(amd):
    q, ql = _padded(1, [8, 5, 3], 8)
    c, cl = _padded(2, [6, 8, 1], 8)
    qt = orc.RepLen(q.cpu().permute(0, 2, 1), ql.tolist())
    ct = orc.RepLen(c.cpu().permute(0, 2, 1), cl.tolist())
    got = torch.ops.aspire.l2max_scores(q, ql, c, cl, True).cpu().numpy()
    np.testing.assert_allclose(got, -orc.allpair_masked_dist_l2max(qt, ct).numpy(), atol=1e-4, rtol=0)
    cross = torch.ops.aspire.l2max_scores(q, ql, c, cl, False).view(3, 3).cpu().numpy()
    np.testing.assert_allclose(np.diag(cross), got, atol=1e-5, rtol=0)
    w = orc.AllPairMaskedWasserstein({})
    sims, extra = w.compute_distance(qt, ct, return_pair_sims=True)
    out = torch.ops.aspire.ot_sinkhorn_scores(q, ql, c, cl, 0.05, 0.9, 1.0, 3, 1, True, True)
    np.testing.assert_allclose(out[1].cpu().numpy(), extra[0].numpy(), atol=1e-5, rtol=0)      # query_distr
    np.testing.assert_allclose(out[2].cpu().numpy(), extra[1].numpy(), atol=1e-5, rtol=0)      # cand_distr
    np.testing.assert_allclose(out[3].cpu().numpy(), extra[2].numpy(), atol=1e-4, rtol=0)      # pair_sims
    np.testing.assert_allclose(out[0].cpu().numpy(), sims.numpy(), atol=1e-2, rtol=0)           # plan-weighted similarity (fp32 conditioning)
    dist = torch.ops.aspire.ot_sinkhorn_scores(q, ql, c, cl, 0.05, 0.9, 1.0, 3, 0, True, False)[0].cpu().numpy()
    np.testing.assert_allclose(dist, w.compute_distance(qt, ct).numpy(), atol=1e-4, rtol=0)
    torch.library.opcheck(torch.ops.aspire.l2max_scores, (q, ql, c, cl, False), test_utils=('test_schema', 'test_faketensor'))
    torch.library.opcheck(torch.ops.aspire.ot_sinkhorn_scores, (q, ql, c, cl, 0.05, 0.9, 1.0, 0, 2, False, True),
                          test_utils=('test_schema', 'test_faketensor'))


def test_pool_rank_and_batch_ops(amd):
    g = torch.Generator().manual_seed(4)
    hidden = torch.randn(2, 20, 768, generator=g).cuda()
    from aspire_amd.batch_prep import spans_to_csr
    idxs = [[list(range(3, 9)), list(range(9, 19))], [list(range(2, 7))]]
    tok_idx, span_off = spans_to_csr(idxs, 2)
    cls, sent = torch.ops.aspire.span_mean_pool(hidden, tok_idx.cuda(), span_off.cuda(), 2)
    wcls, wsent = orc.span_mean_pool(hidden.cpu(), idxs, [2, 1])
    assert torch.allclose(sent.cpu(), wsent, atol=1e-5) and torch.equal(cls.cpu(), wcls)
    torch.library.opcheck(torch.ops.aspire.span_mean_pool, (hidden, tok_idx.cuda(), span_off.cuda(), 2),
                          test_utils=('test_schema', 'test_faketensor'))
    s = torch.randn(3, 6000, generator=g)
    s[:, ::5] = 0.5
    ts, ti = torch.ops.aspire.topk_desc(s.cuda(), 6000, 0)
    for r in range(3):
        assert ti[r].cpu().tolist() == np.argsort(-s[r].numpy().astype(np.float64), kind='stable').tolist()
    k0 = torch.ops.aspire.topk_keys(s[:, :3000].contiguous().cuda(), 50, 0)
    k1 = torch.ops.aspire.topk_keys(s[:, 3000:].contiguous().cuda(), 50, 3000)
    ms, mi = torch.ops.aspire.topk_merge(torch.stack([k0, k1]), 50)
    assert torch.equal(mi, ti[:, :50]) and torch.equal(ms, ts[:, :50])
    torch.library.opcheck(torch.ops.aspire.topk_desc, (s.cuda(), 100, 0), test_utils=('test_schema', 'test_faketensor'))
    # CSR ops against the ctypes front end (the same C-ABI calls)
    docs = [torch.randn(int(n), 768, generator=g) for n in (8, 3, 5, 8, 1, 7, 2)]
    qs = amd.ops.DeviceRepSet.from_list(docs[:2])
    cs = amd.ops.DeviceRepSet.from_list(docs[2:])
    a = torch.ops.aspire.ot_scores_csr(qs.rows, qs.start, qs.len, qs.max_len, cs.rows, cs.start, cs.len, cs.max_len, 0.05, 0.9, 1.0, 0, 2)
    assert torch.equal(a, amd.ops.ot_sinkhorn(qs, cs, want=amd.lib.OT_SIMILARITY))
    b = torch.ops.aspire.l2max_scores_csr(qs.rows, qs.start, qs.len, qs.max_len, cs.rows, cs.start, cs.len, cs.max_len)
    assert torch.equal(b, amd.ops.l2max_scores(qs, cs))
    job_off = torch.tensor([0, 2, 5], dtype=torch.int32).cuda()
    sc, t_s, t_i = torch.ops.aspire.ot_rank_batch(qs.rows, qs.start, qs.len, qs.max_len, cs.rows, cs.start, cs.len, cs.max_len, job_off,
                                                  3, 3, 0.05, 0.9, 1.0, 2)
    assert torch.equal(sc[:2], a.view(2, 5)[0, :2]) and torch.equal(sc[2:], a.view(2, 5)[1, 2:])
    assert t_i[0, 2].item() == -1 and sorted(t_i[1].tolist()) == [0, 1, 2]


def test_bert_encoder_op_matches_the_encoder_class(amd):
    from transformers import BertConfig, BertModel
    from aspire_amd.encoder import HipBertEncoder
    torch.manual_seed(0)
    m = BertModel(BertConfig(vocab_size=500, num_hidden_layers=2, max_position_embeddings=64), add_pooling_layer=False).eval()
    enc = HipBertEncoder(m)
    tok = torch.randint(0, 500, (2, 33))
    seg = torch.zeros_like(tok)
    mask = torch.ones_like(tok)
    mask[1, 20:] = 0
    want = enc.forward_hidden(tok, seg, mask)
    sd = m.state_dict()
    w = [sd['embeddings.word_embeddings.weight'], sd['embeddings.position_embeddings.weight'], sd['embeddings.token_type_embeddings.weight'],
         sd['embeddings.LayerNorm.weight'], sd['embeddings.LayerNorm.bias']]
    for i in range(2):
        p = f'encoder.layer.{i}.'
        a = p + 'attention.self.'
        w += [torch.cat([sd[a + 'query.weight'], sd[a + 'key.weight'], sd[a + 'value.weight']], 0),
              torch.cat([sd[a + 'query.bias'], sd[a + 'key.bias'], sd[a + 'value.bias']], 0),
              sd[p + 'attention.output.dense.weight'], sd[p + 'attention.output.dense.bias'],
              sd[p + 'attention.output.LayerNorm.weight'], sd[p + 'attention.output.LayerNorm.bias'],
              sd[p + 'intermediate.dense.weight'], sd[p + 'intermediate.dense.bias'],
              sd[p + 'output.dense.weight'], sd[p + 'output.dense.bias'],
              sd[p + 'output.LayerNorm.weight'], sd[p + 'output.LayerNorm.bias']]
    w = [t.detach().float().cuda().contiguous() for t in w]
    got = torch.ops.aspire.bert_encoder_forward(tok.cuda(), seg.cuda(), mask.cuda(), w, 12, 1e-12)
    # the op takes bare weight tensors (no prepared bf16 planes: its GEMMs split their operands on the fly), the class streams
    # pre-split operands: the same six products per term in another launch geometry -- equal to fp32 rounding, and bit-equal once
    # the class is pinned to the on-the-fly form
    assert torch.allclose(got, want, atol=2e-5, rtol=0)
    from aspire_amd._lib import pinned
    with pinned(GEMM='bf16x3'):
        assert torch.equal(got, enc.forward_hidden(tok, seg, mask))
