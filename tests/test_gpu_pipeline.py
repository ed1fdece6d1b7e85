"""The device-resident pipeline: token ids -> BERT forward -> span pooling written straight into a rows + CSR rep store
in HBM (AspireConSent.encode_to_pool) -> scoring -> ranked lists, against the reference's host-out steps
(caching_encode, disent_models.py:344-371; AspireModel.encode, models.py:199-209) and against
HF BertModel + the oracle's pooling + the oracle's OT + Python's stable sort end to end."""
import numpy as np
import pytest
import torch

from oracle import aspire_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _bert(n_layers, seed=0, vocab=3000):
    from transformers import BertConfig, BertModel
    torch.manual_seed(seed)
    cfg = BertConfig(vocab_size=vocab, hidden_size=768, num_hidden_layers=n_layers, num_attention_heads=12,
                     intermediate_size=3072, max_position_embeddings=512)
    m = BertModel(cfg, add_pooling_layer=False).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if 'LayerNorm' in n or n.endswith('.bias'):
                p.add_(0.1 * torch.randn_like(p))
    return m


def _doc_batches(seed, n_docs, batch, vocab, max_sents, l_max=120):
    """synthetic prepare_abstracts outputs: ragged sentence counts and lengths, contiguous spans after a title"""
    g = torch.Generator().manual_seed(seed)
    out = []
    for b0 in range(0, n_docs, batch):
        docs = []
        for _ in range(min(batch, n_docs - b0)):
            ns = int(torch.randint(1, max_sents + 1, (1,), generator=g))
            title = int(torch.randint(2, 8, (1,), generator=g))
            lens = torch.randint(2, max(3, (l_max - title - 2) // ns), (ns,), generator=g).tolist()
            spans, pos = [], 1 + title
            for n in lens:
                spans.append(list(range(pos, pos + n)))
                pos += n
            docs.append((pos + 1, spans))
        seq_lens = [d[0] for d in docs]
        L = max(seq_lens)
        tok = torch.zeros(len(docs), L, dtype=torch.int64)
        mask = torch.zeros_like(tok)
        for i, (n, _) in enumerate(docs):
            tok[i, :n] = torch.randint(5, vocab, (n,), generator=g)
            mask[i, :n] = 1
        bert_batch = {'tokid_tt': tok, 'seg_tt': torch.zeros_like(tok), 'attnmask_tt': mask, 'seq_lens': seq_lens}
        out.append((bert_batch, [len(d[1]) for d in docs], [d[1] for d in docs]))
    return out


def _oracle_reps(model, batches):
    reps = []
    with torch.no_grad():
        for bb, abs_lens, idxs in batches:
            h = model(bb['tokid_tt'], token_type_ids=bb['seg_tt'], attention_mask=bb['attnmask_tt']).last_hidden_state
            _, sent = orc.span_mean_pool(h, idxs, abs_lens)
            reps += [sent[i, :n] for i, n in enumerate(abs_lens)]
    return reps


def test_encode_to_pool_is_forward_without_the_padding():
    from aspire_amd import AspireConSent
    m = _bert(2, seed=1)
    model = AspireConSent(bert_model=m)
    batches = _doc_batches(3, 23, 8, 3000, 9)
    pool, cls = model.encode_to_pool(batches, pids=[f'd{i}' for i in range(23)], want_cls=True, docs_per_forward=None)     # one encoder call per batch, as given
    assert len(pool) == 23 and pool.pids[5] == 'd5'
    rows, start, lens = pool.repset.rows.cpu(), pool.repset.start.cpu().tolist(), pool.repset.len.cpu().tolist()
    assert pool.repset.rows.is_cuda and rows.shape[0] == sum(lens)
    d = 0
    for bb, abs_lens, idxs in batches:
        wcls, wsent = model.forward(bb, abs_lens, idxs)          # the drop-in, host-out call on the same batch
        ce = model.caching_encode({'bert_batch': bb, 'abs_lens': abs_lens, 'senttok_idxs': idxs})
        for i, n in enumerate(abs_lens):
            assert lens[d] == n
            assert torch.equal(rows[start[d]:start[d] + n], wsent[i, :n]), d
            assert torch.equal(cls[d].cpu(), wcls[i])
            assert ce[i]['sent_reps'].shape == (n, 768) and np.array_equal(ce[i]['sent_reps'], wsent[i, :n].numpy())
            assert np.array_equal(ce[i]['doc_cls_reps'], wcls[i].numpy())
            d += 1
    want = _oracle_reps(m, batches)
    for d in range(23):
        np.testing.assert_allclose(rows[start[d]:start[d] + lens[d]].numpy(), want[d].numpy(), atol=TOL, rtol=0)
    # by default consecutive batches are joined into encoder calls of up to 64 documents (the shorter ones' token tensors padded):
    # the same rows -- to rounding, the joined call's row count can select other GEMM kernel forms
    for per in (64, 16):
        joined = model.encode_to_pool(batches, docs_per_forward=per, sort_by_length=False)
        assert torch.equal(joined.repset.start.cpu(), pool.repset.start.cpu())
        np.testing.assert_allclose(joined.repset.rows.cpu().numpy(), rows.numpy(), atol=2e-5, rtol=0)
    # sort_by_length (the default): the documents of all batches regrouped by token length (fewer pad tokens per encoder call); the store keeps
    # the corpus order, CLS rows included
    for per in (64, 8, 5):
        regrouped, rcls = model.encode_to_pool(batches, pids=[f'd{i}' for i in range(23)], want_cls=True, docs_per_forward=per, sort_by_length=True)
        assert regrouped.pids == pool.pids
        assert torch.equal(regrouped.repset.start.cpu(), pool.repset.start.cpu()) and torch.equal(regrouped.repset.len.cpu(), pool.repset.len.cpu())
        np.testing.assert_allclose(regrouped.repset.rows.cpu().numpy(), rows.numpy(), atol=2e-5, rtol=0)
        np.testing.assert_allclose(rcls.cpu().numpy(), cls.cpu().numpy(), atol=2e-5, rtol=0)
    groups, ids = AspireConSent._regroup_by_length(batches, 5)
    assert sorted(i for g in ids for i in g) == list(range(23)) and [len(g) for g in ids] == [5, 5, 5, 5, 3]
    lens_sorted = [n for bb, _, _ in groups for n in bb['seq_lens']]
    assert lens_sorted == sorted(lens_sorted, reverse=True)
    assert all(bb['tokid_tt'].shape[1] == max(bb['seq_lens']) for bb, _, _ in groups)


def test_token_ids_to_ranked_list_end_to_end():
    """ids -> (HIP BERT -> HIP pooling into the resident store -> HIP OT -> HIP rank) vs
    ids -> (HF BertModel -> oracle pooling -> oracle OT -> Python's stable sort)"""
    from aspire_amd import AspireConSent, scorer
    m = _bert(2, seed=7)
    model = AspireConSent(bert_model=m)
    cand_batches = _doc_batches(11, 40, 16, 3000, 8)
    query_batches = _doc_batches(12, 3, 3, 3000, 8)
    pool = model.encode_to_pool(cand_batches)
    qpool = model.encode_to_pool(query_batches)
    qreps = [qpool.repset.rows[s:s + n] for s, n in zip(qpool.repset.start.tolist(), qpool.repset.len.tolist())]
    ranked = scorer.rank_pool(qreps, pool, k=None, method='ot', schedule='pair')
    want_c = _oracle_reps(m, cand_batches)
    want_q = _oracle_reps(m, query_batches)
    for qi in range(3):
        sims = np.array([orc.get_similarity(want_q[qi], c) for c in want_c], dtype=np.float32)
        got = dict(ranked[qi])
        np.testing.assert_allclose(np.array([got[i] for i in range(40)]), sims, atol=2e-4, rtol=0)     # 1e-4 reps through a 1e-4 OT
        order = orc.rank_descending(sims.tolist())
        mine = [i for i, _ in ranked[qi]]
        # identical order wherever the oracle's own scores are separated by more than the tolerance
        for a, b in zip(mine, order):
            assert a == b or abs(sims[a] - sims[b]) < 4e-4, (qi, a, b)
    # tsAspire on the same resident reps
    l2 = scorer.score_pool(qreps, pool, method='l2max').cpu().numpy()
    for qi in range(3):
        ref = [-orc.allpair_masked_dist_l2max(orc.RepLen(want_q[qi][None].permute(0, 2, 1), [len(want_q[qi])]),
                                              orc.RepLen(c[None].permute(0, 2, 1), [len(c)])).item() for c in want_c]
        np.testing.assert_allclose(l2[qi], np.array(ref, dtype=np.float32), atol=2e-4, rtol=0)


def test_encode_drop_in_signature(tmp_path, golden_dir):
    """AspireModel.encode's shape contract (models.py:199-209) with an offline tokenizer: list of [abs_len, 768] tensors"""
    import json, os
    from transformers import BertTokenizer
    from aspire_amd import AspireConSent
    z = json.load(open(os.path.join(golden_dir, 'prep.json')))
    (tmp_path / 'vocab.txt').write_text('\n'.join(z['vocab']) + '\n')
    tok = BertTokenizer(str(tmp_path / 'vocab.txt'), do_lower_case=True)
    model = AspireConSent(bert_model=_bert(1, seed=2, vocab=len(z['vocab'])))
    docs = [z['docs'][0], z['docs'][4], z['docs'][1]]
    reps = model.encode(docs, tok)
    assert len(reps) == 3 and all(r.shape[1] == 768 and r.shape[0] >= 1 for r in reps)
    from aspire_amd import prepare_abstracts
    _, abs_lens, _ = prepare_abstracts(docs, tok)
    assert [r.shape[0] for r in reps] == abs_lens


@pytest.mark.parametrize('b,l', [(32, 256), (2, 502)])
def test_bert_12_layers_at_config5_geometry(b, l):
    """encoder parity at the config-5 geometry (12 layers, B = 32, L = 256) and at the reference's token cap (L = 502)"""
    from aspire_amd.encoder import HipBertEncoder
    m = _bert(12, seed=4)
    g = torch.Generator().manual_seed(l)
    tok = torch.randint(5, 3000, (b, l), generator=g)
    lens = torch.randint(l // 2, l + 1, (b,), generator=g)
    lens[0] = l
    mask = (torch.arange(l)[None, :] < lens[:, None]).long()
    tok = tok * mask
    with torch.no_grad():
        want = m(tok, token_type_ids=torch.zeros_like(tok), attention_mask=mask).last_hidden_state
    got = HipBertEncoder(m)(tok, token_type_ids=torch.zeros_like(tok), attention_mask=mask).last_hidden_state.cpu()
    err = (got - want).abs()[mask.bool()].max().item()
    assert err < TOL, err


def test_baseline_config5_one_gpu_slice_at_size():
    """BASELINE config 5, ONE GPU's whole slice of the scoring: 128 queries x 125 000 candidates x 12 sentences (4.6 GB of
    reps resident in HBM, 16 M pairs), otAspire and tsAspire.  Size-independent properties: every score finite, a
    sub-pool scored alone on the same kernel family reproduces its slice, the per-query rank is the stable sort of the
    scores, oracle spot checks."""
    from aspire_amd import ops, _lib
    nq, nc, s = 128, 125000, 12
    gen = torch.Generator(device='cuda').manual_seed(5)
    qrows = torch.randn(nq * s, 768, device='cuda', generator=gen)
    crows = torch.randn(nc * s, 768, device='cuda', generator=gen)
    mk = lambda rows, n: ops.DeviceRepSet(rows, (torch.arange(n, device='cuda', dtype=torch.int32) * s).contiguous(),
                                          torch.full((n,), s, device='cuda', dtype=torch.int32), ext=0, max_len=s)
    q, c = mk(qrows, nq), mk(crows, nc)
    scores, top_s, top_i = ops.ot_rank(q, c, 100, want=_lib.OT_SIMILARITY)
    assert torch.isfinite(scores).all() and (scores < 0).all()
    ref_s, ref_i = torch.sort(scores, dim=1, descending=True, stable=True)
    assert torch.equal(top_i, ref_i[:, :100]) and torch.equal(top_s, ref_s[:, :100])
    pick = torch.arange(60000, 62048, device='cuda')
    sub_rows = crows.view(nc, s, 768)[pick].reshape(-1, 768).contiguous()
    with _lib.pinned(COST_PATH='mfma'):
        sub = ops.ot_sinkhorn(q, mk(sub_rows, len(pick)), want=_lib.OT_SIMILARITY).view(nq, len(pick))
    assert torch.equal(scores[:, pick], sub)
    for qi, ci in [(0, 0), (127, 124999), (64, 61000)]:
        want = orc.get_similarity(qrows[qi * s:(qi + 1) * s].cpu(), crows[ci * s:(ci + 1) * s].cpu())
        assert scores[qi, ci].item() == pytest.approx(want, abs=TOL)
    l2 = ops.l2max_scores(q, c).view(nq, nc)
    assert torch.isfinite(l2).all()
    with _lib.pinned(COST_PATH='mfma'):
        l2sub = ops.l2max_scores(q, mk(sub_rows, len(pick))).view(nq, len(pick))
    assert torch.equal(l2[:, pick], l2sub)
    qt = orc.RepLen(qrows[:s].cpu()[None].permute(0, 2, 1), [s])
    ct = orc.RepLen(crows[:s].cpu()[None].permute(0, 2, 1), [s])
    assert l2[0, 0].item() == pytest.approx(-orc.allpair_masked_dist_l2max(qt, ct).item(), abs=TOL)
