"""GPU parity of the encoder path (A1-A3): HIP BERT forward + span pooling against HuggingFace BertModel on
the CPU (the third-party arithmetic the reference calls at ex_aspire_consent.py:72) and the oracle's pooling.
Weights are random-init from a fixed seed (no checkpoints offline).  Tolerance 1e-4 on hidden states and
sentence reps (BASELINE.json north_star)."""
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
    # random-init LayerNorms are identity and biases zero: perturb them so every parameter matters
    with torch.no_grad():
        for n, p in m.named_parameters():
            if 'LayerNorm' in n or n.endswith('.bias'):
                p.add_(0.1 * torch.randn_like(p))
    return m


def _batch(b, l, vocab, seed, ragged=True):
    g = torch.Generator().manual_seed(seed)
    tok = torch.randint(5, vocab, (b, l), generator=g)
    lens = torch.randint(l // 3, l + 1, (b,), generator=g) if ragged else torch.full((b,), l)
    lens[0] = l
    mask = (torch.arange(l)[None, :] < lens[:, None]).long()
    tok = tok * mask  # pad id 0, like the reference's batches
    return tok, torch.zeros_like(tok), mask, lens.tolist()


@pytest.mark.parametrize('n_layers,b,l', [(0, 2, 16), (1, 3, 37), (2, 4, 128), (12, 2, 64), (1, 3, 129), (1, 2, 300)])
def test_bert_forward_matches_transformers(n_layers, b, l):
    from aspire_amd.encoder import HipBertEncoder
    m = _bert(n_layers)
    tok, seg, mask, _ = _batch(b, l, 3000, seed=l)
    with torch.no_grad():
        want = m(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state
    got = HipBertEncoder(m)(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state.cpu()
    valid = mask.bool()
    err = (got - want).abs()[valid].max().item()
    assert err < TOL, err
    # padded query rows still attend to the real keys in BertModel; they agree too
    assert (got - want).abs().max().item() < TOL


def test_fused_attention_matches_three_kernel_form():
    """aspire_debug_set("ATTN", "gemm") runs attention as QK^T GEMM + masked soft-max + PV GEMM; the fused kernels (the default
    on fp16 planes, and "f32") must give the same hidden states (key tiles of 128: lengths on, just past and between tile edges, ragged masks)."""
    from aspire_amd._lib import pinned
    from aspire_amd.encoder import HipBertEncoder
    m = _bert(2, seed=5)
    enc = HipBertEncoder(m)
    for l in (128, 131, 257):
        tok, seg, mask, _ = _batch(3, l, 3000, seed=100 + l)
        fused = enc(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state.cpu()
        with pinned(ATTN='gemm'):
            ref = enc(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state.cpu()
        assert (fused - ref).abs().max().item() < 2e-5, l
        # ... and the fused kernel's fp32-input form (round 2; the default splits Q, K, V and the probabilities into fp16 planes)
        with pinned(ATTN='f32'):
            f32 = enc(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state.cpu()
        assert (fused - f32).abs().max().item() < 2e-5, l
        assert (f32 - ref).abs().max().item() < 2e-5, l


def test_bert_full_length_512():
    from aspire_amd.encoder import HipBertEncoder
    m = _bert(2, seed=3)
    tok, seg, mask, _ = _batch(2, 502, 3000, seed=9)
    with torch.no_grad():
        want = m(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state
    got = HipBertEncoder(m)(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state.cpu()
    assert (got - want).abs().max().item() < TOL


def test_consent_forward_matches_reference_path():
    """AspireConSent.forward end to end: BertModel (CPU) + the oracle's mask-multiply pooling."""
    from aspire_amd import AspireConSent
    m = _bert(12, seed=1)
    b, l = 3, 96
    tok, seg, mask, lens = _batch(b, l, 3000, seed=4)
    idxs = []
    for bi in range(b):
        n = lens[bi] - 1
        cuts = list(range(6, n, 13)) + [n]
        idxs.append([list(range(cuts[i], cuts[i + 1])) for i in range(len(cuts) - 1)][:7])
    abs_lens = [len(x) for x in idxs]
    batch = {'tokid_tt': tok, 'seg_tt': seg, 'attnmask_tt': mask, 'seq_lens': [l] + lens[1:]}
    batch['seq_lens'][0] = l
    model = AspireConSent(bert_model=m)
    cls, sent = model.forward(batch, abs_lens, idxs)
    with torch.no_grad():
        hidden = m(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state
    wcls, wsent = orc.span_mean_pool(hidden, idxs, abs_lens)
    assert cls.shape == (b, 768) and sent.shape == (b, max(abs_lens), 768)
    assert not cls.is_cuda and not sent.is_cuda          # CPU in, CPU out like the reference's example
    np.testing.assert_allclose(sent.numpy(), wsent.numpy(), atol=TOL, rtol=0)
    np.testing.assert_allclose(cls.numpy(), wcls.numpy(), atol=TOL, rtol=0)
    for bi in range(b):
        assert torch.all(sent[bi, abs_lens[bi]:] == 0)


def test_readme_example_shapes(tmp_path, golden_dir):
    """BASELINE config 1 (README.md:58-93 with an offline tokenizer): 2 abstracts through
    prepare_abstracts -> AspireConSent -> tsAspire / otAspire scores."""
    import json, os
    from transformers import BertTokenizer
    from aspire_amd import AspireConSent, prepare_abstracts, AllPairMaskedWasserstein, rep_len_tup
    z = json.load(open(os.path.join(golden_dir, 'prep.json')))
    (tmp_path / 'vocab.txt').write_text('\n'.join(z['vocab']) + '\n')
    tok = BertTokenizer(str(tmp_path / 'vocab.txt'), do_lower_case=True)
    docs = [z['docs'][0], z['docs'][4]]
    bert_batch, abs_lens, sent_token_idxs = prepare_abstracts(docs, tok)
    model = AspireConSent(bert_model=_bert(2, seed=5, vocab=len(z['vocab'])))
    cls, sent = model.forward(bert_batch, abs_lens, sent_token_idxs)
    assert cls.shape == (2, 768) and sent.shape == (2, max(abs_lens), 768)
    q = rep_len_tup(embed=sent[:1].permute(0, 2, 1), abs_lens=abs_lens[:1])
    c = rep_len_tup(embed=sent[1:].permute(0, 2, 1), abs_lens=abs_lens[1:])
    sims, extras = AllPairMaskedWasserstein({}).compute_distance(q, c, return_pair_sims=True)
    assert sims.shape == (1,) and extras[3].shape == (1, sent.shape[1], sent.shape[1])
    assert np.isfinite(sims.numpy()).all()


def test_gemm_bf16x3_form_is_fp32_accurate():
    """the three-way bf16 split on the bf16 matrix pipe against float64 and against the fp32-input MFMA form: same error
    level (the dropped cross terms are 2^-24 relative); odd M / N edges, bias; asymmetric operands (a transposed or
    mis-paired fragment cannot pass)"""
    import ctypes
    from aspire_amd import _lib
    f = _lib.lib.aspire_debug_gemm_f32
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
    g = torch.Generator().manual_seed(5)
    for M, N, K in ((8192, 2304, 768), (1000, 768, 3072), (130, 70, 768), (4096, 3072, 768)):
        A = (torch.randn(M, K, generator=g) * torch.linspace(0.1, 3.0, K)).cuda()
        B = (torch.randn(N, K, generator=g) + 0.5).cuda()
        bias = torch.randn(N, generator=g).cuda()
        ref = A[:256].double() @ B.double().T + bias.double()
        out = {}
        for form in ('f32', 'bf16x3'):
            C = torch.empty(M, N, device='cuda')
            with _lib.pinned(GEMM=form):
                assert f(A.data_ptr(), B.data_ptr(), C.data_ptr(), bias.data_ptr(), M, N, K,
                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
            torch.cuda.synchronize()
            out[form] = C
        e32 = (out['f32'][:256].double() - ref).abs().max().item()
        e3 = (out['bf16x3'][:256].double() - ref).abs().max().item()
        scale = ref.abs().max().item()
        assert e3 <= max(2.0 * e32, 3e-7 * scale), (M, N, K, e32, e3, scale)
        assert (out['f32'] - out['bf16x3']).abs().max().item() <= 4e-6 * scale


def test_gemm_on_presplit_operands_is_fp32_accurate():
    """The P-layout GEMM (operands pre-split into two fp16 planes: the weights at model load, an activation by the kernel that
    produces it; LDS-DMA tiles; three exact products per term) against float64 and against the on-the-fly bf16x3 split (six
    products): no larger an error, on operands far from the encoder's comfortable ranges (activations to +-12, weights O(1)); M edges
    (rows past the matrix feed unstored outputs only), both ring depths, 128 x 64 column tiles, and the GELU -> P-layout epilogue
    of the first FFN GEMM (its output read back through a second P-layout GEMM against the identity)."""
    import ctypes
    from aspire_amd import _lib
    L = _lib.lib
    L.aspire_debug_gemm_f32.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
    L.aspire_debug_planes_bytes.restype = ctypes.c_size_t
    L.aspire_debug_planes_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64]
    L.aspire_debug_split_planes.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.aspire_debug_gemm_planes.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(7)

    def planes(X, weight=0):                       # weight: the B side (split as 64 x, the GEMM's epilogue takes the factor off)
        r, k = X.shape
        P = torch.empty(L.aspire_debug_planes_bytes(r, k), dtype=torch.uint8, device='cuda')     # slack rows left uninitialised
        assert L.aspire_debug_split_planes(X.data_ptr(), r, k, P.data_ptr(), weight, st) == 0
        return P

    for M, N, K in ((8192, 2304, 768), (1000, 768, 3072), (130, 768, 768), (4100, 3072, 768)):
        A = (torch.randn(M, K, generator=g) * torch.linspace(0.1, 3.0, K)).cuda()
        B = (torch.randn(N, K, generator=g) + 0.5).cuda()
        bias = torch.randn(N, generator=g).cuda()
        ref = A.double() @ B.double().T + bias.double()
        scale = ref.abs().max().item()
        C0 = torch.empty(M, N, device='cuda')
        with _lib.pinned(GEMM='bf16x3'):
            assert L.aspire_debug_gemm_f32(A.data_ptr(), B.data_ptr(), C0.data_ptr(), bias.data_ptr(), M, N, K, st) == 0
        Ap, Bp = planes(A), planes(B, 1)
        for pin in ({}, {'GEMM_RING': '3'}, {'GEMM_TILE': '64'}, {'GEMM_RING': '113'}, {'GEMM_TILE': '256'}):      # 113: the persistent tile loop (> 768 tiles); 256: 256 x 128 tiles on eight waves
            C = torch.full((M, N), float('nan'), device='cuda')
            with _lib.pinned(**pin):
                assert L.aspire_debug_gemm_planes(Ap.data_ptr(), Bp.data_ptr(), C.data_ptr(), None, bias.data_ptr(), M, N, K, 0, st) == 0
            torch.cuda.synchronize()
            assert torch.isfinite(C).all(), (M, N, K, pin)
            err = (C.double() - ref).abs().max().item()
            err0 = (C0.double() - ref).abs().max().item()
            assert err <= max(1.5 * err0, 3e-7 * scale), (M, N, K, pin, err, err0)
            assert (C - C0).abs().max().item() <= 2e-6 * scale, (M, N, K, pin)
    # outliers, as trained encoders have them: a few activation columns in the hundreds, a few weight rows in the tens, most entries
    # small (their low planes sit in fp16's subnormal range: an absolute 2^-25 each) -- still no worse than the fp32-input forms
    M, N, K = 2048, 768, 3072
    A = 0.05 * torch.randn(M, K, generator=g)
    A[:, ::257] *= 4000.0
    B = 0.02 * torch.randn(N, K, generator=g)
    B[::97] *= 400.0
    A, B = A.cuda(), B.cuda()
    ref = A.double() @ B.double().T
    scale = ref.abs().max().item()
    C0, C = torch.empty(M, N, device='cuda'), torch.empty(M, N, device='cuda')
    with _lib.pinned(GEMM='bf16x3'):
        assert L.aspire_debug_gemm_f32(A.data_ptr(), B.data_ptr(), C0.data_ptr(), None, M, N, K, st) == 0
    assert L.aspire_debug_gemm_planes(planes(A).data_ptr(), planes(B, 1).data_ptr(), C.data_ptr(), None, None, M, N, K, 0, st) == 0
    torch.cuda.synchronize()
    err, err0 = (C.double() - ref).abs().max().item(), (C0.double() - ref).abs().max().item()
    assert torch.isfinite(C).all() and err <= max(1.5 * err0, 3e-7 * scale), (err, err0, scale)
    # ... and row by row against the row's own magnitude (small outputs next to big ones)
    rel = ((C.double() - ref).abs().amax(1) / ref.abs().amax(1)).max().item()
    rel0 = ((C0.double() - ref).abs().amax(1) / ref.abs().amax(1)).max().item()
    assert rel <= max(1.5 * rel0, 1e-6), (rel, rel0)
    # GELU epilogue into the P layout: H = GELU(A . B^T + bias) as planes, read back as H . I^T
    M, N, K = 300, 256, 768
    A = torch.randn(M, K, generator=g).cuda()
    B = (0.05 * torch.randn(N, K, generator=g)).cuda()
    bias = torch.randn(N, generator=g).cuda()
    Hp = torch.empty(L.aspire_debug_planes_bytes(M, N), dtype=torch.uint8, device='cuda')
    assert L.aspire_debug_gemm_planes(planes(A).data_ptr(), planes(B, 1).data_ptr(), None, Hp.data_ptr(), bias.data_ptr(), M, N, K, 1, st) == 0
    eye = torch.eye(N, device='cuda')
    H = torch.empty(M, N, device='cuda')
    assert L.aspire_debug_gemm_planes(Hp.data_ptr(), planes(eye, 1).data_ptr(), H.data_ptr(), None, None, M, N, N, 0, st) == 0
    torch.cuda.synchronize()
    want = torch.nn.functional.gelu(A.double() @ B.double().T + bias.double())
    assert (H.double() - want).abs().max().item() < 2e-6 * max(1.0, want.abs().max().item())
    # the same epilogue from the persistent tile loop (780 tiles on 768 resident workgroups, a ragged last row of tiles): the same bits
    M, N, K = 8300, 1536, 768
    A = torch.randn(M, K, generator=g).cuda()
    B = (0.05 * torch.randn(N, K, generator=g)).cuda()
    bias = torch.randn(N, generator=g).cuda()
    Ap, Bp, eyep = planes(A), planes(B, 1), planes(torch.eye(N, device='cuda'), 1)
    out = []
    for pin in ({}, {'GEMM_RING': '113'}):
        Hp = torch.zeros(L.aspire_debug_planes_bytes(M, N), dtype=torch.uint8, device='cuda')
        H = torch.empty(M, N, device='cuda')
        with _lib.pinned(**pin):
            assert L.aspire_debug_gemm_planes(Ap.data_ptr(), Bp.data_ptr(), None, Hp.data_ptr(), bias.data_ptr(), M, N, K, 1, st) == 0
        assert L.aspire_debug_gemm_planes(Hp.data_ptr(), eyep.data_ptr(), H.data_ptr(), None, None, M, N, N, 0, st) == 0
        torch.cuda.synchronize()
        out.append(H)
    assert torch.equal(out[0], out[1])
    want = torch.nn.functional.gelu(A[-200:].double() @ B.double().T + bias.double())
    assert (out[1][-200:].double() - want).abs().max().item() < 2e-6 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize('n_layers,b,l', [(2, 4, 128), (12, 2, 64), (1, 2, 300), (2, 2, 502), (1, 3, 37)])
def test_bert_forward_on_presplit_operands_small_shapes(n_layers, b, l):
    """The forward on pre-split operands (P layout, LDS-DMA GEMM, planes written by the LayerNorm / attention / GELU epilogues) is the
    default from ~3000 token rows on (the 12-layer B = 32, L = 256 test of test_gpu_pipeline.py runs it); pinned here at small
    shapes -- row counts that are no multiple of the 128-row tile, L = 502 -- against HuggingFace BertModel at 1e-4 and against the
    on-the-fly-split form."""
    from aspire_amd._lib import pinned
    from aspire_amd.encoder import HipBertEncoder
    m = _bert(n_layers, seed=7)
    tok, seg, mask, _ = _batch(b, l, 3000, seed=200 + l)
    with torch.no_grad():
        want = m(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state
    enc = HipBertEncoder(m)
    with pinned(GEMM='planes'):
        got = enc(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state.cpu()
    with pinned(GEMM='bf16x3'):
        other = enc(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state.cpu()
    assert (got - want).abs().max().item() < TOL
    assert (got - other).abs().max().item() < 2e-5


@pytest.mark.parametrize('n_layers,b,l', [(2, 9, 130), (3, 64, 256), (1, 3, 37), (2, 40, 100), (1, 128, 500)])
def test_layernorm_in_the_gemm_epilogue_matches_the_separate_pass(n_layers, b, l):
    """LayerNorm inside the N = 768 GEMMs' epilogue (gemm_p_kernel's LN form: the six / twelve column tiles of a row block exchange
    (mean, sum of squared deviations) through global memory and normalise out of their accumulators; the residual stream lives in the
    P layout only) against the same forward with ASPIRE_HIP_GEMM_LN=off (layernorm_kernel as its own pass) and against HuggingFace: row counts that are no multiple of 128, a
    launch that fills every workgroup slot (64 x 256 = 16 384 rows: 768 tiles) and one of four rounds (128 x 500 = 64 000 rows: tiles wait at the
    launch frontier while later ones start), 128 x 64 tiles pinned as well, run twice --
    the exchange counters are zeroed per forward -- with equal bits both times."""
    from aspire_amd._lib import pinned
    from aspire_amd.encoder import HipBertEncoder
    m = _bert(n_layers, seed=11)
    tok, seg, mask, _ = _batch(b, l, 3000, seed=300 + l)
    enc = HipBertEncoder(m)
    with pinned(GEMM='planes', GEMM_LN='on'):
        got = enc(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state.cpu()
        again = enc(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state.cpu()
        with pinned(GEMM_TILE='64'):
            narrow = enc(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state.cpu()
    with pinned(GEMM='planes', GEMM_LN='off'):
        sep = enc(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state.cpu()
    with pinned(GEMM='planes'):
        default = enc(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state.cpu()
    assert torch.isfinite(got).all()
    assert torch.equal(got, again)
    assert torch.equal(default, got if (b * l + 127) // 128 >= 48 else sep)          # the default: fused from 48 row tiles on
    assert (got - sep).abs().max().item() < 5e-6, (got - sep).abs().max().item()
    assert (narrow - sep).abs().max().item() < 5e-6
    if b * l <= 4096:
        with torch.no_grad():
            want = m(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state
        assert (got - want).abs().max().item() < TOL


@pytest.mark.timeout(240)
def test_layernorm_epilogue_forwards_on_two_streams_at_once():
    """Two encoders' forwards in flight at the same time on two streams (each with its own workspace): the LayerNorm-epilogue GEMMs WAIT
    inside the kernel for their row block's other column tiles, so two such launches share the chip's workgroup slots while tiles of
    both wait -- the launch-order argument (a waiting tile's partners are resident or next in line in their own launch) has to hold
    with slots taken by the other launch.  Same bits as each forward alone; 2 x 96 x 128 rows = more tiles than slots for either."""
    from aspire_amd._lib import pinned
    from aspire_amd.encoder import HipBertEncoder
    encs = [HipBertEncoder(_bert(2, seed=21 + i)) for i in range(2)]
    ins = [_batch(96, 128, 3000, seed=500 + i) for i in range(2)]
    with pinned(GEMM='planes', GEMM_LN='on'):
        alone = [e.forward_hidden(t[0], t[1], t[2]).clone() for e, t in zip(encs, ins)]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream() for _ in range(2)]
        dev_in = [tuple(x.cuda() for x in t[:3]) for t in ins]
        outs = [[], []]
        for _ in range(6):
            for k in range(2):
                with torch.cuda.stream(streams[k]):
                    outs[k].append(encs[k].forward_hidden(*dev_in[k], check_ids=False))
        torch.cuda.synchronize()
    for k in range(2):
        for o in outs[k]:
            assert torch.equal(o, alone[k])


def test_weights_beyond_the_fp16_planes_are_left_to_the_fp32_input_kernels():
    """aspire_bert_prepare_planes rejects a weight beyond +-1023 (64 w must stay inside fp16); HipBertEncoder then runs without
    planes -- the on-the-fly bf16x3 GEMMs take any fp32 -- and still matches HuggingFace."""
    from aspire_amd.encoder import HipBertEncoder
    m = _bert(1, seed=9)
    with torch.no_grad():
        m.encoder.layer[0].intermediate.dense.weight[5, 7] = 3000.0
    tok, seg, mask, _ = _batch(8, 128, 3000, seed=77)
    with pytest.warns(UserWarning, match='1023'):
        enc = HipBertEncoder(m)
    assert not enc._w.planes
    with torch.no_grad():
        want = m(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state
    got = enc.forward_hidden(tok.cuda(), seg.cuda(), mask.cuda()).cpu()
    assert (got - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())


def test_checkpoint_path_from_a_local_save_pretrained_directory(tmp_path):
    """AspireConSent(hf_model_name) as examples/ex_aspire_consent.py:33 calls it -- AutoModel.from_pretrained -- on a LOCAL
    directory (no network): a BERT-base-geometry model with SciBERT's vocabulary size (31 090) saved WITH its pooler, as the
    published Aspire checkpoints are.  The forward equals HF's at 1e-4; the pooler is computed by the reference and unused
    (ex_aspire_consent.py:72-76), here it is not even loaded."""
    from transformers import BertConfig, BertModel, AutoModel
    from aspire_amd import AspireConSent
    torch.manual_seed(5)
    cfg = BertConfig(vocab_size=31090, hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=3072,
                     max_position_embeddings=512)
    src = BertModel(cfg, add_pooling_layer=True).eval()
    src.save_pretrained(str(tmp_path / 'ckpt'))
    model = AspireConSent(str(tmp_path / 'ckpt'))
    assert model.bert_encoder.config.vocab_size == 31090
    g = torch.Generator().manual_seed(6)
    tok = torch.randint(0, 31090, (3, 40), generator=g)
    tok[:, -1] = 31089                                           # the last row of the embedding table
    mask = torch.ones_like(tok)
    mask[1, 30:] = 0
    hf = AutoModel.from_pretrained(str(tmp_path / 'ckpt')).eval()
    with torch.no_grad():
        want = hf(tok, token_type_ids=torch.zeros_like(tok), attention_mask=mask).last_hidden_state
    got = model.bert_encoder.forward_hidden(tok, torch.zeros_like(tok), mask).cpu()
    np.testing.assert_allclose(got[0].numpy(), want[0].numpy(), atol=1e-4, rtol=0)
    np.testing.assert_allclose(got[1, :30].numpy(), want[1, :30].numpy(), atol=1e-4, rtol=0)
    # the drop-in forward on top of it
    bert_batch = {'tokid_tt': tok, 'seg_tt': torch.zeros_like(tok), 'attnmask_tt': mask, 'seq_lens': [40, 30, 40]}
    cls, sent = model.forward(bert_batch, [2, 1, 2], [[list(range(3, 9)), list(range(9, 20))], [list(range(2, 7))],
                                                        [list(range(1, 5)), list(range(5, 39))]])
    np.testing.assert_allclose(cls.numpy(), want[:, 0].numpy(), atol=1e-4, rtol=0)
    np.testing.assert_allclose(sent[2, 1].numpy(), want[2, 5:39].mean(0).numpy(), atol=1e-4, rtol=0)


def test_state_dict_with_the_bert_prefix():
    """a BertForPreTraining-style module: every encoder weight under 'bert.' (+ heads the encoder ignores) -- the weight-name
    mapping of HipBertEncoder finds them"""
    from transformers import BertConfig, BertForPreTraining
    from aspire_amd.encoder import HipBertEncoder
    torch.manual_seed(7)
    cfg = BertConfig(vocab_size=500, hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=3072,
                     max_position_embeddings=64)
    full = BertForPreTraining(cfg).eval()
    assert all(k.startswith(('bert.', 'cls.')) for k in full.state_dict())
    enc = HipBertEncoder(full)
    tok = torch.randint(0, 500, (2, 16), generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        want = full.bert(tok, token_type_ids=torch.zeros_like(tok), attention_mask=torch.ones_like(tok)).last_hidden_state
    got = enc.forward_hidden(tok, torch.zeros_like(tok), torch.ones_like(tok)).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=1e-4, rtol=0)


def test_activation_beyond_the_fp16_planes_falls_back_to_the_full_range_kernels():
    """the default encoder path keeps activations as two fp16 planes (|x| <= 65504).  A checkpoint with an outlier
    feature beyond that -- here one channel of the embedding LayerNorm scaled by 2e5, its input weights scaled down so that the
    layer stays well conditioned -- produces inf on that path; AspireConSent.forward notices the non-finite reps and runs the
    forward again on the kernels that take any fp32 value (bf16x3 GEMMs, fp32-input attention), which match HF"""
    from transformers import BertConfig, BertModel
    from aspire_amd import AspireConSent
    torch.manual_seed(11)
    cfg = BertConfig(vocab_size=400, hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=3072,
                     max_position_embeddings=128)
    m = BertModel(cfg, add_pooling_layer=False).eval()
    ch = 5
    with torch.no_grad():
        m.embeddings.LayerNorm.weight[ch] = 2e5
        lyr = m.encoder.layer[0]
        for lin in (lyr.attention.self.query, lyr.attention.self.key, lyr.attention.self.value, lyr.intermediate.dense):
            lin.weight[:, ch] *= 1e-5
    model = AspireConSent(bert_model=m)
    tok = torch.randint(0, 400, (8, 128), generator=torch.Generator().manual_seed(12))      # 1024 token rows: the fp16-plane GEMMs
    mask = torch.ones_like(tok)
    hidden_default = model.bert_encoder.forward_hidden(tok, torch.zeros_like(tok), mask)
    assert not bool(torch.isfinite(hidden_default).all())            # the fp16-plane path does overflow on this model
    with torch.no_grad():
        want = m(tok, token_type_ids=torch.zeros_like(tok), attention_mask=mask).last_hidden_state
    full = model.bert_encoder.forward_full_range(tok, torch.zeros_like(tok), mask).cpu()
    scale = float(want.abs().max())
    assert torch.isfinite(full).all() and float((full - want).abs().max()) < 2e-5 * scale
    bert_batch = {'tokid_tt': tok, 'seg_tt': torch.zeros_like(tok), 'attnmask_tt': mask, 'seq_lens': [128] * 8}
    idxs = [[list(range(1, 12)), list(range(12, 31))]] * 8
    cls, sent = model.forward(bert_batch, [2] * 8, idxs)
    assert torch.isfinite(sent).all() and torch.isfinite(cls).all()
    np.testing.assert_allclose(sent[1, 1].numpy(), want[1, 12:31].mean(0).numpy(), atol=2e-5 * scale, rtol=0)
    with pytest.warns(UserWarning, match='non-finite'):
        pool = model.encode_to_pool([(bert_batch, [2] * 8, idxs)])
    rows = pool.repset.rows.cpu()
    assert torch.isfinite(rows).all()
    np.testing.assert_allclose(rows[3].numpy(), want[1, 12:31].mean(0).numpy(), atol=2e-5 * scale, rtol=0)
