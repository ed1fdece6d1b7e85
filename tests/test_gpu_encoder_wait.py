"""GPU: the bounded wait of the LayerNorm-epilogue GEMMs (encoder.hip: gemm_p_ln_kernel; VERDICT r5 item 5 / ADVICE r5): a tile that
waits for its row block's partners gives up after 20 ms and says so in the encoder's status word (include/aspire_hip.h:
aspire_bert_status); the host layer then runs the forward again with the LayerNorm as its own pass.  Reference call site of the
forward: examples/ex_aspire_consent.py:72-73."""
import time

import pytest
import torch

from test_gpu_encoder import _batch, _bert

pytestmark = pytest.mark.gpu


def _spans(lens, n_sents=3):
    """n_sents contiguous equal spans over tokens 1 .. len-1 of every document"""
    out = []
    for n in lens:
        per = max(1, (n - 1) // n_sents)
        out.append([list(range(1 + s * per, 1 + (s + 1) * per)) for s in range(n_sents)])
    return out


@pytest.mark.timeout(300)
def test_a_broken_exchange_times_out_and_the_host_falls_back():
    """Fault injection (GEMM_PROBE=48: the first column tile of every row block never counts itself, so its five partners can never
    see the full count): the launch ENDS (bounded wait), the status word reads ASPIRE_BERT_STATUS_LN_TIMEOUT and is cleared by the
    read; AspireConSent.forward notices, warns, and returns the reps of the separate-LayerNorm path (equal to HuggingFace)."""
    from aspire_amd._lib import pinned
    from aspire_amd.consent import AspireConSent
    from aspire_amd.encoder import HipBertEncoder
    m = _bert(1, seed=31)
    enc = HipBertEncoder(m)
    tok, seg, mask, lens = _batch(52, 128, 3000, seed=41)          # 6 656 rows = 52 row tiles
    assert enc.status() == 0
    with pinned(GEMM='planes', GEMM_LN='on'):
        good = enc.forward_hidden(tok, seg, mask).cpu()
        assert enc.status() == 0
        with pinned(GEMM_PROBE='48'):
            t0 = time.time()
            enc.forward_hidden(tok, seg, mask)
            st = enc.status()                                        # synchronises
            took = time.time() - t0
    assert st == 1, st
    assert took < 5.0, took                                          # two launches x 20 ms, not a hang
    assert enc.status() == 0                                         # cleared by the read
    with pinned(GEMM='planes', GEMM_LN='on'):                        # and the next forward is fine again
        assert torch.equal(enc.forward_hidden(tok, seg, mask).cpu(), good)
    assert enc.status() == 0
    # the host layer's fall-back: forward() under the fault returns what the separate pass computes
    model = AspireConSent(bert_model=m)
    bb = {'tokid_tt': tok, 'seg_tt': seg, 'attnmask_tt': mask, 'seq_lens': lens}
    spans = _spans(lens)
    with pinned(GEMM='planes', GEMM_LN='off'):
        want_cls, want_sent = model.forward(bb, [3] * len(lens), spans)
    with pinned(GEMM='planes', GEMM_LN='on', GEMM_PROBE='48'):
        with pytest.warns(UserWarning, match='timed out'):
            got_cls, got_sent = model.forward(bb, [3] * len(lens), spans)
    assert torch.equal(got_cls, want_cls) and torch.equal(got_sent, want_sent)
    assert model.bert_encoder.status() == 0


@pytest.mark.timeout(600)
def test_eight_forwards_on_eight_streams_beside_a_long_scoring_call():
    """Eight LayerNorm-epilogue forwards in flight on eight streams (one workspace each: HipBertEncoder keeps one per stream) while a
    ninth stream runs a many-query otAspire call that holds every CU for milliseconds: nobody times out, every forward has the bits it
    has alone."""
    from aspire_amd import ops, scorer
    from aspire_amd._lib import pinned
    from aspire_amd.encoder import HipBertEncoder
    enc = HipBertEncoder(_bert(2, seed=51))
    ins = [tuple(x.cuda() for x in _batch(56, 128, 3000, seed=600 + i)[:3]) for i in range(8)]
    g = torch.Generator().manual_seed(5)
    pool = scorer.CandidatePool([torch.randn(8, 768, generator=g) for _ in range(20000)]).prepare_planes()
    qs = ops.DeviceRepSet.from_list([torch.randn(8, 768, generator=g) for _ in range(32)])
    with pinned(GEMM='planes', GEMM_LN='on'):
        alone = [enc.forward_hidden(*t, check_ids=False).clone() for t in ins]
        score_alone = ops.ot_sinkhorn(qs, pool.repset).clone()
        torch.cuda.synchronize()
        assert enc.status() == 0
        streams = [torch.cuda.Stream() for _ in range(9)]
        outs = [[] for _ in range(8)]
        scores = []
        for rep in range(3):
            with torch.cuda.stream(streams[8]):
                scores.append(ops.ot_sinkhorn(qs, pool.repset))
            for k in range(8):
                with torch.cuda.stream(streams[k]):
                    outs[k].append(enc.forward_hidden(*ins[k], check_ids=False))
        torch.cuda.synchronize()
        assert enc.status() == 0
    assert len(enc._ws) >= 9                                         # a workspace per stream
    for k in range(8):
        for o in outs[k]:
            assert torch.equal(o, alone[k])
    for s in scores:
        assert torch.equal(s, score_alone)


@pytest.mark.timeout(600)
def test_encode_to_pool_on_two_streams_fills_the_same_store():
    """encode_to_pool(streams=2): forwards alternate over two side streams, forked from and joined to the caller's stream; the store,
    the CLS reps and the planes are the bits of the one-stream run (a document's reps do not depend on what runs beside it)."""
    from aspire_amd.consent import AspireConSent
    model = AspireConSent(bert_model=_bert(2, seed=61))
    batches = []
    for i in range(12):
        tok, seg, mask, lens = _batch(32, 96 + 8 * i, 3000, seed=700 + i)
        batches.append(({'tokid_tt': tok, 'seg_tt': seg, 'attnmask_tt': mask, 'seq_lens': lens}, [3] * 32, _spans(lens)))
    one, cls1 = model.encode_to_pool(batches, want_cls=True, rows_per_forward=8192)
    two, cls2 = model.encode_to_pool(batches, want_cls=True, rows_per_forward=8192, streams=2)
    three = model.encode_to_pool(batches, rows_per_forward=8192, streams=3, planes=True)      # (the same forwards: another row count per forward is another LayerNorm form)
    torch.cuda.synchronize()
    assert torch.equal(one.repset.rows, two.repset.rows) and torch.equal(cls1, cls2)
    assert torch.equal(one.repset.rows, three.repset.rows)
    assert model.bert_encoder.status() == 0
