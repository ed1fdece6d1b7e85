"""GPU: round 6's attention path -- the QKV GEMM writes the attention's operands as fp16 planes per head ([plane][Q | K | V][head][M][64]),
flash_attn_p_kernel stages K and V tiles by LDS-DMA and reads V^T fragments through the LDS transpose read (ds_read_b64_tr_b16) -- against
round 5's (ASPIRE_HIP_ATTN=f16x2: fp32 Q / K / V split inside the attention kernel).  The planes are the same split of the same fp32
values, the key tiles and every sum are the same: the SAME BITS at every length.  Reference arithmetic: HuggingFace BertModel
(examples/ex_aspire_consent.py:72)."""
import pytest
import torch

from test_gpu_encoder import _batch, _bert

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n_layers,b,l', [(2, 8, 128), (1, 64, 256), (2, 4, 512), (1, 16, 64), (2, 3, 400), (1, 52, 128)])
def test_plane_attention_has_the_bits_of_the_round5_kernel(n_layers, b, l):
    from aspire_amd._lib import pinned
    from aspire_amd.encoder import HipBertEncoder
    enc = HipBertEncoder(_bert(n_layers, seed=40 + l))
    tok, seg, mask, _ = _batch(b, l, 3000, seed=900 + l)
    with pinned(GEMM='planes'):
        new = enc.forward_hidden(tok, seg, mask).cpu()
        again = enc.forward_hidden(tok, seg, mask).cpu()
        with pinned(ATTN='f16x2'):
            old = enc.forward_hidden(tok, seg, mask).cpu()
    assert torch.isfinite(new).all()
    assert torch.equal(new, again)
    assert torch.equal(new, old), (new - old).abs().max().item()


@pytest.mark.parametrize('n_layers,b,l', [(2, 9, 130), (1, 3, 37), (2, 40, 100), (1, 5, 500), (1, 7, 301), (12, 6, 173)])
def test_plane_attention_on_lengths_that_move_the_tile_edges(n_layers, b, l):
    """lengths that are no multiple of 128 / 16 / 4: the last key tile of a document runs into the NEXT document's rows (weighted exactly 0),
    the last document's past the last token row (clamped); ragged masks; against round 5's kernel (same bits) and HuggingFace"""
    from aspire_amd._lib import pinned
    from aspire_amd.encoder import HipBertEncoder
    m = _bert(n_layers, seed=50 + l)
    enc = HipBertEncoder(m)
    tok, seg, mask, _ = _batch(b, l, 3000, seed=950 + l)
    with pinned(GEMM='planes'):
        new = enc.forward_hidden(tok, seg, mask).cpu()
        with pinned(ATTN='f16x2'):
            old = enc.forward_hidden(tok, seg, mask).cpu()
    assert torch.isfinite(new).all()
    assert torch.equal(new, old), (new - old).abs().max().item()
    with torch.no_grad():
        want = m(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state
    assert (new - want).abs().max().item() < 1e-4


def test_neighbours_with_huge_rows_do_not_leak_into_a_document():
    """a tile that runs into the NEXT document multiplies that document's V rows by probabilities of exactly 0: make the
    neighbour's values large (token embeddings x 300 on its ids) and compare the small document's rows with a forward of it alone"""
    from aspire_amd._lib import pinned
    from aspire_amd.encoder import HipBertEncoder
    m = _bert(1, seed=77)
    with torch.no_grad():
        m.embeddings.word_embeddings.weight[2000:] *= 300.0            # (LayerNorm brings the hidden state back; K / V see other directions)
    enc = HipBertEncoder(m)
    g = torch.Generator().manual_seed(3)
    l = 40                                                             # starts 0, 40, 80, ...: most are no multiple of 16
    tok = torch.randint(5, 1500, (32, l), generator=g)
    tok[::2] = torch.randint(2000, 3000, (16, l), generator=g)         # every other document: the huge ids
    mask = torch.ones_like(tok)
    with pinned(GEMM='planes'):
        both = enc.forward_hidden(tok, None, mask).cpu()
        with pinned(ATTN='f16x2'):
            old = enc.forward_hidden(tok, None, mask).cpu()
    assert torch.isfinite(both).all()
    assert torch.equal(both, old)


@pytest.mark.parametrize('n_layers,b,l', [(2, 8, 128), (1, 64, 256), (2, 3, 400), (1, 3, 37), (2, 9, 130), (12, 6, 173)])
def test_plane_attention_on_64_key_tiles(n_layers, b, l):
    """ASPIRE_HIP_ATTN=p64: the same kernel on 64-key tiles (33 KB of LDS, three workgroups per CU): other tile edges, so other online-soft-max
    groupings -- equal to the 128-key form to rounding, and to HuggingFace at 1e-4"""
    from aspire_amd._lib import pinned
    from aspire_amd.encoder import HipBertEncoder
    m = _bert(n_layers, seed=60 + l)
    enc = HipBertEncoder(m)
    tok, seg, mask, _ = _batch(b, l, 3000, seed=980 + l)
    with pinned(GEMM='planes'):
        ref = enc.forward_hidden(tok, seg, mask).cpu()
        with pinned(ATTN='p64'):
            got = enc.forward_hidden(tok, seg, mask).cpu()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 5e-6 * max(1, n_layers // 2), (got - ref).abs().max().item()     # (rounding differences add up over layers)
    if b * l <= 4096:
        with torch.no_grad():
            want = m(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state
        assert (got - want).abs().max().item() < 1e-4
