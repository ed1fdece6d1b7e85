"""A BERT-base whose weights carry the statistics of a TRAINED checkpoint instead of the N(0, 0.02) init: the
reference loads `allenai/aspire-contextualsentence-multim-*` (examples/ex_aspire_consent.py:33, 72), which cannot be downloaded here; what
trained BERTs are known to have, and what the fp16 `h + l` planes, the 64 w weight scale and the overflow fall-backs are sensitive to, is
made synthetically:
  * LayerNorm gains with a few entries x 10 - 30, and two or three OUTLIER hidden dimensions whose residual values sit at 30 - 100
    (a large gain and an O(30 - 100) bias on the same few dimensions of EVERY LayerNorm, as the emergent outlier features of trained
    models are);
  * attention logits of +-50 (query / key projections of some heads scaled up);
  * FFN rows scaled x 20 (a few dozen intermediate units with large pre-activations; their output columns scaled down so that the layer
    stays conditioned like a trained one);
  * biases of O(1).
Shared by tests/test_gpu_encoder_heavy.py and tools/fuzz_encoder.py (mode `heavy`)."""
import torch

OUTLIER_DIMS = (77, 308, 588)


def heavy_tailed_bert(n_layers=12, seed=0, vocab=3000, ffn_overflow=False):
    from transformers import BertConfig, BertModel
    torch.manual_seed(seed)
    cfg = BertConfig(vocab_size=vocab, hidden_size=768, num_hidden_layers=n_layers, num_attention_heads=12, intermediate_size=3072,
                     max_position_embeddings=512)
    m = BertModel(cfg, add_pooling_layer=False).eval()
    g = torch.Generator().manual_seed(seed + 1)

    def rnd(*shape):
        return torch.randn(*shape, generator=g)

    with torch.no_grad():
        lns = [m.embeddings.LayerNorm] + [ln for ly in m.encoder.layer for ln in (ly.attention.output.LayerNorm, ly.output.LayerNorm)]
        for li, ln in enumerate(lns):
            # The outlier dimensions carry most of a row's variance (sum of squares ~ 16 500 of ~ 17 500), so a LayerNorm divides the
            # ordinary dimensions by ~ 4.8: a trained model's gains undo that (ordinary gains around 3.5 here), or every layer would
            # shrink the signal 5 x.  The outliers keep gain ~ 1 and a large bias: (x - mean) / std ~ 8 - 20 on them, + bias = 30 - 100.
            ln.weight.copy_(3.5 + 0.5 * rnd(768))
            ln.bias.copy_(0.3 * rnd(768))                              # biases of O(1)
            big = torch.randperm(768, generator=g)[:5]
            ln.weight[big] = torch.tensor([10.0, 14.0, 18.0, 24.0, 30.0])    # a few gains of 10 - 30
            for k, d in enumerate(OUTLIER_DIMS):                       # the outlier dimensions: the same ones in every LayerNorm
                ln.weight[d] = (1.0, 1.2, 0.8)[k]
                ln.bias[d] = (30.0, -55.0, 80.0)[k] * (1.0 if li % 2 == 0 else 0.9)
        for li, ly in enumerate(m.encoder.layer):
            att = ly.attention.self
            for lin in (att.query, att.key, att.value, ly.attention.output.dense, ly.intermediate.dense, ly.output.dense):
                lin.bias.copy_(0.5 * rnd(lin.bias.shape[0]))           # O(1) biases
            # trained projections do not read the outlier dimensions at full weight (they would drown everything else)
            for lin in (att.query, att.key, att.value, ly.intermediate.dense):
                lin.weight[:, list(OUTLIER_DIMS)] *= 0.05
            # FFN rows x 20 (their output columns scaled down: the unit still matters, the layer stays conditioned)
            units = torch.randperm(3072, generator=g)[:40]
            ly.intermediate.dense.weight[units] *= 20.0
            ly.output.dense.weight[:, units] *= 0.1
            if ffn_overflow and li == n_layers - 1:
                # one unit whose activation leaves fp16's range (|x| > 65504): the fp16-plane path must hand over to the full-range kernels
                u = int(units[0])
                ly.intermediate.dense.weight[u] *= 60.0
                ly.intermediate.dense.bias[u] = 90000.0
                ly.output.dense.weight[:, u] *= 1e-4
        # attention logits of +-50: the query / key projections of four heads per layer scaled until the largest |q.k / 8| over a
        # calibration batch is 50 (layer by layer: a layer's input depends on the layers below it)
        gc = torch.Generator().manual_seed(seed + 2)
        tok = torch.randint(5, vocab, (4, 96), generator=gc)
        mask = (torch.arange(96)[None, :] < torch.tensor([96, 70, 96, 33])[:, None]).long()
        tok, seg = tok * mask, torch.zeros_like(tok)
        for li, ly in enumerate(m.encoder.layer):
            att = ly.attention.self
            hs = m(tok, token_type_ids=seg, attention_mask=mask, output_hidden_states=True).hidden_states[li]
            q = att.query(hs).view(4, 96, 12, 64).transpose(1, 2)
            k = att.key(hs).view(4, 96, 12, 64).transpose(1, 2)
            sc = ((q @ k.transpose(-1, -2)) / 8.0).masked_fill(~mask.bool()[:, None, None, :], 0.0).masked_fill(~mask.bool()[:, None, :, None], 0.0)
            for h in ((li + 0) % 12, (li + 5) % 12, (li + 7) % 12, (li + 10) % 12):
                f = float((50.0 / sc[:, h].abs().max().clamp_min(1e-3)).sqrt().clamp(1.0, 40.0))
                for lin in (att.query, att.key):
                    lin.weight[64 * h:64 * h + 64] *= f
                    lin.bias[64 * h:64 * h + 64] *= f
    return m


def attention_logit_range(m, tok, seg, mask, layer=0):
    """(min, max) of q.k / 8 over the real keys of `layer` (what the heavy-tailed weights are calibrated for)"""
    with torch.no_grad():
        hs = m(tok, token_type_ids=seg, attention_mask=mask, output_hidden_states=True).hidden_states[layer]
        att = m.encoder.layer[layer].attention.self
        b, l, _ = hs.shape
        q = att.query(hs).view(b, l, 12, 64).transpose(1, 2)
        k = att.key(hs).view(b, l, 12, 64).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) / 8.0
        s = s.masked_fill(~mask.bool()[:, None, None, :], 0.0)
        return float(s.min()), float(s.max())
