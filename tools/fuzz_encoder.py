"""Randomised encoder parity: HipBertEncoder against HuggingFace BertModel (fp32, CPU) on random (layers, batch, length) with
ragged masks -- row counts around the 1024-row switch to the fp16-plane path, lengths that are no multiple of 4 / 128, key tiles
on and past their edges.   python tools/fuzz_encoder.py [n_cases] [seed] [heavy]
`heavy`: the models carry the statistics of a trained checkpoint (tests/heavy_bert.py: LayerNorm gains of 10 - 30, outlier hidden
dimensions at 30 - 100, attention logits of +-50, FFN rows x 20, O(1) biases) and the bar is the distance from HuggingFace FLOAT64:
max(1e-4, 1.5 x the fp32 CPU forward's own distance from float64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from transformers import BertConfig, BertModel
from aspire_amd.encoder import HipBertEncoder

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
heavy = len(sys.argv) > 3 and sys.argv[3] == 'heavy'
if heavy:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
    from heavy_bert import heavy_tailed_bert
rng = np.random.default_rng(seed)
models = {}


def model(n_layers):
    if n_layers not in models:
        torch.manual_seed(100 + n_layers)
        if heavy:
            m = heavy_tailed_bert(n_layers, seed=100 + n_layers)
        else:
            m = BertModel(BertConfig(vocab_size=3000, num_hidden_layers=n_layers), add_pooling_layer=False).eval()
            with torch.no_grad():
                for n, p in m.named_parameters():
                    if 'LayerNorm' in n or n.endswith('.bias'):
                        p.add_(0.1 * torch.randn_like(p))
        models[n_layers] = (m, HipBertEncoder(m))
        assert not heavy or models[n_layers][1]._w.planes
    return models[n_layers]


worst = 0.0
for case in range(n_cases):
    n_layers = int(rng.choice([1, 2, 3, 12] if heavy else [1, 2, 3]))
    L = int(rng.choice([5, 31, 64, 127, 128, 129, 200, 255, 256, 257, 384, 511, 512]))
    rows = int(rng.choice([300, 1000, 1023, 1024, 1025, 1500, 2047, 2049, 4000]))
    B = max(1, rows // L)
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    tok = torch.randint(5, 3000, (B, L), generator=g)
    lens = torch.randint(max(1, L // 4), L + 1, (B,), generator=g)
    lens[int(rng.integers(B))] = L
    mask = (torch.arange(L)[None, :] < lens[:, None]).long()
    tok = tok * mask
    seg = torch.zeros_like(tok)
    m, enc = model(n_layers)
    with torch.no_grad():
        want = m(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state
    got = enc(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state.cpu()
    assert torch.isfinite(got).all(), (case, n_layers, B, L)
    real = mask.bool()
    bar = 1e-4
    if heavy:
        with torch.no_grad():
            w64 = m.double()(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state
            m.float()
        bar = max(1e-4, 1.5 * float((want.double() - w64)[real].abs().max()))
        e = float((got.double() - w64)[real].abs().max())
    else:
        e = float((got - want)[real].abs().max())
    worst = max(worst, e / bar * 1e-4)
    assert e <= bar, (case, n_layers, B, L, e, bar)
    print(f'case {case}: layers={n_layers} B={B} L={L} rows={B * L} max|err| on real tokens {e:.2e} ok', flush=True)
print(f'{n_cases} cases ok; worst {worst:.2e}')
