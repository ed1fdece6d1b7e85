"""Adversarial check of the attention kernels' mask handling (round 6): attention masks with holes (not the contiguous prefix prepare_abstracts makes), one document with a
single real token, lengths off the tile edges -- HipBertEncoder (plane path: flash_attn_p_kernel, its 64-key form, round 5's kernel) against HuggingFace fp32 on every
position.   python tools/experiments/randmask.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
from test_gpu_encoder import _bert
from aspire_amd.encoder import HipBertEncoder
from aspire_amd._lib import pinned

worst = 0.0
for n_layers, b, l, seed in ((2, 8, 128, 1), (1, 5, 300, 2), (2, 16, 64, 3), (1, 3, 511, 4), (3, 12, 130, 5)):
    m = _bert(n_layers, seed=seed)
    enc = HipBertEncoder(m)
    g = torch.Generator().manual_seed(100 + seed)
    tok = torch.randint(5, 3000, (b, l), generator=g)
    mask = (torch.rand(b, l, generator=g) < 0.7).long()
    mask[:, 0] = 1
    mask[1] = 0
    mask[1, 0] = 1                       # one real token
    mask[2] = 1                          # a full document
    seg = torch.zeros_like(tok)
    with torch.no_grad():
        want = m(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state
    for form in ('', 'p64', 'f16x2', 'f32'):
        with pinned(GEMM='planes', **({'ATTN': form} if form else {})):
            got = enc.forward_hidden(tok, seg, mask).cpu()
        e = float((got - want).abs().max())
        worst = max(worst, e)
        print(f'layers {n_layers} B {b} L {l} attention {form or "default"}: max |HIP - HF| over ALL positions {e:.2e}' + ('' if e < 1e-4 else '   <<<<<< FAIL'), flush=True)
print('worst', worst)
