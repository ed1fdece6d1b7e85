"""12-layer encoder error against HF BertModel (fp32 CPU) for the two GEMM forms.  usage: python tools/encerr.py [B L]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from aspire_amd import _lib
from aspire_amd.encoder import HipBertEncoder
from test_gpu_pipeline import _bert

b, l = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (8, 256)
m = _bert(12, seed=4)
g = torch.Generator().manual_seed(l)
tok = torch.randint(5, 3000, (b, l), generator=g)
lens = torch.randint(l // 2, l + 1, (b,), generator=g)
lens[0] = l
mask = (torch.arange(l)[None, :] < lens[:, None]).long()
tok = tok * mask
with torch.no_grad():
    want = m(tok, token_type_ids=torch.zeros_like(tok), attention_mask=mask).last_hidden_state
    want64 = m.double()(tok, token_type_ids=torch.zeros_like(tok), attention_mask=mask).last_hidden_state
m.float()
for form in ('f32', 'bf16x3'):
    with _lib.pinned(GEMM=form):
        got = HipBertEncoder(m)(tok, token_type_ids=torch.zeros_like(tok), attention_mask=mask).last_hidden_state.cpu()
    e32 = (got - want).abs()[mask.bool()].max().item()
    e64 = (got.double() - want64).abs()[mask.bool()].max().item()
    print(f'GEMM={form:7s}: max |hip - HF fp32| {e32:.2e}   max |hip - HF float64| {e64:.2e}')
print(f'HF fp32 vs HF float64: {(want.double() - want64).abs()[mask.bool()].max().item():.2e}')
