import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from test_gpu_encoder import _bert
from aspire_amd.encoder import HipBertEncoder
m = _bert(2, seed=3); enc = HipBertEncoder(m)
g = torch.Generator().manual_seed(9)
for b, l in ((8, 128), (3, 77), (2, 16)):
    tok = torch.randint(5, 3000, (b, l), generator=g); seg = torch.randint(0, 2, (b, l), generator=g); mask = torch.ones_like(tok)
    with torch.no_grad(): want = m(tok, token_type_ids=seg, attention_mask=mask).last_hidden_state
    got = enc.forward_hidden(tok, seg, mask).cpu(); print(b, l, 'seg random', float((got-want).abs().max()))
    with torch.no_grad(): want = m(tok, attention_mask=mask).last_hidden_state
    got = enc.forward_hidden(tok, None, mask).cpu(); print(b, l, 'seg None', float((got-want).abs().max()))
    got = enc.forward_hidden(tok, None, None).cpu(); print(b, l, 'mask None', float((got-want).abs().max()))
