"""Which columns differ between the fused LayerNorm epilogue and the separate pass (debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from transformers import BertConfig, BertModel
from aspire_amd.encoder import HipBertEncoder
from aspire_amd._lib import pinned
torch.manual_seed(0)
for nl in (1, 2):
    m = BertModel(BertConfig(vocab_size=3000, num_hidden_layers=nl), add_pooling_layer=False).eval()
    enc = HipBertEncoder(m)
    tok = torch.randint(0, 3000, (10, 128)); mask = torch.ones_like(tok)
    with pinned(GEMM='planes', GEMM_LN='on'):
        a = enc(tok, attention_mask=mask).last_hidden_state.cpu()
    with pinned(GEMM='planes', GEMM_LN='off'):
        b = enc(tok, attention_mask=mask).last_hidden_state.cpu()
    d = (a - b).abs().reshape(-1, 768)
    print('layers', nl, 'max', d.max().item())
    print(' by col % 32:', [round(x, 3) for x in d.reshape(-1, 24, 32).amax((0, 1)).tolist()])
    print(' by row % 8 :', [round(x, 3) for x in d.reshape(-1, 8, 768).amax((0, 2)).tolist()])
