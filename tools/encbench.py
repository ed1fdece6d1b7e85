"""Encoder throughput (GPU box): docs/s and fp32-MFMA TFLOP/s of aspire_bert_forward_f32 + pooling."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformers import BertConfig, BertModel
from aspire_amd.encoder import HipBertEncoder

def main():
    B, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 256)
    torch.manual_seed(0)
    m = BertModel(BertConfig(vocab_size=31090), add_pooling_layer=False).eval()
    enc = HipBertEncoder(m)
    tok = torch.randint(1000, 30000, (B, L)).cuda()
    mask = torch.ones_like(tok)
    for _ in range(3): enc.forward_hidden(tok, None, mask)
    torch.cuda.synchronize()
    n = 10
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): enc.forward_hidden(tok, None, mask)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    flops = B * 12 * L * (14155776 + 3072 * L)
    print(f'B={B} L={L}: {ms:.2f} ms/batch  {B/ms*1e3:.0f} docs/s  {flops/ms/1e9:.1f} TFLOP/s ({flops/ms/1e9/157.3*100:.1f}% of 157.3 fp32-MFMA peak)')

main()
