"""Span mean-pooling throughput (GPU box): achieved HBM GB/s of aspire_span_mean_pool_f32.
usage: python tools/poolbench.py [B L S]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import ops
from aspire_amd.batch_prep import spans_to_csr


def main():
    B, L, S = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (256, 512, 12)
    hidden = torch.randn(B, L, 768, device='cuda')
    per = (L - 2) // S
    idxs = [[list(range(1 + s * per, 1 + (s + 1) * per)) for s in range(S)] for _ in range(B)]   # contiguous equal spans
    tok_idx, span_off = spans_to_csr(idxs, S)
    tok_idx, span_off = tok_idx.cuda(), span_off.cuda()
    for _ in range(5):
        ops.span_mean_pool(hidden, tok_idx, span_off, S)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    a.record()
    for _ in range(n):
        ops.span_mean_pool(hidden, tok_idx, span_off, S)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / n * 1e3
    nbytes = 4 * 768 * (L + S) * B         # every token row read once, every sentence rep written once
    print(f'B={B} L={L} S={S}: {us:.1f} us/call  {nbytes / us / 1e3:.0f} GB/s algorithmic ({nbytes / us / 1e3 / 8000 * 100:.1f}% of 8 TB/s)')


if __name__ == '__main__':
    main()
