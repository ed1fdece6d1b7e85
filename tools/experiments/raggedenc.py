"""encode_to_pool on documents of uneven token length (abstracts: 60 .. 500 tokens), the reference's batches of 32 in corpus order
against sort_by_length (documents of all batches regrouped by length before encoding).   python tools/experiments/raggedenc.py [n_docs]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from transformers import BertConfig, BertModel
from aspire_amd import AspireConSent

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = 8
rng = np.random.default_rng(0)
# token lengths roughly like abstracts: log-normal around 220, clipped to [60, 500]
lens = np.clip(np.exp(rng.normal(np.log(220), 0.35, n_docs)).astype(int), 60, 500)
batches = []
for lo in range(0, n_docs, 32):
    ls = lens[lo:lo + 32]
    b, L = len(ls), int(ls.max())
    tok = torch.zeros(b, L, dtype=torch.long)
    mask = torch.zeros(b, L, dtype=torch.long)
    idxs = []
    for i, n in enumerate(ls):
        tok[i, :n] = torch.from_numpy(rng.integers(1000, 30000, n))
        mask[i, :n] = 1
        edges = np.linspace(1, n - 1, S + 1).astype(int)
        idxs.append([list(range(edges[s], edges[s + 1])) for s in range(S)])
    batches.append(({'tokid_tt': tok, 'seg_tt': torch.zeros_like(tok), 'attnmask_tt': mask, 'seq_lens': [int(n) for n in ls]}, [S] * b, idxs))
torch.manual_seed(0)
model = AspireConSent(bert_model=BertModel(BertConfig(vocab_size=31090), add_pooling_layer=False).eval())
pad_corpus = sum(len(a) * bb['tokid_tt'].shape[1] for bb, a, _ in batches)
print(f'{n_docs} docs, {int(lens.sum())} real tokens, {pad_corpus} token rows in the given batches of 32 ({pad_corpus / lens.sum():.2f} x)')
res = {}
for sort in (False, True, False, True):
    model.encode_to_pool(batches[:8], sort_by_length=sort)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pool = model.encode_to_pool(batches, sort_by_length=sort)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res[sort] = pool.repset.rows
    print(f'sort_by_length={sort}: {dt:.3f} s = {n_docs / dt:.0f} docs/s')
print('max |difference| between the two stores:', (res[True] - res[False]).abs().max().item())
