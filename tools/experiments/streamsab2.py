import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from tools.e2ebench import synthetic_batches
from transformers import BertConfig, BertModel
from aspire_amd.consent import AspireConSent
torch.manual_seed(0)
model = AspireConSent(bert_model=BertModel(BertConfig(vocab_size=31090), add_pooling_layer=False).eval())
batches = synthetic_batches(8192, 256, 12, 2)
for bb, _, _ in batches:
    for key in ('tokid_tt', 'seg_tt', 'attnmask_tt'):
        bb[key] = bb[key].cuda()
forms = [(1, 16384), (2, 16384), (2, 8192)]
for s, r in forms:
    model.encode_to_pool(batches[:8 * s], streams=s, rows_per_forward=r)
torch.cuda.synchronize()
for rnd in range(3):
    for s, r in forms:
        t0 = time.perf_counter(); model.encode_to_pool(batches, streams=s, rows_per_forward=r); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f'{os.environ.get("ASPIRE_HIP_ATTN","default")} {os.environ.get("ASPIRE_HIP_LIB","product")[-30:]} round {rnd} streams {s} rows {r}: {8192/dt:8.1f} docs/s', flush=True)
