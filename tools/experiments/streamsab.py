"""A/B of encode_to_pool's forwards in flight: one forward at a time against two / three on their own streams, at
8 192 and 16 384 token rows per forward; config 5's documents (256 tokens, 12 sentences), alternating, one box.
  python tools/experiments/streamsab.py [N_DOCS] [ROUNDS]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from tools.e2ebench import synthetic_batches


def main():
    from transformers import BertConfig, BertModel
    from aspire_amd.consent import AspireConSent
    n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    torch.manual_seed(0)
    model = AspireConSent(bert_model=BertModel(BertConfig(vocab_size=31090), add_pooling_layer=False).eval())
    batches = synthetic_batches(n_docs, 256, 12, 2)
    for bb, _, _ in batches:
        for key in ('tokid_tt', 'seg_tt', 'attnmask_tt'):
            bb[key] = bb[key].cuda()
    forms = [(1, 16384), (2, 8192), (2, 16384), (3, 8192), (1, 32768), (2, 32768)]
    ref = None
    for s, r in forms:                                    # warm-up: workspaces, clocks
        model.encode_to_pool(batches[:8 * s], streams=s, rows_per_forward=r)
    torch.cuda.synchronize()
    for rnd in range(rounds):
        for s, r in forms:
            t0 = time.perf_counter()
            pool = model.encode_to_pool(batches, streams=s, rows_per_forward=r)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if ref is None:
                ref = pool.repset.rows.clone()
            same = bool(torch.equal(ref, pool.repset.rows))
            print(f'round {rnd}  streams {s}  rows/forward {r:6d}: {n_docs / dt:8.1f} docs/s  ({dt * 1e3:7.1f} ms)  same bits as the first run: {same}', flush=True)
    print('status word:', model.bert_encoder.status())


if __name__ == '__main__':
    main()
