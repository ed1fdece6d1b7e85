"""encode_to_pool on documents of uneven token length (abstracts: 60 .. 500 tokens), the reference's batches of 32 in corpus order
against sort_by_length (documents of all batches regrouped by length before encoding; tools/e2ebench.py: run_ragged -- the same
measurement rides in bench.py's e2e block).   python tools/experiments/raggedenc.py [n_docs]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from transformers import BertConfig, BertModel
from aspire_amd import AspireConSent
import e2ebench

torch.manual_seed(0)
model = AspireConSent(bert_model=BertModel(BertConfig(vocab_size=31090), add_pooling_layer=False).eval())
print(json.dumps(e2ebench.run_ragged(model, n_docs=int(sys.argv[1]) if len(sys.argv) > 1 else 4096), indent=1))
