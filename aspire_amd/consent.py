"""AspireConSent: contextual sentence encoder, drop-in for the class of the same name in
examples/ex_aspire_consent.py:25-101 and examples/ex_aspire_consent_multimatch.py:30-106
(original: WordSentAlignBiEnc.partial_forward / consent_reps_bert,
src/learning/facetid_models/disent_models.py:470-535).

    model = AspireConSent('allenai/aspire-contextualsentence-multim-compsci')
    bert_batch, abs_lens, sent_token_idxs = prepare_abstracts(batch_abs, tokenizer)
    clsreps, contextual_sent_reps = model.forward(bert_batch, abs_lens, sent_token_idxs)

The BERT forward and the span pooling both run in libaspire_hip.so; outputs are returned on the device of
``bert_batch['tokid_tt']`` (CPU tensors in, CPU tensors out, as in the reference's examples).
"""
import torch

from . import ops
from .batch_prep import spans_to_csr
from .encoder import HipBertEncoder


class AspireConSent:
    def __init__(self, hf_model_name=None, bert_model=None):
        """
        :param hf_model_name: HuggingFace model name or path, loaded like the reference does (:33).
        :param bert_model: an already constructed transformers BertModel (weights are copied to the GPU).
        """
        self.bert_encoding_dim = 768
        self.bert_layer_count = 12 + 1  # plus 1 for the bottom most layer.
        if bert_model is None:
            from transformers import AutoModel
            bert_model = AutoModel.from_pretrained(hf_model_name)
        self.bert_encoder = HipBertEncoder(bert_model)

    def eval(self):
        return self

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def forward(self, bert_batch, abs_lens, sent_tok_idxs):
        """
        :return: doc_cls_reps [batch_size x 768], sent_reps [batch_size x max_sents x 768]
        """
        doc_cls_reps, sent_reps = self.consent_reps_bert(bert_batch=bert_batch, num_sents=abs_lens,
                                                         batch_senttok_idxs=sent_tok_idxs)
        return doc_cls_reps, sent_reps

    def consent_reps_bert(self, bert_batch, batch_senttok_idxs, num_sents):
        """
        :param bert_batch: dict('tokid_tt', 'seg_tt', 'attnmask_tt', 'seq_lens')
        :param batch_senttok_idxs: list(list(list(int))); batch_size([num_sents_per_abs[num_tokens_in_sent]])
        :param num_sents: list(int); number of sentences in each example in the batch passed.
        """
        seq_lens = bert_batch['seq_lens']
        batch_size, max_seq_len = len(seq_lens), max(seq_lens)
        max_sents = max(num_sents)
        tokid_tt, seg_tt, attnmask_tt = bert_batch['tokid_tt'], bert_batch['seg_tt'], bert_batch['attnmask_tt']
        out_dev = tokid_tt.device
        assert tokid_tt.shape == (batch_size, max_seq_len)
        final_hidden_state = self.bert_encoder.forward_hidden(tokid_tt, token_type_ids=seg_tt,
                                                              attention_mask=attnmask_tt)
        for doc in batch_senttok_idxs:
            for span in doc:
                if span and (min(span) < 0 or max(span) >= max_seq_len):
                    raise IndexError('sentence token index out of range')   # numpy fancy indexing raises too
        tok_idx, span_off = spans_to_csr(batch_senttok_idxs, max_sents)
        dev = final_hidden_state.device
        doc_cls_reps, sent_reps = ops.span_mean_pool(final_hidden_state, tok_idx.to(dev), span_off.to(dev), max_sents)
        # the reference squeezes and re-unsqueezes (:76, :46-49): shapes are [B,768] and [B,S,768] for every B.
        return doc_cls_reps.to(out_dev), sent_reps.to(out_dev)
