"""A0: batch preparation for the contextual sentence encoder (host side, integer only).

Same contract as the reference's ``prepare_abstracts`` / ``prepare_bert_sentences``
(examples/ex_aspire_consent.py:107-212; originals src/learning/batchers.py:525-630):

  * sentence 0 of every document is ``TITLE + ' [SEP] '``; its token positions are NOT returned
  * positions count from 1 (position 0 is [CLS])
  * at most 500 word pieces per document: the sentence that crosses the cap is kept only up to the cap
    (dropped entirely if nothing of it fits) and everything after it is discarded
  * ids / segment ids / attention mask are right-padded with ``tokenizer.pad_token_id``

Besides the reference's return values, ``spans_to_csr`` flattens the ragged index lists into the
(tok_idx, span_off) arrays aspire_span_mean_pool_f32 consumes.
"""
import torch

MAX_NUM_TOKS = 500


def _with_special_tokens(tokenizer, ids):
    # transformers 4.5.1 (the reference's pin) has build_inputs_with_special_tokens; transformers >= 5
    # removed it from BertTokenizer.  For a single sequence both mean [CLS] ids [SEP].
    build = getattr(tokenizer, 'build_inputs_with_special_tokens', None)
    if build is not None:
        return build(token_ids_0=ids)
    return [tokenizer.cls_token_id] + ids + [tokenizer.sep_token_id]


def _word_pieces(tokenizer, sents, want_text=True):
    """tokenizer.tokenize + convert_tokens_to_ids of every sentence (ex_aspire_consent.py:135-137), as (pieces, ids) per sentence.
    A fast (Rust) tokenizer takes all the sentences of the batch in ONE call -- the same encode its tokenize() runs per sentence --
    instead of one Python round trip each: the reference's loop prepares ~700 documents/s per core, a ninth of what one GPU encodes."""
    if getattr(tokenizer, 'is_fast', False) and sents:
        ids = tokenizer(list(sents), add_special_tokens=False, return_attention_mask=False, return_token_type_ids=False,
                        verbose=False)['input_ids']
        # (the word-piece strings only where the caller returns them: prepare_abstracts does not)
        return [(tokenizer.convert_ids_to_tokens(x) if want_text else x, list(x)) for x in ids]
    out = []
    for sent in sents:
        pieces = tokenizer.tokenize(sent)
        out.append((pieces, tokenizer.convert_tokens_to_ids(pieces)))
    return out


def prepare_bert_sentences(batch_doc_sents, tokenizer, want_text=True):
    """
    :param batch_doc_sents: list(list(string)); per document: title sentence then abstract sentences.
    :param want_text: False: batch_tokenized_text comes back as ids instead of word-piece strings (prepare_abstracts drops it).
    :return: bert_batch dict('tokid_tt', 'seg_tt', 'attnmask_tt', 'seq_lens'),
             batch_tokenized_text list(list(string)),
             batch_sent_token_idxs list(list(list(int))) -- title excluded.
    """
    docs_ids, docs_text, docs_spans = [], [], []
    tokenized = iter(_word_pieces(tokenizer, [sent for doc_sents in batch_doc_sents for sent in doc_sents], want_text))
    for doc_sents in batch_doc_sents:
        ids, text, spans = [], [], []
        used = 0
        doc_pieces = [next(tokenized) for _ in doc_sents]
        for pieces, piece_ids in doc_pieces:
            room = MAX_NUM_TOKS - used
            keep = min(len(pieces), room)
            overflow = len(pieces) > room
            if keep > 0 or not overflow:
                # (an empty sentence that fits still gets its empty span, like the reference)
                spans.append(list(range(used + 1, used + 1 + keep)))
                text.extend(pieces[:keep])
                ids.extend(piece_ids[:keep])
            if overflow:
                break
            used += keep
        docs_text.append(text)
        docs_spans.append(spans[1:])
        docs_ids.append(_with_special_tokens(tokenizer, ids))
    seq_lens = [len(x) for x in docs_ids]
    max_seq_len = max(seq_lens) if seq_lens else 0
    pad = tokenizer.pad_token_id
    tok, seg, att = [], [], []
    for ids in docs_ids:
        n_pad = max_seq_len - len(ids)
        tok.append(ids + [pad] * n_pad)
        seg.append([0] * len(ids) + [pad] * n_pad)
        att.append([1] * len(ids) + [pad] * n_pad)
    bert_batch = {'tokid_tt': torch.tensor(tok), 'seg_tt': torch.tensor(seg), 'attnmask_tt': torch.tensor(att),
                  'seq_lens': seq_lens}
    return bert_batch, docs_text, docs_spans


def prepare_abstracts(batch_abs, pt_lm_tokenizer):
    """
    :param batch_abs: list(dict) with 'TITLE' (str) and 'ABSTRACT' (list of sentence strings).
    :return: bert_batch, abs_lens list(int), sent_token_idxs list(list(list(int)))
    """
    batch_abs_seqs = [[ex_abs['TITLE'] + ' [SEP] '] + list(ex_abs['ABSTRACT']) for ex_abs in batch_abs]
    bert_batch, _, sent_token_idxs = prepare_bert_sentences(batch_doc_sents=batch_abs_seqs, tokenizer=pt_lm_tokenizer, want_text=False)
    abs_lens = []
    for abs_sent_tok_idxs in sent_token_idxs:
        num_sents = len(abs_sent_tok_idxs)
        abs_lens.append(num_sents)
        assert (num_sents > 0)   # ex_aspire_consent.py:210
    return bert_batch, abs_lens, sent_token_idxs


def spans_to_csr(batch_senttok_idxs, max_sents):
    """Ragged [B][<=S][tokens] position lists -> (tok_idx int32 [N], span_off int32 [B*S+1]) on the host.
    Slots beyond a document's sentence count are empty (they pool to exact zeros)."""
    flat, off = [], [0]
    for doc in batch_senttok_idxs:
        for s in range(max_sents):
            if s < len(doc):
                flat.extend(doc[s])
            off.append(len(flat))
    return (torch.tensor(flat if flat else [0], dtype=torch.int32)[:len(flat)].contiguous(),
            torch.tensor(off, dtype=torch.int32))
