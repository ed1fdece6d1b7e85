"""AspireConSent: contextual sentence encoder, drop-in for the class of the same name in
examples/ex_aspire_consent.py:25-101 and examples/ex_aspire_consent_multimatch.py:30-106
(original: WordSentAlignBiEnc.partial_forward / consent_reps_bert,
src/learning/facetid_models/disent_models.py:470-535).

    model = AspireConSent('allenai/aspire-contextualsentence-multim-compsci')
    bert_batch, abs_lens, sent_token_idxs = prepare_abstracts(batch_abs, tokenizer)
    clsreps, contextual_sent_reps = model.forward(bert_batch, abs_lens, sent_token_idxs)

The BERT forward and the span pooling both run in libaspire_hip.so; outputs are returned on the device of
``bert_batch['tokid_tt']`` (CPU tensors in, CPU tensors out, as in the reference's examples).

Beyond the drop-in ``forward`` the class carries the callers' encode steps in two forms:
  * the reference's own, host-out ones -- ``caching_encode`` (WordSentAlignBiEnc.caching_encode,
    src/learning/facetid_models/disent_models.py:344-371) and ``encode`` (AspireModel.encode,
    src/evaluation/utils/models.py:199-209): un-padded per-document reps;
  * the device-resident one -- ``forward_device`` / ``encode_to_pool``: token ids go in, the pooled sentence reps are
    written by the pooling kernel straight into a rows + CSR rep store in HBM (no padding rows, no host round trip) and
    come back as a ``CandidatePool`` that the scoring calls read in place.  1 M documents x 12 sentences = 36.9 GB fit one
    MI355X's 288 GB seven times over.
"""
import numpy as np
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
        if self.bert_encoder.status():
            # a LayerNorm-epilogue GEMM gave up waiting for its row block (encoder.hip: gemm_p_ln_kernel's bounded wait): once more with
            # the LayerNorm as its own pass
            from ._lib import pinned
            import warnings
            warnings.warn('AspireConSent: the fused GEMM + LayerNorm exchange timed out; encoding again with ASPIRE_HIP_GEMM_LN=off')
            with pinned(GEMM_LN='off'):
                final_hidden_state = self.bert_encoder.forward_hidden(tokid_tt, token_type_ids=seg_tt, attention_mask=attnmask_tt,
                                                                      check_ids=False)
            doc_cls_reps, sent_reps = ops.span_mean_pool(final_hidden_state, tok_idx.to(dev), span_off.to(dev), max_sents)
        if not bool(torch.isfinite(sent_reps).all() & torch.isfinite(doc_cls_reps).all()):      # (the CLS token belongs to no sentence span)
            # an activation beyond the fp16 planes' range (encoder.py: forward_full_range): once more on the full-range kernels
            import warnings
            warnings.warn('AspireConSent.forward: non-finite sentence reps on the fp16-plane encoder path (an activation beyond 65504); '
                          'encoding the batch again with ASPIRE_HIP_GEMM=bf16x3, ASPIRE_HIP_ATTN=f32')
            final_hidden_state = self.bert_encoder.forward_full_range(tokid_tt, seg_tt, attnmask_tt)
            doc_cls_reps, sent_reps = ops.span_mean_pool(final_hidden_state, tok_idx.to(dev), span_off.to(dev), max_sents)
        # the reference squeezes and re-unsqueezes (:76, :46-49): shapes are [B,768] and [B,S,768] for every B.
        return doc_cls_reps.to(out_dev), sent_reps.to(out_dev)

    # ---- the callers' encode steps -------------------------------------------------------------------------------------
    def forward_device(self, bert_batch, abs_lens, sent_tok_idxs):
        """``forward`` with the outputs left on the GPU: (doc_cls_reps [B, 768], sent_reps [B, max_sents, 768])."""
        gpu = ops.require_gpu()
        dev_batch = dict(bert_batch)
        for k in ('tokid_tt', 'seg_tt', 'attnmask_tt'):
            dev_batch[k] = bert_batch[k].to(gpu)
        return self.consent_reps_bert(bert_batch=dev_batch, num_sents=abs_lens, batch_senttok_idxs=sent_tok_idxs)

    def caching_encode(self, batch_dict):
        """WordSentAlignBiEnc.caching_encode (disent_models.py:344-371): batch_dict with 'bert_batch', 'abs_lens',
        'senttok_idxs' -> list of {'doc_cls_reps': np [768], 'sent_reps': np [num_sents, 768]} (un-padded)."""
        doc_cls_reps, sent_reps = self.forward_device(batch_dict['bert_batch'], batch_dict['abs_lens'], batch_dict['senttok_idxs'])
        sent_reps = sent_reps.cpu().numpy()
        doc_cls_reps = doc_cls_reps.cpu().numpy()
        return [{'doc_cls_reps': doc_cls_reps[i, :], 'sent_reps': sent_reps[i, :num_sents, :]}
                for i, num_sents in enumerate(batch_dict['abs_lens'])]

    def encode(self, batch_papers, pt_lm_tokenizer):
        """AspireModel.encode (src/evaluation/utils/models.py:199-209): papers -> list of [abs_len, 768] tensors."""
        from .batch_prep import prepare_abstracts
        bert_batch, abs_lens, sent_token_idxs = prepare_abstracts(batch_abs=batch_papers, pt_lm_tokenizer=pt_lm_tokenizer)
        _, batch_reps_sent = self.forward(bert_batch=bert_batch, abs_lens=abs_lens, sent_tok_idxs=sent_token_idxs)
        return [batch_reps_sent[i, :abs_lens[i]] for i in range(len(abs_lens))]

    @staticmethod
    def _merge_batches(batches, docs_per_forward):
        """Consecutive prepare_abstracts batches joined into forwards of up to docs_per_forward documents: token tensors padded
        (id 0, mask 0) to the group's longest sequence -- a padded key weighs exactly 0 in the attention and every other step is
        per token row, so a document's reps do not depend on what it is batched with.  (The reference encodes 32 at a time,
        pp_gen_nearest.py:141-160, for its GPU's memory; at 64 x 256 tokens the GEMMs' tile counts fill whole rounds of the chip:
        B = 32 5 900 docs/s, B = 64 6 470.)"""
        out, cur = [], []

        def flush():
            if len(cur) == 1:
                out.append(cur[0])
            elif cur:
                lmax = max(bb['tokid_tt'].shape[1] for bb, _, _ in cur)
                pad = lambda t: torch.nn.functional.pad(t, (0, lmax - t.shape[1]))
                merged = {k: torch.cat([pad(bb[k]) for bb, _, _ in cur], 0) for k in ('tokid_tt', 'seg_tt', 'attnmask_tt')}
                merged['seq_lens'] = [n for bb, _, _ in cur for n in bb['seq_lens']]
                out.append((merged, [n for _, a, _ in cur for n in a], [i for _, _, idx in cur for i in idx]))
            cur.clear()

        n = 0
        for item in batches:
            b = len(item[1])
            if cur and n + b > docs_per_forward:
                flush()
                n = 0
            cur.append(item)
            n += b
        flush()
        return out

    @staticmethod
    def _regroup_by_length(batches, docs_per_forward, window=8192, rows_per_forward=None):
        """All windows of _regroup_windows at once: (forwards, ids)."""
        out, ids = [], []
        for o, i in AspireConSent._regroup_windows(batches, docs_per_forward, window, rows_per_forward):
            out += o
            ids += i
        return out, ids

    @staticmethod
    def _regroup_windows(batches, docs_per_forward, window=8192, rows_per_forward=None, first_window=None):
        """The documents of all prepare_abstracts batches regrouped into forwards of docs_per_forward documents of SIMILAR token
        length (longest first), each padded to its own longest sequence: on abstracts of 100 - 500 tokens the reference's batches
        in corpus order (pp_gen_nearest.py:141-160) spend a third and more of the encoder's work on pad tokens.  A document's reps
        do not depend on what it is batched with (_merge_batches).  Returns (batches, ids): ids[g][j] = the corpus position of
        document j of group g -- its rows of the store stay where the corpus order puts them.  Documents are sorted inside windows
        of ~`window` consecutive documents (whole batches): the host never holds more than a window's token tensors twice.
        A GENERATOR, one (forwards, ids) pair per window: encode_to_pool regroups window w + 1 while the GPU encodes window w, and asks
        for a small first window (`first_window` documents) so that the first encoder call goes out after a few milliseconds of host work
        instead of after the regrouping of the whole corpus (~2 ms per 1000 documents: a few thousand small tensor operations).
        rows_per_forward: a forward takes as many documents as fit that many TOKEN ROWS at its longest document's length instead of
        a fixed docs_per_forward (the encoder's GEMMs fill the chip's workgroup slots in whole rounds at 16 384 rows: 64 x 256,
        128 x 128 and 256 x 64 all run at ~337 TFLOP/s; 64 x 128 at 297, 128 x 64 at 289)."""
        b0, doc0 = 0, 0
        while b0 < len(batches):
            out, ids = [], []
            b1, n = b0, 0
            limit = first_window if (first_window and b0 == 0) else window
            while b1 < len(batches) and (n == 0 or n + len(batches[b1][1]) <= limit):
                n += len(batches[b1][1])
                b1 += 1
            part = batches[b0:b1]
            lmax = max(bb['tokid_tt'].shape[1] for bb, _, _ in part)
            pad = lambda t: torch.nn.functional.pad(t, (0, lmax - t.shape[1]))
            big = {k: torch.cat([pad(bb[k]) for bb, _, _ in part], 0) for k in ('tokid_tt', 'seg_tt', 'attnmask_tt')}
            seq_lens = [int(v) for bb, _, _ in part for v in bb['seq_lens']]
            abs_lens = [v for _, a, _ in part for v in a]
            spans = [i for _, _, idx in part for i in idx]
            order = sorted(range(n), key=lambda d: -seq_lens[d])                  # stable: equal lengths keep corpus order
            as_given = order == list(range(n))                                     # documents of one length: slices, no gather
            order_t = None if as_given else torch.tensor(order, dtype=torch.long).to(big['tokid_tt'].device)      # ONE upload per window
            g0 = 0
            while g0 < n:
                # (longest first: the group's first document sets its padded length)
                take = max(1, int(rows_per_forward) // max(1, seq_lens[order[g0]])) if rows_per_forward else docs_per_forward
                sel = order[g0:g0 + take]
                L = max(seq_lens[d] for d in sel)
                if as_given:
                    bb = {k: big[k][g0:g0 + take, :L].contiguous() for k in big}
                else:
                    bb = {k: big[k].index_select(0, order_t[g0:g0 + take])[:, :L].contiguous() for k in big}
                bb['seq_lens'] = [seq_lens[d] for d in sel]
                out.append((bb, [abs_lens[d] for d in sel], [spans[d] for d in sel]))
                ids.append([doc0 + d for d in sel])
                g0 += take
            b0, doc0 = b1, doc0 + n
            yield out, ids

    def encode_to_pool(self, batches, pids=None, want_cls=False, docs_per_forward=64, planes=False, sort_by_length=True,
                       _full_range=False, stage_events=None, rows_per_forward=16384, streams=1, _ln_off=False):
        """Encode document batches straight into a resident candidate pool.

        batches: iterable of (bert_batch, abs_lens, sent_tok_idxs) as prepare_abstracts returns them (it is consumed
        twice when it is a list; a generator is materialised).  The store's row matrix [sum(abs_lens), 768] is allocated
        once in HBM; for every batch the encoder runs and the pooling kernel writes each sentence's mean straight into the
        document's rows (aspire_span_mean_pool_rows_f32) -- no padded tensor, no copy back to the host.
        docs_per_forward: consecutive batches are joined into encoder calls of up to this many documents (_merge_batches; None or 0:
        one call per batch as given).
        sort_by_length (default; needs docs_per_forward): the documents of the batches regrouped by token length before encoding
        (_regroup_by_length: fewer pad tokens per forward -- 4096 abstract-length documents 3 320 -> 6 100 docs/s, the same bits;
        documents of one length keep the given grouping).  The store keeps the corpus order.  False: consecutive batches joined as
        given (_merge_batches).
        rows_per_forward (with sort_by_length; None: docs_per_forward documents per forward whatever their length): a forward takes
        as many documents as fit this many token rows at its longest document's length -- 64 documents of 256 tokens, 128 of 128,
        32 of 512: the shapes at which the encoder's GEMMs fill whole rounds of the chip.
        planes: also keep the rows as fp16 planes (CandidatePool.prepare_planes: one more pass over the finished store, ~2.5 ms per
        GB) for the many-query cost tiles.
        streams: encoder calls alternate over this many HIP streams (each with its own encoder workspace; forked from and joined to the
        current stream inside this call): a forward's one-round launches end with their tiles' epilogues under an idle matrix pipe, a
        second forward in flight fills those slots.  1: everything on the current stream.
        stage_events: a list that receives one (start, encoded, pooled) triple of HIP events per encoder call, recorded on the
        current stream INSIDE this call (tools/e2ebench.py: the stage's own time split; the caller reads them after a sync).
        Returns a scorer.CandidatePool (and the [N, 768] CLS reps on the GPU with want_cls)."""
        from .scorer import CandidatePool
        dev = ops.require_gpu()
        batches = list(batches)
        given = batches
        all_lens = [int(n) for _, abs_lens, _ in batches for n in abs_lens]       # corpus order
        n_docs, total = len(all_lens), int(sum(all_lens))
        lens_t = torch.tensor(all_lens, dtype=torch.int32)
        start_t = (torch.cumsum(lens_t, 0) - lens_t).to(torch.int32)
        rows = torch.empty(max(total, 1), 768, device=dev, dtype=torch.float32)[:total]
        cls_all = torch.empty(n_docs, 768, device=dev, dtype=torch.float32) if want_cls else None
        # token ids of every batch validated with ONE device round trip (nn.Embedding raises IndexError on the reference path);
        # per batch that check is a host sync in front of every encoder call
        if batches:
            # (64 batches' ids flattened into one tensor per reduction: per batch it was two tiny kernels each, ~1000 launches in
            # front of the first encoder call of a 16 384-document corpus)
            lo_hi = torch.stack([torch.stack(torch.aminmax(torch.cat([bb['tokid_tt'].reshape(-1) for bb, _, _ in batches[i:i + 64]]))).to(dev)
                                 for i in range(0, len(batches), 64)])
            if int(lo_hi[:, 0].min()) < 0 or int(lo_hi[:, 1].max()) >= self.bert_encoder.config.vocab_size:
                raise IndexError('token id out of range')
        # Per group of batches: (host) every batch's pooling tables -- token positions (CSR), slot -> store row -- uploaded as ONE int32
        # buffer, then the group's kernels.  (Uploaded batch by batch from pageable memory, each copy waited behind the encoder
        # kernels queued before it, and the next batch's kernels were launched late: the GPU sat idle ~0.5 ms per batch of 32
        # documents, 5 % of the encode stage.)  A group's tables are built while the GPU works on the group before it; the first
        # groups are small so that it starts at once (all tables up front: ~80 ms of host loops per 16 384 documents with the GPU idle).
        # The forwards, one window of the corpus at a time (a generator: the next window is regrouped while the GPU encodes this one; the
        # first window is small so that the GPU starts after a few milliseconds of host work).  doc_ids: per forward the corpus positions
        # of its documents (None: consecutive).
        if sort_by_length and docs_per_forward and batches:
            window_iter = self._regroup_windows(batches, docs_per_forward, rows_per_forward=rows_per_forward, first_window=1024)
        elif docs_per_forward:
            window_iter = iter([(self._merge_batches(batches, docs_per_forward), None)])
        else:
            window_iter = iter([(batches, None)])
        start_np = start_t.numpy()
        doc0 = 0
        first = True
        cur = torch.cuda.current_stream()
        n_streams = max(1, int(streams))
        side = [cur] if n_streams == 1 else [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
        if n_streams > 1:
            fork = cur.record_event()            # the store's allocation and the id check precede every side stream's work
            for st in side:
                st.wait_event(fork)
        n_fwd = 0
        for batches, doc_ids in window_iter:
            bounds, nxt = [0], 2 if first else 64
            first = False
            while bounds[-1] < len(batches):
                bounds.append(min(len(batches), bounds[-1] + nxt))
                nxt = min(64, nxt * 4)
            for g0, g1 in zip(bounds[:-1], bounds[1:]):
                chunk, parts, offs, o = [], [], [], 0
                for bi in range(g0, g1):
                    bert_batch, abs_lens, sent_tok_idxs = batches[bi]
                    b = len(abs_lens)
                    ids = np.asarray(doc_ids[bi], dtype=np.int64) if doc_ids is not None else np.arange(doc0, doc0 + b)
                    max_sents = max(abs_lens)
                    max_seq_len = max(bert_batch['seq_lens'])
                    assert bert_batch['tokid_tt'].shape == (b, max_seq_len)
                    tok_idx, span_off = spans_to_csr(sent_tok_idxs, max_sents)
                    if tok_idx.numel() and (int(tok_idx.min()) < 0 or int(tok_idx.max()) >= max_seq_len):
                        raise IndexError('sentence token index out of range')
                    # slot (b, s) -> row of the store, -1 beyond the document's sentence count
                    lens_b = np.asarray(abs_lens, dtype=np.int32)[:, None]
                    slot = np.arange(max_sents, dtype=np.int32)[None, :]
                    out_row = np.where(slot < lens_b, start_np[ids, None] + slot, -1).astype(np.int32)
                    for arr in (tok_idx.numpy(), span_off.numpy(), out_row.reshape(-1)):
                        offs.append((o, o + arr.size))
                        parts.append(arr.astype(np.int32, copy=False))
                        o += arr.size
                    chunk.append((max_sents, doc0, b))
                    doc0 += b
                flat = torch.from_numpy(np.concatenate(parts) if parts else np.zeros(0, np.int32)).to(dev)
                if n_streams > 1:
                    up = cur.record_event()          # the tables' upload happens on the caller's stream
                    for st in side:
                        st.wait_event(up)
                        flat.record_stream(st)
                for i, (bert_batch, _, _) in enumerate(batches[g0:g1]):
                    max_sents, d0, b = chunk[i]
                    (t0, t1), (s0, s1), (r0, r1) = offs[3 * i:3 * i + 3]
                    with torch.cuda.stream(side[n_fwd % n_streams]):
                        n_fwd += 1
                        if stage_events is not None:
                            evs = tuple(torch.cuda.Event(enable_timing=True) for _ in range(3))
                            evs[0].record()
                        hidden = self.bert_encoder.forward_hidden(bert_batch['tokid_tt'], token_type_ids=bert_batch['seg_tt'],
                                                                  attention_mask=bert_batch['attnmask_tt'], check_ids=False)
                        if stage_events is not None:
                            evs[1].record()
                        if want_cls and doc_ids is not None:       # regrouped documents: the forward's CLS rows go to their corpus positions
                            cls_b = torch.empty(b, 768, device=dev, dtype=torch.float32)
                            ops.span_mean_pool_rows(hidden, flat[t0:t1], flat[s0:s1], max_sents, flat[r0:r1], rows, cls_b)
                            cls_all.index_copy_(0, torch.tensor(doc_ids[g0 + i], dtype=torch.long, device=dev), cls_b)
                        else:
                            ops.span_mean_pool_rows(hidden, flat[t0:t1], flat[s0:s1], max_sents, flat[r0:r1], rows,
                                                    cls_all[d0:d0 + b] if want_cls else None)
                        if stage_events is not None:
                            evs[2].record()
                            stage_events.append(evs)
        if n_streams > 1:
            for st in side:                          # join: what follows on the caller's stream sees the finished store
                cur.wait_event(st.record_event())
                rows.record_stream(st)
                if cls_all is not None:
                    cls_all.record_stream(st)
        if total and not _ln_off and self.bert_encoder.status():
            # a LayerNorm-epilogue GEMM gave up waiting for its row block (encoder.hip: gemm_p_ln_kernel's bounded wait): the whole corpus
            # once more with the LayerNorm as its own pass (one sync over the finished store, as the range check below)
            from ._lib import pinned
            import warnings
            warnings.warn('AspireConSent.encode_to_pool: the fused GEMM + LayerNorm exchange timed out; encoding again with '
                          'ASPIRE_HIP_GEMM_LN=off')
            with pinned(GEMM_LN='off'):
                if stage_events is not None:
                    del stage_events[:]
                return self.encode_to_pool(given, pids=pids, want_cls=want_cls, docs_per_forward=docs_per_forward, planes=planes,
                                           sort_by_length=sort_by_length, _full_range=_full_range, stage_events=stage_events,
                                           rows_per_forward=rows_per_forward, streams=streams, _ln_off=True)
        if total and not _full_range and not bool(torch.isfinite(rows).all() & (torch.isfinite(cls_all).all() if want_cls else True)):
            # an activation left the fp16 planes' range somewhere (one check over the finished store): encode again on the kernels
            # that take any fp32 value
            from ._lib import pinned
            import warnings
            warnings.warn('AspireConSent.encode_to_pool: non-finite sentence reps on the fp16-plane encoder path; encoding again with '
                          'ASPIRE_HIP_GEMM=bf16x3, ASPIRE_HIP_ATTN=f32')
            with pinned(GEMM='bf16x3', ATTN='f32'):
                if stage_events is not None:
                    del stage_events[:]
                return self.encode_to_pool(given, pids=pids, want_cls=want_cls, docs_per_forward=docs_per_forward, planes=planes,
                                           sort_by_length=sort_by_length, _full_range=True, stage_events=stage_events,
                                           rows_per_forward=rows_per_forward, streams=streams, _ln_off=True)
        repset = ops.DeviceRepSet(rows, start_t.to(dev), lens_t.to(dev), ext=0, max_len=max(all_lens) if all_lens else 0,
                                  lens_host=all_lens)
        pool = CandidatePool.from_repset(repset, pids=pids)
        if planes and total:
            pool.prepare_planes()
        return (pool, cls_all) if want_cls else pool
