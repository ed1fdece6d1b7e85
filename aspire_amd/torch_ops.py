"""PyTorch-ROCm custom ops over the C ABI ("what the build's extension must export").

``import aspire_amd.torch_ops`` registers the operators below in the ``aspire`` namespace with ``torch.library``; every
one is a thin shim over one entry point of include/aspire_hip.h (no arithmetic here), has a GPU ("cuda" = HIP on ROCm)
implementation ONLY -- calling one with CPU tensors fails in the dispatcher, there is no CPU kernel -- and a fake (meta)
implementation so that shapes propagate under FakeTensor / torch.compile tracing.

    torch.ops.aspire.span_mean_pool(hidden, tok_idx, span_off, max_sents) -> (cls, sent)         A2/A3  ex_aspire_consent.py:75-100
    torch.ops.aspire.bert_encoder_forward(ids, type_ids, mask, weights, n_heads, ln_eps) -> hidden  A1   ex_aspire_consent.py:72-73
    torch.ops.aspire.l2max_scores(q, q_lens, c, c_lens, paired) -> scores                          A9   pair_distances.py:138-186
    torch.ops.aspire.ot_sinkhorn_scores(q, q_lens, c, c_lens, blur, scaling, temp, group, want, paired, extras)
                                         -> (scores, q_distr, c_distr, pair_sims, plan)           A5-A8 pair_distances.py:21-92
    torch.ops.aspire.topk_desc(scores, k, idx_base) -> (top_scores, top_idx)                       A12  evaluate.py:76
    torch.ops.aspire.topk_keys(scores, k, idx_base) -> keys          } the shard merge of section 8(e): local top-k in key
    torch.ops.aspire.topk_merge(gathered_keys, k) -> (top_scores, top_idx)  } form, (all-gather by the caller), merge
  resident CSR pools (rows + start + len, struct aspire_repset):
    torch.ops.aspire.l2max_scores_csr / ot_scores_csr(q_rows, q_start, q_len, q_max, c_rows, c_start, c_len, c_max, ...) -> [Q * C]
    torch.ops.aspire.ot_rank_batch(q_rows, q_start, q_len, q_max, c_rows, c_start, c_len, c_max, job_off, max_job, k, ...)
                                         -> (scores [C], top_scores [J, k], top_idx [J, k])        evaluate.py:58-76, batched

Padded inputs are [n, S, 768] fp32 with int32 lens [n] (the reference's RepLen after its permute, disent_models.py:15);
``paired`` False scores every query against every candidate ([Q * C], query-major), True scores pair p (Q == C).
"""
import ctypes
from typing import List, Tuple

import torch

from . import _lib, ops
from ._lib import lib, check

Tensor = torch.Tensor
_D = 768


def _padded_repset(t, lens):
    assert t.dim() == 3 and t.shape[-1] == _D, 'padded reps must be [n, S, 768]'
    n, s, _ = t.shape
    t = t.contiguous()
    start = torch.arange(n, device=t.device, dtype=torch.int32) * s
    return ops.DeviceRepSet(t.view(n * s, _D), start, lens.to(torch.int32).contiguous(), ext=s)


def _csr_repset(rows, start, lens, max_len):
    return ops.DeviceRepSet(rows, start, lens, ext=0, max_len=max_len)


# ---------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op('aspire::span_mean_pool', mutates_args=(), device_types='cuda')
def span_mean_pool(hidden: Tensor, tok_idx: Tensor, span_off: Tensor, max_sents: int) -> Tuple[Tensor, Tensor]:
    cls, sent = ops.span_mean_pool(hidden.contiguous(), tok_idx, span_off, max_sents)
    return cls, sent


@span_mean_pool.register_fake
def _(hidden, tok_idx, span_off, max_sents):
    b, _, d = hidden.shape
    return hidden.new_empty(b, d), hidden.new_empty(b, max_sents, d)


# weights: [word_emb, pos_emb, type_emb, emb_ln_g, emb_ln_b] + per layer [w_qkv, b_qkv, w_o, b_o, ln1_g, ln1_b, w_ffn1,
# b_ffn1, w_ffn2, b_ffn2, ln2_g, ln2_b] (struct aspire_bert_layer, nn.Linear layout)
@torch.library.custom_op('aspire::bert_encoder_forward', mutates_args=(), device_types='cuda')
def bert_encoder_forward(ids: Tensor, type_ids: Tensor, mask: Tensor, weights: List[Tensor], n_heads: int, ln_eps: float) -> Tensor:
    assert (len(weights) - 5) % 12 == 0 and len(weights) >= 5, 'weights: 5 embedding tensors + 12 per layer'
    n_layers = (len(weights) - 5) // 12
    w = [t.contiguous() for t in weights]
    assert all(t.is_cuda and t.dtype == torch.float32 for t in w)
    layers = (_lib.BertLayer * max(n_layers, 1))()
    for i in range(n_layers):
        for f, t in zip(_lib.BertLayer._fields_, w[5 + 12 * i:17 + 12 * i]):
            setattr(layers[i], f[0], ctypes.c_void_p(t.data_ptr()))
    hidden_size = w[0].shape[1]
    ffn = w[5 + 6].shape[0] if n_layers else 4 * hidden_size
    bw = _lib.BertWeights(*(ctypes.c_void_p(t.data_ptr()) for t in w[:5]), layers, n_layers, n_heads, hidden_size, ffn,
                          w[0].shape[0], w[1].shape[0], w[2].shape[0], float(ln_eps))
    ids = ids.to(torch.int64).contiguous()
    b, l = ids.shape
    out = torch.empty(b, l, hidden_size, device=ids.device, dtype=torch.float32)
    need = lib.aspire_bert_workspace_bytes(ctypes.byref(bw), b, l)
    ws = torch.empty(max(need, 16), device=ids.device, dtype=torch.uint8)
    check(lib.aspire_bert_forward_f32(ctypes.byref(bw), ops._ptr(ids), ops._ptr(type_ids.to(torch.int64).contiguous()),
                                      ops._ptr(mask.to(torch.int64).contiguous()), b, l, ops._ptr(out), ops._ptr(ws), ws.numel(),
                                      ops._stream()))
    return out


@bert_encoder_forward.register_fake
def _(ids, type_ids, mask, weights, n_heads, ln_eps):
    return weights[0].new_empty(ids.shape[0], ids.shape[1], weights[0].shape[1])


def _npairs(qn, cn, paired):
    if paired:
        assert qn == cn, 'paired scoring needs equal batch sizes'      # pair_distances.py:46
        return qn
    return qn * cn


@torch.library.custom_op('aspire::l2max_scores', mutates_args=(), device_types='cuda')
def l2max_scores(q: Tensor, q_lens: Tensor, c: Tensor, c_lens: Tensor, paired: bool) -> Tensor:
    return ops.l2max_scores(_padded_repset(q, q_lens), _padded_repset(c, c_lens),
                            pairing=_lib.PAIR_PAIRED if paired else _lib.PAIR_CROSS)


@l2max_scores.register_fake
def _(q, q_lens, c, c_lens, paired):
    return q.new_empty(_npairs(q.shape[0], c.shape[0], paired))


# group: 0 = one epsilon schedule per pair (models.py:190-197); n > 0 = one per consecutive group of n candidates
# (caching_score's padded batches, pp_gen_nearest.py:182-196; paired: one per n pairs).  want: 0 distance, 1 plan-weighted
# similarity, 2 -distance.  extras: also return query_distr, cand_distr, pair_sims, transport plan (else empty tensors).
@torch.library.custom_op('aspire::ot_sinkhorn_scores', mutates_args=(), device_types='cuda')
def ot_sinkhorn_scores(q: Tensor, q_lens: Tensor, c: Tensor, c_lens: Tensor, blur: float, scaling: float, temp: float,
                       group: int, want: int, paired: bool, extras: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    qs, cs = _padded_repset(q, q_lens), _padded_repset(c, c_lens)
    pairing = _lib.PAIR_PAIRED if paired else _lib.PAIR_CROSS
    diam = ops.group_diameter(qs, cs, pairing, group) if group > 0 else None
    out = ops.ot_sinkhorn(qs, cs, pairing=pairing, blur=blur, scaling=scaling, sent_sm_temp=temp, diameter=diam,
                          diam_group=group, want=want, want_extras=extras)
    if extras:
        scores, (qd, cd, ps, plan) = out
        return scores, qd, cd, ps, plan
    e = q.new_empty(0)
    return out, e, e.clone(), e.clone(), e.clone()


@ot_sinkhorn_scores.register_fake
def _(q, q_lens, c, c_lens, blur, scaling, temp, group, want, paired, extras):
    p = _npairs(q.shape[0], c.shape[0], paired)
    if not extras:
        return q.new_empty(p), q.new_empty(0), q.new_empty(0), q.new_empty(0), q.new_empty(0)
    sq, sc = q.shape[1], c.shape[1]
    return q.new_empty(p), q.new_empty(p, sq), q.new_empty(p, sc), q.new_empty(p, sq, sc), q.new_empty(p, sq, sc)


@torch.library.custom_op('aspire::topk_desc', mutates_args=(), device_types='cuda')
def topk_desc(scores: Tensor, k: int, idx_base: int) -> Tuple[Tensor, Tensor]:
    return ops.topk_desc(scores.contiguous(), k, idx_base)


@topk_desc.register_fake
def _(scores, k, idx_base):
    return scores.new_empty(scores.shape[0], k), scores.new_empty(scores.shape[0], k, dtype=torch.int64)


@torch.library.custom_op('aspire::topk_keys', mutates_args=(), device_types='cuda')
def topk_keys(scores: Tensor, k: int, idx_base: int) -> Tensor:
    return ops.topk_keys(scores.contiguous(), k, idx_base)


@topk_keys.register_fake
def _(scores, k, idx_base):
    return scores.new_empty(scores.shape[0], k, dtype=torch.int64)


@torch.library.custom_op('aspire::topk_merge', mutates_args=(), device_types='cuda')
def topk_merge(gathered_keys: Tensor, k: int) -> Tuple[Tensor, Tensor]:
    return ops.topk_merge_keys(gathered_keys.contiguous(), k)


@topk_merge.register_fake
def _(gathered_keys, k):
    qn = gathered_keys.shape[1]
    return gathered_keys.new_empty(qn, k, dtype=torch.float32), gathered_keys.new_empty(qn, k, dtype=torch.int64)


# ---- resident CSR pools -------------------------------------------------------------------------------------------------
@torch.library.custom_op('aspire::l2max_scores_csr', mutates_args=(), device_types='cuda')
def l2max_scores_csr(q_rows: Tensor, q_start: Tensor, q_len: Tensor, q_max: int, c_rows: Tensor, c_start: Tensor,
                     c_len: Tensor, c_max: int) -> Tensor:
    return ops.l2max_scores(_csr_repset(q_rows, q_start, q_len, q_max), _csr_repset(c_rows, c_start, c_len, c_max))


@l2max_scores_csr.register_fake
def _(q_rows, q_start, q_len, q_max, c_rows, c_start, c_len, c_max):
    return q_rows.new_empty(q_start.shape[0] * c_start.shape[0])


@torch.library.custom_op('aspire::ot_scores_csr', mutates_args=(), device_types='cuda')
def ot_scores_csr(q_rows: Tensor, q_start: Tensor, q_len: Tensor, q_max: int, c_rows: Tensor, c_start: Tensor, c_len: Tensor,
                  c_max: int, blur: float, scaling: float, temp: float, group: int, want: int) -> Tensor:
    qs, cs = _csr_repset(q_rows, q_start, q_len, q_max), _csr_repset(c_rows, c_start, c_len, c_max)
    diam = ops.group_diameter(qs, cs, _lib.PAIR_CROSS, group) if group > 0 else None
    return ops.ot_sinkhorn(qs, cs, pairing=_lib.PAIR_CROSS, blur=blur, scaling=scaling, sent_sm_temp=temp, diameter=diam,
                           diam_group=group, want=want)


@ot_scores_csr.register_fake
def _(q_rows, q_start, q_len, q_max, c_rows, c_start, c_len, c_max, blur, scaling, temp, group, want):
    return q_rows.new_empty(q_start.shape[0] * c_start.shape[0])


@torch.library.custom_op('aspire::ot_rank_batch', mutates_args=(), device_types='cuda')
def ot_rank_batch(q_rows: Tensor, q_start: Tensor, q_len: Tensor, q_max: int, c_rows: Tensor, c_start: Tensor, c_len: Tensor,
                  c_max: int, job_off: Tensor, max_job: int, k: int, blur: float, scaling: float, temp: float,
                  want: int) -> Tuple[Tensor, Tensor, Tensor]:
    qs, cs = _csr_repset(q_rows, q_start, q_len, q_max), _csr_repset(c_rows, c_start, c_len, c_max)
    return ops.ot_rank_batch(qs, cs, job_off, max_job, k, blur=blur, scaling=scaling, sent_sm_temp=temp, want=want)


@ot_rank_batch.register_fake
def _(q_rows, q_start, q_len, q_max, c_rows, c_start, c_len, c_max, job_off, max_job, k, blur, scaling, temp, want):
    j = q_start.shape[0]
    return (q_rows.new_empty(c_start.shape[0]), q_rows.new_empty(j, k), q_rows.new_empty(j, k, dtype=torch.int64))


OPS = ('span_mean_pool', 'bert_encoder_forward', 'l2max_scores', 'ot_sinkhorn_scores', 'topk_desc', 'topk_keys', 'topk_merge',
       'l2max_scores_csr', 'ot_scores_csr', 'ot_rank_batch')
