"""A1: BERT-base encoder on the GPU through libaspire_hip.so (aspire_bert_forward_f32).

``HipBertEncoder`` takes its weights from a HuggingFace ``BertModel`` (the object the reference builds with
``AutoModel.from_pretrained`` at examples/ex_aspire_consent.py:33) and exposes the one call the reference
makes on it: ``encoder(tokid_tt, token_type_ids=seg_tt, attention_mask=attnmask_tt).last_hidden_state``
(:72-73).  Weights are copied to HBM once, in nn.Linear layout; query/key/value are concatenated so the three
projections are one GEMM.
"""
import ctypes
from types import SimpleNamespace

import torch

from . import ops
from ._lib import BertLayer, BertWeights, lib, check


class HipBertEncoder:
    def __init__(self, bert_model):
        dev = ops.require_gpu()
        cfg = bert_model.config
        if cfg.hidden_size != 768 or cfg.num_attention_heads != 12:
            raise NotImplementedError('only BERT-base geometry (hidden 768, 12 heads) is built')
        if cfg.hidden_act != 'gelu':
            raise NotImplementedError(f'hidden_act {cfg.hidden_act!r}: only erf-GELU is built')
        if getattr(cfg, 'position_embedding_type', 'absolute') != 'absolute':
            raise NotImplementedError('only absolute position embeddings are built')
        sd = {k: v.detach() for k, v in bert_model.state_dict().items()}
        self.config = cfg
        self.device = dev
        self._keep = []

        def put(t):
            t = t.to(device=dev, dtype=torch.float32).contiguous()
            self._keep.append(t)
            return ctypes.c_void_p(t.data_ptr())

        pre = 'bert.' if any(k.startswith('bert.') for k in sd) else ''
        emb = pre + 'embeddings.'
        n_layers = cfg.num_hidden_layers
        self._layers = (BertLayer * max(n_layers, 1))()
        for i in range(n_layers):
            p = f'{pre}encoder.layer.{i}.'
            att = p + 'attention.self.'
            ly = self._layers[i]
            ly.w_qkv = put(torch.cat([sd[att + 'query.weight'], sd[att + 'key.weight'], sd[att + 'value.weight']], 0))
            ly.b_qkv = put(torch.cat([sd[att + 'query.bias'], sd[att + 'key.bias'], sd[att + 'value.bias']], 0))
            ly.w_o, ly.b_o = put(sd[p + 'attention.output.dense.weight']), put(sd[p + 'attention.output.dense.bias'])
            ly.ln1_g = put(sd[p + 'attention.output.LayerNorm.weight'])
            ly.ln1_b = put(sd[p + 'attention.output.LayerNorm.bias'])
            ly.w_ffn1, ly.b_ffn1 = put(sd[p + 'intermediate.dense.weight']), put(sd[p + 'intermediate.dense.bias'])
            ly.w_ffn2, ly.b_ffn2 = put(sd[p + 'output.dense.weight']), put(sd[p + 'output.dense.bias'])
            ly.ln2_g, ly.ln2_b = put(sd[p + 'output.LayerNorm.weight']), put(sd[p + 'output.LayerNorm.bias'])
        self._w = BertWeights(
            put(sd[emb + 'word_embeddings.weight']), put(sd[emb + 'position_embeddings.weight']),
            put(sd[emb + 'token_type_embeddings.weight']), put(sd[emb + 'LayerNorm.weight']),
            put(sd[emb + 'LayerNorm.bias']), self._layers, n_layers, cfg.num_attention_heads, cfg.hidden_size,
            cfg.intermediate_size, cfg.vocab_size, cfg.max_position_embeddings, cfg.type_vocab_size,
            float(cfg.layer_norm_eps), None)
        self._ws = {}                       # one workspace per HIP stream: forwards on different streams overlap
        # the nn.Linear weights' fp16 planes, formed once (include/aspire_hip.h: aspire_bert_prepare_planes)
        nbytes = lib.aspire_bert_planes_bytes(ctypes.byref(self._w))
        if nbytes and cfg.intermediate_size % 128 == 0:
            self._planes = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            try:
                check(lib.aspire_bert_prepare_planes(ctypes.byref(self._w), ops._ptr(self._planes), nbytes, ops._stream()))
                self._w.planes = ctypes.c_void_p(self._planes.data_ptr())
            except NotImplementedError as e:        # a weight beyond the fp16 planes' range: the on-the-fly bf16x3 GEMMs take any fp32
                import warnings
                warnings.warn(f'HipBertEncoder: {e}; running without pre-split weights')
                self._planes = None

    def eval(self):
        return self

    def forward_full_range(self, tokid_tt, token_type_ids=None, attention_mask=None):
        """The forward on the kernels that take ANY fp32 activation: GEMM operands split into three bf16 planes on the fly, attention
        on the fp32-input MFMA.  The default path keeps activations as two fp16 planes (|x| <= 65504: far above what BERT-base
        checkpoints produce, but a fine-tuned model with an outlier feature beyond it turns into inf there); callers that find
        non-finite hidden states (AspireConSent.forward / encode_to_pool check what they hand out) come here."""
        from ._lib import pinned
        with pinned(GEMM='bf16x3', ATTN='f32'):
            return self.forward_hidden(tokid_tt, token_type_ids, attention_mask, check_ids=False)

    def forward_hidden(self, tokid_tt, token_type_ids=None, attention_mask=None, check_ids=True):
        """int64 [B, L] tensors (any device) -> last_hidden_state [B, L, 768] on the GPU.  check_ids=False: the caller has
        validated the token ids already (encode_to_pool checks all its batches with one device round trip)."""
        dev = self.device
        tok = tokid_tt.to(device=dev, dtype=torch.int64).contiguous()
        b, l = tok.shape
        if check_ids and (int(tok.max()) >= self.config.vocab_size or int(tok.min()) < 0):
            raise IndexError('token id out of range')   # nn.Embedding raises IndexError on the reference path
        typ = token_type_ids.to(device=dev, dtype=torch.int64).contiguous() if token_type_ids is not None else None
        msk = attention_mask.to(device=dev, dtype=torch.int64).contiguous() if attention_mask is not None \
            else torch.ones_like(tok)
        out = torch.empty(b, l, 768, device=dev, dtype=torch.float32)
        need = lib.aspire_bert_workspace_bytes(ctypes.byref(self._w), b, l)
        sid = torch.cuda.current_stream().cuda_stream
        ws = self._ws.get(sid)
        if ws is None or ws.numel() < need:
            self._ws[sid] = ws = torch.empty(need, device=dev, dtype=torch.uint8)
        check(lib.aspire_bert_forward_f32(ctypes.byref(self._w), ops._ptr(tok), ops._ptr(typ), ops._ptr(msk), b, l,
                                          ops._ptr(out), ops._ptr(ws), ws.numel(), ops._stream()))
        return out

    @staticmethod
    def status():
        """The encoder kernels' sticky status word (include/aspire_hip.h: aspire_bert_status), read and cleared; synchronises the current
        stream.  Non-zero: a LayerNorm-epilogue GEMM gave up waiting for its row block (ASPIRE_BERT_STATUS_LN_TIMEOUT) -- the forwards
        since the last check are invalid; the callers in consent.py run them again under pinned(GEMM_LN='off')."""
        v = ctypes.c_int32(0)
        check(lib.aspire_bert_status(ctypes.byref(v), ops._stream()))
        return int(v.value)

    def __call__(self, tokid_tt, token_type_ids=None, attention_mask=None):
        return SimpleNamespace(last_hidden_state=self.forward_hidden(tokid_tt, token_type_ids, attention_mask))
