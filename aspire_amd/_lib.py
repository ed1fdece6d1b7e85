"""ctypes binding of libaspire_hip.so (the C ABI declared in include/aspire_hip.h).

The library is the product: there is no CPU or PyTorch-eager fallback.  If it is missing the import
fails loudly; if it is present but no GPU is visible, every compute entry point raises.
"""
import ctypes
import os

# PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so).  It must be in the process BEFORE
# libaspire_hip.so is dlopen'ed so that both resolve the same runtime: device pointers and streams that
# torch hands out are only meaningful to the runtime that created them.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# ASPIRE_HIP_LIB: developer override (instrumented builds under build/); the product path is in-tree.
LIB_PATH = os.environ.get('ASPIRE_HIP_LIB') or os.path.join(_HERE, 'lib', 'libaspire_hip.so')

c_void_p, c_int, c_int32, c_int64, c_double, c_size_t = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int32,
                                                         ctypes.c_int64, ctypes.c_double, ctypes.c_size_t)

ASPIRE_OK, ASPIRE_ERR_INVALID_ARG, ASPIRE_ERR_UNSUPPORTED, ASPIRE_ERR_HIP = 0, 1, 2, 3
CDIST_AUTO, CDIST_DIRECT, CDIST_MM = 0, 1, 2
CDIST_ONE_FORM = 0x100       # or'ed into cdist_mode of the max-sim entry points: one kernel form whatever the call's size
OT_FLAG_ONE_FORM = 1         # aspire_ot_params.flags: the same for otAspire
CDIST_CENTER = 0x200         # rows with a large common component: subtract the query's mean row before the expansion (max-sim entries)
OT_FLAG_CENTER = 2           # the same for otAspire
PAIR_CROSS, PAIR_PAIRED = 0, 1
OT_DISTANCE, OT_PLAN_SIM, OT_SIMILARITY = 0, 1, 2
AGG_MAX, AGG_TOP2, AGG_ATTENTION = 0, 1, 2


class RepPlanes(ctypes.Structure):
    """struct aspire_rep_planes"""
    _fields_ = [('planes', c_void_p), ('row_nrm', c_void_p), ('row_iscale', c_void_p), ('mu', c_void_p),
                ('total_rows', c_int64), ('plane_rows', c_int64)]


class RepSet(ctypes.Structure):
    """struct aspire_repset"""
    _fields_ = [('rows', c_void_p), ('start', c_void_p), ('len', c_void_p), ('n', c_int64),
                ('ext', c_int32), ('max_len', c_int32), ('planes', ctypes.POINTER(RepPlanes)), ('doc_box', c_void_p)]


class OtParams(ctypes.Structure):
    """struct aspire_ot_params"""
    _fields_ = [('blur', c_double), ('scaling', c_double), ('sent_sm_temp', c_double), ('cdist_mode', c_int32),
                ('flags', c_int32)]


class BertLayer(ctypes.Structure):
    """struct aspire_bert_layer"""
    _fields_ = [(n, c_void_p) for n in ('w_qkv', 'b_qkv', 'w_o', 'b_o', 'ln1_g', 'ln1_b', 'w_ffn1', 'b_ffn1',
                                        'w_ffn2', 'b_ffn2', 'ln2_g', 'ln2_b')]


class BertWeights(ctypes.Structure):
    """struct aspire_bert_weights"""
    _fields_ = [('word_emb', c_void_p), ('pos_emb', c_void_p), ('type_emb', c_void_p), ('emb_ln_g', c_void_p),
                ('emb_ln_b', c_void_p), ('layers', ctypes.POINTER(BertLayer)), ('n_layers', c_int32),
                ('n_heads', c_int32), ('hidden', c_int32), ('ffn_dim', c_int32), ('vocab', c_int32),
                ('max_pos', c_int32), ('n_types', c_int32), ('ln_eps', ctypes.c_float), ('planes', c_void_p)]


class AspireHipError(RuntimeError):
    pass


# name -> (restype, argtypes); must list every function include/aspire_hip.h declares.
SIGNATURES = {
    'aspire_abi_version': (c_int, []),
    'aspire_last_error': (ctypes.c_char_p, []),
    'aspire_max_sents': (c_int, []),
    'aspire_span_mean_pool_f32': (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64,
                                          c_void_p, c_void_p, c_void_p]),
    'aspire_span_mean_pool_rows_f32': (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64,
                                               c_void_p, c_void_p, c_void_p, c_void_p]),
    'aspire_cls_l2_f32': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_double, c_void_p, c_void_p]),
    'aspire_bert_planes_bytes': (c_size_t, [ctypes.POINTER(BertWeights)]),
    'aspire_bert_prepare_planes': (c_int, [ctypes.POINTER(BertWeights), c_void_p, c_size_t, c_void_p]),
    'aspire_bert_workspace_bytes': (c_size_t, [ctypes.POINTER(BertWeights), c_int64, c_int64]),
    'aspire_bert_forward_f32': (c_int, [ctypes.POINTER(BertWeights), c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                        c_void_p, c_void_p, c_size_t, c_void_p]),
    'aspire_bert_status': (c_int, [ctypes.POINTER(ctypes.c_int32), c_void_p]),
    'aspire_rep_planes_bytes': (c_size_t, [c_int64]),
    'aspire_rep_planes_prepare': (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_size_t, ctypes.POINTER(RepPlanes),
                                          c_void_p]),
    'aspire_repset_boxes_f32': (c_int, [ctypes.POINTER(RepSet), c_int64, c_void_p, c_void_p]),
    'aspire_l2max_scores_f32': (c_int, [ctypes.POINTER(RepSet), ctypes.POINTER(RepSet), c_int64, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p]),
    'aspire_l2agg_scores_f32': (c_int, [ctypes.POINTER(RepSet), ctypes.POINTER(RepSet), c_int64, c_int, c_int, c_int,
                                        ctypes.c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    'aspire_ot_sinkhorn_f32': (c_int, [ctypes.POINTER(RepSet), ctypes.POINTER(RepSet), c_int64, c_int,
                                       ctypes.POINTER(OtParams), c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'aspire_ot_workspace_bytes': (c_size_t, [ctypes.POINTER(RepSet), ctypes.POINTER(RepSet), c_int]),
    'aspire_group_diameter_f32': (c_int, [ctypes.POINTER(RepSet), ctypes.POINTER(RepSet), c_int64, c_int, c_int64,
                                          c_void_p, c_void_p]),
    'aspire_topk_workspace_bytes': (c_size_t, [c_int64, c_int64, c_int64]),
    'aspire_topk_desc_f32': (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                     c_size_t, c_void_p]),
    'aspire_ot_rank_workspace_bytes': (c_size_t, [ctypes.POINTER(RepSet), ctypes.POINTER(RepSet), c_int64]),
    'aspire_ot_rank_f32': (c_int, [ctypes.POINTER(RepSet), ctypes.POINTER(RepSet), c_int64, ctypes.POINTER(OtParams), c_void_p,
                                   c_int64, c_int, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                   c_void_p]),
    'aspire_topk_keys_f32': (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    'aspire_topk_merge_keys': (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    'aspire_ot_rank_batch_workspace_bytes': (c_size_t, [ctypes.POINTER(RepSet), ctypes.POINTER(RepSet), c_int64, c_int64]),
    'aspire_ot_rank_batch_f32': (c_int, [ctypes.POINTER(RepSet), ctypes.POINTER(RepSet), c_int64, c_void_p, c_int64,
                                         ctypes.POINTER(OtParams), c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_size_t, c_void_p]),
    'aspire_l2max_rank_batch_workspace_bytes': (c_size_t, [ctypes.POINTER(RepSet), ctypes.POINTER(RepSet), c_int64, c_int64]),
    'aspire_l2max_rank_batch_f32': (c_int, [ctypes.POINTER(RepSet), ctypes.POINTER(RepSet), c_int64, c_void_p, c_int64, c_int,
                                            c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'aspire_debug_set': (c_int, [ctypes.c_char_p, ctypes.c_char_p]),
    'aspire_debug_get': (c_int, [ctypes.c_char_p, ctypes.c_char_p, c_size_t]),
    'aspire_debug_ot_cost_stage_f32': (c_int, [ctypes.POINTER(RepSet), ctypes.POINTER(RepSet), c_int64, c_int,
                                               ctypes.POINTER(OtParams), c_void_p, c_void_p, c_size_t, c_void_p]),
    'aspire_debug_ot_rank_batch_stages_f32': (c_int, [ctypes.POINTER(RepSet), ctypes.POINTER(RepSet), c_int64, c_void_p, c_int64,
                                                      ctypes.POINTER(OtParams), c_int, c_void_p, c_int64, c_void_p, c_void_p,
                                                      c_void_p, c_size_t, c_void_p, c_int]),
    'aspire_debug_clock_probe': (c_int, [c_void_p, ctypes.c_longlong, c_void_p]),
    'aspire_selftest_xlane': (c_int, [ctypes.POINTER(c_int)]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} is missing: build it first with `python -c "import __graft_entry__ as g; g.build()"` '
            f'(hipcc --offload-arch=gfx950).  aspire_amd has no CPU fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(status):
    if status != ASPIRE_OK:
        msg = lib.aspire_last_error().decode('utf-8', 'replace')
        if status == ASPIRE_ERR_INVALID_ARG:
            # the reference signals these with `assert`, keep the exception type
            raise AssertionError(msg)
        if status == ASPIRE_ERR_UNSUPPORTED:
            raise NotImplementedError(msg)
        raise AspireHipError(msg)


class pinned:
    """Context manager over aspire_debug_set: pin diagnostic switches (kernel forms, grids) for a block of calls, e.g.
    ``with pinned(SINKHORN='block', COST_PATH='valu'): ...``; on exit every switch goes back to what it was before the
    block (an enclosing pin, an ASPIRE_HIP_* environment setting, or the default)."""

    def __init__(self, **kv):
        self.kv = kv
        self.before = {}

    def __enter__(self):
        buf = ctypes.create_string_buffer(64)
        for k, v in self.kv.items():
            check(lib.aspire_debug_get(k.encode(), buf, len(buf)))
            self.before[k] = buf.value
            check(lib.aspire_debug_set(k.encode(), str(v).encode()))
        return self

    def __exit__(self, *exc):
        for k, v in self.before.items():
            lib.aspire_debug_set(k.encode(), v if v else None)
        self.before = {}
        return False
