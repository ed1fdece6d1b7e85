"""Torch-tensor front end of the C ABI: device memory and streams come from PyTorch-ROCm, every
computation happens in libaspire_hip.so.  All tensors given here must already live on the GPU."""
import ctypes

import torch

from . import _lib
from ._lib import RepSet, OtParams, lib, check

D = 768


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError('aspire_amd needs an AMD GPU (PyTorch-ROCm sees none); there is no CPU fallback.')
    return torch.device('cuda', torch.cuda.current_device())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _f32(t, name):
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), f'{name}: need contiguous fp32 GPU tensor'
    return t


def _i32(t, name):
    assert t.is_cuda and t.dtype == torch.int32 and t.is_contiguous(), f'{name}: need contiguous int32 GPU tensor'
    return t


# None: every rep set decides from a sample of its rows (center_hint: a few small kernels + one host sync per NEW row matrix);
# True / False: the answer for every call (set_center_hint) -- e.g. False for a process that knows its vectors are isotropic,
# True for one that scores a sentence-embedding space, so that all shards and call forms of a job use one arithmetic.
CENTER_HINT = None


def set_center_hint(mode):
    global CENTER_HINT
    assert mode in (None, True, False)
    CENTER_HINT = mode


def _rows_key(rows):
    """What a cache derived from a row matrix is valid for: the same memory, the same number of rows, not written since.  Two
    counters: the library's own generation (`_aspire_gen` on the tensor object, bumped by _rows_written: the entry points of this
    module that write rows through the C ABI report theirs) and torch's count of in-place writes (rows._version, shared by a matrix
    and its views) where torch keeps one -- a tensor made under torch.inference_mode() tracks no version (reading it raises): for
    those only the library's own writes are seen, a torch write into such a store needs drop_planes() / prepare_planes() by hand."""
    version = None if rows.is_inference() else rows._version
    return (rows.data_ptr(), int(rows.shape[0]), version, getattr(rows, '_aspire_gen', 0))


def _rows_written(t):
    """The library wrote into `t` behind torch's back (a kernel given its data pointer): caches keyed by _rows_key(t) are stale."""
    t._aspire_gen = getattr(t, '_aspire_gen', 0) + 1
    if not t.is_inference():
        torch.autograd.graph.increment_version(t)


_REBUILD_WARNED = False


def _warn_rebuild(what):
    """An automatic rebuild of a cache (planes: ~2.5 ms per GB of rows; boxes: 6 KB per document) inside a scoring call: said once."""
    global _REBUILD_WARNED
    if not _REBUILD_WARNED:
        _REBUILD_WARNED = True
        import warnings
        warnings.warn(f'aspire_amd: the rows of a resident store were written after its {what} were prepared; they are formed again '
                      'inside this scoring call (and on every call that follows a write -- torch counts writes through ANY view of the '
                      'buffer).  Prepare the caches after the last write to keep this out of the scoring path.')


def _match_planes(q, c, pairing):
    """An all-against-all call on a big pool that carries fp16 planes (DeviceRepSet.prepare_planes): the queries get
    theirs, around the pool's centre, unless they are rows of the same matrix already (then nothing happens) or carry planes the
    CALLER prepared around another store's centre (left alone: the call takes the kernels that read the fp32 rows).  Planes this
    function made earlier are kept on the query matrix and reused only while they are valid: same rows, not written since
    (_rows_key), same centre as THIS pool -- otherwise they are made again.  ~10 us for 256 query rows."""
    if pairing != _lib.PAIR_CROSS or c.planes is None or q.ext or c.ext:
        return
    slot = max(8, (c.max_len + 3) // 4 * 4)
    if c.n * slot < 128 * 128 or max(q.max_len, c.max_len) > 32:          # fewer than 128 candidate tiles: the small-pool kernels
        return
    qp = q.planes
    if qp is None or (qp.auto and qp.mu.data_ptr() != c.planes.mu.data_ptr()):
        q.prepare_planes(like=c)
        q.planes.auto = True


class RowPlanes:
    """struct aspire_rep_planes of a row matrix + the device blob it points into."""

    def __init__(self, rows, mu=None):
        _f32(rows, 'rows')
        n = int(rows.shape[0])
        nbytes = lib.aspire_rep_planes_bytes(n)
        self.blob = torch.empty(nbytes, device=rows.device, dtype=torch.uint8)
        self.mu = mu                       # tensor [768] (the store's common vector); None: formed from these rows
        self.mu_given = mu is not None
        self.auto = False                  # made by _match_planes for a call (not by the caller)
        self.key = _rows_key(rows)
        self.c = _lib.RepPlanes()
        check(lib.aspire_rep_planes_prepare(_ptr(rows), n, D, _ptr(_f32(mu, 'mu')) if mu is not None else None, _ptr(self.blob),
                                            nbytes, ctypes.byref(self.c), _stream()))
        if mu is None:
            self.mu = self.blob[:4 * D].view(torch.float32)


class DeviceRepSet:
    """Rows + CSR view of sentence reps resident in HBM (struct aspire_repset).

    rows [total, 768] fp32; start/len int32 [n]; ext > 0 marks a padded [n, ext, 768] tensor."""

    def __init__(self, rows, start, lens, ext=0, max_len=None, lens_host=None):
        self.lens_host = list(lens_host) if lens_host is not None else None      # host copy of len[], when the caller has one
        self.rows = _f32(rows, 'rows')
        assert rows.shape[-1] == D, f'encoding dim must be {D}'
        self.start = _i32(start, 'start')
        self.len = _i32(lens, 'len')
        self.n = int(start.numel())
        assert lens.numel() == self.n
        self.ext = int(ext)
        self.max_len = int(max_len if max_len is not None else (ext if ext > 0 else (int(lens.max()) if self.n else 0)))
        # A document without sentences: the reference raises (torch.max over an empty dimension, pair_distances.py:57 via
        # models.py:190-197); scored here it would come back as OT distance 0 = the best possible similarity.  Checked
        # wherever the lengths are known on the host (every constructor of the host layer passes them).
        if self.lens_host is not None and any(n <= 0 for n in self.lens_host):
            raise ValueError('a document without sentence rows cannot be scored (the reference raises on it: '
                             'pair_distances.py:57); drop it from the pool')

    @classmethod
    def from_padded(cls, reps, abs_lens):
        """reps [B, S, 768] (any device) + abs_lens list -> padded repset on the GPU."""
        dev = require_gpu()
        reps = reps.to(device=dev, dtype=torch.float32).contiguous()
        b, s, _ = reps.shape
        start = torch.arange(b, device=dev, dtype=torch.int32) * s
        lens_host = [int(n) for n in abs_lens]
        lens = torch.as_tensor(lens_host, dtype=torch.int32).to(dev)
        assert lens.numel() == b, 'abs_lens must have one entry per batch element'
        return cls(reps.view(b * s, D), start, lens, ext=s, lens_host=lens_host)

    @classmethod
    def from_list(cls, reps_list):
        """list of [S_i, 768] arrays/tensors -> CSR repset on the GPU (no padding rows)."""
        dev = require_gpu()
        ts = [torch.as_tensor(r, dtype=torch.float32) for r in reps_list]
        lens_host = [int(t.shape[0]) for t in ts]
        rows = torch.cat(ts, dim=0).to(dev).contiguous() if ts else torch.zeros(0, D, device=dev)
        lens = torch.tensor(lens_host, dtype=torch.int32)
        start = (torch.cumsum(lens, 0) - lens).to(torch.int32)
        return cls(rows, start.to(dev), lens.to(dev), ext=0, max_len=max(lens_host) if lens_host else 0, lens_host=lens_host)

    def struct(self):
        planes = self.planes                                     # kept on the matrix: index lists and slices of it share them
        return RepSet(_ptr(self.rows), _ptr(self.start), _ptr(self.len), self.n, self.ext, self.max_len,
                      ctypes.pointer(planes.c) if planes is not None else None, _ptr(self._fresh_boxes()))

    def _fresh_boxes(self):
        box = getattr(self, 'doc_box', None)
        if box is not None and getattr(self, '_doc_box_key', None) not in (None, _rows_key(self.rows)):
            self.doc_box = None                                  # the rows were written since: formed again
            _warn_rebuild('per-document boxes')
            self.prepare_boxes()
            box = self.doc_box
        return box

    def prepare_boxes(self):
        """The documents' per-coordinate bounding boxes [n, 2, 768], kept with this rep set (include/aspire_hip.h:
        aspire_repset.doc_box): the many-query otAspire calls on a resident pool then skip their pass over every candidate row
        (geomloss's diameter).  6 KB per document; formed again when the rows have been written since (_rows_key).  Returns self."""
        if self.n and getattr(self, 'doc_box', None) is None:
            box = torch.empty(self.n, 2, D, device=self.rows.device, dtype=torch.float32)
            s = RepSet(_ptr(self.rows), _ptr(self.start), _ptr(self.len), self.n, self.ext, self.max_len, None, None)
            check(lib.aspire_repset_boxes_f32(ctypes.byref(s), D, _ptr(box), _stream()))
            self.doc_box = box
            self._doc_box_key = _rows_key(self.rows)
        return self

    def slice(self, lo, hi):
        r = DeviceRepSet(self.rows, self.start[lo:hi].contiguous(), self.len[lo:hi].contiguous(), self.ext,
                         self.max_len, lens_host=self.lens_host[lo:hi] if self.lens_host is not None else None)
        if self._fresh_boxes() is not None:
            r.doc_box = self.doc_box[lo:hi]
            r._doc_box_key = self._doc_box_key
        return r

    @property
    def planes(self):
        """The matrix's fp16 planes, or None.  Planes are a cache of the rows: when the rows have been written since they were
        made (or the tensor now points elsewhere), planes the caller prepared are made again the same way (own centre / the centre
        given then), planes made for a call (_match_planes) are dropped.  The rebuild keeps the CENTRE the planes had (a copy of it):
        the shards of one pool share rank 0's centre (parallel.py), and a shard that formed a new one on its own would leave the
        bit-equality of sharded and un-sharded scores behind; any common vector serves as a centre, L2 distances do not move."""
        pl = getattr(self.rows, '_aspire_planes', None)
        if pl is not None and pl.key != _rows_key(self.rows):
            if pl.auto:
                self.rows._aspire_planes = pl = None
            else:
                _warn_rebuild('fp16 planes')
                was_given = pl.mu_given
                self.rows._aspire_planes = pl = RowPlanes(self.rows, pl.mu.clone())
                pl.mu_given = was_given
        return pl

    def prepare_planes(self, like=None, mu=None):
        """The row matrix a second time as fp16 planes (include/aspire_hip.h: aspire_rep_planes): once per resident store;
        for query sets that are not part of the store, per call with like = the store's rep set (its common vector).  The
        many-query cost tiles then run on the fp16 matrix pipe.  Returns self."""
        if like is not None:
            assert like.planes is not None, 'prepare the store\'s planes first'
            mu = like.planes.mu
        self.rows._aspire_planes = RowPlanes(self.rows, mu)
        return self

    def drop_planes(self):
        if getattr(self.rows, '_aspire_planes', None) is not None:
            self.rows._aspire_planes = None
        return self

    def center_hint(self):
        if CENTER_HINT is not None:
            return CENTER_HINT
        return self._center_hint_sampled()

    def _center_hint_sampled(self):
        """True when the rows share a large common component (|mean row|^2 > 0.25 x the mean squared norm, i.e. a mean cosine of
        roughly 0.25 between unrelated rows -- sentence-embedding spaces are usually far above that): the scoring calls then set
        ASPIRE_OT_FLAG_CENTER / ASPIRE_CDIST_CENTER (include/aspire_hip.h).  From a sample of up to 512 rows, once per rep set."""
        if getattr(self, '_center_hint', None) is None:
            hint = getattr(self.rows, '_aspire_center_hint', None)     # kept on the matrix: index lists into it share the answer
            if hint is None:
                n = int(self.rows.shape[0])
                if n == 0:
                    hint = False
                else:
                    sample = self.rows[::max(1, n // 512)][:512]
                    m = sample.mean(0)
                    hint = bool((m * m).sum() > 0.25 * (sample * sample).sum(1).mean())      # (elementwise + reductions: no rocBLAS in the product path)
                self.rows._aspire_center_hint = hint
            self._center_hint = hint
        return self._center_hint

    def host_lens(self):
        if self.lens_host is None:
            self.lens_host = self.len.cpu().tolist()
        return self.lens_host


def _npairs(q, c, pairing):
    return q.n if pairing == _lib.PAIR_PAIRED else q.n * c.n


def span_mean_pool(hidden, tok_idx, span_off, max_sents, want_cls=True):
    """A2/A3 (ex_aspire_consent.py:75-100).  hidden [B,L,768] GPU fp32 -> (cls [B,768], sent [B,S,768])."""
    _f32(hidden, 'hidden')
    b, l, d = hidden.shape
    sent = torch.empty(b, max_sents, d, device=hidden.device, dtype=torch.float32)
    cls = torch.empty(b, d, device=hidden.device, dtype=torch.float32) if want_cls else None
    check(lib.aspire_span_mean_pool_f32(_ptr(hidden), b, l, d, _ptr(_i32(tok_idx, 'tok_idx')),
                                        _ptr(_i32(span_off, 'span_off')), max_sents, _ptr(sent), _ptr(cls), _stream()))
    return cls, sent


def span_mean_pool_rows(hidden, tok_idx, span_off, max_sents, out_row, rows, cls=None):
    """The pooling written straight into a resident rep store (include/aspire_hip.h: aspire_span_mean_pool_rows_f32): slot
    (b, s) -> rows[out_row[b * S + s]] (skipped when negative); cls [B, 768] optional.  Nothing is returned: `rows` (a view
    into the store's row matrix) and `cls` are filled in place."""
    _f32(hidden, 'hidden')
    _f32(rows, 'rows')
    b, l, d = hidden.shape
    assert out_row.numel() == b * max_sents
    check(lib.aspire_span_mean_pool_rows_f32(_ptr(hidden), b, l, d, _ptr(_i32(tok_idx, 'tok_idx')), _ptr(_i32(span_off, 'span_off')),
                                             max_sents, _ptr(_i32(out_row, 'out_row')), _ptr(rows), _ptr(cls), _stream()))
    _rows_written(rows)


def cls_l2(q_cls, c_cls, pairing=_lib.PAIR_PAIRED, eps=1e-6):
    """functional.pairwise_distance(q_cls, c_cls, p=2.0) (disent_models.py:306): ||q - c + eps||_2, [P] on the GPU."""
    _f32(q_cls, 'q_cls')
    _f32(c_cls, 'c_cls')
    qn, cn = q_cls.shape[0], c_cls.shape[0]
    out = torch.empty(cn if pairing == _lib.PAIR_PAIRED else qn * cn, device=q_cls.device, dtype=torch.float32)
    check(lib.aspire_cls_l2_f32(_ptr(q_cls), qn, _ptr(c_cls), cn, D, pairing, float(eps), _ptr(out), _stream()))
    return out


def l2max_scores(q, c, pairing=_lib.PAIR_CROSS, cdist_mode=_lib.CDIST_AUTO, want_pair_sims=False, one_form=False):
    """A9 (pair_distances.py:138-186).  Returns sims [P] (and pair_sims [P, q.ext, c.ext]).  one_form: see
    include/aspire_hip.h, ASPIRE_CDIST_ONE_FORM."""
    if one_form:
        cdist_mode |= _lib.CDIST_ONE_FORM
    if c.center_hint():
        cdist_mode |= _lib.CDIST_CENTER
    p = _npairs(q, c, pairing)
    dev = q.rows.device
    scores = torch.empty(p, device=dev, dtype=torch.float32)
    pair = torch.empty(p, q.ext, c.ext, device=dev, dtype=torch.float32) if want_pair_sims else None
    _match_planes(q, c, pairing)
    qs, cs = q.struct(), c.struct()
    check(lib.aspire_l2max_scores_f32(ctypes.byref(qs), ctypes.byref(cs), D, pairing, cdist_mode, _ptr(scores),
                                      _ptr(pair), _stream()))
    return (scores, pair) if want_pair_sims else scores


def l2agg_scores(q, c, agg, temp=1.0, pairing=_lib.PAIR_CROSS, cdist_mode=_lib.CDIST_AUTO, want_pair_sims=False):
    """Sibling aggregations of the masked -cdist block: agg = _lib.AGG_TOP2 (pair_distances.py:295-345) or
    _lib.AGG_ATTENTION (pair_distances.py:95-135; temp = cdatt_sm_temp).  Returns sims [P]; with want_pair_sims also
    pair_sims [P, q.ext, c.ext] and, for attention, pair_softmax [P, q.ext, c.ext]."""
    p = _npairs(q, c, pairing)
    dev = q.rows.device
    scores = torch.empty(p, device=dev, dtype=torch.float32)
    pair = torch.empty(p, q.ext, c.ext, device=dev, dtype=torch.float32) if want_pair_sims else None
    soft = torch.empty(p, q.ext, c.ext, device=dev, dtype=torch.float32) \
        if want_pair_sims and agg == _lib.AGG_ATTENTION else None
    qs, cs = q.struct(), c.struct()
    check(lib.aspire_l2agg_scores_f32(ctypes.byref(qs), ctypes.byref(cs), D, pairing, cdist_mode, agg,
                                      ctypes.c_double(temp), _ptr(scores), _ptr(pair), _ptr(soft), _stream()))
    if not want_pair_sims:
        return scores
    return (scores, pair, soft) if agg == _lib.AGG_ATTENTION else (scores, pair)


def group_diameter(q, c, pairing, group):
    ngroups = (c.n + group - 1) // group
    n = ngroups if pairing == _lib.PAIR_PAIRED else q.n * ngroups
    out = torch.empty(max(n, 1), device=q.rows.device, dtype=torch.float32)
    qs, cs = q.struct(), c.struct()
    check(lib.aspire_group_diameter_f32(ctypes.byref(qs), ctypes.byref(cs), D, pairing, group, _ptr(out), _stream()))
    return out


def ot_sinkhorn(q, c, pairing=_lib.PAIR_CROSS, blur=0.05, scaling=0.9, sent_sm_temp=1.0, cdist_mode=_lib.CDIST_AUTO,
                diameter=None, diam_group=0, want=_lib.OT_DISTANCE, want_extras=False, out=None, workspace=None, one_form=False):
    """A5-A8 (pair_distances.py:21-92).  Returns scores [P]; with want_extras also
    (query_distr [P,q.ext], cand_distr [P,c.ext], pair_sims [P,q.ext,c.ext], plan [P,q.ext,c.ext])."""
    p = _npairs(q, c, pairing)
    dev = q.rows.device
    scores = out if out is not None else torch.empty(p, device=dev, dtype=torch.float32)
    assert scores.numel() >= p
    extras = [None] * 4
    if want_extras:
        extras = [torch.empty(p, q.ext, device=dev), torch.empty(p, c.ext, device=dev),
                  torch.empty(p, q.ext, c.ext, device=dev), torch.empty(p, q.ext, c.ext, device=dev)]
    prm = OtParams(float(blur), float(scaling), float(sent_sm_temp), cdist_mode, (_lib.OT_FLAG_ONE_FORM if one_form else 0) | (_lib.OT_FLAG_CENTER if c.center_hint() else 0))
    _match_planes(q, c, pairing)
    qs, cs = q.struct(), c.struct()
    nbytes = lib.aspire_ot_workspace_bytes(ctypes.byref(qs), ctypes.byref(cs), pairing)
    ws = workspace if workspace is not None else torch.empty(max(nbytes, 8), device=dev, dtype=torch.uint8)
    check(lib.aspire_ot_sinkhorn_f32(ctypes.byref(qs), ctypes.byref(cs), D, pairing, ctypes.byref(prm),
                                     _ptr(diameter), diam_group, want, _ptr(scores), _ptr(extras[0]), _ptr(extras[1]),
                                     _ptr(extras[2]), _ptr(extras[3]), _ptr(ws), ws.numel(), _stream()))
    return (scores, extras) if want_extras else scores


def ot_rank(q, c, k, blur=0.05, scaling=0.9, sent_sm_temp=1.0, cdist_mode=_lib.CDIST_AUTO, diameter=None, diam_group=0,
            want=_lib.OT_DISTANCE, idx_base=0, key_form=False, one_form=False):
    """The ranking step in one call (include/aspire_hip.h: aspire_ot_rank_f32): otAspire scores [Q, C] of every
    query against every candidate and their per-query stable descending rank.  Returns (scores [Q, C], top_scores
    [Q, k], top_idx [Q, k]) -- `scores` holds the raw kernel output (positive distances for OT_DISTANCE, so the rank
    is of the OUTPUT; pass want=OT_PLAN_SIM for similarities) -- or (scores, keys [Q, k]) with key_form."""
    dev = q.rows.device
    scores = torch.empty(q.n, c.n, device=dev, dtype=torch.float32)
    prm = OtParams(float(blur), float(scaling), float(sent_sm_temp), cdist_mode, (_lib.OT_FLAG_ONE_FORM if one_form else 0) | (_lib.OT_FLAG_CENTER if c.center_hint() else 0))
    _match_planes(q, c, _lib.PAIR_CROSS)
    qs, cs = q.struct(), c.struct()
    nbytes = lib.aspire_ot_rank_workspace_bytes(ctypes.byref(qs), ctypes.byref(cs), k)
    ws = torch.empty(max(nbytes, 8), device=dev, dtype=torch.uint8)
    top_s = top_i = keys = None
    if key_form:
        keys = torch.empty(q.n, k, device=dev, dtype=torch.int64)
    else:
        top_s = torch.empty(q.n, k, device=dev, dtype=torch.float32)
        top_i = torch.empty(q.n, k, device=dev, dtype=torch.int64)
    check(lib.aspire_ot_rank_f32(ctypes.byref(qs), ctypes.byref(cs), D, ctypes.byref(prm), _ptr(diameter), diam_group, want,
                                 _ptr(scores), k, idx_base, _ptr(top_s), _ptr(top_i), _ptr(keys), _ptr(ws), ws.numel(), _stream()))
    return (scores, keys) if key_form else (scores, top_s, top_i)


def ot_rank_batch(q, c, job_off, max_job, k, blur=0.05, scaling=0.9, sent_sm_temp=1.0, cdist_mode=_lib.CDIST_AUTO,
                  want=_lib.OT_SIMILARITY, out=None, workspace=None, job_base=None, key_form=False, one_form=False):
    """J independent (query, pool) re-ranks in ONE call (include/aspire_hip.h: aspire_ot_rank_batch_f32; the per-query
    loop of evaluate.py:58-76 batched over queries).  q: J queries; c: every job's candidates back to back; job_off int32
    GPU tensor [J + 1]; max_job: host-known bound of a pool's size.  Returns (scores [C], top_scores [J, k], top_idx [J, k])
    with top_idx = position inside the job's own pool (+ job_base[j], int32 GPU tensor [J], when this rank holds one
    block of every pool); `out` = preallocated (scores, top_scores, top_idx).  key_form: (scores, keys [J, k]) -- the
    sortable keys of topk_keys, what a shard contributes to the all-gather."""
    dev = q.rows.device
    _i32(job_off, 'job_off')
    assert job_off.numel() == q.n + 1, 'job_off must have one entry per job plus one'
    keys = None
    if out is not None and key_form:
        scores, keys = out
        top_s = top_i = None
    elif out is not None:
        scores, top_s, top_i = out
    else:
        scores = torch.empty(c.n, device=dev, dtype=torch.float32)
        top_s = torch.empty(q.n, k, device=dev, dtype=torch.float32) if k > 0 and not key_form else None
        top_i = torch.empty(q.n, k, device=dev, dtype=torch.int64) if k > 0 and not key_form else None
        keys = torch.empty(q.n, k, device=dev, dtype=torch.int64) if k > 0 and key_form else None
    prm = OtParams(float(blur), float(scaling), float(sent_sm_temp), cdist_mode, (_lib.OT_FLAG_ONE_FORM if one_form else 0) | (_lib.OT_FLAG_CENTER if c.center_hint() else 0))
    qs, cs = q.struct(), c.struct()
    if workspace is None:
        nbytes = lib.aspire_ot_rank_batch_workspace_bytes(ctypes.byref(qs), ctypes.byref(cs), max_job, k)
        workspace = torch.empty(max(nbytes, 16), device=dev, dtype=torch.uint8)
    check(lib.aspire_ot_rank_batch_f32(ctypes.byref(qs), ctypes.byref(cs), D, _ptr(job_off), max_job, ctypes.byref(prm), want,
                                       _ptr(scores), k, _ptr(job_base), _ptr(top_s), _ptr(top_i), _ptr(keys), _ptr(workspace),
                                       workspace.numel(), _stream()))
    return (scores, keys) if key_form else (scores, top_s, top_i)


def l2max_rank_batch(q, c, job_off, max_job, k, cdist_mode=_lib.CDIST_AUTO, out=None, workspace=None, job_base=None, key_form=False,
                     one_form=False):
    """tsAspire over J independent (query, pool) jobs in ONE call (include/aspire_hip.h: aspire_l2max_rank_batch_f32); arguments
    and returns as ot_rank_batch: (scores [C], top_scores [J, k], top_idx [J, k]) or (scores, keys [J, k]) with key_form."""
    dev = q.rows.device
    if one_form:
        cdist_mode |= _lib.CDIST_ONE_FORM
    if c.center_hint():
        cdist_mode |= _lib.CDIST_CENTER
    _i32(job_off, 'job_off')
    assert job_off.numel() == q.n + 1, 'job_off must have one entry per job plus one'
    keys = None
    if out is not None and key_form:
        scores, keys = out
        top_s = top_i = None
    elif out is not None:
        scores, top_s, top_i = out
    else:
        scores = torch.empty(c.n, device=dev, dtype=torch.float32)
        top_s = torch.empty(q.n, k, device=dev, dtype=torch.float32) if k > 0 and not key_form else None
        top_i = torch.empty(q.n, k, device=dev, dtype=torch.int64) if k > 0 and not key_form else None
        keys = torch.empty(q.n, k, device=dev, dtype=torch.int64) if k > 0 and key_form else None
    qs, cs = q.struct(), c.struct()
    if workspace is None:
        nbytes = lib.aspire_l2max_rank_batch_workspace_bytes(ctypes.byref(qs), ctypes.byref(cs), max_job, k)
        workspace = torch.empty(max(nbytes, 16), device=dev, dtype=torch.uint8)
    check(lib.aspire_l2max_rank_batch_f32(ctypes.byref(qs), ctypes.byref(cs), D, _ptr(job_off), max_job, cdist_mode, _ptr(scores), k,
                                          _ptr(job_base), _ptr(top_s), _ptr(top_i), _ptr(keys), _ptr(workspace), workspace.numel(),
                                          _stream()))
    return (scores, keys) if key_form else (scores, top_s, top_i)


def topk_desc(scores, k, idx_base=0):
    """A12 (evaluate.py:76): scores [Q, C] -> (top_scores [Q,k], top_idx [Q,k] int64), stable descending."""
    _f32(scores, 'scores')
    qn, cn = scores.shape
    top_s = torch.empty(qn, k, device=scores.device, dtype=torch.float32)
    top_i = torch.empty(qn, k, device=scores.device, dtype=torch.int64)
    nbytes = lib.aspire_topk_workspace_bytes(qn, cn, k)
    ws = torch.empty(max(nbytes, 8), device=scores.device, dtype=torch.uint8)
    check(lib.aspire_topk_desc_f32(_ptr(scores), qn, cn, k, idx_base, _ptr(top_s), _ptr(top_i), _ptr(ws), nbytes,
                                   _stream()))
    return top_s, top_i


def topk_keys(scores, k, idx_base=0, out=None):
    """Per-query local top-k of scores [Q, C] as sortable int64 keys [Q, k] that carry the GLOBAL candidate index
    (include/aspire_hip.h: aspire_topk_keys_f32) -- what a shard contributes to the all-gather of section 8(e)."""
    _f32(scores, 'scores')
    qn, cn = scores.shape
    keys = out if out is not None else torch.empty(qn, k, device=scores.device, dtype=torch.int64)
    nbytes = lib.aspire_topk_workspace_bytes(qn, cn, k)
    ws = torch.empty(max(nbytes, 8), device=scores.device, dtype=torch.uint8)
    check(lib.aspire_topk_keys_f32(_ptr(scores), qn, cn, k, idx_base, _ptr(keys), _ptr(ws), nbytes, _stream()))
    return keys


def topk_merge_keys(keys, k):
    """keys [R, Q, k_in] (R shards' blocks as an all-gather leaves them) -> (top_scores [Q, k], top_idx [Q, k])."""
    require_gpu()
    assert keys.dtype == torch.int64 and keys.dim() == 3 and keys.is_contiguous() and keys.is_cuda, 'keys: int64 [R, Q, k_in]'
    r, qn, k_in = keys.shape
    top_s = torch.empty(qn, k, device=keys.device, dtype=torch.float32)
    top_i = torch.empty(qn, k, device=keys.device, dtype=torch.int64)
    check(lib.aspire_topk_merge_keys(_ptr(keys), r, qn, k_in, k, _ptr(top_s), _ptr(top_i), _stream()))
    return top_s, top_i


def selftest_xlane():
    require_gpu()
    n = (ctypes.c_int * 16)()
    check(lib.aspire_selftest_xlane(n))
    return sum(n), list(n)


def clock_under(fn, wall_us=4000, reps=None):
    """GHz the shader engines hold while fn() runs repeatedly on the current stream (aspire_debug_clock_probe on a side stream)."""
    side = torch.cuda.Stream()
    out = torch.zeros(2, device='cuda', dtype=torch.int64)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    check(lib.aspire_debug_clock_probe(_ptr(out), int(wall_us), ctypes.c_void_p(side.cuda_stream)))
    if reps is None:
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        fn()
        t1.record()
        t1.synchronize()
        reps = max(2, int(1.3 * wall_us / 1e3 / max(t0.elapsed_time(t1), 1e-3)))
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ticks, wall = out.tolist()
    return ticks / max(wall, 1) * 0.1
