"""Ranking-loop entry points: the callers either side of the distance functions.

Reference call stacks:
  evaluate.py:58-76      per query: get_similarity(query, cand) for every candidate, one pair per call
                         (src/evaluation/utils/models.py:190-197), then a stable descending sort
  pp_gen_nearest.py:131-204  per query: caching_score on consecutive groups of 64 candidates
                         (src/learning/facetid_models/disent_models.py:256-342), then sorted(...)

Here a whole candidate pool is scored by ONE launch against sentence reps that stay resident in HBM; the
two reference loops differ only in how geomloss's epsilon schedule is grouped, which is the
``schedule`` argument.
"""
import ctypes

import numpy as np
import torch

from . import _lib, ops
from .pair_distances import (rep_len_tup, AllPairMaskedWasserstein, AllPairMaskedAttention, allpair_masked_dist_l2max,
                             allpair_masked_dist_l2topk)


class CandidatePool:
    """Sentence reps of a candidate pool, resident on the GPU as rows + CSR (struct aspire_repset)."""

    def __init__(self, reps_list, pids=None):
        self.repset = ops.DeviceRepSet.from_list(reps_list)
        self.pids = list(pids) if pids is not None else list(range(self.repset.n))
        assert len(self.pids) == self.repset.n

    @classmethod
    def from_repset(cls, repset, pids=None):
        """Wrap reps that are already resident (AspireConSent.encode_to_pool writes them there): no copy."""
        self = cls.__new__(cls)
        self.repset = repset
        self.pids = list(pids) if pids is not None else list(range(repset.n))
        assert len(self.pids) == repset.n
        return self

    def __len__(self):
        return self.repset.n

    def prepare_planes(self, mu=None):
        """Keep the pool's rows a second time as fp16 planes (ops.DeviceRepSet.prepare_planes; include/aspire_hip.h:
        aspire_rep_planes): calls with many queries (more than 64 query rows) then form their costs on the fp16 matrix pipe --
        32 queries x 50 000 candidates max-sim 1.08 -> 0.49 ms.  Once per resident pool; 4 B per element more HBM.
        mu: the common vector to centre on (a [768] GPU tensor); default: the mean of a sample of the pool's rows."""
        self.repset.prepare_planes(mu=mu)
        self.repset.prepare_boxes()          # + the documents' boxes (geomloss's diameter in the many-query otAspire calls): 6 KB each
        return self


def _as_pool(x):
    return x if isinstance(x, CandidatePool) else CandidatePool(x)


class PoolBatch:
    """J candidate pools over ONE resident row matrix, laid back to back the way aspire_ot_rank_batch_f32 takes them: the jobs'
    (start, len) index lists and job_off, uploaded ONCE as one int32 buffer and kept.  RepStore.pool_batch builds (and caches)
    them, so that a score step over the same pools -- another facet, another aggregation, the next repetition -- re-uses the
    device tables instead of rebuilding them per call (torch.cat + cumsum + small uploads were ~100 x the kernels' time on the
    config-4 shape)."""

    def __init__(self, rows, starts, lens, pids_list):
        dev = rows.device
        self.sizes = [len(x) for x in lens]
        self.pids = [np.asarray(p, dtype=object) for p in pids_list]
        c_total = int(sum(self.sizes))
        flat_start = np.concatenate(starts).astype(np.int32) if c_total else np.zeros(0, np.int32)
        flat_len = np.concatenate(lens).astype(np.int32) if c_total else np.zeros(0, np.int32)
        if c_total and int(flat_len.min()) <= 0:
            raise ValueError('a document without sentence rows cannot be scored (the reference raises on it: pair_distances.py:57)')
        job_off = np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int32)
        meta = torch.from_numpy(np.concatenate([flat_start, flat_len, job_off])).to(dev)
        self.max_job = max(self.sizes) if self.sizes else 0
        self.job_off = meta[2 * c_total:]
        self.c = ops.DeviceRepSet(rows, meta[:c_total], meta[c_total:2 * c_total], ext=0,
                                  max_len=int(flat_len.max()) if c_total else 0) if c_total else None
        self._out = {}          # (k, key_form) -> preallocated outputs + workspace of the last call with that k


_WORKSPACE = {}


def _shared_workspace(dev, nbytes):
    """ONE grow-only scratch buffer per device for the batched score + rank calls of this module: they run on the caller's stream
    and hand their results to the host before they return, so consecutive calls can share it (a cached PoolBatch used to keep
    its own -- slot_bytes x candidates, ~8 KB per candidate at 32 rows -- for as long as the RepStore cached the batch)."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    ws = _WORKSPACE.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _WORKSPACE[key] = torch.empty(max(int(nbytes), 16), device=dev, dtype=torch.uint8)
    return ws


def rank_pool_batch(query_reps_list, batch, k=None, hparams=None, method='ot', deterministic=False, sign=1.0):
    """rank_pools on a prepared PoolBatch (RepStore.pool_batch): ONE upload of the queries (rows + index list), one library call,
    two small downloads.  Returns per query [(pid, sign * score), ...] as rank_pools does (evaluate.py:77 stores -similarity:
    sign = -1 negates on the way out, one numpy multiply instead of a Python loop over the pairs)."""
    hparams = hparams or {}
    if method not in ('ot', 'l2max'):
        raise ValueError(f'Unknown aggregation: {method}')
    if hparams.get('geoml_reach', None) is not None:
        raise NotImplementedError('unbalanced OT (geoml_reach) is not built')
    assert len(query_reps_list) == len(batch.sizes), 'one pool per query'
    if batch.max_job == 0:
        return [[] for _ in batch.sizes]
    k = batch.max_job if k is None else min(k, batch.max_job)
    dev = batch.c.rows.device
    q_lens = [int(np.shape(r)[0]) for r in query_reps_list]
    if min(q_lens) <= 0:
        raise ValueError('a document without sentence rows cannot be scored (the reference raises on it: pair_distances.py:57)')
    q_rows = torch.from_numpy(np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=np.float32) for r in query_reps_list], 0))).to(dev)
    q_len_np = np.asarray(q_lens, dtype=np.int32)
    q_meta = torch.from_numpy(np.concatenate([np.cumsum(q_len_np) - q_len_np, q_len_np]).astype(np.int32)).to(dev)
    j = len(q_lens)
    q = ops.DeviceRepSet(q_rows, q_meta[:j], q_meta[j:], ext=0, max_len=max(q_lens))
    slot = batch._out.get((k, method))
    if slot is None:
        slot = batch._out[(k, method)] = {
            'out': (torch.empty(batch.c.n, device=dev), torch.empty(j, k, device=dev), torch.empty(j, k, device=dev, dtype=torch.int64)),
            }
    # the workspace depends on the QUERIES too (longest document of the call: slot size, record count): asked for on every call
    # -- a host-side computation -- and grown when another facet's queries need more than the last call's
    qs, cs = q.struct(), batch.c.struct()
    need = (_lib.lib.aspire_l2max_rank_batch_workspace_bytes if method == 'l2max' else _lib.lib.aspire_ot_rank_batch_workspace_bytes)(
        ctypes.byref(qs), ctypes.byref(cs), batch.max_job, k)
    slot['ws'] = _shared_workspace(dev, need)
    if method == 'l2max':
        _, top_s, top_i = ops.l2max_rank_batch(q, batch.c, batch.job_off, batch.max_job, k, out=slot['out'], workspace=slot['ws'],
                                               one_form=deterministic)
    else:
        _, top_s, top_i = ops.ot_rank_batch(q, batch.c, batch.job_off, batch.max_job, k, blur=hparams.get('geoml_blur', 0.05),
                                            scaling=hparams.get('geoml_scaling', 0.9), sent_sm_temp=hparams.get('sent_sm_temp', 1.0),
                                            want=_lib.OT_SIMILARITY, out=slot['out'], workspace=slot['ws'], one_form=deterministic)
    top_s, top_i = top_s.cpu().numpy().astype(np.float64) * sign, top_i.cpu().numpy()
    ranked = []
    for pids, n, rs, ri in zip(batch.pids, batch.sizes, top_s, top_i):
        kk = min(k, n)
        ranked.append(list(zip(pids[ri[:kk]].tolist(), rs[:kk].tolist())))
    return ranked


def _cdist_runs(q, c, group):
    """torch.cdist picks its formula from the extents of the tensors it is given (direct differences up to 25 rows on
    both sides, the matmul expansion beyond).  caching_score hands it one padded group of `group` candidates at a time
    (disent_models.py:269-297 via pp_gen_nearest.py:182-196), so EVERY pair of a group whose longest candidate -- or whose
    query -- has more than 25 rows takes the expansion, short documents included (2.7e-5 apart on N(0,1) data: enough to swap
    near-ties).  Returns [(lo, hi, cdist_mode)] runs of whole groups; one run (AUTO) when no document is that long."""
    if c.max_len <= 25 and q.max_len <= 25:
        return [(0, c.n, _lib.CDIST_AUTO)]
    if q.n == 1 and q.max_len > 25:
        return [(0, c.n, _lib.CDIST_MM)]
    assert q.max_len <= 25, 'several queries of which some are long are scored one query per call (score_pool does)'
    lens = c.host_lens()
    runs = []
    for lo in range(0, c.n, group):
        hi = min(c.n, lo + group)
        mode = _lib.CDIST_MM if max(lens[lo:hi]) > 25 else _lib.CDIST_DIRECT
        if runs and runs[-1][2] == mode:
            runs[-1] = (runs[-1][0], hi, mode)
        else:
            runs.append((lo, hi, mode))
    return runs


def score_pool(query_reps_list, pool, method='ot', schedule='pair', hparams=None, score_batch_size=64, deterministic=False):
    """Scores [Q, C] (GPU tensor, higher = more similar) of every query against every candidate.

    deterministic ('ot' with the per-pair schedule, 'l2max'): every pair through ONE kernel form (include/aspire_hip.h:
    ASPIRE_OT_FLAG_ONE_FORM), so that a pair's score -- and with it the order of near-ties -- does not depend on the size of
    the call it is scored in: score_pool / rank_pool per query and rank_pools over all queries then agree bit for bit.
    Several times slower than the default, which picks the kernel family by grid size (scores a few 1e-5 apart).

    method   'ot'     otAspire.  schedule 'pair': -OT_eps distance, one epsilon schedule per pair, exactly
                      AspireModel.get_similarity (models.py:190-197).  schedule 'batch': plan-weighted
                      similarity with one schedule per consecutive group of `score_batch_size`
                      candidates, exactly caching_score with return_pair_sims=True
                      (disent_models.py:297, pp_gen_nearest.py:182-196).
             'l2max'  tsAspire max-sim (caching_score's 'l2lse' branch, disent_models.py:294-295).
             'l2top2' / 'l2attention'  the sibling aggregations (disent_models.py:238-245); hparams['cdatt_sm_temp'].
    """
    hparams = hparams or {}
    pool = _as_pool(pool)
    q = ops.DeviceRepSet.from_list(query_reps_list)
    c = pool.repset
    if method not in ('ot', 'l2max', 'l2top2', 'l2attention'):
        raise ValueError(f'Unknown aggregation: {method}')
    if schedule not in ('pair', 'batch'):
        raise ValueError(f'Unknown schedule: {schedule}')
    # schedule 'pair': one pair per reference call (evaluate.py) -> cdist's formula per pair (AUTO).  'batch': per padded
    # group of score_batch_size candidates (caching_score) -> per group, see _cdist_runs.
    if schedule == 'batch' and q.n > 1 and q.max_len > 25 and c.n > 0:
        # the formula is per (query, group) and a launch takes one mode: long queries go one per call
        return torch.cat([score_pool([qr], pool, method, schedule, hparams, score_batch_size, deterministic) for qr in query_reps_list],
                         dim=0)
    runs = _cdist_runs(q, c, score_batch_size) if schedule == 'batch' and q.n > 0 and c.n > 0 else [(0, c.n, _lib.CDIST_AUTO)]
    if len(runs) > 1:
        parts = [_score_run(q, c.slice(lo, hi), method, schedule, hparams, score_batch_size, mode, deterministic)
                 for lo, hi, mode in runs]
        return torch.cat(parts, dim=1)
    return _score_run(q, c, method, schedule, hparams, score_batch_size, runs[0][2], deterministic)


def _score_run(q, c, method, schedule, hparams, score_batch_size, cdist_mode, deterministic=False):
    if deterministic and not (method == 'l2max' or (method == 'ot' and schedule == 'pair')):
        raise ValueError("deterministic=True is built for method 'ot' with schedule 'pair' and for 'l2max'")
    if method == 'l2max':
        return ops.l2max_scores(q, c, pairing=_lib.PAIR_CROSS, cdist_mode=cdist_mode, one_form=deterministic).view(q.n, c.n)
    if method == 'l2top2':
        return ops.l2agg_scores(q, c, _lib.AGG_TOP2, pairing=_lib.PAIR_CROSS, cdist_mode=cdist_mode).view(q.n, c.n)
    if method == 'l2attention':
        return ops.l2agg_scores(q, c, _lib.AGG_ATTENTION, temp=hparams.get('cdatt_sm_temp', 1.0),
                                pairing=_lib.PAIR_CROSS, cdist_mode=cdist_mode).view(q.n, c.n)
    kw = dict(blur=hparams.get('geoml_blur', 0.05), scaling=hparams.get('geoml_scaling', 0.9),
              sent_sm_temp=hparams.get('sent_sm_temp', 1.0), cdist_mode=cdist_mode)
    if hparams.get('geoml_reach', None) is not None:
        raise NotImplementedError('unbalanced OT (geoml_reach) is not built')
    if schedule == 'pair':
        if 1 < q.n <= 32 and q.max_len <= 8 and 8 < c.max_len <= 32 and c.ext == 0 and c.n >= 256 and not deterministic:
            # a few short queries (the facets of one paper, models.py:127-163) against ONE pool of whole abstracts: as batched jobs
            # over the same index list -- the fused kernel's CHUNK form scores each of them in the one launch (the cross-product
            # entry would take the Gram tiles + a Sinkhorn launch, built for many queries)
            dev = c.rows.device
            cc = ops.DeviceRepSet(c.rows, c.start.repeat(q.n), c.len.repeat(q.n), ext=0, max_len=c.max_len)
            qs, cs = q.struct(), cc.struct()
            need = _lib.lib.aspire_ot_rank_batch_workspace_bytes(ctypes.byref(qs), ctypes.byref(cs), c.n, 0)
            if need <= (1 << 30):      # (the batched entry reserves pair slots for EVERY candidate: beyond 1 GiB the chunked path below)
                job_off = (torch.arange(q.n + 1, dtype=torch.int64) * c.n).to(torch.int32).to(dev)
                sims, _, _ = ops.ot_rank_batch(q, cc, job_off, c.n, 0, want=_lib.OT_SIMILARITY, workspace=_shared_workspace(dev, need), **kw)
                return sims.view(q.n, c.n)
        dist = ops.ot_sinkhorn(q, c, pairing=_lib.PAIR_CROSS, want=_lib.OT_DISTANCE, one_form=deterministic, **kw)
        return (-dist).view(q.n, c.n)
    diam = ops.group_diameter(q, c, _lib.PAIR_CROSS, group=score_batch_size)
    sims = ops.ot_sinkhorn(q, c, pairing=_lib.PAIR_CROSS, want=_lib.OT_PLAN_SIM, diameter=diam,
                           diam_group=score_batch_size, **kw)
    return sims.view(q.n, c.n)


def rank_pool(query_reps_list, pool, k=None, method='ot', schedule='pair', hparams=None, score_batch_size=64, deterministic=False):
    """Per query: [(pid, score), ...] best first, ties in pool order (evaluate.py:76).  otAspire goes through ONE
    C-ABI call that scores and ranks (aspire_ot_rank_f32).  deterministic: see score_pool."""
    if deterministic and not (method == 'l2max' or (method == 'ot' and schedule == 'pair')):
        raise ValueError("deterministic=True is built for method 'ot' with schedule 'pair' and for 'l2max'")
    pool = _as_pool(pool)
    if len(pool) == 0:
        return [[] for _ in query_reps_list]
    k = len(pool) if k is None else min(k, len(pool))
    q = ops.DeviceRepSet.from_list(query_reps_list)
    # one fused score + rank call unless caching_score's per-group cdist formula differs between groups (documents of more
    # than 25 rows, see _cdist_runs)
    one_call = method == 'ot' and (schedule == 'pair' or (schedule == 'batch' and (
        q.max_len <= 25 or q.n == 1) and len(_cdist_runs(q, pool.repset, score_batch_size)) == 1))
    if one_call:
        hparams = hparams or {}
        if hparams.get('geoml_reach', None) is not None:
            raise NotImplementedError('unbalanced OT (geoml_reach) is not built')
        kw = dict(blur=hparams.get('geoml_blur', 0.05), scaling=hparams.get('geoml_scaling', 0.9),
                  sent_sm_temp=hparams.get('sent_sm_temp', 1.0))
        if schedule == 'pair':
            _, top_s, top_i = ops.ot_rank(q, pool.repset, k, want=_lib.OT_SIMILARITY, one_form=deterministic, **kw)
        else:
            diam = ops.group_diameter(q, pool.repset, _lib.PAIR_CROSS, group=score_batch_size)
            _, top_s, top_i = ops.ot_rank(q, pool.repset, k, want=_lib.OT_PLAN_SIM, diameter=diam,
                                          diam_group=score_batch_size,
                                          cdist_mode=_cdist_runs(q, pool.repset, score_batch_size)[0][2], **kw)
    else:
        scores = score_pool(query_reps_list, pool, method=method, schedule=schedule, hparams=hparams,
                            score_batch_size=score_batch_size, deterministic=deterministic)
        top_s, top_i = ops.topk_desc(scores.contiguous(), k)
    top_s, top_i = top_s.cpu().numpy(), top_i.cpu().numpy()
    return [[(pool.pids[i], float(s)) for s, i in zip(rs, ri) if i >= 0] for rs, ri in zip(top_s, top_i)]


def _launch_rank_pools(query_reps_list, pools, k, hparams, method='ot', deterministic=False):
    """Uploads + the one library call of rank_pools on the CURRENT stream; returns (pools, top_scores, top_idx) GPU tensors
    (None for the tensors when every pool is empty)."""
    hparams = hparams or {}
    if method not in ('ot', 'l2max'):
        raise ValueError(f'Unknown aggregation: {method}')
    if hparams.get('geoml_reach', None) is not None:
        raise NotImplementedError('unbalanced OT (geoml_reach) is not built')
    assert len(query_reps_list) == len(pools), 'one pool per query'
    pools = [_as_pool(p) for p in pools]
    sizes = [len(p) for p in pools]
    max_job = max(sizes) if sizes else 0
    if max_job == 0:
        return pools, None, None
    k = max_job if k is None else min(k, max_job)
    dev = ops.require_gpu()
    q = ops.DeviceRepSet.from_list(query_reps_list)
    nonempty = [p.repset for p in pools if len(p)]
    if all(r.rows.data_ptr() == nonempty[0].rows.data_ptr() and r.rows.shape == nonempty[0].rows.shape for r in nonempty):
        # every pool indexes ONE resident matrix (RepStore.to_device): the jobs' candidates are its index lists back to back
        c = ops.DeviceRepSet(nonempty[0].rows, torch.cat([r.start for r in nonempty]).contiguous(),
                             torch.cat([r.len for r in nonempty]).contiguous(), ext=0, max_len=max(r.max_len for r in nonempty))
    else:
        # pools over several row matrices (some resident, some uploaded per pool): every DISTINCT matrix is concatenated once --
        # a resident pool's `rows` is the whole store, and one copy of it per pool would be tens of GB for a real dataset
        mats, base_of, total = [], {}, 0
        for r in nonempty:
            key = (r.rows.data_ptr(), tuple(r.rows.shape))
            if key not in base_of:
                base_of[key] = total
                mats.append(r.rows)
                total += int(r.rows.shape[0])
        bases = [base_of[(r.rows.data_ptr(), tuple(r.rows.shape))] for r in nonempty]
        c = ops.DeviceRepSet(torch.cat(mats, 0),
                             torch.cat([r.start + int(b) for r, b in zip(nonempty, bases)]).to(torch.int32).contiguous(),
                             torch.cat([r.len for r in nonempty]).contiguous(), ext=0, max_len=max(r.max_len for r in nonempty))
    job_off = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
    if method == 'l2max':
        _, top_s, top_i = ops.l2max_rank_batch(q, c, job_off, max_job, k, one_form=deterministic)
        return pools, top_s, top_i
    _, top_s, top_i = ops.ot_rank_batch(q, c, job_off, max_job, k, blur=hparams.get('geoml_blur', 0.05),
                                        scaling=hparams.get('geoml_scaling', 0.9), sent_sm_temp=hparams.get('sent_sm_temp', 1.0),
                                        want=_lib.OT_SIMILARITY, one_form=deterministic)
    return pools, top_s, top_i


def _ranked_lists(pools, top_s, top_i):
    if top_s is None:
        return [[] for _ in pools]
    top_s, top_i = top_s.cpu().numpy(), top_i.cpu().numpy()
    return [[(p.pids[i], float(sc)) for sc, i in zip(rs, ri) if i >= 0] for p, rs, ri in zip(pools, top_s, top_i)]


def rank_pools(query_reps_list, pools, k=None, hparams=None, method='ot', deterministic=False):
    """The whole per-query loop of evaluate.py:58-76 in ONE library call: query j is scored against ITS OWN pool
    pools[j] (every query of a dataset has its own candidate pool, evaluate.py:60-62) with otAspire, one epsilon schedule
    per pair (AspireModel.get_similarity, models.py:190-197) -- or, method='l2max', tsAspire's max-sim -- and each pool is
    ranked on its own (stable descending, evaluate.py:76).  pools: list of CandidatePool or lists of [S_i, 768] arrays.
    Returns per query [(pid, score), ...].  deterministic: one kernel form whatever the batch's size -- the same bits and the
    same order as rank_pool(..., deterministic=True) query by query (see score_pool)."""
    if not pools:
        return []
    return _ranked_lists(*_launch_rank_pools(query_reps_list, pools, k, hparams, method, deterministic))


class InFlightRanker:
    """rank_pools for a process that serves independent requests: up to `n_lanes` calls in flight, each on its own stream
    with its own buffers.  The library queues everything on the caller's stream and keeps no state between calls, so calls on
    different streams overlap -- the end of one call (last Sinkhorn solves, rank launch: no HBM traffic) runs beside the next
    call's streaming (bench.py: 93 instead of 105 us per 20-query call).

        ranker = InFlightRanker()
        tickets = [ranker.submit(queries_b, pools_b) for queries_b, pools_b in requests]
        ranked = [ranker.result(t) for t in tickets]          # same lists as rank_pools(queries_b, pools_b)
    """

    def __init__(self, n_lanes=3, k=None, hparams=None, method='ot'):
        dev = ops.require_gpu()
        self.k, self.hparams, self.method = k, hparams, method
        self.lanes = [torch.cuda.Stream(dev) for _ in range(max(1, int(n_lanes)))]
        self.busy = [None] * len(self.lanes)          # the lane's last ticket (its stream runs calls in order)
        self.turn = 0

    def submit(self, query_reps_list, pools):
        lane = self.turn % len(self.lanes)
        self.turn += 1
        with torch.cuda.stream(self.lanes[lane]):
            out = _launch_rank_pools(query_reps_list, pools, self.k, self.hparams, self.method) if pools else ([], None, None)
            done = torch.cuda.Event()
            done.record()
        ticket = {'lane': lane, 'out': out, 'done': done}
        self.busy[lane] = ticket
        return ticket

    def result(self, ticket):
        ticket['done'].synchronize()
        with torch.cuda.stream(self.lanes[ticket['lane']]):
            return _ranked_lists(*ticket['out'])


def get_similarity(x, y, hparams=None):
    """AspireModel.get_similarity (src/evaluation/utils/models.py:190-197): one pair, -OT distance."""
    dist_func = AllPairMaskedWasserstein(hparams or {})
    x, y = torch.as_tensor(x), torch.as_tensor(y)
    xt = rep_len_tup(embed=x[None, :].permute(0, 2, 1), abs_lens=[len(x)])
    yt = rep_len_tup(embed=y[None, :].permute(0, 2, 1), abs_lens=[len(y)])
    ot_dist = dist_func.compute_distance(query=xt, cand=yt).item()
    return -ot_dist


def caching_score(query_encode_ret_dict, cand_encode_ret_dicts, score_agg_type='l2wasserstein', hparams=None,
                  sent_loss_prop=1.0, abs_loss_prop=0.0):
    """WordSentAlignBiEnc.caching_score (src/learning/facetid_models/disent_models.py:256-342): the sentence-level term
    scaled by sent_loss_prop (:300-304) plus, when abs_loss_prop > 0 (:305-307; the model classes read both from their
    hyper-parameters, disent_models.py:583, 714 -- every published config sets abs_loss_prop 0.0), abs_loss_prop times the
    negative L2 distance of the documents' CLS reps ('doc_cls_reps' of the encode dicts).
    Returns {'batch_scores': np.ndarray [B], 'pair_scores': un-padded per-candidate extras}."""
    query_sent_reps = np.asarray(query_encode_ret_dict['sent_reps'])
    cand_sent_reps = [np.asarray(d['sent_reps']) for d in cand_encode_ret_dicts]
    batch_size = len(cand_sent_reps)
    cand_lens = [r.shape[0] for r in cand_sent_reps]
    cmax_sents = max(cand_lens)
    qmax_sents, encoding_dim = query_sent_reps.shape
    query_lens = [qmax_sents] * batch_size
    dev = ops.require_gpu()
    padded_cand = torch.zeros(batch_size, cmax_sents, encoding_dim, device=dev)
    for bi, ex_reps in enumerate(cand_sent_reps):
        padded_cand[bi, :cand_lens[bi], :] = torch.as_tensor(ex_reps, dtype=torch.float32)
    padded_query = torch.as_tensor(query_sent_reps, dtype=torch.float32, device=dev).unsqueeze(0) \
        .expand(batch_size, -1, -1).contiguous()
    qt = rep_len_tup(embed=padded_query.permute(0, 2, 1), abs_lens=query_lens)
    ct = rep_len_tup(embed=padded_cand.permute(0, 2, 1), abs_lens=cand_lens)
    if score_agg_type in {'l2lse', 'l2max'}:
        batch_sent_sims, pair_sims = allpair_masked_dist_l2max(query=qt, cand=ct, return_pair_sims=True)
    elif score_agg_type == 'l2top2':
        batch_sent_sims, pair_sims = allpair_masked_dist_l2topk(query=qt, cand=ct, return_pair_sims=True)
    elif score_agg_type == 'l2attention':
        batch_sent_sims, pair_sims = AllPairMaskedAttention(hparams or {}).compute_distance(
            query=qt, cand=ct, return_pair_sims=True)
    elif score_agg_type == 'l2wasserstein':
        batch_sent_sims, pair_sims = AllPairMaskedWasserstein(hparams or {}).compute_distance(
            query=qt, cand=ct, return_pair_sims=True)
    else:
        raise ValueError(f'Unknown aggregation: {score_agg_type}')
    batch_scores = sent_loss_prop * batch_sent_sims
    if abs_loss_prop > 0.0:
        query_cls_reps = torch.as_tensor(np.vstack([query_encode_ret_dict['doc_cls_reps']] * batch_size), dtype=torch.float32, device=dev)
        cand_cls_reps = torch.as_tensor(np.vstack([d['doc_cls_reps'] for d in cand_encode_ret_dicts]), dtype=torch.float32, device=dev)
        batch_doc_sims = -1 * ops.cls_l2(query_cls_reps.contiguous(), cand_cls_reps.contiguous(), pairing=_lib.PAIR_PAIRED)
        batch_scores = batch_scores + abs_loss_prop * batch_doc_sims.to(batch_scores.device)
    batch_scores = batch_scores.cpu().numpy()
    if isinstance(pair_sims, list):
        pair_sims = [t.cpu().numpy() for t in pair_sims]
    else:
        pair_sims = pair_sims.cpu().numpy()
    unpadded = []
    for i, (clen, qlen) in enumerate(zip(cand_lens, query_lens)):
        if len(pair_sims) == 5 and isinstance(pair_sims, list):
            upsm = [pair_sims[0][i, :qlen], pair_sims[1][i, :clen], pair_sims[2][i, :qlen, :clen],
                    pair_sims[3][i, :qlen, :clen], pair_sims[4][i, :qlen, :clen]]
        elif isinstance(pair_sims, list):   # attention distance (disent_models.py:329-332)
            upsm = [pair_sims[0][i, :qlen, :clen], pair_sims[1][i, :qlen, :clen], pair_sims[2][i, :qlen, :clen]]
        else:
            upsm = pair_sims[i, :qlen, :clen]
        unpadded.append(upsm)
    return {'batch_scores': batch_scores, 'pair_scores': unpadded}
