"""Candidate-pool sharding across the GPUs of one node.

Every (query, candidate) pair is independent, so the pool is split into contiguous blocks in pool
order, one per rank (one process per GPU); queries are replicated.  Each rank scores its block with the
HIP kernels and keeps a per-query local top-k; the only exchange is one all-gather of k (score, global
index) pairs per query per rank -- RCCL over xGMI on the GPUs (torch.distributed backend "nccl"),
gloo in the CPU tests -- followed by a k-way merge.  The reference has no collective on this path
(its ranking loops are single process: evaluate.py:58-76, pp_gen_nearest.py:131-204).

Tie rule: equal scores are ordered by ascending GLOBAL candidate index, which is what Python's stable
sorted(..., reverse=True) over the un-sharded pool gives (evaluate.py:76).

Which split a BASELINE config takes:
  configs 2, 3, 5 (few / many queries against ONE big pool)   candidate blocks: ShardedPoolRanker (queries replicated, top-k merge)
  config 4 (CSFCube: every query has its OWN pool of ~125 candidates, evaluate.py:58-76)   JOB blocks: rank_pools_sharded /
      evaluate.score(..., group=...) -- a pool is never cut (125 candidates over 8 ranks on multiples of 64 would leave six ranks
      empty); the 50 (query, pool) jobs are dealt out in contiguous blocks, each rank makes ONE aspire_ot_rank_batch_f32 call on its
      block, and ONE all-gather of the ranked [jobs, k] lists (score bits + in-pool index in one int64 each) gives every rank
      every job's ranking.  No merge: a job's ranking is complete on the rank that owns it.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, world_size, rank, multiple=1):
    """Contiguous block [lo, hi) of rank `rank`; block edges fall on multiples of `multiple` (64 keeps
    caching_score's batch-of-64 epsilon-schedule groups intact across shards, pp_gen_nearest.py:182)."""
    units = (n_items + multiple - 1) // multiple
    base, extra = divmod(units, world_size)
    lo_u = rank * base + min(rank, extra)
    hi_u = lo_u + base + (1 if rank < extra else 0)
    return min(lo_u * multiple, n_items), min(hi_u * multiple, n_items)


def all_gather_flat(out, inp, group=None):
    """dist.all_gather_into_tensor(out, inp) for 1-D tensors.  RCCL ("nccl") gathers GPU tensors in place over xGMI; the gloo
    backend (CPU tests, the one-GPU rehearsals of the multi-rank code) has no GPU all-gather, so there the data takes a
    round trip through host memory."""
    if dist.get_backend(group) == 'gloo' and inp.is_cuda:
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, inp.cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)
    return out


def merge_topk(scores, idx, k):
    """scores/idx [Q, M] candidates from all shards (idx = global ids, -1 = padding) ->
    (top scores [Q, k], top idx [Q, k]) descending, ties by ascending global index.
    Plain torch ops on whatever device the tensors live on: M = world_size * k is tiny."""
    pad = idx < 0
    scores = torch.where(pad, torch.full_like(scores, float('-inf')), scores)
    # order by index first, then a stable descending sort on score keeps index order inside ties
    key = torch.where(pad, torch.full_like(idx, torch.iinfo(idx.dtype).max), idx)
    by_idx = torch.argsort(key, dim=1, stable=True)
    s1, i1 = torch.gather(scores, 1, by_idx), torch.gather(idx, 1, by_idx)
    by_score = torch.argsort(s1, dim=1, descending=True, stable=True)
    s2, i2 = torch.gather(s1, 1, by_score), torch.gather(i1, 1, by_score)
    k = min(k, scores.shape[1])
    return s2[:, :k].contiguous(), i2[:, :k].contiguous()


def all_gather_topk(local_scores, local_idx, k, group=None):
    """All-gather every rank's per-query local top-k and merge.  local_* are [Q, k_local]."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return merge_topk(local_scores, local_idx, k)
    world = dist.get_world_size(group)
    qn, kl = local_scores.shape
    # one fused buffer per rank: scores as fp32 bits next to int64 indices would need two collectives;
    # pack both into int64 (score bits in the low word) so ONE all_gather moves everything.
    packed = torch.stack([local_scores.contiguous().view(torch.int32).to(torch.int64), local_idx], dim=-1)
    flat = packed.contiguous().view(-1)
    out = torch.empty(world * flat.numel(), dtype=packed.dtype, device=packed.device)
    all_gather_flat(out, flat, group)                      # rank-major concatenation on every backend
    out = out.view(world, qn, kl, 2).permute(1, 0, 2, 3).reshape(qn, world * kl, 2)
    scores = out[..., 0].to(torch.int32).view(torch.float32)
    return merge_topk(scores, out[..., 1].contiguous(), k)


def all_gather_topk_keys(local_keys, k, group=None):
    """GPU form of the exchange: every rank contributes its local top-k as sortable keys that already carry global
    candidate indices (ops.topk_keys); ONE all-gather, then one merge kernel (ops.topk_merge_keys).
    local_keys int64 [Q, k_local] on the GPU -> (top scores [Q, k], top idx [Q, k])."""
    from . import ops
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        world = dist.get_world_size(group)
        gathered = torch.empty((world,) + tuple(local_keys.shape), dtype=local_keys.dtype, device=local_keys.device)
        all_gather_flat(gathered.view(-1), local_keys.contiguous().view(-1), group)
    else:
        gathered = local_keys.contiguous().unsqueeze(0)
    return ops.topk_merge_keys(gathered, k)


class ShardedPoolRanker:
    """Holds this rank's block of a candidate pool resident in HBM and ranks queries against the whole
    pool.  `pool_reps` is the FULL pool (list of [S_i, 768] arrays) or, with `presharded=True`, only this
    rank's block together with `global_offset` (and `n_total` for rank_queries_full, which also requires the block to be
    exactly shard_bounds(n_total, world, rank, multiple))."""

    def __init__(self, pool_reps, presharded=False, global_offset=0, multiple=64, group=None, n_total=None):
        from .scorer import CandidatePool
        self.group = group
        self.multiple = multiple
        self.n_total = n_total if presharded else len(pool_reps)
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if presharded:
            self.lo = global_offset
            block = pool_reps
        else:
            self.lo, hi = shard_bounds(len(pool_reps), self.world, self.rank, multiple)
            block = pool_reps[self.lo:hi]
        self.pool = CandidatePool(block, pids=list(range(self.lo, self.lo + len(block))))

    @classmethod
    def from_resident(cls, pool, global_offset, n_total, multiple=64, group=None, planes=False):
        """Wrap a block that is ALREADY resident on this rank's GPU -- what AspireConSent.encode_to_pool or RepStore.pool return
        (config 5: every rank encodes its own block of the corpus straight into HBM, src/pre_process/pp_gen_nearest.py:141-202
        per rank) -- as this rank's shard: documents [global_offset, global_offset + len(pool)) of a pool of n_total.  No copy.
        planes: keep the block's rows as fp16 planes too (CandidatePool.prepare_planes), centred on RANK 0's common vector,
        broadcast once: every shard then rounds its rows around the same point."""
        self = cls.__new__(cls)
        self.group, self.multiple, self.n_total = group, multiple, n_total
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.lo = int(global_offset)
        self.pool = pool          # (its pids stay the caller's; the rankings carry GLOBAL positions lo + i: rank_queries, idx_base)
        if planes:
            self.prepare_planes()
        return self

    def prepare_planes(self):
        """fp16 planes of this rank's block around ONE centre for all shards: rank 0 forms it from a sample of its rows, one
        768-float broadcast hands it to the others (an empty rank-0 block: zeros)."""
        from . import ops
        dev = ops.require_gpu()
        mu = None
        if self.world > 1:
            if self.rank == 0 and len(self.pool) > 0:
                # rank 0 prepares its planes ONCE, around the centre it forms from its own rows; the others get that centre (given
                # the same vector the split is the same arithmetic: one pass here, not a sample pass and then the real one)
                self.pool.prepare_planes()
                mu = self.pool.repset.planes.mu.clone()
            elif self.rank == 0:
                mu = torch.zeros(768, device=dev)
            else:
                mu = torch.empty(768, device=dev)
            if dist.get_backend(self.group) == 'gloo':
                host = mu.cpu()
                dist.broadcast(host, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
                mu = host.to(dev)
            else:
                dist.broadcast(mu, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            if self.rank != 0 and len(self.pool) > 0:
                self.pool.prepare_planes(mu=mu)
        elif len(self.pool) > 0:
            self.pool.prepare_planes()
        self.mu = mu
        return self

    def rank_queries(self, query_reps_list, k, **score_kw):
        from . import ops
        from .scorer import score_pool
        key_form = self.world * k <= 4096      # one chunk of the merge kernel holds every rank's k keys
        if len(self.pool) > 0:
            scores = score_pool(query_reps_list, self.pool, **score_kw).contiguous()
            if key_form:
                return all_gather_topk_keys(ops.topk_keys(scores, k, idx_base=self.lo), k, self.group)
            ls, li = ops.topk_desc(scores, k, idx_base=self.lo)
        else:
            dev = ops.require_gpu()
            if key_form:   # an empty shard contributes padding keys
                return all_gather_topk_keys(torch.zeros(len(query_reps_list), k, dtype=torch.int64, device=dev), k, self.group)
            ls = torch.full((len(query_reps_list), k), float('-inf'), device=dev)
            li = torch.full((len(query_reps_list), k), -1, dtype=torch.int64, device=dev)
        return all_gather_topk(ls, li, k, self.group)

    def rank_queries_full(self, query_reps_list, **score_kw):
        """Full re-rank (the CSFCube / evaluate.py:76 case: the WHOLE pool is sorted, SURVEY.md 8e "all-gather all
        Q * C / 8 scores instead"): every rank scores its block, ONE all-gather moves the blocks' scores (padded to the
        longest block), and every rank sorts the un-sharded [Q, C] score matrix with the full stable sort.  Blocks are
        contiguous in pool order, so rank-major concatenation IS pool order and ties keep it.
        Returns (scores [Q, C] sorted descending, global candidate idx [Q, C])."""
        from . import ops
        from .scorer import score_pool
        dev = ops.require_gpu()
        qn = len(query_reps_list)
        sizes = [shard_bounds(self.n_total, self.world, r, self.multiple) for r in range(self.world)] if self.n_total is not None else None
        assert sizes is not None, 'rank_queries_full needs the full pool size (construct without presharded=True or pass n_total)'
        lens = [hi - lo for lo, hi in sizes]
        # a presharded caller must have cut the pool exactly as shard_bounds does (same `multiple`): the gathered [Q, C] matrix is
        # assembled from those bounds, and a block cut elsewhere would silently misalign its columns
        assert self.lo == sizes[self.rank][0] and len(self.pool) == lens[self.rank], (
            f'rank {self.rank}: block [{self.lo}, {self.lo + len(self.pool)}) is not shard_bounds({self.n_total}, {self.world}, '
            f'{self.rank}, {self.multiple}) = {sizes[self.rank]}')
        width = max(max(lens), 1)
        local = torch.full((qn, width), float('-inf'), device=dev)
        if len(self.pool) > 0:
            local[:, :len(self.pool)] = score_pool(query_reps_list, self.pool, **score_kw)
        if self.world > 1:
            gathered = torch.empty(self.world, qn, width, device=dev)
            all_gather_flat(gathered.view(-1), local.view(-1), self.group)
        else:
            gathered = local.unsqueeze(0)
        full = torch.cat([gathered[r, :, :lens[r]] for r in range(self.world)], dim=1).contiguous()     # [Q, C] in pool order
        if full.shape[1] == 0:
            return full, torch.empty(qn, 0, dtype=torch.int64, device=dev)
        return ops.topk_desc(full, full.shape[1])


# ---- config 4: per-query pools, sharded by JOB ----------------------------------------------------------------------------------
def job_bounds(n_jobs, world_size, rank):
    """Contiguous block [lo, hi) of the (query, pool) jobs that rank `rank` owns: sizes differ by at most one, earlier ranks take
    the longer blocks (50 jobs over 8 ranks: 7 7 6 6 6 6 6 6)."""
    return shard_bounds(n_jobs, world_size, rank, 1)


def pack_ranked(top_s, top_i):
    """[J, k] fp32 scores + [J, k] in-pool indices (-1 = beyond the pool's end) -> [J, k] int64: score bits in the high word, the
    index in the low word -- what one all-gather moves per job and rank."""
    bits = top_s.contiguous().view(torch.int32).to(torch.int64)
    return (bits << 32) | (top_i.to(torch.int64) & 0xffffffff)


def unpack_ranked(packed):
    """Inverse of pack_ranked: (scores fp32 [J, k], idx int64 [J, k])."""
    scores = (packed >> 32).to(torch.int32).view(torch.float32)
    idx = (packed & 0xffffffff).to(torch.int32).to(torch.int64)          # the low word, sign-extended: -1 stays -1
    return scores, idx


def all_gather_ranked_jobs(local_s, local_i, n_jobs, k, group=None, device=None):
    """Every rank holds the rankings of ITS block of jobs (job_bounds): local_s / local_i [jobs of this rank, <= k] (fewer columns
    when the rank's longest pool is shorter than k; None for a rank without jobs or whose pools are all empty).  ONE all-gather of
    [ceil(n_jobs / world), k] int64 per rank (50 jobs x 125 at world 8: 7 KB per rank) -> (scores [n_jobs, k], idx [n_jobs, k]) on every
    rank, jobs in the caller's order, columns beyond a pool's size (-inf, -1)."""
    dev = device if device is not None else (local_s.device if local_s is not None else ops_device())
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = job_bounds(n_jobs, world, rank)
    per = (n_jobs + world - 1) // world if n_jobs else 0
    s = torch.full((per, k), float('-inf'), device=dev, dtype=torch.float32)
    i = torch.full((per, k), -1, device=dev, dtype=torch.int64)
    if local_s is not None and hi > lo:
        assert local_s.shape[0] == hi - lo, (local_s.shape, lo, hi)
        kl = min(k, local_s.shape[1])
        s[:hi - lo, :kl] = local_s[:, :kl]
        i[:hi - lo, :kl] = local_i[:, :kl]
    packed = pack_ranked(s, i)
    if world > 1 and per * k:
        out = torch.empty(world * per * k, dtype=torch.int64, device=dev)
        all_gather_flat(out, packed.view(-1), group)
        out = out.view(world, per, k)
        packed = torch.cat([out[r, :job_bounds(n_jobs, world, r)[1] - job_bounds(n_jobs, world, r)[0]] for r in range(world)], 0)
    else:
        packed = packed[:n_jobs]
    return unpack_ranked(packed.contiguous())


def rank_pools_sharded(query_reps_list, pools, k=None, hparams=None, method='ot', deterministic=False, group=None, pool_sizes=None):
    """scorer.rank_pools across the ranks of `group` (config 4: 50 CSFCube queries, each against its own ~125-candidate pool,
    evaluate.py:58-76, on the 8 GPUs of a node): job j = (query j, pools[j]).  Rank r owns the contiguous block job_bounds(J, world, r)
    and only ever touches ITS queries and pools -- entries of `query_reps_list` / `pools` outside the block may be None (with
    `pool_sizes`, the pools' lengths, given on every rank), so a rank uploads only its block's candidates.  Each rank makes one
    aspire_ot_rank_batch_f32 (or aspire_l2max_rank_batch_f32) call; one all-gather exchanges the ranked lists; no merge.
    Returns (scores [J, k], idx [J, k]) on every rank: idx = position in the job's own pool, best first, ties in pool order
    (evaluate.py:76), (-inf, -1) beyond a pool's size.  k: default the longest pool (full ranking).  ranked_lists_from maps them
    to the [(pid, score), ...] lists rank_pools returns."""
    from . import scorer
    n_jobs = len(pools)
    assert len(query_reps_list) == n_jobs, 'one pool per query'
    sizes = [int(n) for n in pool_sizes] if pool_sizes is not None else [len(p) for p in pools]
    assert len(sizes) == n_jobs
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    max_job = max(sizes) if sizes else 0
    k = max_job if k is None else min(int(k), max_job)
    if n_jobs == 0 or k == 0:
        dev = ops_device()
        return torch.empty(n_jobs, 0, device=dev), torch.empty(n_jobs, 0, dtype=torch.int64, device=dev)
    lo, hi = job_bounds(n_jobs, world, rank)
    top_s = top_i = None
    if hi > lo:
        assert all(len(pools[j]) == sizes[j] for j in range(lo, hi)), 'pool_sizes disagree with this rank\'s pools'
        _, top_s, top_i = scorer._launch_rank_pools(query_reps_list[lo:hi], pools[lo:hi], k, hparams, method, deterministic)
    return all_gather_ranked_jobs(top_s, top_i, n_jobs, k, group)


def ops_device():
    from . import ops
    return ops.require_gpu()


def ranked_lists_from(pids_lists, top_s, top_i):
    """(scores [J, k], idx [J, k]) of rank_pools_sharded + every job's candidate ids -> per job [(pid, score), ...], as
    scorer.rank_pools returns them."""
    top_s, top_i = top_s.cpu().numpy(), top_i.cpu().numpy()
    return [[(pids[i], float(sc)) for sc, i in zip(rs, ri) if i >= 0] for pids, rs, ri in zip(pids_lists, top_s, top_i)]
