"""bench.py -- query x candidate OT alignments per second (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Workload at N=1 = BASELINE.json configs[1]: otAspire, 1 query x 1000 candidates, 8 sentences x 768 d, reps ~ N(0,1)
fp32, resident in HBM before the timed region.  One STEP = one query re-ranked against ITS OWN 1000-candidate pool
(evaluate.py:58-76: every query of a dataset has its own pool): cost matrix + marginals + Sinkhorn (one epsilon schedule
per pair, AspireModel.get_similarity) of the 1000 pairs, then the stable descending rank (top-100).  The K steps of a run
are K independent (query, pool) jobs and go through ONE library call, aspire_ot_rank_batch_f32 -- a small launch that
builds the job tables and query boxes (batches of more than 64 jobs only), ONE scoring launch over the K x 1000 pairs (costs and Sinkhorn solves fused: a
wave streams four candidates' rows, then solves those four pairs from its registers while other waves stream), one
K-workgroup rank launch -- two launches per call.  No hipGraphs.  Consecutive calls are independent requests and go round-robin over --streams
caller streams (default 3), each with its own outputs and workspace: the end of one call (the last Sinkhorn solves,
the rank launch -- neither touches HBM) overlaps the next call's streaming.  `one_stream` in the JSON is the same schedule
with one call at a time.

Timing: the K-step schedule is repeated R times back to back (R chosen so that the timed region is >= 50 ms: K = 20 steps
alone are ~0.1 ms) between barrier + torch.cuda.synchronize() on both sides; ms_per_step = elapsed / (K * R), value =
N * K * R * 1000 / elapsed.  Every job has its own query and its own pool, and consecutive repetitions walk through a pool
store of >= 2 K (at least 24) distinct pools (> 256 MiB, the Infinity Cache): the candidate reads come from HBM, not L3.

At N > 1 every rank holds its own 1000-candidate block of each job's N * 1000-candidate pool (weak scaling), ranks its
block (key form, global indices), and the per-job top-k lists are merged with ONE RCCL all-gather per call + one merge
kernel.  No data-path collective.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (the scoring kernel: it streams every rep once)
against HBM: algorithmic bytes per launch = 24 605 B/pair x pairs per launch over the kernel's duration
measured live with HIP events around launches of that stage alone on the launch stream
(aspire_debug_ot_rank_batch_stages_f32), on the rotating (cold) pools; `l3_resident_frac` is the same on ONE pool set small
enough to stay in the Infinity Cache; `step` prices the whole step (all kernels, timed region) against the same bytes;
`two_kernel_form` gives the durations of the separate cost and Sinkhorn kernels (OT_FORM=tile) for comparison.  `cpu_baseline` times the CPU oracle (a
port of the reference's PyTorch CPU path; its Sinkhorn solver is a parity-unpinned restatement of geomloss 0.2.4) on the
host cores of this box, with 1 thread and with all cores.

Beside the headline (N = 1): `config3` (tsAspire 32 x 50 000 x 8, blocks + single calls), `config4` (CSFCube shape), `e2e` (config 5's one-GPU
slice: tokens -> encoder -> store -> 128 queries otAspire + top-100), each with its own roofline.  The calling patterns and those figures are
ALSO plain numbers inside `roofline` (`one_stream_value`, `one_stream_frac`, `single_job_us`) and `config` (`config3_us`, `config3_us_min`,
`config3_us_p90`, `config4_ot_us`, `e2e_docs_per_s`, `e2e_encoder_frac`).  At N > 1: `rccl` (what the backend saw), `config4` (the jobs dealt
out over the ranks, one all-gather of ranked lists: config4_sharded_probe) and `e2e` (every rank encodes its block, shards, merges).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist


def _pinned(**kw):
    from aspire_amd._lib import pinned
    return pinned(**kw)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NC, S, D = 1000, 8, 768          # candidates per pool, sentences per document, encoding dim
TOPK = 100
if os.environ.get('ASPIRE_BENCH_SHAPE'):      # "C,S": tuning experiments at other pool shapes (invalid as a result)
    NC, S = (int(v) for v in os.environ['ASPIRE_BENCH_SHAPE'].split(','))
    TOPK = min(TOPK, NC)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
L3_BYTES = 256 << 20
MIN_TIMED_S = 0.05


def algorithmic_bytes(n_jobs):
    """every rep read once, every score written once -- per job 4*D*(NC*S + S) + 4*NC = 24 605 B x 1000."""
    return n_jobs * (4 * D * (NC * S + S) + 4 * NC)


_CPU_WORKER = r"""
import sys, time, torch
sys.path.insert(0, sys.argv[1])
from oracle import aspire_oracle as orc
torch.set_num_threads(1)
g = torch.Generator().manual_seed(int(sys.argv[2]))
q = torch.randn(8, 768, generator=g)
c = torch.randn(64, 8, 768, generator=g)
orc.get_similarity(q, c[0])
print('ready', flush=True)
sys.stdin.readline()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < float(sys.argv[3]):
    orc.get_similarity(q, c[n % 64])
    n += 1
print(n, time.perf_counter() - t0, flush=True)
"""


def cpu_baseline(query, cands, budget_s=30.0):
    """The oracle (a PyTorch CPU port of the reference path) on this box's host cores, same workload (one job's pairs), bounded
    wall time.  Both calling patterns of the reference -- one pair per call (evaluate.py:72-74 via models.py:190-197) and
    caching_score groups of 64 (pp_gen_nearest.py:182-202) -- over a sweep of intra-op thread counts, plus the process-parallel
    form of the per-pair pattern (one single-threaded worker per core, what a user with many cores would run).  `value` is the
    fastest single-process figure."""
    import subprocess
    from oracle import aspire_oracle as orc
    ncpu = os.cpu_count() or 1
    q = query.cpu().view(S, D)
    c = cands.cpu().view(NC, S, D)
    qn = q.numpy()
    sweep = sorted({t for t in (1, 8, 32, ncpu) if t <= ncpu})
    slice_s = budget_s / (2 * len(sweep) + 3)
    pair, batch = {}, {}
    for nt in sweep:
        torch.set_num_threads(nt)
        orc.get_similarity(q, c[0])
        t0 = time.perf_counter()
        n = 0
        while n < NC and time.perf_counter() - t0 < slice_s:
            orc.get_similarity(q, c[n])
            n += 1
        pair[nt] = n / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        nb = 0
        while time.perf_counter() - t0 < slice_s:
            lo = nb % NC
            orc.caching_score(qn, [c[i].numpy() for i in range(lo, min(NC, lo + 64))])
            nb += min(NC, lo + 64) - lo
        batch[nt] = nb / (time.perf_counter() - t0)
    best_pair, best_batch = max(pair, key=pair.get), max(batch, key=batch.get)
    # process-parallel: min(cores, 32) workers (each imports torch: memory bounds the count), one thread each, released together
    n_workers = min(ncpu, 32)
    procs = [subprocess.Popen([sys.executable, '-c', _CPU_WORKER, ROOT, str(w), str(slice_s)], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                              text=True, env=dict(os.environ, OMP_NUM_THREADS='1', MKL_NUM_THREADS='1', HIP_VISIBLE_DEVICES=''))
             for w in range(n_workers)]
    par = None
    try:
        for pr in procs:
            assert pr.stdout.readline().strip() == 'ready'
        for pr in procs:
            pr.stdin.write('go\n')
            pr.stdin.flush()
        res = [pr.stdout.readline().split() for pr in procs]
        par = sum(int(n) / float(dt) for n, dt in res)
    except Exception as e:          # a reported baseline, not the measurement: never fail the bench over it
        par = None
        par_err = repr(e)
    finally:
        for pr in procs:
            pr.kill()
    torch.set_num_threads(ncpu)
    value = max(batch[best_batch], pair[best_pair])
    return {
        'value': value, 'unit': 'alignments/s', 'cores': best_batch if batch[best_batch] >= pair[best_pair] else best_pair, 'kind': 'port',
        'sample': f'caching_score groups of 64 (disent_models.py:256, pp_gen_nearest.py:182) of one step\'s {NC} pairs for {slice_s:.1f} s per '
                  f'thread count, torch.set_num_threads({best_batch}) the fastest; one pair per call in per_pair_by_threads',
        'host_cores': ncpu,
        'batched64_by_threads': {str(t): batch[t] for t in sweep},
        'per_pair_by_threads': {str(t): pair[t] for t in sweep},
        'per_pair_value': pair[best_pair],
        'process_parallel': {'value': par, 'workers': n_workers, 'threads_per_worker': 1,
                             'what': f'{n_workers} processes x one thread, one pair per call (models.py:190-197), {slice_s:.1f} s, released together'}
                            if par is not None else {'value': None, 'error': par_err},
        'note': 'oracle = PyTorch CPU port of the reference path; Sinkhorn = restated geomloss 0.2.4 (parity unpinned)',
    }


def cpu_l2max(device_q_rows, device_c_rows, nq, s, budget_s=4.0):
    """tsAspire on the host cores (config 3's CPU leg): the oracle's allpair_masked_dist_l2max through caching_score's groups of 64
    candidates per query (disent_models.py:294-295) on a bounded sample of the same rows; torch intra-op threads swept over
    1 / 8 / 32 / all cores (a quarter of the budget each), the fastest reported."""
    from oracle import aspire_oracle as orc
    q = device_q_rows[:nq * s].cpu().view(nq, s, D).numpy()
    c = device_c_rows[:64 * 64 * s].cpu().view(64 * 64, s, D)
    groups = [[c[i].numpy() for i in range(g0, g0 + 64)] for g0 in range(0, 64 * 64, 64)]
    orc.caching_score(q[0], groups[0], score_agg_type='l2lse')
    ncpu = os.cpu_count() or 1
    sweep = sorted({t for t in (1, 8, 32, ncpu) if t <= ncpu})
    rates, calls = {}, {}
    for t in sweep:
        torch.set_num_threads(t)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < budget_s / len(sweep):
            orc.caching_score(q[n % nq], groups[(n // nq) % len(groups)], score_agg_type='l2lse')
            n += 1
        rates[t], calls[t] = 64 * n / (time.perf_counter() - t0), n
    best = max(rates, key=rates.get)
    torch.set_num_threads(min(ncpu, 8))
    return {'value': rates[best], 'unit': 'pairs/s', 'cores': best, 'kind': 'port', 'by_threads': {str(t): rates[t] for t in sweep},
            'sample': f'{calls[best]} caching_score(l2lse) calls of 64 candidates each in {budget_s / len(sweep):.1f} s at torch.set_num_threads({best}) '
                      f'(the fastest of {sweep} on {ncpu} host cores)'}


def config3_probe(device, Q=32, C=50000, s=8, reps=30, cpu=True):
    """BASELINE config 3 on one GPU (outside the timed headline): tsAspire, a batch of 32 queries x 50 000 candidates of 8
    sentences, max-sim single match (pair_distances.py:138-186 via disent_models.py:294-295) -- ONE aspire_l2max_scores_f32
    call over a resident store that carries its rows as fp16 planes (aspire_rep_planes, prepared once), timed with HIP events;
    every call prepares the planes of a fresh batch of queries (~10 us) as a serving loop would.  The x.y term of the 102 M
    sentence pairs is a [400 000 x 256] x 768 GEMM: matrix-pipe bound.  Roofline: algorithmic flops 2 Sq Sc D per pair
     against the dense fp16 MFMA peak / 3 -- fp32 accuracy on the fp16 pipe takes three products per term -- and,
    beside it, the executed MFMA rate against what the pipe sustains on random operands with NO data movement (tools/ubench/mfmapeak:
    the chip is power-limited there to ~1.6 PFLOP/s at ~1.65 GHz: profiles/r04_mfmapeak.txt)."""
    from aspire_amd import ops
    g = torch.Generator().manual_seed(1)
    crows = torch.empty(C * s, D, device=device)
    for lo in range(0, C * s, 1 << 16):
        crows[lo:lo + (1 << 16)] = torch.randn(min(1 << 16, C * s - lo), D, generator=g).to(device)
    qrows = torch.randn(Q * s, D, generator=g).to(device)
    mk = lambda rows, n: ops.DeviceRepSet(rows, (torch.arange(n, device=device, dtype=torch.int32) * s).contiguous(),
                                          torch.full((n,), s, device=device, dtype=torch.int32), ext=0, max_len=s, lens_host=[s] * n)
    c, q = mk(crows, C), mk(qrows, Q)
    t0 = time.perf_counter()
    c.prepare_planes()
    torch.cuda.synchronize()
    t_prepare = time.perf_counter() - t0

    def call():
        q.drop_planes()                 # a new batch of queries each call: their planes are part of the call
        return ops.l2max_scores(q, c)
    # warm-up: the first few dozen calls of a process run ~15 % slower than the steady state (the clock governor under a new kind of
    # load: tools/experiments/c3probe.py -- 544 us for the first block of 30, 458 - 482 afterwards, whatever the block does); three
    # timed blocks, the median reported, all three listed
    for _ in range(40):
        sc = call()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    blocks = []
    for _ in range(3):
        a.record()
        for _ in range(reps):
            call()
        b.record()
        torch.cuda.synchronize()
        blocks.append(a.elapsed_time(b) / reps * 1e3)
    us = sorted(blocks)[1]
    # ... and call by call: the kernel runs power-limited and its duration follows the clock governor -- 420 us for the
    # first calls of a cold process, up to ~600 a few calls later, 450 - 490 settled, with periodic excursions (profiles/r06_config3_per_call_trace*):
    # a block average depends on where the block falls; min / median / p90 of 60 single calls say what the spread is
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
    for ea, eb in evs:
        ea.record()
        call()
        eb.record()
    torch.cuda.synchronize()
    per_call = sorted(ea.elapsed_time(eb) * 1e3 for ea, eb in evs)
    ghz = ops.clock_under(call)
    with _pinned(GEMM='bf16x3'):
        for _ in range(2):
            ops.l2max_scores(q, c)
        torch.cuda.synchronize()
        a.record()
        for _ in range(10):
            ops.l2max_scores(q, c)
        b.record()
        torch.cuda.synchronize()
        us_rows = a.elapsed_time(b) / 10 * 1e3
    # spot check against float64 on the first rows
    want = -torch.cdist(qrows[:s].double(), crows[:64 * s].double()).view(s, 64, s).permute(1, 0, 2).reshape(64, -1).min(1).values
    err = float((sc.view(Q, C)[0, :64].double() - want).abs().max())
    flop = 2.0 * s * s * D * Q * C
    nbytes = 4 * D * (C * s + Q * s) + 4 * Q * C
    peak = 2500.0 / 3
    res = {
        'workload': f'tsAspire biomed: {Q} queries x {C} candidates, {s} sents x {D}d, max-sim single match, one call; resident store with fp16 '
                    f'planes (prepared once: {t_prepare * 1e3:.1f} ms), query planes prepared per call; reps ~ N(0,1); {nbytes / 2**20:.0f} MiB > L3',
        'us_per_call': us, 'us_per_call_blocks': blocks, 'pairs_per_s': Q * C / (us * 1e-6), 'clock_ghz_under_kernel': ghz,
        'us_per_single_call': {'min': per_call[0], 'median': per_call[30], 'p90': per_call[54], 'max': per_call[-1], 'n': 60,
                               'what': 'single calls (query planes + the scoring kernel) under their own HIP events, after the blocks'},
        'us_per_call_fp32_row_tiles': us_rows, 'max_abs_err_vs_float64_on_64_pairs': err,
        'roofline': {'bound': 'mfma', 'achieved': flop / (us * 1e-6) / 1e12, 'peak': peak, 'unit': 'TFLOP/s', 'frac': flop / (us * 1e-6) / 1e12 / peak,
                     'kernel': 'pair_gram_p_kernel<128,128,3,true>', 'algorithmic_flop_per_call': flop,
                     'denominator': 'dense fp16 MFMA peak 2500 TFLOP/s / 3 products per term (h.h + h.l + l.h: fp32 accuracy on the fp16 pipe)',
                     'executed_mfma_tflops': 3 * flop / (us * 1e-6) / 1e12,
                     'executed_frac_of_random_operand_ceiling': 3 * flop / (us * 1e-6) / 1e12 / 1600.0,
                     'random_operand_ceiling': '1600 TFLOP/s: v_mfma_f32_32x32x16_f16 from registers, random operands, no memory traffic, 1.65 GHz '
                                               'under power limit (profiles/r04_mfmapeak.txt)',
                     'hbm_view': {'achieved_GBs': nbytes / (us * 1e-6) / 1e9, 'frac': nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                  'algorithmic_bytes_per_call': nbytes}},
    }
    # HBM traffic of the kernel from the committed counter passes (not measured in this run): 2 x FETCH_SIZE + WRITE_SIZE
    tname = next((n for n in ('r06_planes_32x50000x8_fetch_write_size.txt', 'r05_planes_32x50000x8_fetch_write_size.txt',
                              'r04_planes_32x50000x8_fetch_write_size.txt')
                  if os.path.exists(os.path.join(ROOT, 'profiles', n))), None)
    tpath = os.path.join(ROOT, 'profiles', tname or '')
    if tname and (Q, C, s) == (32, 50000, 8):
        kb = {}
        for line in open(tpath):
            f = line.split()
            if len(f) >= 5 and f[0] in ('FETCH_SIZE', 'WRITE_SIZE'):
                kb[f[0]] = float(f[-1])
        if len(kb) == 2:
            res['roofline']['traffic'] = int((2 * kb['FETCH_SIZE'] + kb['WRITE_SIZE']) * 1024)
            res['roofline']['traffic_source'] = f'profiles/{tname} (2 x FETCH_SIZE + WRITE_SIZE, KB per launch)'
    if cpu:
        res['cpu_baseline'] = cpu_l2max(qrows, crows, Q, s)
    return res


def config4_probe(device, J=50, NCAND=125, smax=20, reps=40):
    """BASELINE config 4's shape on one GPU (outside the timed headline): the CSFCube re-rank -- 50 (query, facet) jobs, each query
    (facet-selected rows, 1 .. 8) against its own pool of 125 abstracts of 3 .. 20 sentences (pp_settings.py:2-3, evaluate.py:58-76,
    models.py:127-163), full ranking -- as ONE aspire_ot_rank_batch_f32 call, timed with HIP events on the launch stream; its own
    roofline: algorithmic bytes = 4 D (sum of candidate rows + query rows) + 4 C over the call's duration."""
    from aspire_amd import ops
    g = torch.Generator().manual_seed(4)
    c_lens = torch.randint(3, smax + 1, (J * NCAND,), generator=g)
    q_lens = torch.randint(1, 9, (J,), generator=g)

    def repset(lens):
        start = torch.cumsum(lens, 0) - lens
        rows = torch.randn(int(lens.sum()), D, generator=g).to(device)
        return ops.DeviceRepSet(rows, start.to(torch.int32).to(device), lens.to(torch.int32).to(device), ext=0, max_len=int(lens.max()))

    c, q = repset(c_lens), repset(q_lens)
    job_off = (torch.arange(J + 1, dtype=torch.int32) * NCAND).to(device)
    res = {}
    for name, fn in (('otAspire', ops.ot_rank_batch), ('tsAspire', ops.l2max_rank_batch)):
        out = fn(q, c, job_off, NCAND, NCAND)
        # warm-up BY TIME, as the headline's settle loop: the clock governor takes ~15 - 30 ms of a new kind of load (config3_probe;
        # 40 calls of this one are 3.6 ms: the first block then reads 89 - 93 us where the steady state is 82 - 85); three blocks, the
        # median reported, all three listed
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.08:
            for _ in range(50):
                fn(q, c, job_off, NCAND, NCAND, out=out)
            torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        blocks = []
        for _ in range(3):
            a.record()
            for _ in range(reps):
                fn(q, c, job_off, NCAND, NCAND, out=out)
            b.record()
            torch.cuda.synchronize()
            blocks.append(a.elapsed_time(b) / reps * 1e3)
        us = sorted(blocks)[1]
        nbytes = 4 * D * (int(c_lens.sum()) + int(q_lens.sum())) + 4 * J * NCAND
        gbs = nbytes / (us * 1e-6) / 1e9
        res[name] = {'us_per_call': us, 'us_per_call_blocks': blocks, 'pairs_per_s': J * NCAND / (us * 1e-6),
                     'roofline': {'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
                                  'algorithmic_bytes_per_call': nbytes,
                                  'what': 'the whole call (item sort + scoring launch + rank), back to back on one stream; the data (< 256 MiB) sits in the Infinity Cache: HBM is not what bounds it -- see issue_floor'}}
    # the scoring kernel against its VALU issue floor (the data sits in the Infinity Cache; what bounds the launch is issue slots and
    # the 1.5 rounds of items): SQ_ACTIVE_INST_VALU of the committed counter pass = cycles in which some wave of a SIMD issued a
    # VALU instruction, summed over SIMDs; spread evenly over 1024 SIMDs at 2.4 GHz it is the time below which no schedule of the
    # same instruction stream gets
    for name in ('r06_csf_50x125_ot_sq_counters.txt', 'r05_csf_50x125_ot_sq_counters.txt', 'r04_csf_50x125_ot_sq_counters.txt',
                 'r03_csf_50x125_ot_sq_counters.txt'):
        cpath = os.path.join(ROOT, 'profiles', name)
        if not os.path.exists(cpath):
            continue
        in_kernel, quad = False, None
        for line in open(cpath):
            if 'dispatches' in line:
                in_kernel = 'pair_fused_kernel' in line
            elif in_kernel and 'SQ_ACTIVE_INST_VALU' in line:
                quad = float(line.split()[-1])
        if quad:
            floor_us = quad * 4 / 1024 / 2.4e3
            res['otAspire']['issue_floor'] = {'bound': 'valu-issue', 'floor_us': floor_us, 'frac': floor_us / res['otAspire']['us_per_call'],
                                              'source': f'profiles/{name}: SQ_ACTIVE_INST_VALU {quad:.0f} quad-cycles per launch x 4 / (1024 SIMDs x 2.4 GHz)'}
            break
    res['workload'] = (f'{J} jobs x {NCAND} candidates of 3 .. {smax} sentence rows, facet-selected queries of 1 .. 8 rows, k = {NCAND} '
                       f'(full ranking), reps resident; data {sum(r["roofline"]["algorithmic_bytes_per_call"] for r in res.values()) // 2 / 2**20:.0f} MiB'
                       f' < L3: warm')
    return res


def config4_sharded_probe(device, rank, world, J=50, NCAND=125, smax=20, reps=30):
    """BASELINE config 4 as the N-GPU job runs it (outside the timed headline): every CSFCube query has its OWN pool of ~125 candidates
    (evaluate.py:58-76), so the split is by JOB (aspire_amd/parallel.py: job_bounds -- 50 jobs over 8 ranks: 7 7 6 6 6 6 6 6), a pool is
    never cut: each rank holds only ITS jobs' candidates, makes ONE aspire_ot_rank_batch_f32 call on them, and ONE all-gather of the ranked
    [jobs, 125] lists (score bits + in-pool index, one int64 each) hands every rank the whole result.  The same synthetic jobs as
    config4_probe.  Returns this rank's numbers; rank 0 also ranks all 50 jobs alone and compares."""
    from aspire_amd import ops
    from aspire_amd.parallel import job_bounds, all_gather_ranked_jobs
    g = torch.Generator().manual_seed(4)
    c_lens = torch.randint(3, smax + 1, (J * NCAND,), generator=g)
    q_lens = torch.randint(1, 9, (J,), generator=g)
    c_rows = torch.randn(int(c_lens.sum()), D, generator=g)
    q_rows = torch.randn(int(q_lens.sum()), D, generator=g)
    c_start, q_start = torch.cumsum(c_lens, 0) - c_lens, torch.cumsum(q_lens, 0) - q_lens

    def block(j0, j1):
        """rep sets of jobs [j0, j1): only their rows are uploaded"""
        def cut(rows, start, lens, a, b):
            r0, r1 = (int(start[a]), int(start[b - 1] + lens[b - 1])) if b > a else (0, 0)
            return ops.DeviceRepSet(rows[r0:r1].to(device), (start[a:b] - r0).to(torch.int32).to(device), lens[a:b].to(torch.int32).to(device),
                                    ext=0, max_len=int(lens[a:b].max()) if b > a else 0)
        return (cut(q_rows, q_start, q_lens, j0, j1), cut(c_rows, c_start, c_lens, j0 * NCAND, j1 * NCAND),
                (torch.arange(j1 - j0 + 1, dtype=torch.int32) * NCAND).to(device))

    lo, hi = job_bounds(J, world, rank)
    res = {'rank': rank, 'jobs': [lo, hi]}
    q, c, job_off = block(lo, hi) if hi > lo else (None, None, None)
    out = ops.ot_rank_batch(q, c, job_off, NCAND, NCAND) if hi > lo else None

    def step(exchange=True):
        if hi > lo:
            ops.ot_rank_batch(q, c, job_off, NCAND, NCAND, out=out)
        if exchange:
            return all_gather_ranked_jobs(out[1] if out else None, out[2] if out else None, J, NCAND, device=device)
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.08:          # warm-up by time (config4_probe)
        for _ in range(20):
            step(exchange=False)
        torch.cuda.synchronize()
    if hi > lo:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            step(exchange=False)
        b.record()
        torch.cuda.synchronize()
        res['score_rank_us_per_call'] = a.elapsed_time(b) / reps * 1e3
    ts = []
    for _ in range(12):                                # the whole step: the rank's call + the exchange, every rank starting together
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        top_s, top_i = step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    res['step_us'] = {'median': ts[len(ts) // 2], 'min': ts[0]}
    res['result_digest'] = int((top_i.to(torch.int64) * torch.arange(1, top_i.numel() + 1, device=device).view_as(top_i)).sum().item() % (1 << 31))
    if rank == 0:
        qa, ca, joa = block(0, J)
        _, ws_, wi_ = ops.ot_rank_batch(qa, ca, joa, NCAND, NCAND)
        res['order_equals_one_gpu'] = bool(torch.equal(wi_, top_i))
        res['max_abs_score_diff_vs_one_gpu'] = float((ws_ - top_s).abs().max())
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-probes', action='store_true',
                    help='skip the side measurements that launch the scoring kernel at OTHER sizes / forms (Infinity-Cache-resident job set, '
                         'two-kernel form): for counter passes, whose per-kernel sums should hold the 20-job launches only')
    ap.add_argument('--repeats', type=int, default=0, help='repetitions of the K-step schedule (0 = enough for >= 50 ms)')
    ap.add_argument('--streams', type=int, default=3,
                    help='independent calls in flight: repetitions go round-robin over this many caller streams, each with its own '
                         'outputs and workspace (1 = one call at a time)')
    ap.add_argument('--shard-path', action='store_true',
                    help='run the N > 1 code path (key-form rank, gather, merge kernel) on one GPU, for testing')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    assert args.steps >= 1
    # ASPIRE_BENCH_ONE_GPU=1 (testing only, invalid as a result): every rank on cuda:0 over gloo, so that the N > 1
    # control flow (sharded indices, all-gather layout, merge) can be exercised on a 1-GPU box
    one_gpu_test = os.environ.get('ASPIRE_BENCH_ONE_GPU') == '1'
    if one_gpu_test:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if one_gpu_test:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=device)

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    # ---- N > 1: what the collective backend saw, gathered from every rank (self-verifying first RCCL contact: no 8-GPU node
    # has run this code yet -- DESIGN.md section 5) ---------------------------------------------------------------------------
    rccl = None
    if world > 1:
        dist.barrier()
        props = torch.cuda.get_device_properties(device)
        mine = {'rank': rank, 'local_rank': local_rank, 'device': torch.cuda.get_device_name(device), 'uuid': str(getattr(props, 'uuid', '')),
                'gcn_arch': getattr(props, 'gcnArchName', ''), 'cus': props.multi_processor_count, 'xcds': props.multi_processor_count // 32,
                'hbm_gib': round(props.total_memory / 2**30, 1), 'pid': os.getpid()}
        seen = [None] * world
        dist.all_gather_object(seen, mine)
        rccl = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(), 'ranks_seen': sorted(d['rank'] for d in seen),
                'distinct_devices': len({d['uuid'] or (d['pid'], d['local_rank']) for d in seen}), 'devices': seen}
    from aspire_amd import _lib, ops
    from aspire_amd.parallel import all_gather_flat
    lib = _lib.lib
    shard_path = world > 1 or args.shard_path
    K = args.steps

    # ---- the pool store, resident in HBM: M distinct (query, 1000-candidate pool) jobs; repetition r runs jobs
    # [r*K, r*K + K) mod M.  Rank r of a sharded run owns block r of every job's pool. ----------------------------------
    n_lanes = max(1, args.streams)
    M = max((n_lanes + 1) * K, 24)      # calls in flight never share a pool set
    while M * algorithmic_bytes(1) <= L3_BYTES or M % K:
        M += 1
    g = torch.Generator().manual_seed(1000 * rank)
    queries = torch.randn(M * S, D, generator=torch.Generator().manual_seed(0)).to(device)      # replicated on every rank
    cands = torch.empty(M * NC * S, D, device=device)
    for j in range(M):                                                         # generated pool by pool: bounded host memory
        cands[j * NC * S:(j + 1) * NC * S] = torch.randn(NC * S, D, generator=g).to(device)
    ar = torch.arange(M * NC, device=device, dtype=torch.int32)
    job_off = (torch.arange(K + 1, dtype=torch.int32) * NC).to(device)
    job_base = torch.full((K,), rank * NC, dtype=torch.int32, device=device) if shard_path else None
    prm = _lib.OtParams(0.05, 0.9, 1.0, _lib.CDIST_AUTO)
    null = ctypes.c_void_p(0)

    class JobSet:
        """K consecutive jobs of the store as the two rep sets of one aspire_ot_rank_batch_f32 call."""

        def __init__(self, first, n_jobs=K):
            self.q = ops.DeviceRepSet(queries[first * S:(first + n_jobs) * S], (ar[:n_jobs] * S).contiguous(),
                                      torch.full((n_jobs,), S, device=device, dtype=torch.int32), ext=0, max_len=S)
            self.c = ops.DeviceRepSet(cands[first * NC * S:(first + n_jobs) * NC * S], (ar[:n_jobs * NC] * S).contiguous(),
                                      torch.full((n_jobs * NC,), S, device=device, dtype=torch.int32), ext=0, max_len=S)
            self.qs, self.cs = self.q.struct(), self.c.struct()
            self.n = n_jobs

    sets = [JobSet(f) for f in range(0, M, K)]
    ws_bytes = lib.aspire_ot_rank_batch_workspace_bytes(ctypes.byref(sets[0].qs), ctypes.byref(sets[0].cs), NC, TOPK)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else null
    P_job_off, P_job_base = ptr(job_off), ptr(job_base)

    class Lane:
        """One caller stream with its own outputs and workspace: calls on different lanes are independent requests."""

        def __init__(self):
            self.stream = torch.cuda.Stream(device)
            self.sp = ctypes.c_void_p(self.stream.cuda_stream)
            self.scores = torch.empty(K * NC, device=device, dtype=torch.float32)
            self.top_s = torch.empty(K, TOPK, device=device, dtype=torch.float32)
            self.top_i = torch.empty(K, TOPK, device=device, dtype=torch.int64)
            self.keys = torch.empty(K, TOPK, device=device, dtype=torch.int64) if shard_path else None
            self.gathered = torch.empty(world, K, TOPK, device=device, dtype=torch.int64) if shard_path else None
            self.ws = torch.empty(ws_bytes, device=device, dtype=torch.uint8)
            self.P = [ptr(t) for t in (self.scores, self.top_s, self.top_i, self.keys, self.ws)]

    lanes = [Lane() for _ in range(n_lanes)]
    scores, top_s, top_i, ws = lanes[0].scores, lanes[0].top_s, lanes[0].top_i, lanes[0].ws
    P = [lanes[0].P[0], lanes[0].P[1], lanes[0].P[2], lanes[0].P[3], lanes[0].P[4], P_job_off, P_job_base]

    def stream():
        return lanes[0].sp

    def reduce_max(v):
        """max over ranks of a host scalar (RCCL on the GPUs; gloo, in the one-GPU rehearsal, reduces host tensors)"""
        if world == 1:
            return v
        t = torch.tensor([float(v)], dtype=torch.float64, device='cpu' if one_gpu_test else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item()

    def run_schedule(js, lane=None, exchange=True):
        """K steps = K jobs = ONE C-ABI call on the lane's stream; sharded: + one all-gather of the K x k keys + one merge launch
        (exchange=False: this rank's kernels only -- the stage timings below run on rank 0 alone)."""
        lane = lane or lanes[0]
        rc = lib.aspire_ot_rank_batch_f32(ctypes.byref(js.qs), ctypes.byref(js.cs), D, P_job_off, NC, ctypes.byref(prm), _lib.OT_SIMILARITY,
                                          lane.P[0], TOPK, P_job_base, null if shard_path else lane.P[1], null if shard_path else lane.P[2],
                                          lane.P[3], lane.P[4], ws_bytes, lane.sp)
        if rc:
            _lib.check(rc)
        if shard_path and exchange:
            # on the lane's own stream: torch.distributed orders the collective behind the lane's kernels and the merge behind
            # the collective; collectives of different lanes go out in call order on every rank.  (A separate exchange
            # stream tied to the lanes by events was measured on one GPU, the gather replaced by a copy: 152 instead of 121 us
            # per call -- cross-stream event waits cost more than they hide on this runtime.)
            if world > 1:
                with torch.cuda.stream(lane.stream):
                    all_gather_flat(lane.gathered.view(-1), lane.keys.view(-1))      # -> [world][K][k]; RCCL over xGMI
                src = lane.gathered
            else:
                src = lane.keys
            rc = lib.aspire_topk_merge_keys(ptr(src), world, K, TOPK, TOPK, lane.P[1], lane.P[2], lane.sp)
            if rc:
                _lib.check(rc)

    def run_stage(js, stages):
        rc = lib.aspire_debug_ot_rank_batch_stages_f32(ctypes.byref(js.qs), ctypes.byref(js.cs), D, P[5], NC, ctypes.byref(prm),
                                                       _lib.OT_SIMILARITY, P[0], TOPK, P[1], P[2], P[4], ws.numel(), stream(), stages)
        if rc:
            _lib.check(rc)

    # ---- W untimed warm-up steps, then a calibration pass that sizes R ----------------------------------------------
    for r in range(max(n_lanes, -(-args.warmup // K))):
        run_schedule(sets[r % len(sets)], lanes[r % n_lanes])
    torch.cuda.synchronize()
    if args.repeats > 0:
        R = args.repeats
    else:
        t0 = time.perf_counter()
        n_cal = 4 * n_lanes
        for r in range(n_cal):
            run_schedule(sets[r % len(sets)], lanes[r % n_lanes])
        torch.cuda.synchronize()
        per_call = (time.perf_counter() - t0) / n_cal
        R = max(4, int(1.3 * MIN_TIMED_S / per_call) + 1)
        R = int(reduce_max(R))

    def timed(fn):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()          # every stream of the device
        return reduce_max(time.perf_counter() - t0)

    def all_reps(use=None, reps=None):
        use = use or lanes
        for r in range(reps or R):
            run_schedule(sets[r % len(sets)], use[r % len(use)])

    # untimed: clocks settle on the workload itself (~0.3 s), beyond the W warm-up steps
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < 0.3:
        for r in range(8):
            run_schedule(sets[r % len(sets)], lanes[r % n_lanes], exchange=False)      # time-based loop: no collective inside
        torch.cuda.synchronize()

    elapsed = timed(all_reps)             # EXACTLY K steps x R repetitions, timed once
    # the same schedule with ONE call at a time (lane 0 alone), for the record: not the reported value
    R1 = max(4, R // 2)
    elapsed_one = timed(lambda: all_reps(lanes[:1], R1)) if n_lanes > 1 else elapsed * R1 / R

    # ---- checks on every lane's last outputs ------------------------------------------------------------------------
    torch.cuda.synchronize()
    for ln in lanes:
        assert torch.isfinite(ln.scores).all(), 'non-finite scores'
        assert (ln.top_i >= 0).all() and (ln.top_i < world * NC).all(), 'rank produced out-of-range indices'
        assert (ln.top_s[:, 1:] <= ln.top_s[:, :-1]).all(), 'rank output is not descending'
        if not shard_path or world == 1:
            ref_s, ref_i = torch.sort(ln.scores.view(K, NC), dim=1, descending=True, stable=True)
            assert torch.equal(ln.top_i, ref_i[:, :TOPK]) and torch.equal(ln.top_s, ref_s[:, :TOPK]), 'rank differs from the stable sort'
        else:
            mine = ln.top_i.clone()
            everyone = torch.empty(world, K, TOPK, device=device, dtype=torch.int64)
            all_gather_flat(everyone.view(-1), mine.view(-1))
            agree = all(torch.equal(everyone[0], everyone[r]) for r in range(world))
            if rccl is not None:
                rccl['merged_ranking_agrees_on_all_ranks'] = bool(agree and rccl.get('merged_ranking_agrees_on_all_ranks', True))
                rccl['shards_in_merged_top_k'] = int(len(torch.unique(mine // NC)))
            assert agree, 'ranks disagree on the merged ranking'
            assert len(torch.unique(mine // NC)) > 1, 'merged ranking holds candidates of one shard only'

    if world > 1:
        # the exchange on its own: HIP events on the lane's stream around the all-gather of one call's K x k keys (every rank takes
        # part; rank 0 reports its own view)
        ln = lanes[0]
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        dist.barrier()
        with torch.cuda.stream(ln.stream):
            for a_ev, b_ev in evs:
                a_ev.record(ln.stream)
                all_gather_flat(ln.gathered.view(-1), ln.keys.view(-1))
                b_ev.record(ln.stream)
        torch.cuda.synchronize()
        t = sorted(a_ev.elapsed_time(b_ev) for a_ev, b_ev in evs)
        rccl['all_gather_us'] = {'median': t[len(t) // 2] * 1e3, 'min': t[0] * 1e3, 'bytes_per_rank': K * TOPK * 8,
                                 'what': f'all_gather_into_tensor of {K} x {TOPK} int64 keys per rank on the lane stream, 20 in a row'}
    out = None
    if rank == 0:
        # ---- per-stage kernel durations, live: HIP events on the launch stream around launches of ONE stage alone, on the
        # rotating pools (cold) -- the workspace of a full call holds what the later stages read ------------------------
        def stage_ms(stages, jsets, n=24):
            run_schedule(jsets[0], exchange=False)
            torch.cuda.synchronize()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            for i, (a, b) in enumerate(evs):
                js = jsets[i % len(jsets)]
                if stages != 1:
                    run_stage(js, 1 if stages in (2, 6) else 7)      # this job set's tables and query boxes (+ costs and scores, for the later stages)
                a.record(lanes[0].stream)
                run_stage(js, stages)
                b.record(lanes[0].stream)
            torch.cuda.synchronize()
            t = sorted(a.elapsed_time(b) for a, b in evs)
            return sum(t[:n // 2]) / (n // 2)       # mean of the faster half: launch gaps of a cold queue out of the bracket

        def single_job_probe(n=120):
            """ONE (query, 1000-candidate pool) job per call -- evaluate.py:58-76 for one query: aspire_ot_rank_f32 (scores + top-100),
            rotating pools (cold), each call alone on the stream under its own pair of HIP events."""
            jobs = []
            for j in range(min(M, 48)):
                qj = ops.DeviceRepSet(queries[j * S:(j + 1) * S], ar[:1].contiguous(), torch.full((1,), S, device=device, dtype=torch.int32),
                                      ext=0, max_len=S)
                cj = ops.DeviceRepSet(cands[j * NC * S:(j + 1) * NC * S], (ar[:NC] * S).contiguous(),
                                      torch.full((NC,), S, device=device, dtype=torch.int32), ext=0, max_len=S)
                jobs.append((qj, cj))
            for i in range(24):
                ops.ot_rank(*jobs[i % len(jobs)], TOPK, want=_lib.OT_SIMILARITY)
            torch.cuda.synchronize()
            ts = []
            for i in range(n):
                a_ev, b_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                qj, cj = jobs[i % len(jobs)]
                torch.cuda.synchronize()
                a_ev.record()
                ops.ot_rank(qj, cj, TOPK, want=_lib.OT_SIMILARITY)
                b_ev.record()
                torch.cuda.synchronize()
                ts.append(a_ev.elapsed_time(b_ev) * 1e3)
            ts.sort()
            us = ts[len(ts) // 2]
            return {'what': f'ONE aspire_ot_rank_f32 call of 1 query x {NC} candidates (scores + top-{TOPK}), nothing else in flight, cold '
                            f'rotating pools; median of {n} calls under HIP events (launch latency of its kernels included)',
                    'us_per_call': us, 'us_min': ts[0], 'value': NC / (us * 1e-6), 'unit': 'alignments/s'}

        fused_form = K * ((NC + 3) // 4) >= 512      # the library's own rule (score.hip: ot_rank_batch, kStreamMinGroupsBatch)
        prep_ms = stage_ms(1, sets)
        cost_ms = stage_ms(6, sets)              # the scoring launch (fused form: costs + solves in one kernel)
        rank_ms = stage_ms(8, sets)
        cost_only_ms = solve_only_ms = None
        if not args.no_probes:
            with _lib.pinned(OT_FORM='tile' if fused_form else 'small'):
                cost_only_ms = stage_ms(2, sets)
                solve_only_ms = stage_ms(4, sets)
        run_schedule(sets[0], exchange=False)   # the workspace tables back in the default form's state
        torch.cuda.synchronize()
        # Infinity-Cache-resident variant: ONE small job set (<= 8 jobs, < 200 MB) scored again and again
        n_l3 = min(K, 8)
        l3 = JobSet(0, n_l3)
        job_off_l3 = (torch.arange(n_l3 + 1, dtype=torch.int32) * NC).to(device)

        def cost_l3():
            rc = lib.aspire_debug_ot_rank_batch_stages_f32(ctypes.byref(l3.qs), ctypes.byref(l3.cs), D, ctypes.c_void_p(job_off_l3.data_ptr()),
                                                           NC, ctypes.byref(prm), _lib.OT_SIMILARITY, P[0], 0, null, null, P[4],
                                                           ws.numel(), stream(), 7 if i_l3[0] == 0 else 6)
            i_l3[0] += 1
            if rc:
                _lib.check(rc)
        i_l3 = [0]
        cost_l3_ms = None
        if not args.no_probes:
            for _ in range(4):
                cost_l3()
            torch.cuda.synchronize()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(24)]
            for a, b in evs:
                a.record(lanes[0].stream)
                cost_l3()
                b.record(lanes[0].stream)
            torch.cuda.synchronize()
            t = sorted(a.elapsed_time(b) for a, b in evs)
            cost_l3_ms = sum(t[:12]) / 12

        bytes_per_launch = algorithmic_bytes(K)
        achieved = bytes_per_launch / (cost_ms * 1e-3) / 1e9
        step_achieved = bytes_per_launch * R / elapsed / 1e9
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get('jobs_per_launch') == K:
                traffic = tj.get('cost_kernel_hbm_bytes_per_launch')
        out = {
            'metric': 'query x candidate OT alignments/sec', 'value': world * K * R * NC / elapsed,
            'unit': 'alignments/s', 'n_gpus': world, 'steps': K, 'warmup': args.warmup, 'repeats': R,
            'ms_per_step': elapsed / (K * R) * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'timed_region_ms': elapsed * 1e3,
            'config': {'workload': f'otAspire compsci: 1 query x {NC} candidates per step (per GPU), {S} sents x {D}d, Sinkhorn OT '
                                   f'(blur 0.05, scaling 0.9, one eps schedule per pair) + stable top-{TOPK} rank; every step its own '
                                   f'query and its own pool, {M} distinct pools ({M * algorithmic_bytes(1) / 2**20:.0f} MiB > 256 MiB L3) '
                                   f'walked by consecutive repetitions'
                                   + (f'; sharded: one RCCL all-gather of {K} x {TOPK} keys per call + merge kernel' if shard_path else ''),
                       'queries_per_step': 1, 'candidates_per_gpu_per_step': NC, 'sents': S, 'dim': D, 'topk': TOPK,
                       'parallelism': f'candidate-pool shards x{world}',
                       'launch': f'the {K} steps of a schedule = {K} independent (query, pool) jobs in ONE aspire_ot_rank_batch_f32 call '
                                 f'(eager, no hipGraph); schedule repeated {R}x for a >= 50 ms timed region, the repetitions round-robin on '
                                 f'{n_lanes} caller stream(s) with their own outputs and workspaces = {n_lanes} independent calls in flight '
                                 f'(a call\'s solve tail and rank launch overlap the next call\'s streaming); one call at a time: see one_stream',
                       'streams': n_lanes},
            # the three calling patterns, all at top level: `value` = calls in flight on several caller streams; one call (of K jobs) at a
            # time; ONE job alone (BASELINE config 2 as worded: 1 query x 1000 candidates, one aspire_ot_rank_f32 call, latency bound)
            'one_stream_value': world * K * R1 * NC / elapsed_one,
            'single_job': single_job_probe(),
            'one_stream': {'what': 'the same schedule with ONE call at a time (one caller stream), timed right after the reported region',
                           'value': world * K * R1 * NC / elapsed_one, 'ms_per_call': elapsed_one / R1 * 1e3, 'repeats': R1},
            # Dominant kernel of a step: the cost kernel streams every rep once (the HBM side of the step).
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel': 'pair_fused_kernel (costs + Sinkhorn solves)' if fused_form else 'pair_cost1_kernel + sinkhorn_kernel<1>',
                         'kernel_ms': cost_ms,
                         'algorithmic_bytes_per_launch': bytes_per_launch, 'jobs_per_launch': K, 'data': 'cold (rotating pools, > L3)',
                         'l3_resident_frac': algorithmic_bytes(n_l3) / (cost_l3_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if cost_l3_ms else None,
                         'l3_resident': {'jobs': n_l3, 'kernel_ms': cost_l3_ms, 'bytes': algorithmic_bytes(n_l3)},
                         'stages_ms': {'tables+boxes': prep_ms, 'score': cost_ms, 'rank': rank_ms, 'call_elapsed': elapsed / R * 1e3,
                                       'note': 'score = ONE launch (costs + Sinkhorn solves; a pair whose shifted sums leave fp32 range is re-solved in the max-shifted form by the wave that finds it), rank = the rank launch'},
                         'two_kernel_form': {'cost_ms': cost_only_ms, 'sinkhorn_ms': solve_only_ms,
                                             'cost_frac': bytes_per_launch / (cost_only_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if cost_only_ms else None},
                         'step': {'what': 'all kernels of a schedule (timed region, calls in flight as configured) against the same '
                                          'algorithmic bytes',
                                  'achieved': step_achieved, 'frac': step_achieved / HBM_PEAK_GBS,
                                  'one_stream_frac': bytes_per_launch * R1 / elapsed_one / 1e9 / HBM_PEAK_GBS}},
        }
        spath = os.path.join(ROOT, 'profiles', 'sinkhorn_roofline.json')
        if os.path.exists(spath):
            # the stand-alone Sinkhorn kernel against ITS roof (VALU / transcendental issue), from the committed rocprofv3
            # counter profile of the configs-3/5 shapes -- not measured in this run
            sj = json.load(open(spath))
            out['roofline']['sinkhorn'] = {k: {f: v[f] for f in ('what', 'kernel', 'ns_per_pair', 'valu_wave_instructions_per_pair', 'bound',
                                                                  'achieved_frac', 'issue_floor_us', 'kernel_us_per_call', 'algorithmic_floor') if f in v}
                                           for k, v in sj.items() if isinstance(v, dict)}
            out['roofline']['sinkhorn']['source'] = 'profiles/sinkhorn_roofline.json (committed profile, tools/profile_r5.sh)'
        if rccl is not None:
            out['rccl'] = rccl
        out['cross_check'] = ('per-kernel durations (roofline.kernel_ms, profiles/*_kernel_stats.csv) add up to one_stream.ms_per_call; '
                              '`value` has calls in flight on several streams, where per-kernel durations of overlapping launches mean '
                              'nothing -- verify `one_stream`, treat `value` as the whole-job rate the driver can time from outside')
        if world == 1 and not args.no_probes:
            out['config3'] = config3_probe(device, cpu=not args.no_cpu_baseline)
            out['config4'] = config4_probe(device)
            if not os.environ.get('ASPIRE_BENCH_NO_E2E'):
                sys.path.insert(0, os.path.join(ROOT, 'tools'))
                import e2ebench
                out['e2e'] = e2ebench.run(check=False)
        # the calling patterns and the other configs' figures as PLAIN NUMBERS inside `roofline` and `config` (the driver's record keeps the
        # scalar fields of those two objects; the nested blocks above stay)
        out['roofline']['one_stream_value'] = out['one_stream_value']
        out['roofline']['one_stream_frac'] = out['roofline']['step']['one_stream_frac']
        out['roofline']['single_job_us'] = out['single_job']['us_per_call']
        out['roofline']['calls_in_flight'] = n_lanes
        if 'config3' in out:
            out['config']['config3_us'] = out['config3']['us_per_call']
            out['config']['config3_frac'] = out['config3']['roofline']['frac']
            out['config']['config3_us_min'] = out['config3']['us_per_single_call']['min']
            out['config']['config3_us_p90'] = out['config3']['us_per_single_call']['p90']
        if 'config4' in out:
            out['config']['config4_ot_us'] = out['config4']['otAspire']['us_per_call']
        if 'e2e' in out and 'docs_per_s' in out['e2e']:
            out['config']['e2e_docs_per_s'] = out['e2e']['docs_per_s']
            out['config']['e2e_encoder_frac'] = out['e2e']['encoder_roofline']['frac']
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(queries[:S], cands[:NC * S])
    if world > 1 and not args.no_probes and not os.environ.get('ASPIRE_BENCH_NO_E2E'):
        # config 5's flow on every rank (outside the timed headline; ASPIRE_BENCH_E2E_DOCS=125000 is config 5's per-GPU share of the
        # 1 M-document corpus, the default 8192 a slice of it that keeps the run short).  It holds collectives (a barrier, the centre's
        # broadcast, the keys' all-gather): a rank that fails alone would leave the others waiting, and the headline unprinted -- so the
        # headline is complete BEFORE it starts, and a watchdog prints it and ends the process if the side measurement does not
        # come back in time.
        import threading
        headline = json.dumps(dict(out, e2e={'error': 'the per-rank config-5 side measurement did not finish in time'})) if rank == 0 else None

        def give_up():
            if headline is not None:
                print(headline, flush=True)
            os._exit(0)
        watchdog = threading.Timer(float(os.environ.get('ASPIRE_BENCH_E2E_LIMIT_S', '900')), give_up)
        watchdog.daemon = True
        watchdog.start()
        dist.barrier()                  # (rank 0 took the per-stage timings above while the others waited here)
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import e2ebench
        try:
            c4 = config4_sharded_probe(device, rank, world)
        except Exception as e:
            c4 = {'rank': rank, 'error': repr(e)}
        c4_ranks = [None] * world
        dist.all_gather_object(c4_ranks, c4)
        try:
            mine = e2ebench.run_sharded(rank, world, n_docs=int(os.environ.get('ASPIRE_BENCH_E2E_DOCS', '8192')))
        except Exception as e:
            mine = {'rank': rank, 'error': repr(e)}
        e2e_ranks = [None] * world
        dist.all_gather_object(e2e_ranks, mine)
        watchdog.cancel()
        if rank == 0:
            if any('error' in r for r in c4_ranks):
                out['config4'] = {'errors': [r for r in c4_ranks if 'error' in r]}
            else:
                step_us = max(r['step_us']['median'] for r in c4_ranks)
                out['config4'] = {'what': 'config 4 sharded by JOB (config4_sharded_probe): 50 (query, own pool of 125) jobs in contiguous blocks over the '
                                          'ranks, one aspire_ot_rank_batch_f32 call per rank on its block, ONE all-gather of the ranked lists, no merge; '
                                          'step = call + exchange, every rank starting together, the slowest rank\'s median',
                                  'step_us': step_us, 'pairs_per_s': 50 * 125 / (step_us * 1e-6),
                                  'all_ranks_hold_the_same_result': len({r['result_digest'] for r in c4_ranks}) == 1,
                                  'order_equals_one_gpu': c4_ranks[0].get('order_equals_one_gpu'),
                                  'max_abs_score_diff_vs_one_gpu': c4_ranks[0].get('max_abs_score_diff_vs_one_gpu'), 'ranks': c4_ranks}
                out['config']['config4_sharded_step_us'] = step_us
            if any('error' in r for r in e2e_ranks):
                out['e2e'] = {'errors': [r for r in e2e_ranks if 'error' in r]}
            else:
                out['e2e'] = {'what': 'config 5 per rank (tools/e2ebench.py: run_sharded): each rank encodes its own block into HBM (fp16 planes '
                                      'prepared inside the timed encode stage), ranks 128 replicated queries with otAspire against it, merges the '
                                      'top-100 over one all-gather; weak scaling, the job\'s rates are the sums over ranks; '
                                      'ASPIRE_BENCH_E2E_DOCS=125000 runs config 5\'s full per-GPU share',
                              'docs_per_rank': e2e_ranks[0]['docs'],
                              'docs_per_s': sum(r['docs_per_s'] for r in e2e_ranks),
                              'pairs_per_s': sum(r['pairs_per_s'] for r in e2e_ranks),
                              'merged_top1_agrees': len({r['top1_of_query0'] for r in e2e_ranks}) == 1, 'ranks': e2e_ranks}
                out['config']['e2e_docs_per_s'] = out['e2e']['docs_per_s']
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
