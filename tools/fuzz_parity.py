"""Randomised parity sweep across the dispatch thresholds (pool sizes, document lengths, query counts):
GPU otAspire / tsAspire scores of sampled pairs against the oracle, and the rank of every query against the stable
descending sort of its own scores.   python tools/fuzz_parity.py [n_cases] [seed] [planes]
planes: the pool is a resident CandidatePool that carries fp16 planes and cached boxes (the plane tiles of gramp.hip wherever the
library takes them: pools of >= 128 tiles, any number of queries), query counts up to 40."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aspire_amd import ops, scorer, _lib
from oracle import aspire_oracle as orc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
planes = len(sys.argv) > 3 and sys.argv[3] == 'planes'
rng = np.random.default_rng(seed)
sizes = [1, 5, 100, 256, 257, 511, 512, 1000, 1024, 1025, 2047, 2049, 2500, 4096, 4097, 7001, 8192, 10001, 16001]
worst_ot = worst_l2 = 0.0
for case in range(n_cases):
    nq = int(rng.choice([1, 2, 3, 5, 9, 17, 40] if planes else [1, 1, 1, 2, 3, 5]))
    nc = int(rng.choice([2047, 2049, 2500, 4096, 4097, 7001, 8192] if planes else sizes))
    smax = int(rng.choice([8, 8, 8, 12, 16, 20, 32]))
    if not planes and nq * nc * (smax // 8 + (smax % 8 > 0)) ** 2 > 120000:
        nc = max(1, nc // 8)
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    scale = float(rng.choice([1.0, 1.0, 0.3, 2.0]))
    # a third of the cases: every row carries a common vector (anisotropic embeddings, mean cosine 0.5 .. 0.9: the rows are centred
    # by the kernels that can, the others redo most entries with the direct formula)
    common = float(rng.choice([0.0, 0.0, 1.0, 3.0])) * torch.randn(768, generator=g)
    mk = lambda n: scale * (torch.randn(int(n), 768, generator=g) + common)
    ragged = rng.random() < 0.7
    q = [mk(rng.integers(1, smax + 1) if ragged else smax) for _ in range(nq)]
    c = [mk(rng.integers(1, smax + 1) if ragged else smax) for _ in range(nc)]
    if rng.random() < 0.3 and nc > 2:          # a candidate sharing sentences with the query
        c[1] = torch.cat([q[0][:1], c[1]])[:smax]
    pool = scorer.CandidatePool(c).prepare_planes() if planes else c
    ot = scorer.score_pool(q, pool, method='ot', schedule='pair').cpu().numpy()
    l2 = scorer.score_pool(q, pool, method='l2max').cpu().numpy()
    if not (np.isfinite(ot).all() and np.isfinite(l2).all()):
        bad_ot, bad_l2 = np.argwhere(~np.isfinite(ot)), np.argwhere(~np.isfinite(l2))
        raise AssertionError(('non-finite scores', case, nq, nc, smax, scale, ragged, 'ot', bad_ot[:6].tolist(), len(bad_ot), 'l2max', bad_l2[:6].tolist(),
                              len(bad_l2), 'lens', [(len(q[i]), len(c[j])) for i, j in (bad_ot[:6].tolist() + bad_l2[:6].tolist())]))
    for _ in range(12):
        i, j = int(rng.integers(nq)), int(rng.integers(nc))
        if nc > 2 and rng.random() < 0.2:
            j = 1
        shared = j == 1 and len(c[1]) and torch.equal(c[1][0], q[0][0]) and i == 0
        big = 1.0 + float(common.abs().max() > 0) * 2.0          # a common vector: the reference's own fp32 cost has more rounding to lose
        # coincident sentences: the reference's fp32 cost there is sqrt(clamp(|x|^2 - 2 x.y + |y|^2)) of two equal rows = the square root
        # of a few ulps of |x|^2 (geomloss's expansion: up to 5e-2 by rounding luck), in float64 the clamp's floor 1e-4.  The kernels take
        # the exact sum where the expansion cancels (round 5), so such a pair is held to the FLOAT64 oracle, at the same 1e-4
        # (tests/test_gpu_coincident.py pins every kernel family)
        w = orc.get_similarity(q[i].double(), c[j].double()) if shared else orc.get_similarity(q[i], c[j])
        tol = 1e-4 * big
        e = abs(float(ot[i, j]) - w)
        assert e <= tol, (case, nq, nc, smax, i, j, float(ot[i, j]), w, len(q[i]), len(c[j]))
        if not shared:
            worst_ot = max(worst_ot, e)
        wl = -orc.allpair_masked_dist_l2max(orc.RepLen(q[i][None].permute(0, 2, 1), [len(q[i])]),
                                            orc.RepLen(c[j][None].permute(0, 2, 1), [len(c[j])])).item()
        el = abs(float(l2[i, j]) - wl)
        # coincident sentences under torch.cdist's matmul formula (a side beyond 25 rows): the reference's own value is
        # sqrt(clamp(cancellation noise)), 0 or ~3e-2 by rounding luck
        tol_l2 = 5e-2 if (shared and max(len(q[i]), len(c[j])) > 25) else 1e-4 * big
        assert el <= tol_l2, (case, nq, nc, smax, i, j, float(l2[i, j]), wl, len(q[i]), len(c[j]))
        if tol_l2 <= 3e-4:
            worst_l2 = max(worst_l2, el)
    k = int(min(nc, rng.choice([1, 10, 100, 128])))
    qs, cs = ops.DeviceRepSet.from_list(q), (pool.repset if planes else ops.DeviceRepSet.from_list(c))
    sc, ts, ti = ops.ot_rank(qs, cs, k, want=_lib.OT_SIMILARITY)
    sc, ts, ti = sc.cpu(), ts.cpu(), ti.cpu()
    same = np.ones_like(ot, dtype=bool)
    if nc > 2 and len(c[1]) and torch.equal(c[1][0], q[0][0]):
        same[0, 1] = False            # a coincident sentence: cancellation noise of the cost, another value per kernel family (see above)
        assert abs(float(sc[0, 1]) - float(ot[0, 1])) <= 5e-2 * max(1.0, scale)
    np.testing.assert_allclose(sc.numpy()[same], ot[same], atol=2e-4 * big, rtol=0)
    for i in range(nq):
        order = orc.rank_descending(sc[i].tolist())[:k]
        assert ti[i].tolist() == order, (case, nq, nc, smax, k, i)
        assert torch.equal(ts[i], sc[i][order])
    print(f'case {case}: Q={nq} C={nc} S<={smax} ragged={ragged} k={k} common={float(common.abs().max()) > 0} ok', flush=True)
print(f'{n_cases} cases ok; worst |ot - oracle| {worst_ot:.2e}, worst |l2max - oracle| {worst_l2:.2e}')

# ---- part 2: the padded, paired calling pattern of caching_score (disent_models.py:256-342), groups of <= 64 ----------
worst = {'l2max': 0.0, 'neg': 0.0, 'distr': 0.0, 'plan_sim': 0.0, 'l2top2': 0.0}
n2 = max(4, n_cases // 2)
for case in range(n2):
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    smax = int(rng.choice([8, 8, 12, 20, 32]))
    b = int(rng.choice([1, 2, 7, 33, 64]))
    qlen = int(rng.integers(1, smax + 1))
    qrep = torch.randn(qlen, 768, generator=g).numpy()
    creps = [torch.randn(int(rng.integers(1, smax + 1)), 768, generator=g).numpy() for _ in range(b)]
    # a third of the cases (round 6): candidate 0 repeats the query's first sentence -- its outputs are held to the FLOAT64 evaluation (the fp32
    # reference returns the square root of rounding noise for that entry: include/aspire_hip.h, SHARED SENTENCES)
    shared2 = rng.random() < 0.34
    if shared2:
        creps[0] = np.concatenate([qrep[:1], creps[0]])[:smax]
    qd, cds = {'sent_reps': qrep}, [{'sent_reps': r} for r in creps]

    def f64_pair(i):
        """float64 evaluation of candidate i inside the same padded batch: (score, [q_distr, c_distr, neg, ...])"""
        cmax64 = max(len(r) for r in creps)
        pc64 = torch.zeros(b, cmax64, 768, dtype=torch.float64)
        for t, r in enumerate(creps):
            pc64[t, :len(r)] = torch.as_tensor(r, dtype=torch.float64)
        pq64 = torch.as_tensor(qrep, dtype=torch.float64)[None].expand(b, -1, -1).contiguous()
        w, inter = orc.AllPairMaskedWasserstein({}).compute_distance(
            orc.RepLen(pq64.permute(0, 2, 1), [qlen] * b), orc.RepLen(pc64.permute(0, 2, 1), [len(r) for r in creps]), return_pair_sims=True)
        return float(w[i]), [t[i].numpy() for t in inter]
    for agg in ('l2max', 'l2top2', 'l2wasserstein'):
        if agg == 'l2top2' and qlen * max(len(r) for r in creps) < 2:
            continue                       # torch.topk(k=2) over a single entry raises in the reference too (pair_distances.py:308)
        got = scorer.caching_score(qd, cds, score_agg_type=agg)
        if agg == 'l2top2':
            qt = orc.RepLen(torch.stack([torch.nn.functional.pad(torch.as_tensor(qrep), (0, 0, 0, 0))] * b).permute(0, 2, 1), [qlen] * b)
            cmax = max(len(r) for r in creps)
            pc = torch.zeros(b, cmax, 768)
            for i, r in enumerate(creps):
                pc[i, :len(r)] = torch.as_tensor(r)
            ct = orc.RepLen(pc.permute(0, 2, 1), [len(r) for r in creps])
            want_s, _ = orc.allpair_masked_dist_l2topk(qt, ct, return_pair_sims=True)
            sk = 1 if shared2 else 0            # (candidate 0 of a shared case: beyond 25 rows the fp32 reference's -cdist of the equal rows is sqrt(noise))
            e = float(np.abs(got['batch_scores'][sk:] - want_s.numpy()[sk:]).max()) if b > sk else 0.0
            assert e <= 1e-4, (case, agg, b, qlen, e)
            if shared2:
                assert abs(float(got['batch_scores'][0]) - float(want_s[0])) <= 5e-2, (case, agg, b, qlen)
            worst['l2top2'] = max(worst['l2top2'], e)
            continue
        want_s, want_p = orc.caching_score(qrep, creps, agg)
        skip0 = 1 if shared2 else 0            # candidate 0 of a shared case: against float64 below
        if agg == 'l2max':
            e = float(np.abs(got['batch_scores'][skip0:] - want_s[skip0:]).max()) if b > skip0 else 0.0
            assert e <= 1e-4, (case, agg, b, qlen, e)
            worst['l2max'] = max(worst['l2max'], e)
            for gp, wp in list(zip(got['pair_scores'], want_p))[skip0:]:
                assert np.abs(gp - wp).max() <= 1e-4
            if shared2:         # the shared entry is the best match: -cdist = -0 in float64; beyond 25 rows the fp32 reference itself is sqrt(noise) there
                l64 = -float(torch.cdist(torch.as_tensor(qrep).double(), torch.as_tensor(creps[0]).double()).min())
                assert abs(float(got['batch_scores'][0]) - l64) <= 1e-4, ('shared l2max', case, b, qlen, len(creps[0]), float(got['batch_scores'][0]), l64)
        else:
            for gp, wp in list(zip(got['pair_scores'], want_p))[skip0:]:
                worst['distr'] = max(worst['distr'], float(np.abs(gp[0] - wp[0]).max()), float(np.abs(gp[1] - wp[1]).max()))
                worst['neg'] = max(worst['neg'], float(np.abs(gp[2] - wp[2]).max()))
            if shared2:
                w0, p0 = f64_pair(0)
                gp = got['pair_scores'][0]
                nq0, nc0 = qlen, len(creps[0])
                e_d = max(float(np.abs(gp[0][:nq0] - p0[0][:nq0]).max()), float(np.abs(gp[1][:nc0] - p0[1][:nc0]).max()))
                e_n = float(np.abs(gp[2][:nq0, :nc0] - p0[2][:nq0, :nc0]).max())
                assert e_d <= 1e-4 and e_n <= 1e-4, ('shared pair vs float64', case, b, qlen, nc0, e_d, e_n)
                assert abs(float(got['batch_scores'][0]) - w0) <= 3e-3, ('shared plan-sim vs float64', case, b, qlen, nc0, float(got['batch_scores'][0]), w0)
            e = float(np.abs(got['batch_scores'][skip0:] - want_s[skip0:]).max()) if b > skip0 else 0.0
            worst['plan_sim'] = max(worst['plan_sim'], e)
            assert worst['distr'] <= 1e-4 and worst['neg'] <= 1e-4, (case, b, qlen, worst)
            if e > 3e-3:
                # fp32 plan-weighted similarity, exp((f + g - d) / 0.05) with |f|, |g|, |d| ~ 38: the reference's own fp32
                # value is noisy at this level -- judge both against the float64 evaluation of the same batch
                cmax = max(len(r) for r in creps)
                pc = torch.zeros(b, cmax, 768, dtype=torch.float64)
                for i, r in enumerate(creps):
                    pc[i, :len(r)] = torch.as_tensor(r, dtype=torch.float64)
                pq = torch.as_tensor(qrep, dtype=torch.float64)[None].expand(b, -1, -1).contiguous()
                w64, _ = orc.AllPairMaskedWasserstein({}).compute_distance(
                    orc.RepLen(pq.permute(0, 2, 1), [qlen] * b), orc.RepLen(pc.permute(0, 2, 1), [len(r) for r in creps]),
                    return_pair_sims=True)
                e_gpu = float(np.abs(got['batch_scores'] - w64.numpy()).max())
                e_cpu = float(np.abs(want_s - w64.numpy()).max())
                print(f'   plan-sim vs float64: gpu {e_gpu:.2e}, cpu fp32 {e_cpu:.2e}', flush=True)
                assert e_gpu <= max(3 * e_cpu, 3e-3), (case, b, qlen, e_gpu, e_cpu)
    print(f'caching case {case}: B={b} qlen={qlen} S<={smax} ok', flush=True)
print(f'{n2} caching_score cases ok; worst errors {worst}')

# ---- part 3: batched jobs (aspire_ot_rank_batch_f32 / aspire_l2max_rank_batch_f32): every query against its own pool ----------
n3 = max(4, n_cases // 2)
worst_b = {'ot': 0.0, 'l2max': 0.0}
for case in range(n3):
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    smax = int(rng.choice([8, 8, 8, 12, 16, 20, 32]))
    J = int(rng.choice([1, 2, 5, 20, 50, 70]))
    pool_sizes = [int(rng.choice([0, 1, 3, 50, 125, 400, 1000, 1503])) for _ in range(J)]
    budget = 24000 // (smax // 8 + (smax % 8 > 0)) ** 2              # pairs
    while sum(pool_sizes) > budget:
        pool_sizes = [n // 2 for n in pool_sizes]
    few_long = smax > 8 and smax <= 16 and rng.random() < 0.5           # mostly short documents, a few long ones (the hybrid)
    def doc_len():
        if few_long and rng.random() > 0.01:
            return int(rng.integers(1, 9))
        return int(rng.integers(1, smax + 1))
    bscale = float(rng.choice([1.0, 1.0, 2.0, 3.0]))       # large vectors + a shared sentence: the fused kernel's sums overflow, the repair re-solves
    queries = [bscale * torch.randn(int(rng.integers(1, min(smax, 8) + 1)) if rng.random() < 0.5 else doc_len(), 768, generator=g) for _ in range(J)]
    pools = [[bscale * torch.randn(doc_len(), 768, generator=g) for _ in range(n)] for n in pool_sizes]
    shared = set()
    for j, n in enumerate(pool_sizes):
        if n > 0 and rng.random() < 0.4:                    # candidate 0 of this pool shares a sentence with its query
            pools[j][0] = torch.cat([queries[j][:1], pools[j][0]])[:max(1, min(smax, len(pools[j][0])))]
            shared.add(j)
    k = int(rng.choice([1, 10, 100, 2000]))
    for method in ('ot', 'l2max'):
        pls, top_s, top_i = scorer._launch_rank_pools(queries, pools, k, None, method)
        if top_s is None:
            continue
        top_s, top_i = top_s.cpu(), top_i.cpu()
        ranked = scorer.rank_pools(queries, pools, k=k, method=method)
        for j, n in enumerate(pool_sizes):
            assert len(ranked[j]) == min(k, n), ('length', case, method, j, len(ranked[j]), k, n)
            if n == 0:
                continue
            one = scorer.score_pool([queries[j]], pools[j], method=method, schedule='pair')[0].cpu()
            got = dict(ranked[j])
            # the list = stable descending order of the job's own scores (scores of a batch and of a one-pool call may differ
            # in the last bits -- other kernels --, so the order is checked against the listed scores, the values against both)
            vals = [s for _, s in ranked[j]]
            assert all(vals[t] >= vals[t + 1] for t in range(len(vals) - 1)), ('order', case, method, j, vals[:6])
            assert all(np.isfinite(v) for v in vals), ('non-finite', case, method, j, [(i, v) for i, v in ranked[j] if not np.isfinite(v)][:5], len(queries[j]), [len(pools[j][i]) for i, v in ranked[j] if not np.isfinite(v)][:5], smax, bscale, pool_sizes[j])
            # a shared sentence (round 6: held to the FLOAT64 oracle like part 1 -- every kernel form takes a cancelling entry from the exact sum, at
            # any document length; only tsAspire beyond 25 rows keeps a 5e-2 bar: the fp32 reference's own value there is sqrt(noise))
            is_shared = lambda i: j in shared and i == 0
            loose = lambda i: is_shared(i) and method == 'l2max' and max(len(queries[j]), len(pools[j][0])) > 25
            for i, s in ranked[j]:
                assert abs(s - float(one[i])) <= (5e-2 * bscale if loose(i) else 2e-4 * bscale), (case, method, j, i, s, float(one[i]))
            if len(vals) < n:                   # nothing outside the list beats its last entry
                rest = [float(one[i]) for i in range(n) if i not in got]
                assert max(rest) <= vals[-1] + (5e-2 * bscale if (j in shared and method == 'l2max') else 2e-4 * bscale), ('rest', case, method, j, max(rest), vals[-1])
            for i in ([0] if j in shared else []) + [int(rng.integers(n)) for _ in range(2)]:      # sampled pairs against the oracle
                if i not in got:
                    continue
                qd, cd = (queries[j].double(), pools[j][i].double()) if is_shared(i) else (queries[j], pools[j][i])
                if method == 'ot':
                    try:
                        w = orc.get_similarity(qd, cd)
                    except ValueError:          # a one-sentence candidate equal to its one-sentence query: diameter 0, geomloss's
                        continue                # schedule (arange from log 0) raises -- the GPU side gives the entry's cost (kMinDiameter)
                else:
                    w = -torch.cdist(qd, cd).min().item() if is_shared(i) else -orc.allpair_masked_dist_l2max(
                        orc.RepLen(qd[None].permute(0, 2, 1), [len(qd)]), orc.RepLen(cd[None].permute(0, 2, 1), [len(cd)])).item()
                e = abs(got[i] - w)
                assert e <= (5e-2 * bscale if loose(i) else 1e-4 * bscale), (case, method, j, i, got[i], w, bscale, len(queries[j]), len(pools[j][i]), is_shared(i))
                if not is_shared(i):
                    worst_b[method] = max(worst_b[method], e / bscale)
    print(f'batched case {case}: J={J} pools={pool_sizes[:8]}{"..." if J > 8 else ""} S<={smax} few_long={few_long} k={k} scale={bscale} shared={len(shared)} ok', flush=True)
print(f'{n3} batched cases ok; worst errors {worst_b}')
