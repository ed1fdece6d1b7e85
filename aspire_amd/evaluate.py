"""The two steps either side of scoring in the reference's evaluation driver:

  score     src/evaluation/evaluate.py:36-82     for every query of a test pool: similarity of every candidate,
                                                  stable descending sort, `scores[-facet].json` =
                                                  {query_id: [[cand_id, -similarity], ...]}
  evaluate  src/evaluation/evaluate.py:85-160 +   scores json + gold grades -> per-query metric rows
            utils/utils.py:59-82                  (`query-evaluations[-facet].csv`) and their means per (facet, split)
                                                  (`aggregated-evaluations[-facet].csv`)

The per-pair Python loop of the reference (one get_similarity call per candidate) becomes one rank_pools call per 32 queries
(each against its own pool) on sentence reps resident in HBM.  File names and layouts are the reference's, so its own `evaluate` step can read
what `score` writes here and vice versa.
"""
import codecs
import csv
import json
import os

import numpy as np

from . import metrics as _metrics


def get_scores_filename(results_dir, facet):           # utils/utils.py:59-61
    return os.path.join(results_dir, 'scores.json' if facet is None else f'scores-{facet}.json')


def get_evaluations_filename(results_dir, facet, aggregated):   # utils/utils.py:66-69
    kind = 'aggregated' if aggregated else 'query'
    return os.path.join(results_dir, f'{kind}-evaluations.csv' if facet is None else f'{kind}-evaluations-{facet}.csv')


def score(results_dir, test_pool, rep_store, facet=None, pred_labels=None, method='ot', schedule='pair', hparams=None,
          queries_per_call=32, resident=True, group=None, sharded=None, deterministic=False):
    """evaluate.py:36-82.  test_pool: {query_id: {'cands': [cand_id, ...]}} (the dataset's test-pid json);
    rep_store: aspire_amd.repstore.RepStore of sentence reps.  A faceted query keeps only the sentence rows whose
    predicted label matches the facet (models.py:127-163; pred_labels: {paper_id: [label per sentence]}).
    Writes and returns {query_id: [(cand_id, -sim), ...]}.

    With the per-pair schedule (the reference's own: one get_similarity call per candidate, evaluate.py:68-72) the queries go
    through scorer.rank_pools `queries_per_call` at a time -- every query against ITS OWN pool in one library call; any other
    schedule / aggregation keeps one rank_pool call per query.  resident: the candidates of all pools are uploaded once
    (RepStore.to_device: a paper in many pools is stored once) and the pools are index lists into that matrix.

    sharded (default: whenever torch.distributed is initialised with more than one rank; `group`: the process group): the step on
    the GPUs of a node, one process per GPU -- BASELINE config 4.  Every query has its OWN pool (evaluate.py:60-62), so the split is
    by JOB: rank r takes the contiguous block parallel.job_bounds(len(queries), world, r) of the queries in the test pool's order,
    keeps only ITS pools' candidates resident, ranks them exactly as the single-process step does, and ONE all-gather of the ranked
    lists (parallel.all_gather_ranked_jobs: score bits + position in the job's candidate list, one int64 each) gives every rank
    the whole result; rank 0 writes the file, every rank returns the same dict.  deterministic: one kernel form whatever a call's
    size (scorer.rank_pools), so that the sharded and the un-sharded step write the same bits for any split."""
    from . import scorer
    import torch.distributed as dist
    if sharded is None:
        sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if sharded:
        return _score_sharded(results_dir, test_pool, rep_store, facet, pred_labels, method, schedule, hparams, queries_per_call,
                              resident, group, deterministic)
    results = {}
    query_ids = list(test_pool.keys())
    all_resident = False
    if resident:
        wanted = set().union(*(pool['cands'] for pool in test_pool.values())) if test_pool else set()
        if wanted and not rep_store.resident(wanted):
            rep_store.to_device([c for pool in test_pool.values() for c in pool['cands']])
        all_resident = bool(wanted)

    def query_reps(query_id):
        return rep_store.faceted(query_id, facet, pred_labels[query_id]) if facet is not None else rep_store.get(query_id)

    if schedule == 'pair' and method in ('ot', 'l2max') and queries_per_call > 1:
        for lo in range(0, len(query_ids), queries_per_call):
            ids = query_ids[lo:lo + queries_per_call]
            cand_lists = [test_pool[i]['cands'] for i in ids]
            if resident and all_resident:
                # the pools' device tables are built once and cached (RepStore.pool_batch): one upload of the queries, one call;
                # evaluate.py:77 stores -similarity
                ranked = scorer.rank_pool_batch([query_reps(i) for i in ids], rep_store.pool_batch(cand_lists), method=method,
                                                hparams=hparams, sign=-1.0, deterministic=deterministic)
                results.update(zip(ids, ranked))
                continue
            else:
                pools = [rep_store.pool(list(cl)) for cl in cand_lists]
                ranked = scorer.rank_pools([query_reps(i) for i in ids], pools, method=method, hparams=hparams,
                                           deterministic=deterministic)
            for query_id, r in zip(ids, ranked):
                results[query_id] = [(cid, -1 * sim) for cid, sim in r]     # evaluate.py:77
    else:
        for query_id in query_ids:
            pool = rep_store.pool(list(test_pool[query_id]['cands']))
            ranked = scorer.rank_pool([query_reps(query_id)], pool, method=method, schedule=schedule, hparams=hparams)[0]
            results[query_id] = [(cid, -1 * sim) for cid, sim in ranked]     # evaluate.py:77
    _write_scores(results_dir, facet, results)
    return results


def _write_scores(results_dir, facet, results):
    os.makedirs(results_dir, exist_ok=True)
    with codecs.open(get_scores_filename(results_dir, facet), 'w', 'utf-8') as fp:
        # the same text as json.dump(results, fp) (evaluate.py:80), encoded in one piece.  (~2 of a score step's ~2.6 ms on the CSFCube
        # shape are this call: 6 250 shortest-round-trip float reprs; a hand-assembled text with one dumps call per query was no faster.)
        fp.write(json.dumps(results))


def _score_sharded(results_dir, test_pool, rep_store, facet, pred_labels, method, schedule, hparams, queries_per_call, resident,
                   group, deterministic):
    """score() over the ranks of `group`: jobs in contiguous blocks, no pool cut, one all-gather (see score's docstring)."""
    import torch
    import torch.distributed as dist
    from . import ops, scorer
    from .parallel import job_bounds, all_gather_ranked_jobs
    if schedule != 'pair' or method not in ('ot', 'l2max'):
        raise NotImplementedError("the sharded score step is built for schedule='pair' with method 'ot' or 'l2max' "
                                  '(the reference\'s own evaluate.py loop)')
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    query_ids = list(test_pool.keys())
    n_jobs = len(query_ids)
    sizes = [len(test_pool[i]['cands']) for i in query_ids]
    k = max(sizes) if sizes else 0
    lo, hi = job_bounds(n_jobs, world, rank)
    mine = query_ids[lo:hi]
    dev = ops.require_gpu()
    local_s = local_i = None
    if mine and k:
        if resident:
            wanted = set().union(*(test_pool[i]['cands'] for i in mine))
            if wanted and not rep_store.resident(wanted):
                rep_store.to_device([c for i in mine for c in test_pool[i]['cands']])          # this rank's pools only
        local_s = torch.full((len(mine), k), float('-inf'), device=dev)
        local_i = torch.full((len(mine), k), -1, dtype=torch.int64, device=dev)
        step = max(1, int(queries_per_call))
        for b0 in range(0, len(mine), step):
            ids = mine[b0:b0 + step]
            qreps = [rep_store.faceted(i, facet, pred_labels[i]) if facet is not None else rep_store.get(i) for i in ids]
            pools = [rep_store.pool(list(test_pool[i]['cands'])) for i in ids]
            _, ts, ti = scorer._launch_rank_pools(qreps, pools, None, hparams, method, deterministic)
            if ts is not None:
                local_s[b0:b0 + len(ids), :ts.shape[1]] = ts
                local_i[b0:b0 + len(ids), :ti.shape[1]] = ti
    top_s, top_i = all_gather_ranked_jobs(local_s, local_i, n_jobs, k, group, device=dev)
    top_s, top_i = top_s.cpu().numpy().astype('float64'), top_i.cpu().numpy()
    results = {}
    for qid, n, rs, ri in zip(query_ids, sizes, top_s, top_i):
        cands = test_pool[qid]['cands']
        results[qid] = [(cands[i], -1 * float(sc)) for sc, i in zip(rs[:n], ri[:n])]           # evaluate.py:77
    if rank == 0:
        _write_scores(results_dir, facet, results)
    dist.barrier(group)                                                                        # the file exists when any rank returns
    return results



def load_score_results(results_dir, gold, facet):
    """utils/utils.py:71-82: relevance of every candidate in the order the model ranked them."""
    with codecs.open(get_scores_filename(results_dir, facet), 'r', 'utf-8') as fp:
        model_scores = json.load(fp)
    return {qid: [gold[qid][pid] for pid, _ in cand_scores] for qid, cand_scores in model_scores.items()}


def evaluate(results_dir, gold, facet=None, threshold_grade=2, split=None):
    """evaluate.py:85-160.  gold: {query_id: {cand_id: grade}}; split: optional {query_id: 'dev' | 'test'}.
    Writes the per-query and the aggregated csv; returns (rows, aggregated rows)."""
    facet_key = 'unfaceted' if facet is None else facet
    rows = []
    for query_id, rels in load_score_results(results_dir, gold, facet).items():
        m = _metrics.compute_metrics(rels, pr_atks=[5, 10, 20], threshold_grade=threshold_grade)
        m['facet'] = facet_key
        m['split'] = 'test' if split is None else split[query_id]
        m['paper_id'] = query_id
        rows.append(m)
    metric_cols = [k for k in rows[0] if k not in ('facet', 'split', 'paper_id')] if rows else []
    with open(get_evaluations_filename(results_dir, facet, aggregated=False), 'w', newline='') as fp:
        w = csv.DictWriter(fp, fieldnames=metric_cols + ['facet', 'split', 'paper_id'])
        w.writeheader()
        w.writerows(rows)
    agg = []
    for f in sorted({r['facet'] for r in rows}):
        for sp in sorted({r['split'] for r in rows}):
            sel = [r for r in rows if r['facet'] == f and r['split'] == sp]
            a = {k: round(float(np.mean([r[k] for r in sel])), 4) for k in metric_cols}     # evaluate.py:141 (.round(4))
            a['facet'], a['split'] = f, sp
            agg.append(a)
    with open(get_evaluations_filename(results_dir, facet, aggregated=True), 'w', newline='') as fp:
        w = csv.DictWriter(fp, fieldnames=metric_cols + ['facet', 'split'])
        w.writeheader()
        w.writerows(agg)
    return rows, agg
