"""Ranking metrics for the evaluate step that follows scoring.

Same definitions and return values as the reference's src/evaluation/utils/metrics.py (a copy of
bwhite's rank_metrics gist): mean_reciprocal_rank :7-31, r_precision :34-57, precision_at_k :60-95,
average_precision :98-121, mean_average_precision :124-143, dcg_at_k :146-184, ndcg_at_k :187-220,
recall_at_k :222-239, compute_metrics :241-281.  The reference's dcg_at_k calls np.asfarray, which NumPy 2
removed, so its compute_metrics cannot run on this image; this module uses np.asarray(dtype=float).
Host-side numpy: the inputs are a few hundred relevance grades per query.
"""
import numpy as np


def mean_reciprocal_rank(rs):
    """Mean over queries of 1 / rank of the first relevant (non-zero) item; 0 when there is none."""
    firsts = [np.flatnonzero(np.asarray(r)) for r in rs]
    return float(np.mean([1.0 / (nz[0] + 1) if nz.size else 0.0 for nz in firsts]))


def r_precision(r):
    """Precision at the rank of the last relevant item."""
    rel = np.asarray(r) != 0
    nz = np.flatnonzero(rel)
    if not nz.size:
        return 0.0
    return float(np.mean(rel[:nz[-1] + 1]))


def precision_at_k(r, k):
    assert k >= 1
    rel = np.asarray(r)[:k] != 0
    if rel.size != k:
        raise ValueError('Relevance score length < k')
    return float(np.mean(rel))


def average_precision(r):
    """Mean of precision@rank over the ranks holding a relevant item (area under the PR curve)."""
    rel = np.asarray(r) != 0
    hits = np.flatnonzero(rel)
    if not hits.size:
        return 0.0
    return float(np.mean([np.mean(rel[:h + 1]) for h in hits]))


def mean_average_precision(rs):
    return float(np.mean([average_precision(r) for r in rs]))


def dcg_at_k(r, k, method=1):
    """method 0: weights 1, 1, 1/log2(3), ...; method 1: 1/log2(2), 1/log2(3), ..."""
    g = np.asarray(r, dtype=float)[:k]
    if not g.size:
        return 0.0
    if method == 0:
        return float(g[0] + np.sum(g[1:] / np.log2(np.arange(2, g.size + 1))))
    if method == 1:
        return float(np.sum(g / np.log2(np.arange(2, g.size + 2))))
    raise ValueError('method must be 0 or 1.')


def ndcg_at_k(r, k, method=0):
    best = dcg_at_k(sorted(r, reverse=True), k, method)
    if not best:
        return 0.0
    return dcg_at_k(r, k, method) / best


def recall_at_k(ranked_rel, atk, max_total_relevant):
    total = min(max_total_relevant, sum(ranked_rel))
    return float(sum(ranked_rel[:atk])) / total if total > 0 else 0.0


def compute_metrics(ranked_judgements, pr_atks, threshold_grade):
    """Graded relevances in rank order -> dict of the reference's metric names (metrics.py:241-281).
    CSFCube binarises at grade >= 2 (src/evaluation/utils/datasets.py:125)."""
    graded = list(ranked_judgements)
    binary = [1 if rel >= threshold_grade else 0 for rel in graded]
    out = {}
    for atk in [5, 10, 15, 20, 25]:
        out[f'ndcg%{atk}'] = float(ndcg_at_k(graded, int((atk / 100) * len(binary))))
    total_relevant = sum(binary)
    for atk in pr_atks:
        recall = recall_at_k(binary, atk, total_relevant)
        precision = precision_at_k(binary, atk)
        out[f'precision@{atk}'] = float(precision)
        out[f'recall@{atk}'] = float(recall)
        out[f'f1@{atk}'] = float(2 * precision * recall / (precision + recall)) if precision + recall > 0 else 0.0
    out['r_precision'] = r_precision(binary)
    out['av_precision'] = average_precision(binary)
    out['reciprocal_rank'] = mean_reciprocal_rank([binary])
    out['ndcg'] = float(ndcg_at_k(graded, len(binary)))
    out['ndcg@20'] = float(ndcg_at_k(graded, 20))
    out['ndcg@50'] = float(ndcg_at_k(graded, 50))
    return out


def evaluate_ranked_pool(ranked, gold, pr_atks=(5, 10, 20), threshold_grade=2):
    """ranked: {query_id: [(cand_id, score), ...]} as written by the scoring step (evaluate.py:76-82);
    gold: {query_id: {cand_id: grade}}.  Returns (per-query metrics, aggregate means) like
    load_score_results + evaluate (src/evaluation/utils/utils.py:66-82, evaluate.py:85-160)."""
    per_query = {}
    for qid, cands in ranked.items():
        rels = [gold[qid][cid] for cid, _ in cands]
        per_query[qid] = compute_metrics(rels, [k for k in pr_atks if k <= len(rels)], threshold_grade)
    keys = sorted({k for m in per_query.values() for k in m})
    agg = {k: float(np.mean([m[k] for m in per_query.values() if k in m])) for k in keys}
    agg['map'] = agg.get('av_precision', 0.0)
    return per_query, agg
