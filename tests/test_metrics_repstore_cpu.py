"""CPU: ranking metrics against values produced by the reference's own metrics.py (tests/golden/metrics.json),
and the rep-store loaders / facet row-select."""
import json
import os

import numpy as np
import pytest

from aspire_amd import metrics
from aspire_amd.repstore import RepStore


@pytest.fixture(scope='module')
def kat(golden_dir):
    return json.load(open(os.path.join(golden_dir, 'metrics.json')))


def test_ap_map_doctest_values(kat):
    assert metrics.average_precision(kat['ap_in']) == pytest.approx(0.78333333333333333, abs=1e-15)
    assert metrics.mean_average_precision(kat['map_in']) == pytest.approx(0.39166666666666666, abs=1e-15)
    assert metrics.average_precision(kat['ap_in2']) == pytest.approx(kat['ap_out2'], abs=1e-15)
    assert metrics.average_precision([0, 0, 0]) == 0.0


def test_dcg_ndcg_mrr_rprecision(kat):
    for c in kat['ndcg']:
        assert metrics.dcg_at_k(c['r'], c['k'], c['method']) == pytest.approx(c['dcg'], abs=1e-12)
        assert metrics.ndcg_at_k(c['r'], c['k'], c['method']) == pytest.approx(c['ndcg'], abs=1e-12)
    assert metrics.mean_reciprocal_rank(kat['mrr']['in']) == pytest.approx(kat['mrr']['out'], abs=1e-15)
    for c in kat['r_precision']:
        assert metrics.r_precision(c['in']) == pytest.approx(c['out'], abs=1e-15)
    with pytest.raises(ValueError):
        metrics.precision_at_k([0, 0, 1], 4)
    with pytest.raises(ValueError):
        metrics.dcg_at_k([1, 2], 2, method=2)


def test_compute_metrics_matches_reference(kat):
    for c in kat['compute_metrics']:
        got = metrics.compute_metrics(c['graded'], c['pr_atks'], c['threshold'])
        assert set(got) == set(c['out'])
        for k, v in c['out'].items():
            assert got[k] == pytest.approx(v, abs=1e-12), k


def test_evaluate_ranked_pool():
    ranked = {'q1': [('a', 0.9), ('b', 0.5), ('c', 0.1)], 'q2': [('x', 3.0), ('y', 1.0)]}
    gold = {'q1': {'a': 3, 'b': 0, 'c': 2}, 'q2': {'x': 0, 'y': 1}}
    per_q, agg = metrics.evaluate_ranked_pool(ranked, gold, pr_atks=(1, 2))
    assert per_q['q1']['av_precision'] == pytest.approx((1.0 + 2 / 3) / 2)
    assert per_q['q2']['av_precision'] == 0.0
    assert agg['map'] == pytest.approx(((1.0 + 2 / 3) / 2) / 2)


def test_repstore_roundtrip_and_facets(tmp_path):
    rng = np.random.RandomState(0)
    d = {f'p{i}': {'sent_reps': rng.randn(3 + i, 768).astype(np.float32), 'doc_cls_reps': rng.randn(768)} for i in range(4)}
    import joblib
    jp = tmp_path / 'reps.joblib'
    joblib.dump(d, jp, compress=('gzip', 3))           # the reference's dump format, pp_gen_nearest.py:129
    rs = RepStore.from_joblib(jp)
    assert len(rs) == 4 and 'p2' in rs and rs.get('p2').shape == (5, 768)
    npz = tmp_path / 'reps.npz'
    rs.save_npz(npz)
    rs2 = RepStore.from_npz(npz)
    for pid in d:
        assert np.array_equal(rs2.get(pid), d[pid]['sent_reps'])
    labels = ['background_label', 'objective_label', 'method_label', 'result_label', 'method_label']
    assert rs.faceted('p2', 'method', labels).shape == (2, 768)
    assert np.array_equal(rs.faceted('p2', 'background', labels), d['p2']['sent_reps'][[0, 1]])
    assert rs.faceted('p2', 'all', labels).shape == (5, 768)
    with pytest.raises(ImportError):
        RepStore.from_h5(tmp_path / 'missing.h5')


def test_evaluate_step_reads_scores_json(tmp_path):
    """evaluate.py:85-160 on a scores.json in the reference's layout: per-query rows + aggregated means, csv files
    under the reference's names (no GPU: the score step is covered by the -m gpu suite)."""
    import json
    from aspire_amd import evaluate as ev
    from aspire_amd import metrics as mt
    rng = np.random.RandomState(0)
    cands = [f'c{i}' for i in range(25)]           # the reference's precision@20 needs pools of >= 20 (metrics.py:141)
    gold = {q: {c: int(g) for c, g in zip(cands, rng.randint(0, 4, size=25))} for q in ('q1', 'q2')}
    ranked = {q: [[c, -float(i)] for i, c in enumerate(rng.permutation(cands))] for q in ('q1', 'q2')}
    with open(ev.get_scores_filename(str(tmp_path), None), 'w') as f:
        json.dump(ranked, f)
    rows, agg = ev.evaluate(str(tmp_path), gold, facet=None, threshold_grade=2, split={'q1': 'test', 'q2': 'dev'})
    assert [r['paper_id'] for r in rows] == ['q1', 'q2']
    for row, q in zip(rows, ('q1', 'q2')):
        rels = [gold[q][c] for c, _ in ranked[q]]
        assert row['av_precision'] == pytest.approx(mt.average_precision([1 if r >= 2 else 0 for r in rels]))
        assert row['ndcg'] == pytest.approx(mt.ndcg_at_k(rels, len(rels)))
    assert {(a['facet'], a['split']) for a in agg} == {('unfaceted', 'test'), ('unfaceted', 'dev')}
    assert os.path.exists(os.path.join(tmp_path, 'query-evaluations.csv'))
    assert os.path.exists(os.path.join(tmp_path, 'aggregated-evaluations.csv'))
    assert ev.get_scores_filename('r', 'background') == os.path.join('r', 'scores-background.json')
    assert ev.get_evaluations_filename('r', 'method', True) == os.path.join('r', 'aggregated-evaluations-method.csv')


def test_from_h5_reads_the_reference_layout(tmp_path):
    """the reference's encodings.h5: one dataset per paper id holding [num_sents, 768] (src/evaluation/utils/models.py:68-95,
    file name from utils/utils.py:63-64).  h5py is not in the build image: the test runs wherever it is installed."""
    h5py = pytest.importorskip('h5py')
    from aspire_amd.repstore import RepStore
    rs = np.random.RandomState(0)
    want = {f'pid{i}': rs.randn(n, 768).astype(np.float32) for i, n in enumerate((3, 1, 12))}
    path = str(tmp_path / 'encodings.h5')
    with h5py.File(path, 'w') as f:
        for pid, reps in want.items():
            f.create_dataset(pid, data=reps)
    store = RepStore.from_h5(path)
    assert sorted(store.pid2reps) == sorted(want)
    for pid, reps in want.items():
        assert store.get(pid).dtype == np.float32 and np.array_equal(store.get(pid), reps)
