"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE'S OWN CODE.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
The fixtures are data (inputs + expected outputs); no reference source is copied.

What executes from the reference:
  examples/ex_aspire_consent.py   AspireConSent.forward / consent_reps_bert (pooling, :36-101),
                                  prepare_abstracts / prepare_bert_sentences (:107-212)
  src/learning/facetid_models/pair_distances.py
                                  allpair_masked_dist_l2max (:138-186),
                                  AllPairMaskedWasserstein.compute_distance (:21-92)
  src/evaluation/utils/metrics.py average_precision / mean_average_precision (:98-143)

``geomloss`` (third party, pinned 0.2.4 in requirements.txt:1) is absent from this container and
cannot be installed.  pair_distances.py imports it at module scope, so a stand-in module is
registered whose ``SamplesLoss`` forwards to the oracle's restatement.  Consequently, in ot_*.npz:
  * query_distr, cand_distr, pair_sims (= masked neg L2) come purely from reference code  -> PINNED
  * plan / masked_sims / wasserstein_dists combine reference wrapper code (:73-86) with the
    oracle's solver restatement -> wrapper pinned, SOLVER UNPINNED (see oracle/aspire_oracle.py).
"""
import collections
import importlib
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')
sys.path.insert(0, '/root/reference/examples')

from oracle import aspire_oracle as orc  # noqa: E402

# ---- geomloss: the real package when it is importable (tools/pin_geomloss.sh installs 0.2.4 into a venv and re-runs this
# script: that PINS the solver), else a stand-in whose solver is the oracle's restatement (records nothing else) ----------
try:
    import geomloss as _real_geomloss  # noqa: F401
    SOLVER = 'geomloss-' + getattr(_real_geomloss, '__version__', 'unknown')
except ImportError:
    _real_geomloss = None
    SOLVER = 'oracle-restatement (parity unpinned)'
if os.environ.get('ASPIRE_REQUIRE_GEOMLOSS') and _real_geomloss is None:
    raise SystemExit('ASPIRE_REQUIRE_GEOMLOSS is set but geomloss is not importable')
print('Sinkhorn solver behind the reference wrapper:', SOLVER)
_geomloss = types.ModuleType('geomloss')


class _SamplesLoss:
    def __init__(self, loss, p, blur, reach, scaling, debias, potentials):
        assert loss == 'sinkhorn' and p == 1 and reach is None and debias is False
        self.blur, self.scaling, self.potentials = blur, scaling, potentials

    def __call__(self, a, x, b, y):
        return orc.geomloss_sinkhorn_tensorized(a, x, b, y, p=1, blur=self.blur, scaling=self.scaling,
                                                potentials=self.potentials)


_geomloss.SamplesLoss = _SamplesLoss
if _real_geomloss is None:
    sys.modules['geomloss'] = _geomloss

import ex_aspire_consent as ref_ex  # noqa: E402
ref_pd = importlib.import_module('src.learning.facetid_models.pair_distances')
ref_metrics = importlib.import_module('src.evaluation.utils.metrics')

RepLen = collections.namedtuple('RepLen', ['embed', 'abs_lens'])


def make_pool():
    """A2/A3: hidden states + ragged token index lists -> (cls, sent_reps) via the reference module."""
    class _Out:
        pass

    class _FakeBert(torch.nn.Module):
        def __init__(self, hidden):
            super().__init__()
            self.hidden = hidden

        def forward(self, tokid_tt, token_type_ids=None, attention_mask=None):
            o = _Out()
            o.last_hidden_state = self.hidden
            return o

    def run(hidden, idxs, seq_lens):
        model = ref_ex.AspireConSent.__new__(ref_ex.AspireConSent)
        torch.nn.Module.__init__(model)
        model.bert_encoding_dim = 768
        model.bert_encoder = _FakeBert(hidden)
        b, l, _ = hidden.shape
        batch = {'tokid_tt': torch.zeros(b, l, dtype=torch.long), 'seg_tt': torch.zeros(b, l, dtype=torch.long),
                 'attnmask_tt': torch.ones(b, l, dtype=torch.long), 'seq_lens': seq_lens}
        abs_lens = [len(x) for x in idxs]
        cls, sent = model.forward(batch, abs_lens, idxs)
        return cls.numpy(), sent.numpy()

    g = torch.Generator().manual_seed(1234)
    out = {}
    # case A: B=3 ragged, doc 1 has 1 sentence, doc 2's last sentence truncated to one token,
    #         title tokens (1..4) and SEP excluded from every span.
    hidden = torch.randn(3, 48, 768, generator=g)
    idxs = [
        [list(range(5, 12)), list(range(12, 30)), list(range(30, 31)), list(range(31, 47))],
        [list(range(3, 20))],
        [list(range(6, 9)), list(range(9, 46)), [46]],
    ]
    cls, sent = run(hidden, idxs, [48, 21, 48])
    out.update(a_hidden=hidden.numpy(), a_idxs=json.dumps(idxs), a_cls=cls, a_sent=sent)
    # case B: B=1 (exercises the squeeze/unsqueeze branch of forward, :46-49)
    hidden = torch.randn(1, 17, 768, generator=g)
    idxs = [[list(range(4, 9)), list(range(9, 16))]]
    cls, sent = run(hidden, idxs, [17])
    out.update(b_hidden=hidden.numpy(), b_idxs=json.dumps(idxs), b_cls=cls, b_sent=sent)
    np.savez_compressed(os.path.join(HERE, 'pool.npz'), **out)
    print('pool.npz', {k: getattr(v, 'shape', None) for k, v in out.items()})


VOCAB = ['[PAD]', '[UNK]', '[CLS]', '[SEP]', '[MASK]', '.', ',', 'the', 'of', 'and', 'we', 'a', 'in', 'to',
         'model', 'paper', 'graph', 'neural', 'network', 'learn', '##ing', '##s', 'optimal', 'transport',
         'sentence', 'document', 'similar', '##ity', 'propose', 'method', 'result', 'show', 'that', 'is',
         'data', 'set', 'train', '##ed', 'on', 'with', 'for', 'text', 'align', '##ment', 'score', 'rank',
         'query', 'candidate', 'abstract', 'title', 'science', 'bio', '##medical', 'compute', '##r', 'x', 'y', 'z']


def make_tokenizer(tmpdir):
    from transformers import BertTokenizer
    p = os.path.join(tmpdir, 'vocab.txt')
    with open(p, 'w') as f:
        f.write('\n'.join(VOCAB) + '\n')
    tok = BertTokenizer(p, do_lower_case=True)
    if hasattr(tok, 'build_inputs_with_special_tokens'):
        return tok

    class _Tok451:
        """transformers>=5 dropped BertTokenizer.build_inputs_with_special_tokens, which the reference
        (pinned to transformers 4.5.1) calls at ex_aspire_consent.py:161.  For one sequence 4.5.1's
        BertTokenizer returns [CLS] + ids + [SEP]; this adapter supplies exactly that."""
        def __init__(self, t):
            self._t = t

        def __getattr__(self, k):
            return getattr(self._t, k)

        def build_inputs_with_special_tokens(self, token_ids_0):
            return [self._t.cls_token_id] + token_ids_0 + [self._t.sep_token_id]
    return _Tok451(tok)


def make_prep():
    """A0: prepare_abstracts through the reference with a BertTokenizer over a tiny local vocab."""
    rng = np.random.RandomState(7)
    words = [w for w in VOCAB[5:] if not w.startswith('##')]

    def sent(n):
        return ' '.join(rng.choice(words, size=n)) + ' .'

    docs = [
        {'TITLE': 'optimal transport for document similarity', 'ABSTRACT': [sent(6), sent(9), sent(4)]},
        {'TITLE': 'graph neural networks', 'ABSTRACT': [sent(3)]},
        # crosses 500 word pieces in the middle of a sentence (partial last sentence kept)
        {'TITLE': 'a model', 'ABSTRACT': [sent(100), sent(120), sent(130), sent(90), sent(120), sent(50)]},
        # overflow sentence begins exactly at the 500 cap -> reduced_len == 0 -> dropped, not appended
        {'TITLE': 'x y', 'ABSTRACT': [sent(247), sent(248), sent(30), sent(5)]},
        {'TITLE': 'learning to rank', 'ABSTRACT': [sent(12), sent(1), sent(20), sent(7), sent(9)]},
        # partial last sentence of exactly one word piece (index 500)
        {'TITLE': 'x y', 'ABSTRACT': [sent(247), sent(247), sent(30), sent(5)]},
    ]
    with tempfile.TemporaryDirectory() as td:
        tok = make_tokenizer(td)
        # doc 3: title(2) + [SEP](1) + 248 + 249 = 500 word pieces -> the next sentence has reduced_len == 0
        cases = []
        for group in ([0, 1], [2], [3], [5], [0, 1, 2, 3, 4, 5]):
            batch = [docs[i] for i in group]
            bert_batch, abs_lens, sent_token_idxs = ref_ex.prepare_abstracts(batch, tok)
            cases.append({
                'doc_ids': group,
                'tokid_tt': bert_batch['tokid_tt'].tolist(), 'seg_tt': bert_batch['seg_tt'].tolist(),
                'attnmask_tt': bert_batch['attnmask_tt'].tolist(), 'seq_lens': bert_batch['seq_lens'],
                'abs_lens': abs_lens, 'sent_token_idxs': sent_token_idxs,
            })
    with open(os.path.join(HERE, 'prep.json'), 'w') as f:
        json.dump({'vocab': VOCAB, 'docs': docs, 'cases': cases}, f)
    for c in cases:
        print('prep case', c['doc_ids'], 'seq_lens', c['seq_lens'], 'abs_lens', c['abs_lens'],
              'last span', [(s[-1][0], s[-1][-1]) for s in c['sent_token_idxs']])


def _ragged_batch(g, b, qmax, cmax, qlens, clens, d=768, dup=None):
    q = torch.randn(b, qmax, d, generator=g)
    c = torch.randn(b, cmax, d, generator=g)
    for i in range(b):
        q[i, qlens[i]:] = 0.0
        c[i, clens[i]:] = 0.0
    if dup is not None:  # candidate sentence identical to a query sentence
        bi, qi, ci = dup
        c[bi, ci] = q[bi, qi]
    return q, c


def make_scores():
    g = torch.Generator().manual_seed(99)
    out = {}
    cases = {
        # name: (B, qmax, cmax, qlens, clens)
        's8': (6, 8, 8, [8] * 6, [8, 8, 5, 3, 8, 1]),
        'rag': (5, 7, 12, [7, 3, 1, 7, 5], [12, 6, 12, 1, 9]),
        'one': (1, 7, 6, [7], [6]),                       # the notebook's 7x6 pair shape
        'big': (2, 28, 30, [28, 26], [30, 27]),           # > 25 rows: cdist switches to the mm formula
    }
    for name, (b, qmax, cmax, qlens, clens) in cases.items():
        q, c = _ragged_batch(g, b, qmax, cmax, qlens, clens)
        qt = RepLen(embed=q.permute(0, 2, 1), abs_lens=qlens)
        ct = RepLen(embed=c.permute(0, 2, 1), abs_lens=clens)
        dists = ref_pd.allpair_masked_dist_l2max(qt, ct)
        sims, pair = ref_pd.allpair_masked_dist_l2max(qt, ct, return_pair_sims=True)
        out.update({f'{name}_q': q.numpy(), f'{name}_c': c.numpy(),
                    f'{name}_qlens': np.array(qlens), f'{name}_clens': np.array(clens),
                    f'{name}_l2max_dist': dists.numpy(), f'{name}_l2max_sims': sims.numpy(),
                    f'{name}_l2max_pair': pair.numpy()})
        for temp in (1.0, 5000.0):
            ot = ref_pd.AllPairMaskedWasserstein({'sent_sm_temp': temp})
            wd = ot.compute_distance(qt, ct)
            ws, (qd, cd, ps, plan, ms) = ot.compute_distance(qt, ct, return_pair_sims=True)
            t = 't1' if temp == 1.0 else 't5000'
            out.update({f'{name}_{t}_qdistr': qd.numpy(), f'{name}_{t}_cdistr': cd.numpy(),
                        f'{name}_{t}_pairsims': ps.numpy(), f'{name}_{t}_plan': plan.numpy(),
                        f'{name}_{t}_maskedsims': ms.numpy(), f'{name}_{t}_wsims': ws.numpy(),
                        f'{name}_{t}_wdist': wd.numpy()})
        print(name, 'l2max', dists.numpy()[:3], 'wdist', out[f'{name}_t1_wdist'][:3], 'wsims', out[f'{name}_t1_wsims'][:3])
    out['solver'] = np.array(SOLVER)      # which Sinkhorn solver sat behind the reference wrapper (tests/test_oracle_cpu.py reads it)
    np.savez_compressed(os.path.join(HERE, 'scores.npz'), **out)


def make_siblings():
    """Sibling aggregations of the same masked -cdist block: top-2 sum
    (pair_distances.py:295-345) and the masked 2-D soft-max attention (pair_distances.py:95-135 with
    models_common/activations.py:35-61).  Both run from the reference itself; no third-party arithmetic."""
    g = torch.Generator().manual_seed(123)
    out = {}
    cases = {
        's8': (6, 8, 8, [8] * 6, [8, 8, 5, 3, 8, 2]),
        'rag': (5, 7, 12, [7, 3, 1, 7, 5], [12, 6, 12, 2, 9]),
        'big': (2, 28, 30, [28, 26], [30, 27]),
        'dup': (3, 8, 8, [8, 6, 8], [8, 8, 4]),
    }
    for name, (b, qmax, cmax, qlens, clens) in cases.items():
        q, c = _ragged_batch(g, b, qmax, cmax, qlens, clens, dup=(1, 2, 3) if name == 'dup' else None)
        qt = RepLen(embed=q.permute(0, 2, 1), abs_lens=qlens)
        ct = RepLen(embed=c.permute(0, 2, 1), abs_lens=clens)
        out.update({f'{name}_q': q.numpy(), f'{name}_c': c.numpy(),
                    f'{name}_qlens': np.array(qlens), f'{name}_clens': np.array(clens)})
        out[f'{name}_top2_dist'] = ref_pd.allpair_masked_dist_l2topk(qt, ct).numpy()
        sims, pair = ref_pd.allpair_masked_dist_l2topk(qt, ct, return_pair_sims=True)
        out[f'{name}_top2_sims'], out[f'{name}_top2_pair'] = sims.numpy(), pair.numpy()
        for temp in (1.0, 0.2):
            att = ref_pd.AllPairMaskedAttention({'cdatt_sm_temp': temp})
            t = 't1' if temp == 1.0 else 't02'
            out[f'{name}_att_{t}_dist'] = att.compute_distance(qt, ct).numpy()
            ds, (ps, sm, ms) = att.compute_distance(qt, ct, return_pair_sims=True)
            out.update({f'{name}_att_{t}_sims': ds.numpy(), f'{name}_att_{t}_pair': ps.numpy(),
                        f'{name}_att_{t}_softmax': sm.numpy(), f'{name}_att_{t}_masked': ms.numpy()})
        print(name, 'top2', out[f'{name}_top2_dist'][:3], 'att', out[f'{name}_att_t1_dist'][:3])
    np.savez_compressed(os.path.join(HERE, 'siblings.npz'), **out)


def make_metrics():
    # NumPy 2 removed np.asfarray, which the reference's dcg_at_k calls (metrics.py:179); supply the NumPy 1
    # behaviour so that compute_metrics can run here at all.  Environment shim only -- no arithmetic of ours.
    if not hasattr(np, 'asfarray'):
        np.asfarray = lambda a, dtype=float: np.asarray(a, dtype=dtype)
    r = [1, 1, 0, 1, 0, 1, 0, 0, 0, 1]
    kat = {
        'ap_in': r, 'ap_out': float(ref_metrics.average_precision(r)),
        'map_in': [r, [0]], 'map_out': float(ref_metrics.mean_average_precision([r, [0]])),
        'ap_in2': [0, 0, 1, 0, 1, 1, 0], 'ap_out2': float(ref_metrics.average_precision([0, 0, 1, 0, 1, 1, 0])),
    }
    rng = np.random.RandomState(11)
    cm = []
    for n in (30, 125, 60):
        graded = rng.choice([0, 0, 0, 1, 2, 3], size=n).tolist()
        atks = [5, 10, 20]
        cm.append({'graded': graded, 'pr_atks': atks, 'threshold': 2,
                   'out': ref_metrics.compute_metrics(graded, atks, threshold_grade=2)})
    kat['compute_metrics'] = cm
    kat['ndcg'] = [{'r': [3, 2, 3, 0, 0, 1, 2, 2, 3, 0], 'k': k, 'method': m,
                    'dcg': float(ref_metrics.dcg_at_k([3, 2, 3, 0, 0, 1, 2, 2, 3, 0], k, m)),
                    'ndcg': float(ref_metrics.ndcg_at_k([3, 2, 3, 0, 0, 1, 2, 2, 3, 0], k, m))}
                   for k in (1, 2, 10, 11) for m in (0, 1)]
    kat['mrr'] = {'in': [[0, 0, 1], [0, 1, 0], [1, 0, 0], [0, 0, 0]],
                  'out': float(ref_metrics.mean_reciprocal_rank([[0, 0, 1], [0, 1, 0], [1, 0, 0], [0, 0, 0]]))}
    kat['r_precision'] = [{'in': r_, 'out': float(ref_metrics.r_precision(r_))} for r_ in ([0, 0, 1], [0, 1, 0], [1, 0, 0], [0, 0])]
    with open(os.path.join(HERE, 'metrics.json'), 'w') as f:
        json.dump(kat, f)
    print(kat)


if __name__ == '__main__':
    torch.set_num_threads(4)
    which = sys.argv[1:] or ['pool', 'prep', 'scores', 'siblings', 'metrics']
    for name in which:
        {'pool': make_pool, 'prep': make_prep, 'scores': make_scores, 'siblings': make_siblings,
         'metrics': make_metrics}[name]()
