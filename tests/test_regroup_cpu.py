"""Host logic of AspireConSent.encode_to_pool's sort_by_length: documents of prepare_abstracts batches regrouped by token length
(pp_gen_nearest.py:141-160 encodes 32 at a time in corpus order; the regrouping changes which documents share a forward, never
a document's tokens, spans or its place in the store)."""
import numpy as np
import torch

from aspire_amd.consent import AspireConSent


def _batches(rng, n_batches):
    out = []
    for k in range(n_batches):
        b = int(rng.integers(1, 6))
        ls = rng.integers(5, 40, b)
        L = int(ls.max())
        tok = torch.zeros(b, L, dtype=torch.long)
        mask = torch.zeros(b, L, dtype=torch.long)
        for i, n in enumerate(ls):
            tok[i, :n] = torch.arange(1, n + 1) + 100 * k + 10 * i
            mask[i, :n] = 1
        spans = [[[1, 2], [3 + i]] for i in range(b)]
        out.append(({'tokid_tt': tok, 'seg_tt': torch.zeros_like(tok), 'attnmask_tt': mask, 'seq_lens': [int(n) for n in ls]}, [2] * b, spans))
    return out


def test_regroup_by_length_keeps_every_document_and_its_corpus_position():
    rng = np.random.default_rng(0)
    batches = _batches(rng, 9)
    flat = [(bb['tokid_tt'][i, :bb['seq_lens'][i]], sp[i]) for bb, _, sp in batches for i in range(len(bb['seq_lens']))]
    for window in (8192, 7, 3):
        groups, ids = AspireConSent._regroup_by_length(batches, 4, window)
        seen = []
        for (bb, abs_lens, spans), g in zip(groups, ids):
            assert len(g) <= 4 and len(abs_lens) == len(spans) == len(g) == bb['tokid_tt'].shape[0]
            assert bb['tokid_tt'].shape[1] == max(bb['seq_lens'])                      # padded to the group's own longest sequence
            assert bb['seq_lens'] == sorted(bb['seq_lens'], reverse=True)              # longest first inside a window
            for j, d in enumerate(g):
                n = bb['seq_lens'][j]
                assert torch.equal(bb['tokid_tt'][j, :n], flat[d][0]) and spans[j] == flat[d][1]
                assert int(bb['attnmask_tt'][j].sum()) == n and not bb['tokid_tt'][j, n:].any() and not bb['attnmask_tt'][j, n:].any()
                seen.append(d)
        assert sorted(seen) == list(range(len(flat))), window


def test_regroup_by_length_leaves_equal_lengths_in_corpus_order():
    tok = torch.arange(40).view(10, 4)
    batches = [({'tokid_tt': tok[lo:lo + 5], 'seg_tt': torch.zeros(5, 4, dtype=torch.long), 'attnmask_tt': torch.ones(5, 4, dtype=torch.long),
                 'seq_lens': [4] * 5}, [1] * 5, [[[1]]] * 5) for lo in (0, 5)]
    groups, ids = AspireConSent._regroup_by_length(batches, 4)
    assert ids == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]
    assert torch.equal(torch.cat([bb['tokid_tt'] for bb, _, _ in groups]), tok)


def test_regroup_by_token_rows_sizes_a_forward_by_its_longest_document():
    """rows_per_forward: a forward takes as many documents as fit that many token rows at its longest document's length
    (encode_to_pool's default, 16 384 rows: 64 documents of 256 tokens, 128 of 128); every document kept, corpus positions intact."""
    rng = np.random.default_rng(1)
    batches = _batches(rng, 12)
    n_docs = sum(len(a) for _, a, _ in batches)
    for budget in (100, 64, 39, 8):
        groups, ids = AspireConSent._regroup_by_length(batches, 4, rows_per_forward=budget)
        assert sorted(i for g in ids for i in g) == list(range(n_docs))
        for (bb, _, _), g in zip(groups, ids):
            b, L = bb['tokid_tt'].shape
            assert b == len(g) and L == max(bb['seq_lens']) == bb['seq_lens'][0]
            assert b == max(1, budget // L) or g is ids[-1]          # full groups but the last
            assert b * L <= budget or b == 1


def test_rows_key_under_inference_mode():
    """a store built under torch.inference_mode() tracks no torch version (reading `_version` raises): the cache key falls back
    to the library's own generation counter, which every C-ABI write bumps."""
    import torch
    from aspire_amd import ops
    with torch.inference_mode():
        rows = torch.zeros(16, 768)
        assert rows.is_inference()
        k0 = ops._rows_key(rows)
        ops._rows_written(rows)
        k1 = ops._rows_key(rows)
    assert k0 != k1 and k0[:2] == k1[:2] and k0[2] is None
    plain = torch.zeros(16, 768)
    a = ops._rows_key(plain)
    plain.add_(1.0)                      # a torch write: seen through _version
    b = ops._rows_key(plain)
    ops._rows_written(plain)             # a library write: both counters move
    c = ops._rows_key(plain)
    assert a != b and b != c and c[3] == 1
