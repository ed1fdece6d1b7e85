"""Drop-in replacements for the reference's sentence-set distance functions, same names, argument
meaning and error behaviour; the arithmetic runs in libaspire_hip.so on the GPU.

Reference: src/learning/facetid_models/pair_distances.py (AllPairMaskedWasserstein :14-92,
allpair_masked_dist_l2max :138-186); copy at examples/ex_aspire_consent_multimatch.py:111-189.

Inputs may live on the CPU (as in the reference's examples) or on the GPU; outputs come back on the
device of ``query.embed``.  There is no CPU code path: without a GPU these raise.
"""
import collections

import torch

from . import _lib, ops

rep_len_tup = collections.namedtuple('RepLen', ['embed', 'abs_lens'])


def _to_repsets(query, cand):
    query_reps, cand_reps = query.embed, cand.embed
    qef_batch_size, _, qmax_sents = query_reps.size()
    cef_batch_size, encoding_dim, cmax_sents = cand_reps.size()
    assert (qef_batch_size == cef_batch_size)   # pair_distances.py:46
    # inputs are batch_size x encoding_dim x max_sents, as in the reference; kernels want rows of 768.
    q = ops.DeviceRepSet.from_padded(query_reps.permute(0, 2, 1), query.abs_lens)
    c = ops.DeviceRepSet.from_padded(cand_reps.permute(0, 2, 1), cand.abs_lens)
    return q, c, query_reps.device


class AllPairMaskedWasserstein:
    def __init__(self, model_hparams):
        self.geoml_blur = model_hparams.get('geoml_blur', 0.05)
        self.geoml_scaling = model_hparams.get('geoml_scaling', 0.9)
        self.geoml_reach = model_hparams.get('geoml_reach', None)
        self.sent_sm_temp = model_hparams.get('sent_sm_temp', 1.0)
        if self.geoml_reach is not None:
            # No reference config sets it (config/models_config/**: geoml_reach absent everywhere).
            raise NotImplementedError('unbalanced OT (geoml_reach) is not built')

    def compute_distance(self, query, cand, return_pair_sims=False):
        """
        :param query: namedtuple(embed: batch_size x encoding_dim x q_max_sents; abs_lens: list(int))
        :param cand: namedtuple(embed: batch_size x encoding_dim x c_max_sents; abs_lens: list(int))
        :return: wasserstein distances [batch_size]; with return_pair_sims the plan-weighted similarity and
            [query_distr, cand_distr, pair_sims, transport_plan, masked_sims] (pair_distances.py:86).
        """
        q, c, out_dev = _to_repsets(query, cand)
        # geomloss derives ONE epsilon schedule from the bounding box of the whole batch, pads included.
        diam = ops.group_diameter(q, c, _lib.PAIR_PAIRED, group=max(q.n, 1))
        kw = dict(pairing=_lib.PAIR_PAIRED, blur=self.geoml_blur, scaling=self.geoml_scaling,
                  sent_sm_temp=self.sent_sm_temp, diameter=diam, diam_group=max(q.n, 1))
        if return_pair_sims:
            sims, (qd, cd, pair_sims, plan) = ops.ot_sinkhorn(q, c, want=_lib.OT_PLAN_SIM, want_extras=True, **kw)
            masked_sims = plan * pair_sims
            return sims.to(out_dev), [t.to(out_dev) for t in (qd, cd, pair_sims, plan, masked_sims)]
        return ops.ot_sinkhorn(q, c, want=_lib.OT_DISTANCE, **kw).to(out_dev)


def allpair_masked_dist_l2max(query, cand, return_pair_sims=False):
    """
    :return: positive distances [batch_size] (the smallest sentence-pair L2), or with return_pair_sims
        (sims [batch_size], pair_sims [batch_size, q_max_sents, c_max_sents]).
    """
    q, c, out_dev = _to_repsets(query, cand)
    if return_pair_sims:
        sims, pair = ops.l2max_scores(q, c, pairing=_lib.PAIR_PAIRED, want_pair_sims=True)
        return sims.to(out_dev), pair.to(out_dev)
    return (-1 * ops.l2max_scores(q, c, pairing=_lib.PAIR_PAIRED)).to(out_dev)


def allpair_masked_dist_l2topk(query, cand, return_pair_sims=False):
    """pair_distances.py:295-345 (score_agg_type 'l2top2').
    :return: positive distances [batch_size] (minus the sum of the two largest -cdist entries), or with
        return_pair_sims (sims [batch_size], pair_sims [batch_size, q_max_sents, c_max_sents])."""
    if query.embed.shape[-1] * cand.embed.shape[-1] < 2:
        # torch.topk(k=2) over the [batch, q_max_sents * c_max_sents] view raises for a single entry (pair_distances.py:333)
        raise RuntimeError('selected index k out of range')
    q, c, out_dev = _to_repsets(query, cand)
    if return_pair_sims:
        sims, pair = ops.l2agg_scores(q, c, _lib.AGG_TOP2, pairing=_lib.PAIR_PAIRED, want_pair_sims=True)
        return sims.to(out_dev), pair.to(out_dev)
    return (-1 * ops.l2agg_scores(q, c, _lib.AGG_TOP2, pairing=_lib.PAIR_PAIRED)).to(out_dev)


class AllPairMaskedAttention:
    """pair_distances.py:95-135 (score_agg_type 'l2attention'): -cdist weighted by its masked 2-D soft-max."""

    def __init__(self, model_hparams):
        self.cdatt_sm_temp = model_hparams.get('cdatt_sm_temp', 1.0)

    def compute_distance(self, query, cand, return_pair_sims=False):
        """:return: doc_dists [batch_size]; with return_pair_sims (doc_sims, [pair_sims, pair_softmax, masked_sims])."""
        q, c, out_dev = _to_repsets(query, cand)
        kw = dict(temp=self.cdatt_sm_temp, pairing=_lib.PAIR_PAIRED)
        if return_pair_sims:
            sims, pair, soft = ops.l2agg_scores(q, c, _lib.AGG_ATTENTION, want_pair_sims=True, **kw)
            return sims.to(out_dev), [t.to(out_dev) for t in (pair, soft, soft * pair)]
        return (-1 * ops.l2agg_scores(q, c, _lib.AGG_ATTENTION, **kw)).to(out_dev)
