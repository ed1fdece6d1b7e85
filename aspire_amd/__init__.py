"""aspire_amd -- MI355X-native implementation of Aspire's query-vs-candidate scoring path.

The arithmetic lives in libaspire_hip.so (hand-written HIP for gfx950, C ABI in include/aspire_hip.h);
this package is the host-side mirror of the reference's Python call surface
(examples/ex_aspire_consent.py, examples/ex_aspire_consent_multimatch.py).
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library has not been built)
from .batch_prep import prepare_abstracts, prepare_bert_sentences  # noqa: F401
from .pair_distances import (AllPairMaskedWasserstein, AllPairMaskedAttention, allpair_masked_dist_l2max,  # noqa: F401
                             allpair_masked_dist_l2topk, rep_len_tup)
from .consent import AspireConSent  # noqa: F401
