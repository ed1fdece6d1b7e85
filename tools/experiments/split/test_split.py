"""The role-split kernel's parity test (moved out of tests/ with the kernel in round 6).  Needs the experiment library:
    bash tools/experiments/split/build.sh && ASPIRE_HIP_LIB=build/variants/split/libaspire_hip.so python -m pytest tools/experiments/split/test_split.py -q"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    from aspire_amd import ops, scorer, _lib
    assert torch.cuda.is_available()
    return type('NS', (), dict(ops=ops, scorer=scorer, lib=_lib, pinned=_lib.pinned))


def test_role_split_kernel_gives_the_fused_kernels_bits(amd):
    """split.hip (ASPIRE_HIP_FUSED_SPLIT=1, off by default: NOTES.md round 5): streaming waves hand their items to solver waves through
    an LDS ring -- the same arithmetic on other waves, so the same bits as pair_fused_kernel, for batched jobs (ragged pools and
    documents, more jobs than a workgroup keeps boxes for, a shared sentence) and for one query against a big pool"""
    g = torch.Generator().manual_seed(21)
    sizes = [700, 3, 1200, 0, 450, 64, 1, 900, 333, 12, 800, 90, 5, 610, 77, 300]
    lens_c = torch.randint(1, 9, (sum(sizes),), generator=g)
    lens_q = torch.randint(1, 9, (len(sizes),), generator=g)
    mk = lambda lens: amd.ops.DeviceRepSet.from_list([torch.randn(int(n), 768, generator=g) for n in lens])
    q, c = mk(lens_q), mk(lens_c)
    c.rows[int(c.start[5])] = q.rows[0]                                    # a candidate of job 0 shares the query's first sentence
    job_off = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32).cuda()
    q1 = mk([8])
    res = {}
    for name, pin in (('fused', dict(OT_FORM='fused')), ('split', dict(OT_FORM='fused', FUSED_SPLIT=1))):
        with amd.pinned(**pin):
            sc, ts, ti = amd.ops.ot_rank_batch(q, c, job_off, max(sizes), 10)
            one = amd.ops.ot_sinkhorn(q1, c, want=amd.lib.OT_SIMILARITY)
        res[name] = (sc.cpu(), ts.cpu(), ti.cpu(), one.cpu())
    for a_, b_ in zip(res['fused'], res['split']):
        assert torch.equal(a_, b_)
    assert torch.isfinite(res['split'][0]).all()
