"""CPU: the torch.library registration of SURVEY.md section 8(b) -- every op exists in the `aspire` namespace, its fake
(meta) implementation propagates shapes, and there is NO CPU kernel behind it (the dispatcher raises)."""
import pytest
import torch


def _m(*s, dt=torch.float32):
    return torch.empty(*s, device='meta', dtype=dt)


def test_ops_are_registered_with_fake_impls():
    import aspire_amd.torch_ops as to
    for name in to.OPS:
        assert hasattr(torch.ops.aspire, name), name
    i32 = torch.int32
    assert torch.ops.aspire.l2max_scores(_m(3, 8, 768), _m(3, dt=i32), _m(5, 6, 768), _m(5, dt=i32), False).shape == (15,)
    assert torch.ops.aspire.l2max_scores(_m(4, 8, 768), _m(4, dt=i32), _m(4, 6, 768), _m(4, dt=i32), True).shape == (4,)
    out = torch.ops.aspire.ot_sinkhorn_scores(_m(4, 8, 768), _m(4, dt=i32), _m(4, 6, 768), _m(4, dt=i32), 0.05, 0.9, 1.0, 4, 1, True, True)
    assert [tuple(t.shape) for t in out] == [(4,), (4, 8), (4, 6), (4, 8, 6), (4, 8, 6)]
    out = torch.ops.aspire.ot_sinkhorn_scores(_m(2, 8, 768), _m(2, dt=i32), _m(7, 6, 768), _m(7, dt=i32), 0.05, 0.9, 1.0, 0, 0, False, False)
    assert out[0].shape == (14,) and out[3].numel() == 0
    cls, sent = torch.ops.aspire.span_mean_pool(_m(2, 20, 768), _m(50, dt=i32), _m(7, dt=i32), 3)
    assert cls.shape == (2, 768) and sent.shape == (2, 3, 768)
    ts, ti = torch.ops.aspire.topk_desc(_m(3, 5000), 5000, 0)
    assert ts.shape == (3, 5000) and ti.dtype == torch.int64
    assert torch.ops.aspire.topk_keys(_m(3, 900), 100, 64).shape == (3, 100)
    ts, ti = torch.ops.aspire.topk_merge(_m(8, 3, 100, dt=torch.int64), 100)
    assert ts.shape == (3, 100) and ts.dtype == torch.float32 and ti.dtype == torch.int64
    s, ts, ti = torch.ops.aspire.ot_rank_batch(_m(16, 768), _m(2, dt=i32), _m(2, dt=i32), 8, _m(800, 768), _m(100, dt=i32), _m(100, dt=i32),
                                                8, _m(3, dt=i32), 60, 10, 0.05, 0.9, 1.0, 2)
    assert s.shape == (100,) and ts.shape == (2, 10) and ti.shape == (2, 10)
    w = [_m(100, 768), _m(512, 768), _m(2, 768), _m(768), _m(768)]
    assert torch.ops.aspire.bert_encoder_forward(_m(2, 9, dt=torch.int64), _m(2, 9, dt=torch.int64), _m(2, 9, dt=torch.int64), w, 12,
                                                 1e-12).shape == (2, 9, 768)
    with pytest.raises(AssertionError):      # pair_distances.py:46
        torch.ops.aspire.l2max_scores(_m(3, 8, 768), _m(3, dt=i32), _m(5, 6, 768), _m(5, dt=i32), True)


def test_no_cpu_kernel_behind_the_ops():
    import aspire_amd.torch_ops  # noqa: F401
    z = torch.zeros(1, 2, 768)
    n = torch.ones(1, dtype=torch.int32)
    with pytest.raises(NotImplementedError, match='CPU'):
        torch.ops.aspire.l2max_scores(z, n, z, n, False)
    with pytest.raises(NotImplementedError, match='CPU'):
        torch.ops.aspire.topk_desc(torch.zeros(1, 4), 2, 0)
