"""ONE (query, 1000-candidate pool) job per call (evaluate.py:58-76 for one query): aspire_ot_rank_f32 under HIP events, the kernel forms
the library can take for it pinned one after the other.  usage: singlejob.py [NC S]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aspire_amd import ops, _lib

NC, S = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (1000, 8)
dev = torch.device('cuda')
g = torch.Generator().manual_seed(0)
jobs = []
for j in range(48):
    q = ops.DeviceRepSet.from_list([torch.randn(S, 768, generator=g)])
    c = ops.DeviceRepSet.from_list([torch.randn(S, 768, generator=g) for _ in range(NC)])
    jobs.append((q, c))


def run(label, **pin):
    with _lib.pinned(**pin):
        for i in range(24):
            ops.ot_rank(*jobs[i % len(jobs)], 100, want=_lib.OT_SIMILARITY)
        torch.cuda.synchronize()
        ts = []
        for i in range(150):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            sc, _, _ = ops.ot_rank(*jobs[i % len(jobs)], 100, want=_lib.OT_SIMILARITY)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
    print(f'{label:24s} median {ts[len(ts) // 2]:6.1f} us  min {ts[0]:6.1f}  p90 {ts[int(len(ts) * 0.9)]:6.1f}', flush=True)
    return sc.clone()


base = run('default')
for label, pin in (('small', dict(OT_FORM='small')), ('one', dict(OT_FORM='one')), ('tile', dict(OT_FORM='tile')), ('fused', dict(OT_FORM='fused')),
                   ('default again', {})):
    sc = run(label, **pin)
    print(f'   max |score - default| {float((sc - base).abs().max()):.2e}')
