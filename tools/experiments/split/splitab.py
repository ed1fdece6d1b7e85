"""A/B of the role-split otAspire kernel (split.hip) against the fused kernel (fused.hip) at the bench's shape: J jobs x N candidates
of S rows, scores + top-k in one aspire_ot_rank_batch_f32 call, rotating pools (cold), HIP-event timings; bit comparison of the
scores of the two forms.  usage: splitab.py [J N S npools reps]"""
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from aspire_amd import ops, _lib  # noqa: E402

J, N, S, NP, REPS = (int(x) for x in (sys.argv[1:6] + [20, 1000, 8, 6, 30][len(sys.argv) - 1:]))
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
D = 768


def repset(n, s, ragged=False):
    lens = torch.randint(1, s + 1, (n,), generator=g) if ragged else torch.full((n,), s)
    start = torch.cumsum(lens, 0) - lens
    rows = torch.randn(int(lens.sum()), D, generator=g).to(dev)
    return ops.DeviceRepSet(rows, start.to(torch.int32).to(dev), lens.to(torch.int32).to(dev), ext=0, max_len=int(lens.max()))


pools = [repset(J * N, S) for _ in range(NP)]
q = repset(J, S)
job_off = (torch.arange(J + 1, dtype=torch.int32) * N).to(dev)
outs = [ops.ot_rank_batch(q, c, job_off, N, 100) for c in pools]
import ctypes
ws = torch.empty(max(_lib.lib.aspire_ot_rank_batch_workspace_bytes(ctypes.byref(q.struct()), ctypes.byref(pools[0].struct()), N, 100), 16), device=dev, dtype=torch.uint8)


def timed(pin, label):
    with _lib.pinned(**pin):
        for i in range(3 * NP):
            ops.ot_rank_batch(q, pools[i % NP], job_off, N, 100, out=outs[i % NP], workspace=ws)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(3):
            a.record()
            for i in range(REPS):
                ops.ot_rank_batch(q, pools[i % NP], job_off, N, 100, out=outs[i % NP], workspace=ws)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / REPS * 1e3)
        sc = [o[0].clone() for o in outs]
    print(f'{label:28s} us per call: ' + ' '.join(f'{t:7.1f}' for t in ts) + f'   {J * N / min(ts):.1f} M alignments/s', flush=True)
    return sc


variants = [('fused (FUSED_SPLIT=2)', dict(FUSED_SPLIT=2)), ('split', dict(FUSED_SPLIT=1, SPLIT_PRIO=0)),
            ('split, solves skipped', dict(FUSED_SPLIT=1, SPLIT_PRIO=3)),
            ('fused again', dict(FUSED_SPLIT=2)), ('split again', dict(FUSED_SPLIT=1, SPLIT_PRIO=0)),
            ('split, solves skipped again', dict(FUSED_SPLIT=1, SPLIT_PRIO=3))]
import os
if os.environ.get('SPLITAB_ONLY') == 'fused':
    variants = [('fused (FUSED_SPLIT=2)', dict(FUSED_SPLIT=2)), ('fused again', dict(FUSED_SPLIT=2))]
res = {}
for label, pin in variants:
    res[label] = timed(pin, label)
base = res['fused (FUSED_SPLIT=2)']
for label in res:
    d = max(float((x - y).abs().max()) for x, y in zip(res[label], base))
    same = all(torch.equal(x, y) for x, y in zip(res[label], base))
    print(f'{label:28s} max |score - fused| {d:.3e}  bit-identical {same}')
