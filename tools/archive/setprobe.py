"""Why do alternate bench calls differ?  Times the scoring launch per job set, back to back, in the bench's own layout (one store
of M pools, set s = pools [20 s, 20 s + 20)), for several M and for the cost phase alone.
  python tools/setprobe.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd import _lib, ops  # noqa: E402

lib = _lib.lib
D, S, NC, K, TOPK = 768, 8, 1000, 20, 100
dev = torch.device('cuda')


def probe(M, separate=False, order=None, pins=None, label=''):
    g = torch.Generator().manual_seed(0)
    queries = torch.randn(M * S, D, generator=torch.Generator().manual_seed(0)).to(dev)
    if separate:
        stores = [torch.empty(K * NC * S, D, device=dev) for _ in range(M // K)]
        for t in stores:
            for j in range(K):
                t[j * NC * S:(j + 1) * NC * S] = torch.randn(NC * S, D, generator=g).to(dev)
    else:
        cands = torch.empty(M * NC * S, D, device=dev)
        for j in range(M):
            cands[j * NC * S:(j + 1) * NC * S] = torch.randn(NC * S, D, generator=g).to(dev)
        stores = [cands[f * NC * S:(f + K) * NC * S] for f in range(0, M, K)]
    ar = torch.arange(K * NC, device=dev, dtype=torch.int32)
    sets = []
    for i, st in enumerate(stores):
        q = ops.DeviceRepSet(queries[i * K * S:(i + 1) * K * S], (ar[:K] * S).contiguous(), torch.full((K,), S, device=dev, dtype=torch.int32), 0, S)
        c = ops.DeviceRepSet(st, (ar * S).contiguous(), torch.full((K * NC,), S, device=dev, dtype=torch.int32), 0, S)
        sets.append((q, c, q.struct(), c.struct()))
    job_off = (torch.arange(K + 1, dtype=torch.int32) * NC).to(dev)
    prm = _lib.OtParams(0.05, 0.9, 1.0, 0)
    scores = torch.empty(K * NC, device=dev)
    ts = torch.empty(K, TOPK, device=dev)
    ti = torch.empty(K, TOPK, device=dev, dtype=torch.int64)
    ws = torch.empty(lib.aspire_ot_rank_batch_workspace_bytes(ctypes.byref(sets[0][2]), ctypes.byref(sets[0][3]), NC, TOPK), device=dev, dtype=torch.uint8)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def stage(i, m):
        _, _, qs, cs = sets[i]
        _lib.check(lib.aspire_debug_ot_rank_batch_stages_f32(ctypes.byref(qs), ctypes.byref(cs), D, p(job_off), NC, ctypes.byref(prm), 2,
                                                             p(scores), TOPK, p(ts), p(ti), p(ws), ws.numel(), stream(), m))
    n = len(sets)
    order = order or list(range(n))
    reps = 40
    with _lib.pinned(**(pins or {})):
        for w in range(2):
            evs = []
            for r in range(reps):
                for i in order:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    stage(i, 1)
                    a.record()
                    stage(i, 6)
                    b.record()
                    stage(i, 8)
                    evs.append((i, a, b))
            torch.cuda.synchronize()
    per = {}
    for i, a, b in evs[len(evs) // 2:]:
        per.setdefault(i, []).append(a.elapsed_time(b) * 1e3)
    print(f'M={M:3d} {"separate allocations" if separate else "one store":20s} order={order} {label:14s}: ' +
          '  '.join(f'set{i} {sum(v) / len(v):6.1f} us' for i, v in sorted(per.items())), flush=True)


probe(40)
probe(40, pins=dict(OT_FORM='fused', FUSED_NOSOLVE=1), label='cost only')
probe(60)
