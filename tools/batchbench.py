"""Per-stage and end-to-end timings of aspire_ot_rank_batch_f32 for a few (jobs, pool size) shapes and chunk counts.
  python tools/batchbench.py [J,NC,S ...]
Prints one line per configuration: stage durations (HIP events, stage alone on one stream, rotating cold pools), the
call's end-to-end time for OT_FORM = fused, tile, small (back-to-back calls on one stream) and the host time per call."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd import _lib, ops  # noqa: E402

lib = _lib.lib
D = 768


def bench(J, NC, S, k=100):
    dev = torch.device('cuda')
    nsets = max(2, -(-600_000_000 // (J * NC * S * D * 4)))
    nsets = min(nsets, 12)
    g = torch.Generator().manual_seed(0)
    sets = []
    ar = torch.arange(J * NC, device=dev, dtype=torch.int32)
    for _ in range(nsets):
        qrows = torch.randn(J * S, D, generator=g).to(dev)
        crows = torch.randn(J * NC * S, D, device=dev)
        q = ops.DeviceRepSet(qrows, (ar[:J] * S).contiguous(), torch.full((J,), S, device=dev, dtype=torch.int32), 0, S)
        c = ops.DeviceRepSet(crows, (ar * S).contiguous(), torch.full((J * NC,), S, device=dev, dtype=torch.int32), 0, S)
        sets.append((q, c, q.struct(), c.struct()))
    job_off = (torch.arange(J + 1, dtype=torch.int32) * NC).to(dev)
    prm = _lib.OtParams(0.05, 0.9, 1.0, 0)
    scores = torch.empty(J * NC, device=dev)
    ts = torch.empty(J, k, device=dev)
    ti = torch.empty(J, k, device=dev, dtype=torch.int64)
    ws = torch.empty(lib.aspire_ot_rank_batch_workspace_bytes(ctypes.byref(sets[0][2]), ctypes.byref(sets[0][3]), NC, k),
                     device=dev, dtype=torch.uint8)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def full(i):
        _, _, qs, cs = sets[i % nsets]
        _lib.check(lib.aspire_ot_rank_batch_f32(ctypes.byref(qs), ctypes.byref(cs), D, p(job_off), NC, ctypes.byref(prm), 2, p(scores), k,
                                                None, p(ts), p(ti), None, p(ws), ws.numel(), st()))

    def stage(i, m):
        _, _, qs, cs = sets[i % nsets]
        _lib.check(lib.aspire_debug_ot_rank_batch_stages_f32(ctypes.byref(qs), ctypes.byref(cs), D, p(job_off), NC, ctypes.byref(prm), 2,
                                                             p(scores), k, p(ts), p(ti), p(ws), ws.numel(), st(), m))

    def ev_time(m, n=20):
        full(0)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for i, (a, b) in enumerate(evs):
            if m != 1:
                stage(i, 1 if m in (2, 6) else 7)
            a.record()
            stage(i, m)
            b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in evs)
        return 1e3 * sum(t[:n // 2]) / (n // 2)

    out = {'prep': ev_time(1), 'score': ev_time(6), 'rank': ev_time(8)}
    with _lib.pinned(OT_FORM='tile' if S <= 8 else 'small'):
        out['cost(2k)'] = ev_time(2)
        out['solve(2k)'] = ev_time(4)
    bytes_ = J * NC * (S * D * 4) + J * S * D * 4
    out['cost_TBs'] = bytes_ / out['score'] / 1e6
    e2e = {}
    variants = [('fused', dict(OT_FORM='fused')), ('fused-valu', dict(OT_FORM='fused', FUSED_VALU=1)),
                ('fused-nosolve', dict(OT_FORM='fused', FUSED_NOSOLVE=1)), ('fused-notail', dict(OT_FORM='fused', FUSED_NOSOLVE=2))]
    variants += [('tile', dict(OT_FORM='tile')), ('small', dict(OT_FORM='small'))]
    variants += [(f'fused-w{w}', dict(OT_FORM='fused', FUSED_WAVES=w)) for w in os.environ.get('BATCHBENCH_WAVES', '').split(',') if w]
    if S > 8:
        variants = [('small', dict(OT_FORM='small'))]
    for chunks, pins in variants:
        with _lib.pinned(**pins):
            for i in range(4):
                full(i)
            torch.cuda.synchronize()
            n = 40
            t0 = time.perf_counter()
            for i in range(n):
                full(i)
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            e2e[chunks] = (1e6 * (time.perf_counter() - t0) / n, 1e6 * t_host / n)
    # the same call without its first launch (tables + query boxes left over from the last full call: same layout in every
    # set, boxes of another set's queries) -- what folding that launch into the scoring kernel could save at most
    with _lib.pinned(OT_FORM='fused'):
        full(0)
        for i in range(4):
            stage(i, 14)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40):
            stage(i, 14)
        torch.cuda.synchronize()
        e2e['fused-noprep'] = (1e6 * (time.perf_counter() - t0) / 40, 0.0)
        for i in range(4):
            stage(i, 6)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40):
            stage(i, 6)
        torch.cuda.synchronize()
        e2e['fused-alone'] = (1e6 * (time.perf_counter() - t0) / 40, 0.0)
    print(f'J={J} NC={NC} S={S}: ' + ' '.join(f'{k_}={v:.1f}us' for k_, v in out.items() if k_ != 'cost_TBs') +
          f' score={out["cost_TBs"]:.2f}TB/s | e2e(us)/host(us): ' +
          ' '.join(f'{c}={a:.0f}/{h:.0f}' for c, (a, h) in e2e.items()) +
          f' | best {J * NC / min(a for a, _ in e2e.values()):.0f} M pairs/s', flush=True)


if __name__ == '__main__':
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [(20, 1000, 8), (8, 1000, 8), (40, 1000, 8), (100, 1000, 8), (20, 1000, 12), (50, 125, 8)]
    for s in shapes:
        bench(*s)
