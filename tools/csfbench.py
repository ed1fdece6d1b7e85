"""BASELINE config 4's shape on one GPU: the CSFCube re-rank -- 50 (query, facet) jobs, every query against its own pool of
~125 candidates, ragged abstracts of 3 .. 20 sentences, facet-selected query rows (1 .. 8), full ranking (k = pool size).
Times the batched call (aspire_ot_rank_batch_f32 / aspire_l2max_rank_batch_f32, reps resident) against the per-query loop
(aspire_ot_rank_f32 per query) and prints the launches' share.   python tools/csfbench.py [J NC SMAX]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import _lib, ops

J, NC, SMAX = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (50, 125, 20)
QMAX = int(sys.argv[4]) if len(sys.argv) > 4 else 8          # query rows 1 .. QMAX (8: facet-selected; 20: whole abstracts)
dev = torch.device('cuda')
g = torch.Generator().manual_seed(4)


def repset(lens):
    start = torch.cumsum(lens, 0) - lens
    rows = torch.randn(int(lens.sum()), 768, generator=g).to(dev)
    return ops.DeviceRepSet(rows, start.to(torch.int32).to(dev), lens.to(torch.int32).to(dev), ext=0, max_len=int(lens.max()))


def timed(fn, n=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


c_lens = torch.randint(3, SMAX + 1, (J * NC,), generator=g)
q_lens = torch.randint(1, min(SMAX, QMAX) + 1, (J,), generator=g)
c, q = repset(c_lens), repset(q_lens)
job_off = (torch.arange(J + 1, dtype=torch.int32) * NC).to(dev)
pairs = J * NC
out = ops.ot_rank_batch(q, c, job_off, NC, NC)
t = timed(lambda: ops.ot_rank_batch(q, c, job_off, NC, NC, out=out))
print(f'otAspire {J} x {NC} (rows 3..{SMAX}), one batched call: {t:8.1f} us = {pairs / t:6.2f} M pairs/s', flush=True)
out2 = ops.l2max_rank_batch(q, c, job_off, NC, NC)
t = timed(lambda: ops.l2max_rank_batch(q, c, job_off, NC, NC, out=out2))
print(f'tsAspire {J} x {NC}, one batched call:               {t:8.1f} us = {pairs / t:6.2f} M pairs/s', flush=True)

# the per-query loop: one aspire_ot_rank_f32 call per query on its own pool
qs = [ops.DeviceRepSet(q.rows, q.start[j:j + 1].contiguous(), q.len[j:j + 1].contiguous(), ext=0, max_len=int(q_lens[j])) for j in range(J)]
cs = [ops.DeviceRepSet(c.rows, c.start[j * NC:(j + 1) * NC].contiguous(), c.len[j * NC:(j + 1) * NC].contiguous(), ext=0,
                       max_len=int(c_lens[j * NC:(j + 1) * NC].max())) for j in range(J)]


def loop():
    for j in range(J):
        ops.ot_rank(qs[j], cs[j], NC)


t = timed(loop, n=10)
print(f'otAspire, one call per query:                        {t:8.1f} us = {pairs / t:6.2f} M pairs/s', flush=True)
# batched and per-query calls run different kernels (another summation order): scores agree to a few 1e-5, so near-ties may swap
diff, moved = 0.0, 0
for j in range(J):
    s1, _, i1 = ops.ot_rank(qs[j], cs[j], NC, want=_lib.OT_SIMILARITY)
    diff = max(diff, float((s1[0] - out[0][j * NC:(j + 1) * NC]).abs().max()))
    moved += int((i1[0] != out[2][j]).sum())
print(f'batched vs per-query: max |score difference| {diff:.1e}, {moved} of {pairs} list positions differ (near-ties)')
for form in ('small', 'tile'):
    try:
        with _lib.pinned(OT_FORM=form):
            t = timed(lambda: ops.ot_rank_batch(q, c, job_off, NC, NC, out=out))
        print(f'otAspire batched, OT_FORM={form}: {t:8.1f} us')
    except Exception as e:          # a form that does not take this shape
        print(f'OT_FORM={form}: {type(e).__name__}: {e}')
# the CHUNK form's phases (timing experiments, invalid scores): streaming phase alone / everything but each wave's last solve
for pin in (dict(FUSED_NOSOLVE=1), dict(FUSED_NOSOLVE=2), dict(FUSED_WAVES=1024), dict(FUSED_WAVES=1536), dict(FUSED_WAVES=2560), dict(FUSED_WAVES=3072),
            dict(FUSED_WAVES=4096), dict(), dict(FUSED_WAVES=3072)):
    try:
        with _lib.pinned(**pin):
            t = timed(lambda: ops.ot_rank_batch(q, c, job_off, NC, NC, out=out))
        print(f'otAspire batched, {pin}: {t:8.1f} us')
    except Exception as e:
        print(f'{pin}: {type(e).__name__}: {e}')
