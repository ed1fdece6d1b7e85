import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from aspire_amd import ops, _lib
from tools.kbench import mk
def timed(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / n
res = []
for (Q, C, S) in ((1, 20000, 8), (1, 20000, 12), (1, 1000, 8)):
    q = mk(Q, S, 0); cs = [mk(C, S, 1 + i) for i in range(6)]     # rotating pools (> L3 together)
    out = torch.empty(Q * C, device='cuda'); i = [0]
    def ot():
        i[0] += 1; ops.ot_sinkhorn(q, cs[i[0] % 6], out=out)
    def l2():
        i[0] += 1; ops.l2max_scores(q, cs[i[0] % 6])
    res.append(f'{Q}x{C}x{S}: ot {timed(ot):.1f} us  l2max {timed(l2):.1f} us')
# batch 20 x 1000 x 12 (tile16 MAPPED) and 20 x 1000 x 8 with OT_FORM=tile (pair_tile_kernel)
for S, pins in ((12, {}), (8, dict(OT_FORM='tile'))):
    J, NC = 20, 1000
    sets = [(mk(J, S, 50 + i), mk(J * NC, S, 60 + i)) for i in range(4)]
    job_off = (torch.arange(J + 1, dtype=torch.int32) * NC).cuda(); i = [0]
    outb = ops.ot_rank_batch(sets[0][0], sets[0][1], job_off, NC, 100)
    def b():
        i[0] += 1; qq, cc = sets[i[0] % 4]; ops.ot_rank_batch(qq, cc, job_off, NC, 100, out=outb)
    with _lib.pinned(**pins):
        res.append(f'batch 20x1000x{S} {pins}: {timed(b):.1f} us')
print(' | '.join(res))
