"""Per-wave phase stamps of the fused kernel (debug build: -DASPIRE_PHASE_CLOCK, build/dbg/libaspire_hip_clock.so) on the config-4
shape: when do the waves finish their items' streaming phases, when does the last solve start / end.
  ASPIRE_HIP_LIB=build/dbg/libaspire_hip_clock.so python tools/chunkphases.py [J NC SMAX]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from aspire_amd import _lib, ops

J, NC, SMAX = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (50, 125, 20)
dev = torch.device('cuda')
g = torch.Generator().manual_seed(4)


def repset(lens):
    start = torch.cumsum(lens, 0) - lens
    rows = torch.randn(int(lens.sum()), 768, generator=g).to(dev)
    return ops.DeviceRepSet(rows, start.to(torch.int32).to(dev), lens.to(torch.int32).to(dev), ext=0, max_len=int(lens.max()))


c_lens = torch.randint(3, SMAX + 1, (J * NC,), generator=g)
q_lens = torch.randint(1, min(SMAX, 8) + 1, (J,), generator=g)
c, q = repset(c_lens), repset(q_lens)
job_off = (torch.arange(J + 1, dtype=torch.int32) * NC).to(dev)
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = torch.zeros(2048 * 8, dtype=torch.int64, device=dev)
out = ops.ot_rank_batch(q, c, job_off, NC, NC)
for _ in range(3):
    ops.ot_rank_batch(q, c, job_off, NC, NC, out=out)
torch.cuda.synchronize()
raw.aspire_debug_fused_buffer(ctypes.c_void_p(buf.data_ptr()))
ops.ot_rank_batch(q, c, job_off, NC, NC, out=out)
torch.cuda.synchronize()
raw.aspire_debug_fused_buffer(ctypes.c_void_p(0))
t = buf.cpu().numpy().reshape(2048, 8).astype(np.float64)
live = t[:, 0] > 0
t = t[live]
t0 = t[:, 0].min()
us = lambda x: (x - t0) / 100.0          # 100 MHz
two = t[:, 3] > 0
print(f'{live.sum()} waves, {two.sum()} with two items')
names = ['start', 'item1 streamed', 'solve1 set up', 'item2 streamed', 'solve2 set up', '-', 'last solve begins', 'end']
for k in (0, 1, 2, 3, 4, 6, 7):
    for label, sel in (('two-item waves', two), ('one-item waves', ~two)):
        v = t[sel][:, k]
        v = v[v > 0]
        if len(v):
            print(f'{names[k]:20s} {label}: min {us(v.min()):7.1f}  mean {us(v.mean()):7.1f}  max {us(v.max()):7.1f} us')
d = (t[:, 7] - t[:, 6]) / 100.0
print(f'last solve: two-item waves mean {d[two].mean():.1f} us, one-item waves mean {d[~two].mean():.1f} us')
