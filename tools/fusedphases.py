"""Per-wave phase stamps of the fused kernel (debug build: tools/build_clock.sh) on the bench shape: J jobs x NC candidates of S rows.
  ASPIRE_HIP_LIB=build/dbg/libaspire_hip_clock.so python tools/fusedphases.py [J NC S]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from aspire_amd import _lib, ops

J, NC, S = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (20, 1000, 8)
dev = torch.device('cuda')
g = torch.Generator().manual_seed(4)
sets = []
for rep in range(4):          # rotating pools (cold reads, as the bench)
    crows = torch.randn(J * NC * S, 768, generator=g).to(dev)
    qrows = torch.randn(J * S, 768, generator=g).to(dev)
    ar = torch.arange(J * NC, device=dev, dtype=torch.int32)
    c = ops.DeviceRepSet(crows, (ar * S).contiguous(), torch.full((J * NC,), S, device=dev, dtype=torch.int32), ext=0, max_len=S)
    q = ops.DeviceRepSet(qrows, (ar[:J] * S).contiguous(), torch.full((J,), S, device=dev, dtype=torch.int32), ext=0, max_len=S)
    sets.append((q, c))
job_off = (torch.arange(J + 1, dtype=torch.int32) * NC).to(dev)
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = torch.zeros(2048 * 8, dtype=torch.int64, device=dev)
out = ops.ot_rank_batch(sets[0][0], sets[0][1], job_off, NC, 100)
for r in range(6):
    ops.ot_rank_batch(sets[r % 4][0], sets[r % 4][1], job_off, NC, 100, out=out)
torch.cuda.synchronize()
raw.aspire_debug_fused_buffer(ctypes.c_void_p(buf.data_ptr()))
ops.ot_rank_batch(sets[2][0], sets[2][1], job_off, NC, 100, out=out)
torch.cuda.synchronize()
raw.aspire_debug_fused_buffer(ctypes.c_void_p(0))
t = buf.cpu().numpy().reshape(2048, 8).astype(np.float64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
us = lambda x: (x - t0) / 100.0
three = t[:, 5] > 0
print(f'{len(t)} waves, {three.sum()} with three items; launch ends at {us(t[:, 7].max()):.1f} us')
names = ['start', 'item1 streamed', 'solve1 set up', 'item2 streamed', 'solve2 set up', 'item3 streamed', 'last solve begins', 'end']
for k in range(8):
    for label, sel in (('3-item', three), ('2-item', ~three)):
        v = t[sel][:, k]
        v = v[v > 0]
        if len(v):
            print(f'{names[k]:20s} {label}: min {us(v.min()):7.1f}  mean {us(v.mean()):7.1f}  max {us(v.max()):7.1f} us')
d = (t[:, 7] - t[:, 6]) / 100.0
print(f'last solve: 3-item waves mean {d[three].mean():.1f} us, 2-item waves mean {d[~three].mean():.1f} us')
