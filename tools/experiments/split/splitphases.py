"""Per-wave phase stamps of the role-split otAspire kernel (split.hip; debug build: tools/build_clock.sh) on the bench shape.
  ASPIRE_HIP_LIB=build/dbg/libaspire_hip_clock.so python tools/splitphases.py [J NC S]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
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
NWG, NW = 256, 12
buf = torch.zeros(NWG * NW * 16, dtype=torch.int64, device=dev)
with _lib.pinned(FUSED_SPLIT=1):
    out = ops.ot_rank_batch(sets[0][0], sets[0][1], job_off, NC, 100)
    for r in range(6):
        ops.ot_rank_batch(sets[r % 4][0], sets[r % 4][1], job_off, NC, 100, out=out)
    torch.cuda.synchronize()
    raw.aspire_debug_split_buffer(ctypes.c_void_p(buf.data_ptr()))
    ops.ot_rank_batch(sets[2][0], sets[2][1], job_off, NC, 100, out=out)
    torch.cuda.synchronize()
    raw.aspire_debug_split_buffer(ctypes.c_void_p(0))
t = buf.cpu().numpy().reshape(NWG, NW, 16).astype(np.float64)
t0 = t[:, :, 0][t[:, :, 0] > 0].min()
us = lambda x: (x - t0) / 100.0


def line(name, v):
    v = v[v > 0]
    if len(v):
        print(f'{name:34s} n {len(v):5d}  min {us(v.min()):7.1f}  mean {us(v.mean()):7.1f}  max {us(v.max()):7.1f} us')


st, so = t[:, :8, :], t[:, 8:, :]
print(f'launch ends at {us(t[:, :, 15].max()):.1f} us')
line('start (streamers)', st[:, :, 0])
for k in range(1, 5):
    line(f'streamer: item {k} handed over', st[:, :, k])
line('streamer turns solver', st[:, :, 8])
line('solver waves: boxes done', so[:, :, 8])
for k in range(1, 7):
    line(f'solver waves: solve {k} done', so[:, :, 8 + k])
for k in range(1, 4):
    line(f'ex-streamers: solve {k} done', st[:, :, 8 + k])
line('end (all waves)', t[:, :, 15])
n_sol = (so[:, :, 9:15] > 0).sum(axis=2)
n_ex = (st[:, :, 9:15] > 0).sum(axis=2)
print(f'solves per dedicated solver wave: mean {n_sol.mean():.2f} max {n_sol.max()};  per ex-streamer: mean {n_ex.mean():.2f} max {n_ex.max()}')
# per-item streaming time of a streamer, by its position in the wave's sequence
for k in range(1, 4):
    a, b = st[:, :, k - 1] if k > 1 else st[:, :, 0], st[:, :, k]
    ok = (a > 0) & (b > 0)
    if ok.any():
        d = (b[ok] - a[ok]) / 100.0
        print(f'streamer item {k}: {ok.sum()} waves, duration mean {d.mean():.1f} min {d.min():.1f} max {d.max():.1f} us')
sd = []
for w in range(so.shape[1]):
    for k in range(1, 7):
        a, b = so[:, w, 8 + k - 1], so[:, w, 8 + k]
        ok = (a > 0) & (b > 0)
        sd.extend(((b[ok] - a[ok]) / 100.0).tolist())
if sd:
    sd = np.array(sd)
    print(f'dedicated solver: time between consecutive solve completions mean {sd.mean():.1f} min {sd.min():.1f} max {sd.max():.1f} us')
