"""Phase cycle stamps of ot_kernel block 0 / wave 0 (instrumented build: build/dbg/libaspire_hip_dbg.so).
usage: ASPIRE_HIP_LIB=build/dbg/libaspire_hip_dbg.so python tools/phases.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import _lib, ops
from tools_common import mk  # noqa

names = ['start', 'partials done', 'barrier', 'gather+barrier', 'marginals', 'eps schedule (f64)', '-',
         'sinkhorn solve', 'outputs']
q, c = mk(1, 8, 0), mk(1000, 8, 1)
buf = torch.zeros(32768 + 12 * 1024, dtype=torch.int64, device='cuda')
_lib.lib.aspire_debug_phase_buffer.argtypes = [ctypes.c_void_p]
_lib.lib.aspire_debug_phase_buffer(ctypes.c_void_p(buf.data_ptr()))
for _ in range(3):
    buf.zero_(); buf[44] = 2**62
    ops.ot_sinkhorn(q, c)
torch.cuda.synchronize()
t = buf.cpu().tolist()
for k in range(1, 9):
    print(f'{names[k]:24s} {t[k] - t[k-1]:8d} cycles')
print(f'{"total":24s} {t[8] - t[0]:8d} cycles; wall {(t[24]-t[16])*10} ns -> shader clock {(t[8]-t[0])/((t[24]-t[16])*10e-9)/1e9:.2f} GHz')
import numpy as np
print('per-wave-id last end minus first start (us):', [(t[40+w]-t[44])/100 for w in range(3)])
b = np.array(t[64:64 + 1000]).reshape(-1, 2)
b = b[b[:, 0] > 0]
t0 = b[:, 0].min()
print('blocks', len(b), 'start offsets us: min/med/max', (b[:,0]-t0).min()/100, np.median(b[:,0]-t0)/100, (b[:,0]-t0).max()/100)
print('block durations us: min/med/max', (b[:,1]-b[:,0]).min()/100, np.median(b[:,1]-b[:,0])/100, (b[:,1]-b[:,0]).max()/100, ' last end', (b[:,1].max()-t0)/100)

d = (b[:,1]-b[:,0])/100
order = np.argsort(-d)[:12]
print('slowest blocks (id, dur us, start us):', [(int(i), float(d[i]), float((b[i,0]-t0)/100)) for i in order])
print('fallback solves:', t[46], ' blocks >25us:', int((d>25).sum()))
for trial in range(3):
    buf.zero_(); buf[44] = 2**62
    ops.ot_sinkhorn(q, c); torch.cuda.synchronize()
    t = buf.cpu().tolist()
    b = np.array(t[64:64 + 1000]).reshape(-1, 2); b = b[b[:, 0] > 0]; d = (b[:,1]-b[:,0])/100
    print('trial', trial, 'max dur', d.max(), 'blocks>25us', int((d>25).sum()), 'fallbacks', t[46], 'kernel span', (max(t[40:43])-t[44])/100)

ph = np.array(t[4096:4096 + 16 * 500]).reshape(500, 16)[:, :9]
dur = (ph[:, 8] - ph[:, 0]) / 100
for i in list(np.argsort(-dur)[:4]) + list(np.argsort(dur)[:2]):
    print('block', int(i), 'phase deltas us:', [round(float(x), 2) for x in np.diff(ph[i]) / 100])

w = np.array(t[32768:32768 + 500 * 12]).reshape(500, 3, 4)
hw, xcc = w[:, :, 0], w[:, :, 1] & 0xf
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
t0 = w[:, :, 2].min()
wd = (w[:, :, 3] - w[:, :, 2]) / 100
key = xcc[:, 0] * 10000 + se[:, 0] * 1000 + sh[:, 0] * 100 + cu[:, 0]
from collections import defaultdict
bycu = defaultdict(list)
for bidx in range(500):
    bycu[int(key[bidx])].append(bidx)
print('distinct CUs used:', len(bycu), ' blocks per CU histogram:', np.bincount([len(v) for v in bycu.values()]))
shown = 0
for k, blks in bycu.items():
    if len(blks) > 1 and shown < 6:
        shown += 1
        print('CU', k, [(b_, [int(x) for x in simd[b_]], [round(float(x), 1) for x in wd[b_]]) for b_ in blks])
