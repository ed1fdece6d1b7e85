"""Independent batch calls in flight on 1, 2, 3 caller streams (each with its own workspace and outputs): calls per second.
  python tools/twostream.py [J NC S]"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd import _lib, ops  # noqa: E402

lib = _lib.lib
D = 768
J, NC, S = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (20, 1000, 8)
k = 100
dev = torch.device('cuda')
g = torch.Generator().manual_seed(0)
nsets = 4
ar = torch.arange(J * NC, device=dev, dtype=torch.int32)
sets = []
for _ in range(nsets):
    qrows = torch.randn(J * S, D, generator=g).to(dev)
    crows = torch.randn(J * NC * S, D, device=dev)
    q = ops.DeviceRepSet(qrows, (ar[:J] * S).contiguous(), torch.full((J,), S, device=dev, dtype=torch.int32), 0, S)
    c = ops.DeviceRepSet(crows, (ar * S).contiguous(), torch.full((J * NC,), S, device=dev, dtype=torch.int32), 0, S)
    sets.append((q, c, q.struct(), c.struct()))
job_off = (torch.arange(J + 1, dtype=torch.int32) * NC).to(dev)
prm = _lib.OtParams(0.05, 0.9, 1.0, 0)
p = lambda t: ctypes.c_void_p(t.data_ptr())
wsb = lib.aspire_ot_rank_batch_workspace_bytes(ctypes.byref(sets[0][2]), ctypes.byref(sets[0][3]), NC, k)

for n_streams in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    bufs = [(torch.empty(J * NC, device=dev), torch.empty(J, k, device=dev), torch.empty(J, k, device=dev, dtype=torch.int64),
             torch.empty(wsb, device=dev, dtype=torch.uint8)) for _ in range(n_streams)]

    def call(i):
        s = i % n_streams
        sc, ts, ti, ws = bufs[s]
        _, _, qs, cs = sets[i % nsets]
        _lib.check(lib.aspire_ot_rank_batch_f32(ctypes.byref(qs), ctypes.byref(cs), D, p(job_off), NC, ctypes.byref(prm), 2, p(sc), k,
                                                None, p(ts), p(ti), None, p(ws), ws.numel(), ctypes.c_void_p(streams[s].cuda_stream)))
    for i in range(8):
        call(i)
    torch.cuda.synchronize()
    n = 400
    t0 = time.perf_counter()
    for i in range(n):
        call(i)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'J={J} NC={NC} S={S}: {n_streams} stream(s): {1e6 * dt / n:6.1f} us per call ({J * NC * n / dt / 1e6:6.1f} M pairs/s), host {1e6 * t_host / n:.1f} us per call', flush=True)
