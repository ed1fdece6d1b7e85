"""Per-workgroup time stamps of the P-layout GEMM (debug build: bash tools/build_clock.sh): when does each workgroup start, when has
its first tile landed, when is its main loop done, when are its stores issued -- and on which CU it ran.
  ASPIRE_HIP_LIB=build/dbg/libaspire_hip_clock.so python tools/gemmphases.py [M N K]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from aspire_amd import _lib

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8192, 2304, 768)
L = _lib.lib
L.aspire_debug_planes_bytes.restype = ctypes.c_size_t
L.aspire_debug_planes_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64]
L.aspire_debug_split_planes.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
L.aspire_debug_gemm_planes.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
L.aspire_debug_gemm_buffer.argtypes = [ctypes.c_void_p]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
A = torch.randn(M, K, device='cuda'); B = torch.randn(N, K, device='cuda'); C = torch.empty(M, N, device='cuda')
Ap = torch.zeros(L.aspire_debug_planes_bytes(M, K), dtype=torch.uint8, device='cuda')
Bp = torch.zeros(L.aspire_debug_planes_bytes(N, K), dtype=torch.uint8, device='cuda')
assert L.aspire_debug_split_planes(A.data_ptr(), M, K, Ap.data_ptr(), 0, st) == 0
assert L.aspire_debug_split_planes(B.data_ptr(), N, K, Bp.data_ptr(), 1, st) == 0
run = lambda: L.aspire_debug_gemm_planes(Ap.data_ptr(), Bp.data_ptr(), C.data_ptr(), None, None, M, N, K, 0, st)
for _ in range(20): run()
torch.cuda.synchronize()
n_wg = (M + 127) // 128 * (N // 128)
buf = torch.zeros(n_wg * 16, dtype=torch.int64, device='cuda')
L.aspire_debug_gemm_buffer(ctypes.c_void_p(buf.data_ptr()))
run()
torch.cuda.synchronize()
L.aspire_debug_gemm_buffer(ctypes.c_void_p(0))
t = buf.cpu().numpy().reshape(n_wg, 16)
t0 = t[:, 0].min()
us = lambda x: (x - t0) / 100.0
start, landed, loop_end, end = us(t[:, 0]), us(t[:, 1]), us(t[:, 2]), us(t[:, 3])
hw, xcc = t[:, 4], t[:, 5] & 0xf
cu = ((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15)
print(f'{M} x {N} x {K}: {n_wg} workgroups on {len(set(cu.tolist()))} distinct (xcc, se, sh, cu); launch spans {end.max():.1f} us')
print(f'per workgroup: start -> first tile landed {np.mean(landed - start):.2f} us (max {np.max(landed - start):.2f}); '
      f'main loop {np.mean(loop_end - landed):.2f} (min {np.min(loop_end - landed):.2f}, max {np.max(loop_end - landed):.2f}); '
      f'stores issued in {np.mean(end - loop_end):.2f} (max {np.max(end - loop_end):.2f})')
order = np.argsort(start)
if K // 32 > 9 and os.environ.get('ASPIRE_HIP_GEMM_PROBE') == '20':
    d = lambda a, b: (t[:, b] - t[:, a]) / 100.0
    print(f'inside step 8 (wave 0): barrier -> LDS-DMA issued {d(8, 9).mean():.2f} us, -> MFMAs issued {d(9, 10).mean():.2f}, -> own pieces of the next stage landed '
          f'{d(10, 11).mean():.2f} (max {d(10, 11).max():.2f}), -> through the barrier {d(11, 12).mean():.2f} (max {d(11, 12).max():.2f}); step {d(8, 12).mean():.2f}')
for lo in range(0, n_wg, 256):
    s = order[lo:lo + 256]
    print(f'  workgroups {lo:4d}.. by start time: start {start[s].min():6.1f} .. {start[s].max():6.1f}, loop end {loop_end[s].min():6.1f} .. {loop_end[s].max():6.1f}, '
          f'end {end[s].min():6.1f} .. {end[s].max():6.1f}, per-step {np.mean(loop_end[s] - landed[s]) / (K // 32):.2f} us')
# one CU's timeline
c0 = cu[order[0]]
print('one CU:', ' | '.join(f'wg {i}: {start[i]:.1f} {landed[i]:.1f} {loop_end[i]:.1f} {end[i]:.1f}' for i in order if cu[i] == c0))
# gap between a workgroup's end and the next start on the same CU
gaps = []
for c in set(cu.tolist()):
    idx = [i for i in order if cu[i] == c]
    ends = sorted(end[idx])
    for i in idx[2:]:
        prev = [e for e in ends if e <= start[i] + 1e-9]
        if prev:
            gaps.append(start[i] - max(prev))
print(f'end of a workgroup -> start of the next one on its CU: mean {np.mean(gaps):.2f} us, max {np.max(gaps):.2f}')
