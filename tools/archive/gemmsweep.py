"""The P-layout GEMM at the encoder's four shapes under every (column tile, ring) pin (GPU box).
  python tools/gemmsweep.py [M]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import _lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
L = _lib.lib
L.aspire_debug_planes_bytes.restype = ctypes.c_size_t
L.aspire_debug_planes_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64]
L.aspire_debug_split_planes.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
L.aspire_debug_gemm_planes.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
w = torch.randn(4096, 4096, device='cuda')
for _ in range(100): w @ w
torch.cuda.synchronize()
pins = [dict(GEMM_TILE=t, GEMM_RING=r) for t in ('128', '64') for r in ('22', '23', '13', '14')] if len(sys.argv) < 3 else [eval(sys.argv[2])]
for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    A = torch.randn(M, K, device='cuda'); B = torch.randn(N, K, device='cuda'); C = torch.empty(M, N, device='cuda')
    Ap = torch.zeros(L.aspire_debug_planes_bytes(M, K), dtype=torch.uint8, device='cuda')
    Bp = torch.zeros(L.aspire_debug_planes_bytes(N, K), dtype=torch.uint8, device='cuda')
    assert L.aspire_debug_split_planes(A.data_ptr(), M, K, Ap.data_ptr(), 0, st) == 0
    assert L.aspire_debug_split_planes(B.data_ptr(), N, K, Bp.data_ptr(), 1, st) == 0
    run = lambda: L.aspire_debug_gemm_planes(Ap.data_ptr(), Bp.data_ptr(), C.data_ptr(), None, None, M, N, K, 0, st)
    out = []
    for pin in pins:
        with _lib.pinned(**pin):
            for _ in range(3): assert run() == 0
            torch.cuda.synchronize()
            us = 1e30
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(20): run()
                b.record(); torch.cuda.synchronize()
                us = min(us, a.elapsed_time(b) / 20 * 1e3)
        ideal = M * N * K / 16384 * 3 / 1024 * 32 / 2.4e3        # us at 2.4 GHz, matrix pipe alone
        out.append(f"{pin.get('GEMM_TILE') or 'auto'}/{pin.get('GEMM_RING')}: {us:6.1f} ({ideal / us:.2f})")
    print(f'M={M} N={N} K={K}  tile/ring: us (fraction of the fp16 pipe at 3 products, 2.4 GHz)  ' + '  '.join(out), flush=True)
