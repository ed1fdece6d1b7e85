"""A few launches of the encoder's GEMM kernel at one shape, for rocprofv3 counter passes.
  python tools/gemmprof.py M N K [reps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import _lib

M, N, K = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
f = _lib.lib.aspire_debug_gemm_f32
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
A = torch.randn(M, K, device='cuda'); B = torch.randn(N, K, device='cuda'); C = torch.empty(M, N, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(reps):
    assert f(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, st) == 0
torch.cuda.synchronize()
# the same shape on pre-split operands (P layout): python tools/gemmprof.py M N K reps planes [ring]
if len(sys.argv) > 5 and sys.argv[5] == 'planes':
    L = _lib.lib
    L.aspire_debug_planes_bytes.restype = ctypes.c_size_t
    L.aspire_debug_planes_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64]
    L.aspire_debug_split_planes.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.aspire_debug_gemm_planes.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    Ap = torch.zeros(L.aspire_debug_planes_bytes(M, K), dtype=torch.uint8, device='cuda')
    Bp = torch.zeros(L.aspire_debug_planes_bytes(N, K), dtype=torch.uint8, device='cuda')
    assert L.aspire_debug_split_planes(A.data_ptr(), M, K, Ap.data_ptr(), 0, st) == 0
    assert L.aspire_debug_split_planes(B.data_ptr(), N, K, Bp.data_ptr(), 1, st) == 0
    with _lib.pinned(GEMM_RING=sys.argv[6] if len(sys.argv) > 6 else ''):
        for _ in range(reps):
            assert L.aspire_debug_gemm_planes(Ap.data_ptr(), Bp.data_ptr(), C.data_ptr(), None, None, M, N, K, 0, st) == 0
        torch.cuda.synchronize()
