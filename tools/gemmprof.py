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
