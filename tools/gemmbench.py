"""fp32-MFMA GEMM throughput of the encoder's GEMM kernel at given shapes (GPU box).
usage: python tools/gemmbench.py [M N K ...]   (triples)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import _lib

def main():
    v = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(v[i:i + 3]) for i in range(0, len(v), 3)] or [(8192, 2304, 768), (8192, 3072, 768), (8192, 768, 3072), (8192, 768, 768), (4096, 4096, 4096)]
    f = _lib.lib.aspire_debug_gemm_f32
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
    w = torch.randn(4096, 4096, device='cuda')
    for _ in range(200): w @ w          # clocks up before the first measurement
    torch.cuda.synchronize()
    for M, N, K in shapes:
        A = torch.randn(M, K, device='cuda'); B = torch.randn(N, K, device='cuda'); C = torch.empty(M, N, device='cuda')
        ref = (A[:64].double() @ B.double().T)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        run = lambda: f(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, st)
        for form, tile in (('f32', ''), ('bf16x3', ''), ('bf16x3', '128'), ('bf16x3', '96'), ('bf16x3', '64')):
            if tile == '96' and N % 96:
                continue
            with _lib.pinned(GEMM=form, GEMM_TILE=tile):
                for _ in range(3): assert run() == 0
                torch.cuda.synchronize()
                n, us = 20, 1e30
                for _ in range(3):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(n): run()
                    b.record(); torch.cuda.synchronize()
                    us = min(us, a.elapsed_time(b) / n * 1e3)
            err = (C[:64].double() - ref).abs().max().item()
            print(f'M={M} N={N} K={K} {form:7s} tile {tile or "auto":4s}: {us:8.1f} us  {2 * M * N * K / us / 1e6:6.1f} TFLOP/s ({2 * M * N * K / us / 1e6 / 157.3 * 100:.0f}% of the fp32-MFMA peak)  '
                  f'max|err| vs float64 {err:.2e}', flush=True)
        # pre-split operands (P layout): the planes are formed once (weights at model load, activations by the producing kernel)
        if N % 128 == 0 and K % 32 == 0:
            L = _lib.lib
            L.aspire_debug_planes_bytes.restype = ctypes.c_size_t
            L.aspire_debug_planes_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64]
            L.aspire_debug_split_planes.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
            L.aspire_debug_gemm_planes.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
            Ap = torch.zeros(L.aspire_debug_planes_bytes(M, K), dtype=torch.uint8, device='cuda')
            Bp = torch.zeros(L.aspire_debug_planes_bytes(N, K), dtype=torch.uint8, device='cuda')
            assert L.aspire_debug_split_planes(A.data_ptr(), M, K, Ap.data_ptr(), 0, st) == 0
            assert L.aspire_debug_split_planes(B.data_ptr(), N, K, Bp.data_ptr(), 1, st) == 0
            C.zero_()
            zb = torch.zeros(N, device='cuda')          # (the persistent form, ring 113, wants a bias)
            runp = lambda: L.aspire_debug_gemm_planes(Ap.data_ptr(), Bp.data_ptr(), C.data_ptr(), None, zb.data_ptr(), M, N, K, 0, st)
            # ASPIRE_HIP_GEMM_RING = 10 x (k blocks per stage) + (stages in the ring): 13 = the default (48 KB, three workgroups per CU)
            for ring in ('13', '113', '12'):
                with _lib.pinned(GEMM_RING=ring):
                    for _ in range(10): assert runp() == 0
                    torch.cuda.synchronize()
                    n, us = 20, 1e30
                    for _ in range(3):
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record()
                        for _ in range(n): runp()
                        b.record(); torch.cuda.synchronize()
                        us = min(us, a.elapsed_time(b) / n * 1e3)
                err = (C[:64].double() - ref).abs().max().item()
                errl = (C[-64:].double() - (A[-64:].double() @ B.double().T)).abs().max().item()
                print(f'M={M} N={N} K={K} planes  ring {ring}  : {us:8.1f} us  {2 * M * N * K / us / 1e6:6.1f} TFLOP/s ({2 * M * N * K / us / 1e6 / 157.3 * 100:.0f}% of the fp32-MFMA peak)  '
                      f'max|err| vs float64 {err:.2e} / last rows {errl:.2e}', flush=True)

if __name__ == '__main__':
    main()
