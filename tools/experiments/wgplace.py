"""Where does the hardware put the workgroups of a 768 / 2304-workgroup launch (round 6: can co-resident workgroups be made to share an A row block,
so that the CU's vector L1 serves two of three of their A-tile LDS-DMA reads)?  Per-workgroup HW_ID / XCC_ID stamps of the P-layout GEMM's debug build.
  bash tools/build_clock.sh && ASPIRE_HIP_LIB=build/dbg/libaspire_hip_clock.so python tools/experiments/wgplace.py [M N K]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from aspire_amd import _lib

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16384, 768, 768)
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
for _ in range(5): run()
torch.cuda.synchronize()
n_wg = (M + 127) // 128 * (N // 128)
for rep in range(2):
    buf = torch.zeros(n_wg * 16, dtype=torch.int64, device='cuda')
    L.aspire_debug_gemm_buffer(ctypes.c_void_p(buf.data_ptr()))
    run()
    torch.cuda.synchronize()
    L.aspire_debug_gemm_buffer(ctypes.c_void_p(0))
    t = buf.cpu().numpy().reshape(n_wg, 16)
    hw, xcc = t[:, 4], t[:, 5] & 0xf
    se, sh, cu_in, simd, wave = (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15
    cu = ((xcc * 8 + se) * 2 + sh) * 16 + cu_in
    t0 = t[:, 0].min()
    start = (t[:, 0] - t0) / 100.0
    print(f'launch {rep}: {n_wg} workgroups; wg id mod 8 == xcc for {np.mean((np.arange(n_wg) % 8) == xcc) * 100:.1f} %')
    first = np.arange(min(n_wg, 768))
    print(' first 24 workgroups of XCC 0 (wg = 0, 8, 16, ..): (se, sh, cu, simd-of-wave-0, wave slot, start us)')
    for i in range(0, 24 * 8, 8):
        print(f'   wg {i:4d}: se {se[i]} sh {sh[i]} cu {cu_in[i]:2d} simd {simd[i]} slot {wave[i]:2d} start {start[i]:6.2f}')
    # who shares a CU among the first 768 (one round)?
    groups = {}
    for i in first:
        groups.setdefault(int(cu[i]), []).append(int(i))
    sizes = np.bincount([len(v) for v in groups.values()])
    print(' CUs by number of first-round workgroups:', {k: int(v) for k, v in enumerate(sizes) if v})
    ex = [sorted(v) for v in list(groups.values())[:6]]
    print(' members of six CUs:', ex)
    diffs = np.concatenate([np.diff(sorted(v)) for v in groups.values() if len(v) > 1])
    vals, cnt = np.unique(diffs, return_counts=True)
    print(' id distance between workgroups sharing a CU:', dict(zip(vals.tolist()[:12], cnt.tolist()[:12])))
