# experiment: 4 single-lane graphs replayed concurrently on 4 torch streams vs one 4-branch graph
import ctypes, os, sys, time
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import torch
from aspire_amd import _lib, ops
Q, C, S, D, TOPK = 1, 1000, 8, 768, 100
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(0)
query = torch.randn(Q * S, D, generator=g).to(dev); cands = torch.randn(C * S, D, generator=g).to(dev)
ar = torch.arange(max(Q, C), device=dev, dtype=torch.int32)
qset = ops.DeviceRepSet(query, (ar[:Q] * S).contiguous(), torch.full((Q,), S, device=dev, dtype=torch.int32), ext=0, max_len=S)
cset = ops.DeviceRepSet(cands, (ar[:C] * S).contiguous(), torch.full((C,), S, device=dev, dtype=torch.int32), ext=0, max_len=S)
qs, cs = qset.struct(), cset.struct()
prm = _lib.OtParams(0.05, 0.9, 1.0, _lib.CDIST_AUTO); null = ctypes.c_void_p(0); lib = _lib.lib
def stream(): return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
class Lane:
    def __init__(s):
        s.scores = torch.empty(Q, C, device=dev); s.ts = torch.empty(Q, TOPK, device=dev); s.ti = torch.empty(Q, TOPK, device=dev, dtype=torch.int64)
        s.ws = torch.empty(lib.aspire_ot_rank_workspace_bytes(ctypes.byref(qs), ctypes.byref(cs), TOPK), device=dev, dtype=torch.uint8)
    def step(s):
        rc = lib.aspire_ot_rank_f32(ctypes.byref(qs), ctypes.byref(cs), D, ctypes.byref(prm), null, 0, _lib.OT_SIMILARITY, ctypes.c_void_p(s.scores.data_ptr()), TOPK, 0,
                                    ctypes.c_void_p(s.ts.data_ptr()), ctypes.c_void_p(s.ti.data_ptr()), null, ctypes.c_void_p(s.ws.data_ptr()), s.ws.numel(), stream())
        assert rc == 0
NL = int(sys.argv[1]) if len(sys.argv) > 1 else 4
PER = int(sys.argv[2]) if len(sys.argv) > 2 else 15
lanes = [Lane() for _ in range(NL)]
for ln in lanes: ln.step()
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(NL)]
graphs = []
for k in range(NL):
    gk = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gk, stream=streams[k]):
        for _ in range(PER): lanes[k].step()
    graphs.append(gk)
def round_():
    for k in range(NL):
        with torch.cuda.stream(streams[k]): graphs[k].replay()
for _ in range(50): round_()
torch.cuda.synchronize()
R = 200
t0 = time.perf_counter()
for _ in range(R): round_()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f'{NL} graphs x {PER} steps: {R*NL*PER*Q*C/el/1e6:.1f} M alignments/s  {el/(R*NL*PER)*1e6:.2f} us/step')
