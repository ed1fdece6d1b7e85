import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import _lib, ops
from tools_common import mk
q, c = mk(1, 8, 0), mk(1000, 8, 1)
buf = torch.zeros(64, dtype=torch.int64, device='cuda')
_lib.lib.aspire_debug_k1_buffer.argtypes = [ctypes.c_void_p]
_lib.lib.aspire_debug_k1_buffer(ctypes.c_void_p(buf.data_ptr()))
for _ in range(3):
    ops.ot_sinkhorn(q, c)
torch.cuda.synchronize()
t = buf.cpu().tolist()
names = ['half0 (D2+G acc + 2 reduces)', 'half1', 'norms + bbox', 'barrier 1', 'finish_pair', 'barrier 2', 'cur = nxt']
for k in range(7):
    print(f'{names[k]:32s} {t[k+1]-t[k]:7d} cycles')
print('item total', t[7] - t[0])

print('sinkhorn_kernel<1>, one wave:')
sn = ['marginals', 'schedule scalars (f64 logs)', 'up to solve()', 'solve (77 steps)', 'outputs']
for k in range(5):
    print(f'{sn[k]:32s} {t[32+4+k]-t[32+3+k]:7d} cycles')
