import torch
from aspire_amd import ops
def mk(n, s, seed):
    g = torch.Generator().manual_seed(seed)
    rows = torch.randn(n * s, 768, generator=g).cuda()
    ar = torch.arange(n, device='cuda', dtype=torch.int32)
    return ops.DeviceRepSet(rows, (ar * s).contiguous(), torch.full((n,), s, device='cuda', dtype=torch.int32), ext=0, max_len=s)
