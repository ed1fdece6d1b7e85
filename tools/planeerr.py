"""Error of the many-query max-sim forms against float64: fp16 planes (gramp.hip), bf16x3 and fp32-input MFMA (gram.hip).
usage: python tools/planeerr.py [Q C S shift]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import ops
from aspire_amd._lib import pinned


def main():
    Q, C, S = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 4001, 8)
    shift = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    g = torch.Generator().manual_seed(77)
    qrows = (torch.randn(Q * S, 768, generator=g) + shift).cuda()
    crows = (torch.randn(C * S, 768, generator=g) + shift).cuda()
    mk = lambda rows, n: ops.DeviceRepSet(rows, (torch.arange(n, device='cuda', dtype=torch.int32) * S).contiguous(),
                                          torch.full((n,), S, device='cuda', dtype=torch.int32), ext=0, max_len=S)
    q, c = mk(qrows, Q), mk(crows, C)
    qd, cd = qrows.double(), crows.double()
    d = torch.cdist(qd, cd)                                  # [Q S, C S] float64 (direct differences)
    want = -d.view(Q, S, C, S).permute(0, 2, 1, 3).reshape(Q, C, S * S).min(-1).values
    for form in ('f32', 'bf16x3', 'planes'):
        if form == 'planes':
            c.prepare_planes()
            q.prepare_planes(like=c)
        with pinned(COST_PATH='mfma', GEMM='' if form == 'planes' else form):
            got = ops.l2max_scores(q, c).view(Q, C).double()
        e = (got - want).abs()
        if form == 'planes' and e.max().item() > 2e-5:
            bad = (e > 2e-5).nonzero()
            print('outliers:', bad.shape[0], 'of', e.numel())
            for qi, ci in bad[:24].tolist():
                print(f'   q {qi:3d} c {ci:6d} (tile {ci // 16}, slot {ci % 16}) got {got[qi, ci].item():.6f} want {want[qi, ci].item():.6f}')
            print('   distinct tiles', len(set((bad[:, 1] // 16).tolist())), 'distinct queries', len(set(bad[:, 0].tolist())))
        print(f'{form:8s} max {e.max().item():.3e}  mean {e.mean().item():.3e}  p99.9 {e.flatten().kthvalue(int(0.999 * e.numel())).values.item():.3e}')


if __name__ == '__main__':
    main()
