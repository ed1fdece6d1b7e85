"""Many-query cost tiles: fp16-plane form (gramp.hip) against the bf16x3 form (gram.hip) at a given shape.
usage: python tools/planebench.py [Q C S]   (default: BASELINE config 3, 32 x 50 000 x 8)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import _lib, ops
from aspire_amd._lib import pinned
from kbench import timeit, mk


def main():
    print(f'idle clock {ops.clock_under(lambda: None, wall_us=2000, reps=1):.2f} GHz')
    Q, C, S = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 50000, 8)
    q, c = mk(Q, S, 0), mk(C, S, 1)
    nbytes = 4 * 768 * (C * S + Q * S) + 4 * Q * C
    flop = 2.0 * Q * S * C * S * 768
    out = torch.empty(Q * C, device='cuda')
    res = {}
    for name, form in (('bf16x3 (fp32 rows)', 'bf16x3'), ('fp16 planes', '')):
        if form == '':
            c.prepare_boxes()
            us = timeit(lambda: c.prepare_planes(), n=5, warm=1)
            print(f'prepare_planes({C * S} rows): {us:9.1f} us')
            us = timeit(lambda: q.prepare_planes(like=c), n=20, warm=2)
            print(f'prepare_planes({Q * S} query rows): {us:9.1f} us')
        with pinned(COST_PATH='mfma', GEMM=form):
            us = timeit(lambda: ops.l2max_scores(q, c), n=50, warm=5)
            res[name] = ops.l2max_scores(q, c).clone()
            ghz = ops.clock_under(lambda: ops.l2max_scores(q, c))
            print(f'l2max {name:20s} {us:9.1f} us  {Q*C/us:8.2f} Mpairs/s  {nbytes/us/1e3:8.1f} GB/s algorithmic  {flop/us/1e6:7.1f} TFLOP/s-equivalent  clock {ghz:.2f} GHz')
            us = timeit(lambda: ops.ot_sinkhorn(q, c, out=out), n=20, warm=3)
            print(f'ot    {name:20s} {us:9.1f} us  {Q*C/us:8.2f} Mpairs/s')
    a, b = res.values()
    print('max |planes - bf16x3| =', (a - b).abs().max().item())


if __name__ == '__main__':
    main()
