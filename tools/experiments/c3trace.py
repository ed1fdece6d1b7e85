"""Config 3's kernel call by call (454 us in the bench against 548 us avg in its profile): N calls of
aspire_l2max_scores_f32 on the 32 x 50 000 x 8 plane store from a COLD process, each under its own pair of HIP events, plus the chip's clock
under the kernel; prints the per-call series and min / median / p90 of the first 20 calls and of the rest.
  python tools/experiments/c3trace.py [N_CALLS] [json_out]
Under `rocprofv3 --kernel-trace` the same run gives the profiler's view of the same calls (tools/profile_r6.sh keeps the trace's summary)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


def main():
    from aspire_amd import ops
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    Q, C, s, D = 32, 50000, 8, 768
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(1)
    crows = torch.empty(C * s, D, device=dev)
    for lo in range(0, C * s, 1 << 16):
        crows[lo:lo + (1 << 16)] = torch.randn(min(1 << 16, C * s - lo), D, generator=g).to(dev)
    qrows = torch.randn(Q * s, D, generator=g).to(dev)
    mk = lambda rows, k: ops.DeviceRepSet(rows, (torch.arange(k, device=dev, dtype=torch.int32) * s).contiguous(),
                                          torch.full((k,), s, device=dev, dtype=torch.int32), ext=0, max_len=s, lens_host=[s] * k)
    c, q = mk(crows, C), mk(qrows, Q)
    c.prepare_planes()
    q.prepare_planes(like=c)                  # (the query planes once: the series is the scoring kernel alone)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:                          # back to back, as the bench's blocks and the profiler's loop run them
        a.record()
        ops.l2max_scores(q, c)
        b.record()
    torch.cuda.synchronize()
    us = [a.elapsed_time(b) * 1e3 for a, b in evs]
    since_start = [evs[0][0].elapsed_time(b) for _, b in evs]

    def stats(v):
        v = sorted(v)
        return {'min': v[0], 'median': v[len(v) // 2], 'p90': v[int(0.9 * (len(v) - 1))], 'max': v[-1], 'n': len(v)}
    res = {'what': f'{n} back-to-back aspire_l2max_scores_f32 calls ({Q} x {C} x {s}, plane store), a cold process, each under its own HIP events',
           'us_per_call': [round(x, 1) for x in us], 'ms_since_first_call': [round(x, 2) for x in since_start],
           'first_20': stats(us[:20]), 'rest': stats(us[20:]) if n > 20 else None, 'all': stats(us),
           'clock_ghz_under_kernel_after_the_series': ops.clock_under(lambda: ops.l2max_scores(q, c))}
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 2:
        json.dump(res, open(sys.argv[2], 'w'), indent=1)


if __name__ == '__main__':
    main()
