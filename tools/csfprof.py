"""A few batched calls on BASELINE config 4's shape, for rocprofv3 passes (kernel trace, or --pmc FETCH_SIZE / WRITE_SIZE):
  python tools/csfprof.py [ot|l2max] [reps]      -- 50 jobs x 125 candidates of 3 .. 20 rows, facet-selected queries of 1 .. 8 rows"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aspire_amd import ops

method = sys.argv[1] if len(sys.argv) > 1 else 'ot'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
J, NC, SMAX = 50, 125, 20
dev = torch.device('cuda')
g = torch.Generator().manual_seed(4)


def repset(lens):
    start = torch.cumsum(lens, 0) - lens
    rows = torch.randn(int(lens.sum()), 768, generator=g).to(dev)
    return ops.DeviceRepSet(rows, start.to(torch.int32).to(dev), lens.to(torch.int32).to(dev), ext=0, max_len=int(lens.max()))


c_lens = torch.randint(3, SMAX + 1, (J * NC,), generator=g)
q_lens = torch.randint(1, 9, (J,), generator=g)
c, q = repset(c_lens), repset(q_lens)
job_off = (torch.arange(J + 1, dtype=torch.int32) * NC).to(dev)
fn = ops.ot_rank_batch if method == 'ot' else ops.l2max_rank_batch
out = fn(q, c, job_off, NC, NC)
for _ in range(reps):
    fn(q, c, job_off, NC, NC, out=out)
torch.cuda.synchronize()
print('algorithmic bytes per call', 4 * 768 * (int(c_lens.sum()) + int(q_lens.sum())) + 4 * J * NC)
