cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/one.py <<'PY'
import os, sys
R=os.environ['GRAFT_REPO_ROOT']
sys.path.insert(0, R); sys.path.insert(0, R+'/tools')
import torch
from aspire_amd import ops
from kbench import mk
Q, C, S = (int(x) for x in sys.argv[1:4])
q, c = mk(Q, S, 0), mk(C, S, 1)
c.prepare_planes(); c.prepare_boxes()
out = torch.empty(Q * C, device='cuda')
for _ in range(10): ops.ot_sinkhorn(q, c, out=out)
torch.cuda.synchronize()
PY
for shape in "1 20000 12" "4 20000 8"; do
  n=$(echo $shape | tr " " x)
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sq_$n -o s -- python /tmp/one.py $shape > /dev/null 2>&1
  python - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/sq_$n/**/*kernel_stats.csv',recursive=True)[0]
print('$shape')
for r in list(csv.DictReader(open(f)))[:7]:
    print('  %-80s calls %4s avg %8.1f us' % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
