# LayerNorm in the GEMM epilogue against the separate pass: per-kernel times of one encoder shape under rocprofv3
# usage: bash tools/experiments/lnab.sh B L [extra env assignments for the fused run]
B=${1:-64}; L=${2:-256}; shift 2
cd /tmp; export TMPDIR=/tmp
for v in fused separate; do
  rm -rf /tmp/ln_$v
  if [ $v = separate ]; then env ASPIRE_HIP_GEMM_LN=off rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ln_$v -o e -- python $GRAFT_REPO_ROOT/tools/encbench.py $B $L > /tmp/ln_$v.txt 2>&1
  else env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ln_$v -o e -- python $GRAFT_REPO_ROOT/tools/encbench.py $B $L > /tmp/ln_$v.txt 2>&1; fi
  echo "== $v: $(grep docs/s /tmp/ln_$v.txt)"
  python - $v <<'PY'
import csv, glob, sys
f = glob.glob(f'/tmp/ln_{sys.argv[1]}/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print('  %-110s %6s calls %9.1f us avg %5.1f %%' % (r['Name'][:110], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
done
