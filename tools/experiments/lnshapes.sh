# per-shape durations of the N = 768 GEMMs (out-projection K = 768, FFN2 K = 3072) with and without the LayerNorm epilogue, from a kernel trace
B=${1:-64}; L=${2:-256}
cd /tmp; export TMPDIR=/tmp
for v in fused separate; do
  rm -rf /tmp/lns_$v
  if [ $v = separate ]; then export ASPIRE_HIP_GEMM_LN=off; else unset ASPIRE_HIP_GEMM_LN; fi
  rocprofv3 --kernel-trace --output-format csv -d /tmp/lns_$v -o e -- python $GRAFT_REPO_ROOT/tools/encbench.py $B $L > /dev/null 2>&1
  python - $v <<'PY'
import csv, glob, sys, statistics as st
f = glob.glob(f'/tmp/lns_{sys.argv[1]}/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows]
seq = seq[len(seq) // 2:]            # the timed half
out = {}
for i, (n, d) in enumerate(seq):
    if 'gemm_p_ln_kernel' in n or ('gemm_p_kernel' in n and 'false, false' in n) or 'layernorm_kernel' in n and 'embed' not in n:
        prev = seq[i - 1][0] if i else ''
        key = ('ln-gemm' if 'gemm_p_ln' in n else 'layernorm' if 'layernorm' in n else 'gemm') + ' after ' + ('attention' if 'flash' in prev else 'ffn1' if 'true, false' in prev else 'layernorm/embed' if 'layernorm' in prev else 'gemm' if 'gemm' in prev else prev[:20])
        out.setdefault(key, []).append(d)
for k, v in sorted(out.items()):
    print(f'  {sys.argv[1]:9s} {k:36s} n {len(v):4d}  median {st.median(v):7.1f} us  min {min(v):7.1f}')
PY
done
