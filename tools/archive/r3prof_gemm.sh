# counters of the P-layout GEMM at 8192 x 2304 x 768 (separate --pmc passes, no trace flags) -> gpurun_out/$1/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r3gemm}
RING=${2:-}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 "$@" > $OUT/$name.log 2>&1 || echo "$name: rc=$?"; }
run pmc1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc1 -o pmc -- python $R/tools/gemmprof.py 8192 2304 768 5 planes $RING
run pmc2 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/pmc2 -o pmc -- python $R/tools/gemmprof.py 8192 2304 768 5 planes $RING
for d in pmc1 pmc2; do python $R/tools/pmcsum.py $OUT/$d > $OUT/$d.summary.txt 2>&1; done
find $OUT -name "*.csv" | xargs rm -f
cat $OUT/pmc1.summary.txt $OUT/pmc2.summary.txt | grep -v "split_planes\|at::native\|^$" 
