# counters + kernel stats of the fp16-plane Gram tiles (separate --pmc passes, no trace flags beside them) -> gpurun_out/$1/
# usage: tools/r4prof_planes.sh OUTNAME [Q C S [ot]]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r4planes}
Q=${2:-32}; C=${3:-50000}; S=${4:-8}; OT=${5:-}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 "$@" > $OUT/$name.log 2>&1 || echo "$name: rc=$?"; }
run stats rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python $R/tools/planeprof.py $Q $C $S 20 planes $OT
run pmc1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc1 -o pmc -- python $R/tools/planeprof.py $Q $C $S 5 planes $OT
run pmc2 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/pmc2 -o pmc -- python $R/tools/planeprof.py $Q $C $S 5 planes $OT
run pmc3 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc3 -o pmc -- python $R/tools/planeprof.py $Q $C $S 5 planes $OT
for d in pmc1 pmc2 pmc3; do python $R/tools/pmcsum.py $OUT/$d > $OUT/$d.summary.txt 2>&1; done
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
find $OUT -mindepth 2 -name "*.csv" | xargs rm -f
cat $OUT/kernel_stats.csv | head -8
cat $OUT/pmc1.summary.txt $OUT/pmc2.summary.txt $OUT/pmc3.summary.txt | grep -v "at::native\|^$"
