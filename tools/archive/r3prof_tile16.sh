# counters of pair_tile16_kernel at 1 x 20 000 x 12 (separate --pmc passes, no trace flags) -> gpurun_out/$1/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r3tile16}
SHAPE=${2:-"1 20000 12"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 "$@" > $OUT/$name.log 2>&1 || echo "$name: rc=$?"; }
run pmc1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $OUT/pmc1 -o pmc -- python $R/tools/otprof.py $SHAPE 3
run pmc2 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc2 -o pmc -- python $R/tools/otprof.py $SHAPE 3
run pmc3 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -o pmc -- python $R/tools/otprof.py $SHAPE 3
run pmc4 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -o pmc -- python $R/tools/otprof.py $SHAPE 3
run pmc5 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc5 -o pmc -- python $R/tools/otprof.py $SHAPE 3
for d in pmc1 pmc2 pmc3 pmc4 pmc5; do python $R/tools/pmcsum.py $OUT/$d > $OUT/$d.summary.txt 2>&1; done
find $OUT -name "*.csv" | xargs rm -f
cat $OUT/pmc*.summary.txt | grep -A9 "pair_tile16_kernel<false"
