#!/bin/bash
# A/B counters of the fused kernel's cost phase against pair_tile_kernel (gpurun_out/proff/)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/proff
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/sq_counters_available.txt
for J in 8 20; do
for cfg in "fused 1" "tile 0"; do
  set -- $cfg
  n=${1}_J$J
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$n -o s -- python $R/tools/fusedprof.py $J 1000 $1 8 $2 > $OUT/stats_$n.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc1_$n -o p -- python $R/tools/fusedprof.py $J 1000 $1 4 $2 > $OUT/pmc1_$n.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc2_$n -o p -- python $R/tools/fusedprof.py $J 1000 $1 4 $2 > $OUT/pmc2_$n.log 2>&1
  python $R/tools/pmcsum.py $OUT/pmc1_$n > $OUT/pmc1_$n.txt 2>&1
  python $R/tools/pmcsum.py $OUT/pmc2_$n > $OUT/pmc2_$n.txt 2>&1
done
done
find $OUT -name "*.csv" | grep -v "kernel_stats" | xargs rm -f
find $OUT -name "*kernel_trace*" | xargs rm -f
for f in $OUT/stats_*; do [ -d $f ] && find $f -name "*kernel_stats.csv" -exec sh -c 'echo {}; head -8 {}' \; ; done
cat $OUT/pmc1_*.txt $OUT/pmc2_*.txt | grep -A9 "fused\|pair_tile"
