# The fused kernel built for THREE workgroups per CU (-DASPIRE_FUSED_WAVES3: build/dbg3/libaspire_hip_w3.so) against the product
# build, both on the table-driven form (FUSED_NOSELF=1: 43.5 KB of LDS per workgroup, three fit a CU): kernel durations and the
# wave-cycle counters that say where the time goes.   bash tools/r3prof_waves3.sh   -> gpurun_out/prof3w/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof3w
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-probes"
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES"
SQ2="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
for v in w2 w3; do
  if [ $v = w3 ]; then export ASPIRE_HIP_LIB=$R/build/dbg3/libaspire_hip_w3.so ASPIRE_HIP_FUSED_WAVES=3072; else unset ASPIRE_HIP_LIB ASPIRE_HIP_FUSED_WAVES; fi
  export ASPIRE_HIP_FUSED_NOSELF=1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${v}_stats -o bench -- $BENCH --repeats 40 > $OUT/${v}_stats.log 2>&1
  timeout 300 rocprofv3 --pmc $SQ --output-format csv -d $OUT/${v}_sq -o pmc -- $BENCH --repeats 6 > $OUT/${v}_sq.log 2>&1
  timeout 300 rocprofv3 --pmc $SQ2 --output-format csv -d $OUT/${v}_sq2 -o pmc -- $BENCH --repeats 6 > $OUT/${v}_sq2.log 2>&1
  python $R/tools/statsum.py $OUT/${v}_stats > $OUT/${v}_stats.txt 2>&1
  python $R/tools/pmcsum.py $OUT/${v}_sq > $OUT/${v}_sq.summary.txt 2>&1
  python $R/tools/pmcsum.py $OUT/${v}_sq2 > $OUT/${v}_sq2.summary.txt 2>&1
  $BENCH --repeats 200 > $OUT/${v}_bench.json 2>/dev/null
done
find $OUT -name "*.csv" | grep -v "kernel_stats" | xargs rm -f
for v in w2 w3; do echo "== $v"; head -4 $OUT/${v}_stats.txt | cut -c1-130; grep -A9 "pair_fused_kernel" $OUT/${v}_sq.summary.txt | head -10; grep -A9 "pair_fused_kernel" $OUT/${v}_sq2.summary.txt | head -10; done
