#!/bin/bash
# Produce the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_all.sh        -> gpurun_out/prof/...
# then copy the summaries with tools/make_traffic.py (bench) and by hand (the others) into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 "$@" > $OUT/$name.log 2>&1 || echo "$name: rc=$?"; }
# 1. the bench workload: kernel durations (one stream so that durations are not overlapped), then HBM traffic
run bench_stats rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o bench -- python $R/bench.py --steps 480 --streams 1 --no-cpu-baseline
run pmc_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python $R/bench.py --steps 50 --no-graph --no-cpu-baseline
run pmc_write rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python $R/bench.py --steps 50 --no-graph --no-cpu-baseline
# 2. the big-grid shapes of BASELINE configs 3 and 5 (per GPU) and a long single-query pool
for shape in "32 50000 8" "128 8192 12" "1 20000 8" "1 20000 12"; do
  n=$(echo $shape | tr " " x)
  run ot_$n rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ot_$n -o ot -- python $R/tools/otprof.py $shape 5
done
run pmc_gram rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $OUT/pmc_gram -o pmc -- python $R/tools/otprof.py 32 50000 8 1
# 3. the encoder and the pooling kernel
run pool rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pool -o pool -- python $R/tools/poolbench.py 256 512 12
run enc rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/enc -o enc -- python $R/tools/encbench.py 32 256
find $OUT -name "*.csv" | grep -v "kernel_stats\|counter_collection" | xargs rm -f
find $OUT -name "*kernel_trace*" | xargs rm -f
du -sh $OUT
