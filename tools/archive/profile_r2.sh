#!/bin/bash
# Round-2 rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_r2.sh        -> gpurun_out/prof2/...   (summaries are copied into profiles/ by hand / tools/make_traffic.py)
# Counter passes never share a run with --kernel-trace / --stats (and --pmc runs carry no trace flags at all).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 400 "$@" > $OUT/$name.log 2>&1 || echo "$name: rc=$?"; }
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_[A-Z_0-9]*\|SQ_ACTIVE_INST_[A-Z_0-9]*\|SQ_BUSY_CY[A-Z_]*\|SQ_WAVE_CYCLES\|SQ_INST_CYCLES_[A-Z_]*\|SQ_THREAD_CYCLES_VALU\|SQ_WAIT_[A-Z_]*" | sort -u > $OUT/sq_counters_available.txt
# 1. the bench workload (driver's flags): kernel durations, then HBM traffic of the scoring kernel (separate passes)
# (one call at a time: per-kernel durations of overlapping calls say nothing about a kernel; bench_stats_s3 = the default, three calls in flight)
run bench_stats rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o bench -- python $R/bench.py --steps 20 --warmup 5 --repeats 60 --streams 1 --no-cpu-baseline --no-probes
run bench_stats_s3 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats_s3 -o bench -- python $R/bench.py --steps 20 --warmup 5 --repeats 60 --no-cpu-baseline --no-probes
run pmc_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python $R/bench.py --steps 20 --warmup 5 --repeats 6 --streams 1 --no-cpu-baseline --no-probes
run pmc_write rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python $R/bench.py --steps 20 --warmup 5 --repeats 6 --streams 1 --no-cpu-baseline --no-probes
run pmc_sq_bench rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $OUT/pmc_sq_bench -o pmc -- python $R/bench.py --steps 20 --warmup 5 --repeats 6 --streams 1 --no-cpu-baseline --no-probes
# 2. the Sinkhorn block kernel's issue mix at the config-3 and config-5 shapes
for shape in "32 50000 8" "128 8192 12"; do
  n=$(echo $shape | tr " " x)
  run ot_$n rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ot_$n -o ot -- python $R/tools/otprof.py $shape 5
  run pmc_sink_$n rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS SQ_WAVES --output-format csv -d $OUT/pmc_sink_$n -o pmc -- python $R/tools/otprof.py $shape 1
done
run ot_1x20000x8 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ot_1x20000x8 -o ot -- python $R/tools/otprof.py 1 20000 8 5
run ot_1x20000x12 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ot_1x20000x12 -o ot -- python $R/tools/otprof.py 1 20000 12 5
for d in pmc_fetch pmc_write pmc_sq_bench pmc_sink_32x50000x8 pmc_sink_128x8192x12; do python $R/tools/pmcsum.py $OUT/$d > $OUT/$d.summary.txt 2>&1; done
find $OUT -name "*.csv" | grep -v "kernel_stats" | xargs rm -f
find $OUT -name "*kernel_trace*" | xargs rm -f
du -sh $OUT
