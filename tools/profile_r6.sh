#!/bin/bash
# Round-6 rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_r6.sh        -> gpurun_out/prof6/...   (summaries are copied into profiles/ by tools/collect_r6.py)
# Counter passes (--pmc) never share a run with --kernel-trace / --stats and carry no trace flags at all.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof6
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 400 "$@" > $OUT/$name.log 2>&1 || echo "$name: rc=$?"; }
BENCH="python $R/bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-probes"
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES"
# 1. the bench workload, one call at a time: kernel durations, HBM traffic of the scoring kernel, its issue mix
run bench_stats rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o bench -- $BENCH --repeats 60
run bench_stats_s3 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats_s3 -o bench -- python $R/bench.py --steps 20 --warmup 5 --repeats 60 --no-cpu-baseline --no-probes
run pmc_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $BENCH --repeats 6
run pmc_write rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $BENCH --repeats 6
run pmc_sq_bench rocprofv3 --pmc $SQ --output-format csv -d $OUT/pmc_sq_bench -o pmc -- $BENCH --repeats 6
# 2. config 4's shape: the CHUNK form's launches, their HBM traffic
for m in ot l2max; do
  run csf_${m}_stats rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/csf_${m}_stats -o csf -- python $R/tools/csfprof.py $m 40
  run csf_${m}_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/csf_${m}_fetch -o pmc -- python $R/tools/csfprof.py $m 6
  run csf_${m}_write rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/csf_${m}_write -o pmc -- python $R/tools/csfprof.py $m 6
done
run csf_ot_sq rocprofv3 --pmc $SQ --output-format csv -d $OUT/csf_ot_sq -o pmc -- python $R/tools/csfprof.py ot 6
# 3. the encoder: kernel durations; the P-layout GEMM's counters (two SQ passes + the GRBM pass that gives the effective clock)
run enc_stats rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/enc_stats -o enc -- python $R/tools/encbench.py
GP="python $R/tools/gemmprof.py 8192 2304 768 5 planes"
run gemm_pmc1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/gemm_pmc1 -o pmc -- $GP
run gemm_pmc2 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/gemm_pmc2 -o pmc -- $GP
run gemm_pmc3 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/gemm_pmc3 -o pmc -- $GP
run gemm_stats rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gemm_stats -o gemm -- $GP
run gemmbench python $R/tools/gemmbench.py
# 3b. the attention kernel's counters (none existed): two SQ passes over the encoder at B = 32, L = 256
EB="python $R/tools/encbench.py 32 256"
run attn_pmc1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/attn_pmc1 -o pmc -- $EB
run attn_pmc2 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/attn_pmc2 -o pmc -- $EB
for k in attn_pmc1 attn_pmc2; do python $R/tools/pmcsum.py $OUT/$k > $OUT/$k.summary.txt 2>&1; done
# 4. config 5 end to end (one GPU's slice) and the pooling kernel
run e2e_stats rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e2e_stats -o e2e -- python $R/tools/e2ebench.py
# config 5's FULL per-GPU share (1 M documents / 8 GPUs = 125 000 documents of 256 tokens, 128 queries): encode -> store -> planes -> rank
run e2e_full_share python $R/tools/e2ebench.py 125000 256 12 128
# the old attention path (fp32 Q / K / V split inside the kernel) on the same box: A/B of round 6's plane operands + LDS-DMA staging
ASPIRE_HIP_ATTN=f16x2 timeout 400 python $R/tools/e2ebench.py > $OUT/e2e_attn_f16x2.log 2>&1
run e2e_attn_planes python $R/tools/e2ebench.py
run enc_stats_attn_f16x2 env ASPIRE_HIP_ATTN=f16x2 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/enc_stats_attn_f16x2 -o enc -- python $R/tools/encbench.py 64 256
run enc_stats_64x256 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/enc_stats_64x256 -o enc -- python $R/tools/encbench.py 64 256
run pool_stats rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pool_stats -o pool -- python $R/tools/poolbench.py
run poolbench python $R/tools/poolbench.py
# 5. the stand-alone Sinkhorn kernel's issue mix incl. transcendentals (configs 3 / 5 shapes)
for shape in "32 50000 8" "128 8192 12"; do
  n=$(echo $shape | tr " " x)
  run ot_$n rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ot_$n -o ot -- python $R/tools/otprof.py $shape 5
  # (round 6: the Sinkhorn kernel's counters from the PLANE-STORE call -- what bench.py and config 5 run -- not from the fp32-row call above)
  run pmc_sink_$n rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS SQ_WAVES --output-format csv -d $OUT/pmc_sink_$n -o pmc -- python $R/tools/planeprof.py $shape 1 planes ot
done
run ot_1x20000x12 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ot_1x20000x12 -o ot -- python $R/tools/otprof.py 1 20000 12 5
# 6. the fp16-plane cost tiles (gramp.hip): kernel stats + three counter passes per shape; the OT calls on a plane store; the
#    register-fed MFMA ceiling; the clock the chip holds under the bench call
for shape in "32 50000 8" "1 20000 12"; do
  n=planes_$(echo $shape | tr " " x)
  run ${n}_stats rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${n}_stats -o pl -- python $R/tools/planeprof.py $shape 20 planes
  run ${n}_pmc1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/${n}_pmc1 -o pmc -- python $R/tools/planeprof.py $shape 5 planes
  run ${n}_pmc2 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/${n}_pmc2 -o pmc -- python $R/tools/planeprof.py $shape 5 planes
  run ${n}_pmc3 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${n}_pmc3 -o pmc -- python $R/tools/planeprof.py $shape 5 planes
  for k in pmc1 pmc2 pmc3; do python $R/tools/pmcsum.py $OUT/${n}_$k > $OUT/${n}_$k.summary.txt 2>&1; done
done
run ot_planes_32x50000x8 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ot_planes_32x50000x8 -o ot -- python $R/tools/planeprof.py 32 50000 8 10 planes ot
run ot_planes_128x8192x12 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ot_planes_128x8192x12 -o ot -- python $R/tools/planeprof.py 128 8192 12 10 planes ot
for c in FETCH_SIZE WRITE_SIZE; do
  run planes_$c rocprofv3 --pmc $c --output-format csv -d $OUT/planes_$c -o pmc -- python $R/tools/planeprof.py 32 50000 8 3 planes
  python $R/tools/pmcsum.py $OUT/planes_$c > $OUT/planes_32x50000x8_$c.summary.txt 2>&1
done
# config 3's kernel call by call from a cold process: un-profiled (HIP events), then the same run under the profiler (trace kept as a summary)
run c3trace python $R/tools/experiments/c3trace.py 60 $OUT/c3trace.json
run c3trace_prof rocprofv3 --kernel-trace --output-format csv -d $OUT/c3trace_prof -o c3 -- python $R/tools/experiments/c3trace.py 60
python - <<PY > $OUT/c3trace_prof.summary.txt 2>&1
import csv, glob
f = glob.glob('$OUT/c3trace_prof/**/*kernel_trace.csv', recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if 'pair_gram_p_kernel' in r['Kernel_Name']] if f else []
us = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
def st(v):
    v = sorted(v); return dict(min=round(v[0], 1), median=round(v[len(v) // 2], 1), p90=round(v[int(0.9 * (len(v) - 1))], 1), max=round(v[-1], 1), n=len(v))
print('pair_gram_p_kernel under rocprofv3 --kernel-trace, per call in launch order (us):', [round(x, 1) for x in us])
if us: print('first 20:', st(us[:20])); print('rest:', st(us[20:]) if len(us) > 20 else None); print('all:', st(us))
PY
run planebench python $R/tools/planebench.py
run mfmapeak $R/tools/ubench/mfmapeak
run fusedclock python $R/tools/experiments/fusedclock.py
for d in pmc_fetch pmc_write pmc_sq_bench csf_ot_fetch csf_ot_write csf_l2max_fetch csf_l2max_write csf_ot_sq gemm_pmc1 gemm_pmc2 gemm_pmc3 pmc_sink_32x50000x8 pmc_sink_128x8192x12; do
  python $R/tools/pmcsum.py $OUT/$d > $OUT/$d.summary.txt 2>&1
done
find $OUT -name "*.csv" | grep -v "kernel_stats" | xargs rm -f
find $OUT -name "*kernel_trace*" | xargs rm -f
rm -rf $OUT/c3trace_prof
du -sh $OUT
