#!/bin/bash
# Where the P-layout GEMM's operand path (L2 -> TCP -> LDS-DMA) stalls: TA / TCP / SQ-FIFO counters of gemm_p_kernel at 8192 x 2304 x 768
# (round 6: is the L2 -> CU path, 64 B/clk/CU, co-critical with the matrix pipe?).   bash tools/experiments/gemmpath.sh [M N K]
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/gemmpath; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
GP="python $R/tools/gemmprof.py ${1:-8192} ${2:-2304} ${3:-768} 5 planes"
i=0
# (NOT collected: the TA_* sets -- TA_TA_BUSY_sum / TA_*_STALLED_* / TA_*_WAVEFRONTS_sum -- hang rocprofv3 on this pool until the timeout: 2 x 300 s of box time in round 6)
for set in "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o pmc -- $GP > $OUT/p$i.log 2>&1 || echo "pass $i rc=$?"
  python $R/tools/pmcsum.py $OUT/p$i | grep -A12 "gemm_p_kernel"
done
find $OUT -name "*.csv" | xargs rm -f
