#!/bin/bash
# the attention kernel's issue mix (two SQ passes), the encoder at B = 32, L = 256 (13 forwards of 12 layers), as profiles/r05_attn_B32_L256_sq_counters.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/attnpmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
EB="python $R/tools/encbench.py ${1:-32} ${2:-256}"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/p1 -o pmc -- $EB > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p2 -o pmc -- $EB > /dev/null 2>&1
for k in p1 p2; do python $R/tools/pmcsum.py $OUT/$k | grep -A9 "flash_attn"; done
