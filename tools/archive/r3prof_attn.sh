R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3fa
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc1 -o pmc -- python $R/tools/encbench.py 32 256 > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -o pmc -- python $R/tools/encbench.py 32 256 > $OUT/pmc2.log 2>&1
for d in pmc1 pmc2; do python $R/tools/pmcsum.py $OUT/$d > $OUT/$d.summary.txt 2>&1; done
find $OUT -name "*.csv" | xargs rm -f
grep -A9 "flash_attn" $OUT/pmc1.summary.txt $OUT/pmc2.summary.txt
