# kernel trace of the encoder at B = 32, L = 256 -> gpurun_out/$1/enc_stats.txt   (extra env via $2, e.g. ASPIRE_HIP_GEMM_RING=2)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r3enc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env ${2:-X=1} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/enc_stats -o enc -- python $R/tools/encbench.py > $OUT/enc_prof.log 2>&1
python $R/tools/statsum.py $OUT/enc_stats > $OUT/enc_stats.txt 2>&1
find $OUT -name "*.csv" | grep -v "kernel_stats" | xargs rm -f
head -12 $OUT/enc_stats.txt | cut -c1-150
