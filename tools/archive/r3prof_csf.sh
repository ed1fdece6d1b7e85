# kernel trace of the config-4 shape (tools/csfbench.py) -> gpurun_out/$1/csf_stats.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r3csf}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/csf_stats -o csf -- python $R/tools/csfbench.py ${2:-50} ${3:-125} ${4:-20} > $OUT/csf_prof.log 2>&1
python $R/tools/statsum.py $OUT/csf_stats > $OUT/csf_stats.txt 2>&1
find $OUT -name "*.csv" | grep -v "kernel_stats" | xargs rm -f
