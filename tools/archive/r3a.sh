set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3a
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/tools/csfbench.py > $OUT/csf.txt 2>&1
python $R/tools/csfbench.py 50 125 16 > $OUT/csf16.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/csf_stats -o csf -- python $R/tools/csfbench.py > $OUT/csf_prof.log 2>&1
python $R/tools/statsum.py $OUT/csf_stats > $OUT/csf_stats.txt 2>&1
python $R/tools/otprof.py 1 20000 12 5 > $OUT/ot12.txt 2>&1
python $R/tools/encbench.py > $OUT/enc.txt 2>&1
python $R/tools/gemmbench.py > $OUT/gemm.txt 2>&1
python $R/bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
find $OUT -name "*.csv" | grep -v "kernel_stats" | xargs rm -f
