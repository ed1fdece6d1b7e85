cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof4
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/prof4/planes_$c -o pmc -- python $R/tools/planeprof.py 32 50000 8 3 planes > $R/gpurun_out/prof4/planes_$c.log 2>&1 || echo "$c rc=$?"
  python $R/tools/pmcsum.py $R/gpurun_out/prof4/planes_$c > $R/gpurun_out/prof4/planes_32x50000x8_$c.summary.txt
  find $R/gpurun_out/prof4/planes_$c -name "*.csv" | xargs rm -f
done
cat $R/gpurun_out/prof4/planes_32x50000x8_*SIZE.summary.txt | grep -A2 "pair_gram_p"
python $R/tools/experiments/fusedclock.py > $R/gpurun_out/prof4/fusedclock.log 2>&1; tail -2 $R/gpurun_out/prof4/fusedclock.log
