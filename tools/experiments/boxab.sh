cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
 for v in new old; do
  if [ $v = old ]; then export ASPIRE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/oldbox/libaspire_hip.so; else unset ASPIRE_HIP_LIB; fi
  echo "== $v $r"; python tools/planebench.py 2>&1 | grep "ot    fp16"; python tools/planebench.py 128 8192 12 2>&1 | grep "ot    fp16"
 done
done
cd /tmp; export TMPDIR=/tmp
for v in new old; do
  if [ $v = old ]; then export ASPIRE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/oldbox/libaspire_hip.so; else unset ASPIRE_HIP_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$v -o ot -- python $GRAFT_REPO_ROOT/tools/planeprof.py 32 50000 8 80 planes ot > /dev/null 2>&1
  echo "== profile $v"; grep pair_box $(find /tmp/p_$v -name "*kernel_stats.csv") | cut -d, -f5-
done
