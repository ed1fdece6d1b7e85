# the attention kernel's duration in an encoder forward, two builds of the library on one box: bash tools/experiments/attnab.sh VARIANT_DIR B L
V=$GRAFT_REPO_ROOT/build/variants/$1/libaspire_hip.so; B=${2:-64}; L=${3:-256}
cd /tmp; export TMPDIR=/tmp
for r in 1 2; do for v in new $1; do
  rm -rf /tmp/at_$v
  if [ $v = new ]; then unset ASPIRE_HIP_LIB; else export ASPIRE_HIP_LIB=$V; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/at_$v -o e -- python $GRAFT_REPO_ROOT/tools/encbench.py $B $L > /tmp/at_$v.txt 2>&1
  echo "$v: $(grep docs /tmp/at_$v.txt | cut -c1-50) | $(grep flash_attn $(find /tmp/at_$v -name '*kernel_stats.csv') | awk -F, '{printf "attention avg %.1f us min %.1f", $(NF-4)/1e3, $(NF-2)/1e3}')"
done; done
