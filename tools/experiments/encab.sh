# two builds of the library on one box, alternating: bash tools/experiments/encab.sh VARIANT_DIR B L
V=$GRAFT_REPO_ROOT/build/variants/$1/libaspire_hip.so; B=${2:-64}; L=${3:-256}
for r in 1 2 3; do
  echo "new : $(python tools/encbench.py $B $L 2>/dev/null | grep docs)"
  echo "$1: $(ASPIRE_HIP_LIB=$V python tools/encbench.py $B $L 2>/dev/null | grep docs)"
done
