# two builds of the library on one box, alternating, on any tool: bash tools/experiments/libab.sh VARIANT_DIR "python tools/planebench.py" "grep pattern"
V=$GRAFT_REPO_ROOT/build/variants/$1/libaspire_hip.so
for r in 1 2 3; do
  echo "new : $($2 2>/dev/null | grep "$3" | tr '\n' ' ')"
  echo "$1: $(ASPIRE_HIP_LIB=$V $2 2>/dev/null | grep "$3" | tr '\n' ' ')"
done
