#!/bin/bash
# The product library + round 5's role-split kernel (split.hip, an experiment that lost: NOTES.md round 5): score.hip recompiled with
# -DASPIRE_EXPERIMENT_SPLIT (its two call sites), split.hip beside it -> build/variants/split/libaspire_hip.so  (ASPIRE_HIP_LIB=... to load it)
set -eu
R=$(cd "$(dirname "$0")/../../.." && pwd)
OUT=$R/build/variants/split
mkdir -p $OUT
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DASPIRE_EXPERIMENT_SPLIT -I$R/aspire_amd/csrc"
/opt/rocm/bin/hipcc $F -c $R/aspire_amd/csrc/score.hip -o $OUT/score.hip.o &
/opt/rocm/bin/hipcc $F -c $R/tools/experiments/split/split.hip -o $OUT/split.hip.o &
wait
OBJS=$(ls $R/build/obj/*.o | grep -v "/score.hip.o" | grep -v "/split.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libaspire_hip.so $OBJS $OUT/score.hip.o $OUT/split.hip.o
echo $OUT/libaspire_hip.so
