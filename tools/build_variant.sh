#!/bin/bash
# An instrumented / re-parameterised libaspire_hip.so beside the product one (loaded with ASPIRE_HIP_LIB=...):
#   tools/build_variant.sh NAME FILE.hip "-DFLAG=..." -> build/variants/NAME/libaspire_hip.so
# Only FILE.hip is recompiled; the other objects are the product build's (run __graft_entry__.build() first).
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FILE=$2; shift 2
OUT=$R/build/variants/$NAME
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $R/aspire_amd/csrc/$FILE -o $OUT/$FILE.o
OBJS=$(ls $R/build/obj/*.o | grep -v "/$FILE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libaspire_hip.so $OBJS $OUT/$FILE.o
echo $OUT/libaspire_hip.so
