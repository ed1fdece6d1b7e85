# The instrumented library (per-wave / per-workgroup wall-clock stamps: -DASPIRE_PHASE_CLOCK) -> build/dbg/libaspire_hip_clock.so
# (tools/chunkphases.py, tools/gemmphases.py; run them with ASPIRE_HIP_LIB=build/dbg/libaspire_hip_clock.so)
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build/dbg
objs=""
for s in $R/aspire_amd/csrc/*.hip; do
    o=$R/build/dbg/$(basename $s).o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DASPIRE_PHASE_CLOCK -I$R/include -c $s -o $o &
    objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/dbg/libaspire_hip_clock.so $objs
echo built $R/build/dbg/libaspire_hip_clock.so
