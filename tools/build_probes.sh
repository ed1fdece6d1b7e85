# The library with the P-layout GEMM's timing probes compiled in (-DASPIRE_GEMM_PROBES; no stamps) -> build/probe/libaspire_hip_probe.so
#   ASPIRE_HIP_LIB=build/probe/libaspire_hip_probe.so ASPIRE_HIP_GEMM_PROBE=6 python tools/gemmsweep.py 8192 'dict(GEMM_RING="13")'
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build/probe
objs=""
for s in $R/aspire_amd/csrc/*.hip; do
    o=$R/build/probe/$(basename $s).o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DASPIRE_GEMM_PROBES -I$R/include -c $s -o $o &
    objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/probe/libaspire_hip_probe.so $objs
echo built $R/build/probe/libaspire_hip_probe.so
