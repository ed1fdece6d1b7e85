// Dependent chain of one Sinkhorn step of sinkhorn_kernel<1> (one wave per SIMD), piece by piece (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -I aspire_amd/csrc tools/ubench/chain.hip -o build/dbg/chain && build/dbg/chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "common.h"
using namespace aspire;
#define N 2048
template <int MODE>
__global__ void k(float* out, long long* cyc, float r2, float h, float la, float lb) {
  float phi = -1.0f - threadIdx.x * 1e-3f, f = 0.f, g = 0.f;
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N; ++i) {
    float sc, sr;
    if (MODE == 1 || MODE == 4) { sc = fmaf(phi, r2, la) * 0.999f; sr = fmaf(phi, r2, lb) * 0.999f; }      // no exp
    else { sc = __builtin_amdgcn_exp2f(fmaf(phi, r2, la)); sr = __builtin_amdgcn_exp2f(fmaf(phi, r2, lb)); }
    if (MODE == 2 || MODE == 4) { sc += sc * 0.5f; sr += sr * 0.5f; sc += sc * 0.25f; sr += sr * 0.25f; sc += sc * 0.125f; sr += sr * 0.125f; }  // no cross-lane
    else if (MODE == 3) {   // three DPP levels each (as if no row crossing existed)
      sc += dpp_mov<0x124>(sc, sc); sr += lane_xor<1>(sr);
      sc += dpp_mov<0x128>(sc, sc); sr += lane_xor<2>(sr);
      sc += dpp_mov<0x141>(sc, sc); sr += dpp_mov<0x141>(sr, sr);
    } else {
      sc += dpp_mov<0x124>(sc, sc); sr += lane_xor<1>(sr);
      __builtin_amdgcn_sched_barrier(0);
      sc += dpp_mov<0x128>(sc, sc); sr += lane_xor<2>(sr);
      __builtin_amdgcn_sched_barrier(0);
      sc = swap_add<32>(sc, sc); sr = swap_add<16>(sr, sr);
      __builtin_amdgcn_sched_barrier(0);
    }
    float lc, lr;
    if (MODE == 1 || MODE == 4) { lc = sc * 1e-3f; lr = sr * 1e-3f; }
    else { lc = __builtin_amdgcn_logf(sc); lr = __builtin_amdgcn_logf(sr); }
    phi = fmaf(-h, lr + lc, phi);
    g = fmaf(-h, lc, g);
    f = fmaf(-h, lr, f);
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x * blockDim.x] = phi + f + g;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
  float* o; long long* c; hipMalloc(&o, 4 * 64 * 1024); hipMalloc(&c, 8 * 16); hipMemset(c, 0, 128);
#define RUN(M) k<M><<<1, 64>>>(o, c, 0.7f, 1e-4f, -2.0f, -2.1f);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
  long long hc[16]; hipMemcpy(hc, c, 128, hipMemcpyDeviceToHost);
  const char* names[] = {"full step", "no exp/log", "no cross-lane", "DPP only (no swaps)", "plain VALU only"};
  for (int m = 0; m < 5; ++m) printf("%-24s %7.1f cycles/step\n", names[m], (double)hc[m] / N);
}
