// Dependent-chain latency / issue-rate microbenchmarks for one wave per SIMD (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 4096
template <int MODE>
__global__ void k(float* out, long long* cyc, float a, float b) {
  float x = a + threadIdx.x * 1e-7f, y = b, z = a * 0.5f, w = b * 0.25f;
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
  for (int i = 0; i < N; ++i) {
    if (MODE == 0) x = fmaf(x, b, a);                                  // dependent fma
    if (MODE == 1) x = __builtin_amdgcn_exp2f(x) * 1e-3f;                // dependent exp + mul
    if (MODE == 2) x = __builtin_amdgcn_logf(x) + 3.0f;                  // dependent log + add
    if (MODE == 3) { int ix = __builtin_bit_cast(int, x); x = x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ix, ix, 0xB1, 0xF, 0xF, true)); x *= 0.5f; }  // dpp add + mul
    if (MODE == 4) { int ix = __builtin_bit_cast(int, x); auto r = __builtin_amdgcn_permlane32_swap(ix, ix, false, false); int r0 = r[0], r1 = r[1]; x = (__builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1)) * 0.5f; }  // swap + add + mul
    if (MODE == 5) { x = fmaf(x, b, a); y = fmaf(y, b, a); z = fmaf(z, b, a); w = fmaf(w, b, a); }  // 4 independent fma chains
    if (MODE == 6) { x = __builtin_amdgcn_exp2f(x) * 1e-3f; y = __builtin_amdgcn_exp2f(y) * 1e-3f; z = __builtin_amdgcn_exp2f(z) * 1e-3f; w = __builtin_amdgcn_exp2f(w) * 1e-3f; }
    if (MODE == 7) { int ix = __builtin_bit_cast(int, x); x = x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ix, ix, 0x128, 0xF, 0xF, true)); x *= 0.5f; }  // row_ror:8
    if (MODE == 8) { x = x * b; x = x + a; x = x - y; x = x * 0.999f; }   // 4 dependent plain ops
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x * blockDim.x] = x + y + z + w;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
  float* o; long long* c; hipMalloc(&o, 4 * 64 * 1024); hipMalloc(&c, 8 * 16); hipMemset(c, 0, 128);
#define RUN(M) k<M><<<1, 64>>>(o, c, 1.0001f, 0.9999f);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
  long long h[16]; hipMemcpy(h, c, 128, hipMemcpyDeviceToHost);
  const char* names[] = {"dep fma", "dep exp2+mul", "dep log2+add", "dep dpp-add(quad)+mul", "dep swap32+add+mul", "4 indep fma", "4 indep exp2+mul", "dep dpp-add(ror8)+mul", "4 dep plain ops"};
  for (int m = 0; m < 9; ++m) printf("%-24s %7.1f cycles/iter\n", names[m], (double)h[m] / N);
}
