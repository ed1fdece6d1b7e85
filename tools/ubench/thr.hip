// Issue throughput of cross-lane ops: 8 independent chains per wave, 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 2048
template <int MODE>
__global__ void k(float* out, long long* cyc, float a) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + threadIdx.x * 1e-6f * (i + 1);
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int iv = __builtin_bit_cast(int, v[i]);
      if (MODE == 0) v[i] = fmaf(v[i], 0.999f, a);
      if (MODE == 1) v[i] = v[i] + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, 0xB1, 0xF, 0xF, true));
      if (MODE == 2) v[i] = v[i] + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, 0x128, 0xF, 0xF, true));
      if (MODE == 3) { auto r = __builtin_amdgcn_permlane32_swap(iv, __builtin_bit_cast(int, v[(i + 1) & 7]), false, false); int r0 = r[0], r1 = r[1]; v[i] = __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1); }
      if (MODE == 4) { auto r = __builtin_amdgcn_permlane16_swap(iv, __builtin_bit_cast(int, v[(i + 1) & 7]), false, false); int r0 = r[0], r1 = r[1]; v[i] = __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1); }
      if (MODE == 5) v[i] = (threadIdx.x & 8) ? v[i] * 0.5f : v[(i + 1) & 7];
      if (MODE == 6) v[i] = __builtin_amdgcn_exp2f(v[i]) * 0.001f;
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
  out[threadIdx.x + blockIdx.x * blockDim.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
  float* o; long long* c; hipMalloc(&o, 4 * 1024 * 1024); hipMalloc(&c, 8 * 16);
  const char* names[] = {"fma", "dpp quad add", "dpp ror8 add", "swap32+add", "swap16+add", "cndmask+mul", "exp2+mul"};
  for (int wps = 1; wps <= 4; wps *= 2) {   // waves per SIMD: block of 256*wps threads on one CU
    hipMemset(c, 0, 128);
#define RUN(M) k<M><<<1, 256 * wps>>>(o, c, 1.0001f);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
    long long h[16]; hipMemcpy(h, c, 128, hipMemcpyDeviceToHost);
    printf("waves/SIMD=%d:", wps);
    for (int m = 0; m < 7; ++m) printf("  %s %.1f", names[m], (double)h[m] / N / 8);
    printf("  (cycles per op-group per wave)\n");
  }
}
