#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
  int lane = threadIdx.x;
  int a = lane, b = 100 + lane;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[lane] = r[0]; out[64 + lane] = r[1];
  auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[128 + lane] = s[0]; out[192 + lane] = s[1];
  auto t = __builtin_amdgcn_permlane16_swap(a, a, false, false);
  out[256 + lane] = t[0]; out[320 + lane] = t[1];
}
int main() {
  int* d; hipMalloc(&d, 384 * 4); k<<<1, 64>>>(d); int h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1", "swap16(a,a) r0", "swap16(a,a) r1"};
  for (int j = 0; j < 6; ++j) { printf("%s:", names[j]); for (int i = 0; i < 64; ++i) printf(" %d", h[j * 64 + i]); printf("\n"); }
}
