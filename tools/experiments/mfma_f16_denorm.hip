// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs?  A = one subnormal per row, B = 1.0: D = the subnormal's value or 0.
// hipcc --offload-arch=gfx950 -O2 tools/experiments/mfma_f16_denorm.hip -o /tmp/mfma_f16_denorm && /tmp/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float tiny) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    a[0] = (_Float16)tiny;          // k = 8 * (lane >> 5)
    b[0] = (_Float16)1.0f;
    f16v c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    out[threadIdx.x] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4);
    for (float tiny : {1e-3f, 3e-5f, 1e-6f, 6e-8f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, tiny);
        float h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
        printf("input %g (fp16 %s) -> D[0][0] = %g (expected ~%g: two k groups)\n", tiny, tiny < 6.1e-5f ? "subnormal" : "normal", h[0], 2 * tiny);
    }
    return 0;
}
