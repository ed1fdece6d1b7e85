// ds_read_b64_tr_b16 semantics probe (gfx950): LDS halfword i holds the value i; lane l supplies byte address 8 l (its own four halfwords);
// prints what every lane receives.   hipcc --offload-arch=gfx950 -O2 tools/ubench/trread.hip -o tools/ubench/trread && tools/ubench/trread
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void probe(float* out, int mode) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (_Float16)(float)i;
    __syncthreads();
    const int l = threadIdx.x;
    int off;                                           // in halfwords
    if (mode == 0) off = 4 * l;                        // lane-linear
    else {                                             // the V gather: 16-lane group G: dims 16 (G & 1) .., lane i of the group: key row (i >> 2), columns 4 (i & 3) ..; rows of 64 halfwords
        const int G = l >> 4, i = l & 15;
        off = (i >> 2) * 64 + 16 * (G & 1) + 4 * (i & 3) + (G >> 1) * 512;
    }
    fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(lds + off));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)v[j];
}
int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 4);
    float h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %6.0f %6.0f %6.0f %6.0f\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    }
    return 0;
}
