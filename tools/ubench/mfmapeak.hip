// What the matrix pipe sustains with nothing else going on: W waves per SIMD each issuing independent v_mfma chains from
// registers for a fixed number of iterations; reports TFLOP/s and the shader clock held meanwhile (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -o mfmapeak tools/ubench/mfmapeak.hip && ./mfmapeak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int RANDOM>
__global__ void __launch_bounds__(256) mfma_loop(float* out, long long* clk, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a, b, a2[4], b2[4], a3[16], b3[12];
    bf16x8 ab, bb;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    auto rnd = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (float)(int)(h & 0xFFFF) * (1.0f / 32768.f) - 1.0f; };
    for (int r = 0; r < 8; ++r) {
        a[r] = (_Float16)(threadIdx.x * 0.001f + r);
        b[r] = (_Float16)(1.0f / (1 + r));
        ab[r] = (__bf16)(threadIdx.x * 0.001f + r);
        bb[r] = (__bf16)(1.0f / (1 + r));
        for (int u = 0; u < 4; ++u) {
            a2[u][r] = (_Float16)(RANDOM ? rnd() : 1.0f);
            b2[u][r] = (_Float16)(RANDOM ? rnd() : 1.0f);
        }
        for (int u = 0; u < 16; ++u) a3[u][r] = (_Float16)(RANDOM ? rnd() : 1.0f);
        for (int u = 0; u < 12; ++u) b3[u][r] = (_Float16)(RANDOM ? rnd() : 1.0f);
    }
    long long t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        t0 = (long long)__builtin_amdgcn_s_memtime();
        r0 = (long long)__builtin_amdgcn_s_memrealtime();
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (KIND == 4) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3[(4 * u + i) & 15], b3[(5 * u + 3 * i) % 12], acc[i], 0, 0, 0);
                else if constexpr (KIND == 3) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[(u + i) & 3], b2[u & 3], acc[i], 0, 0, 0);
                else if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
                else if constexpr (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
                else {
                    f32x4 c = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
                    acc[i][0] = c[0]; acc[i][1] = c[1]; acc[i][2] = c[2]; acc[i][3] = c[3];
                }
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = (long long)__builtin_amdgcn_s_memtime() - t0;
        clk[1] = (long long)__builtin_amdgcn_s_memrealtime() - r0;
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int KIND, int RANDOM = 0>
void run(const char* name, double flop_per_mfma, int waves_per_simd, int iters) {
    float* out;
    long long* clk;
    hipMalloc(&out, 4096);
    hipMalloc(&clk, 16);
    const int blocks = 256 * waves_per_simd;      // 256 threads = one wave per SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_loop<KIND, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, clk, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop<KIND, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, clk, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double n_mfma = (double)blocks * 4 * iters * 32;
    printf("%-44s waves/SIMD %d  %8.3f ms  %8.1f TFLOP/s  %6.2f G MFMA/s  clock %.2f GHz  cycles per MFMA and SIMD %.1f\n", name, waves_per_simd, ms,
           n_mfma * flop_per_mfma / ms / 1e9, n_mfma / ms / 1e6, (double)h[0] / h[1] * 0.1,
           ms * 1e-3 * ((double)h[0] / h[1] * 1e8) / (n_mfma / 1024));
    hipFree(out);
    hipFree(clk);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_mfma_f32_32x32x16_f16", 2.0 * 32 * 32 * 16, w, 4000);
        run<1>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, w, 4000);
        run<2>("v_mfma_f32_16x16x32_f16", 2.0 * 16 * 16 * 32, w, 4000);
        run<3, 0>("32x32x16_f16 4 operand sets, ones", 2.0 * 32 * 32 * 16, w, 4000);
        run<3, 1>("32x32x16_f16 4 operand sets, random", 2.0 * 32 * 32 * 16, w, 4000);
        run<4, 1>("32x32x16_f16 16 x 12 operand sets, random", 2.0 * 32 * 32 * 16, w, 4000);
    }
    return 0;
}
