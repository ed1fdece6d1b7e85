// The matrix pipe with operands that arrive from LDS: per iteration READS ds_read_b128 (random fp16 data, rotating addresses) feed
// 12 v_mfma_f32_32x32x16_f16 -- the inner loop of the plane GEMMs without any global memory traffic.  How much of the power-limited
// MFMA rate (tools/ubench/mfmapeak: 51 G MFMA/s on random operands from registers) do the fragment reads cost?
//   hipcc --offload-arch=gfx950 -O3 -o mfmalds tools/ubench/mfmalds.hip && ./mfmalds
#include <hip/hip_runtime.h>
#include <stdio.h>

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int READS, int MODE, int DEPTH = 1>      // DEPTH: iterations between a piece's issue and its wait; MODE 3: as 2 with two of the four pieces streamed from a 4 GB buffer (HBM)
// MODE 0: reads + MFMAs; 1: + one s_barrier per iteration; 2: + four 1 KB LDS-DMA pieces per wave and iteration
                                    // (global -> LDS, L2 / Infinity-Cache resident source), waited one iteration later, + the barrier
__global__ void __launch_bounds__(256, 3) loop(float* out, long long* clk, int iters, const unsigned char* src) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[48 * 1024];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = threadIdx.x; i < 48 * 1024 / 2; i += 256) {
        h ^= h << 13; h ^= h >> 17; h ^= h << 5;
        reinterpret_cast<_Float16*>(lds)[i] = (_Float16)((float)(int)(h & 0xFFFF) * (1.0f / 32768.f) - 1.0f);
    }
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 f[8];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int k = 0; k < 8; ++k) f[k] = *reinterpret_cast<const f16x8*>(lds + ((lane * 16 + k * 1024 + wave * 8192) % (48 * 1024)));
    long long t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        t0 = (long long)__builtin_amdgcn_s_memtime();
        r0 = (long long)__builtin_amdgcn_s_memrealtime();
    }
    unsigned ofs = wave * 8192 + lane * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    // MODE 3: this wave's own stream through the big buffer (64 MB region after the first 64 MB; 3072 waves x 1.3 MB)
    const unsigned long long big = (unsigned long long)(uintptr_t)src + (64ull << 20) + ((size_t)blockIdx.x * 4 + wave) * (size_t)(iters + 8) * 1024;
    unsigned long long gsrc = (unsigned long long)(uintptr_t)src + (size_t)(blockIdx.x % 1024) * 16384 + wave * 4096;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE >= 2) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * DEPTH) : "memory");
        }
        if constexpr (MODE >= 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if constexpr (MODE >= 2) {
            const unsigned dst = lds0 + 16384 + (it % (DEPTH + 1)) * 8192 + wave * 2048 % 8192;
            unsigned keep;
            const unsigned long long res = gsrc + (unsigned long long)(it % 64) * 16777216ull / 64 % 16777216ull;
            // MODE 3: the first of the wave's four pieces comes from its own stream through HBM (1 of 4: what the plane tiles' mix is --
            // a candidate tile once from HBM and once from L2, the query tile from L2), the others from the resident region
            const unsigned long long first = MODE == 3 ? big + (unsigned long long)it * 1024 : res;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:0\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024\n\t"
                         "global_load_lds_dwordx4 %1, %3 offset:2048\n\tglobal_load_lds_dwordx4 %1, %3 offset:3072\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane * 16), "s"(first), "s"(res), "s"(dst) : "memory");
        }
#pragma unroll
        for (int k = 0; k < READS; ++k) f[k & 7] = *reinterpret_cast<const f16x8*>(lds + ((ofs + k * 1024) % (48 * 1024)));
        ofs += 16 * 1024;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 12; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[m & 7], f[(m * 3 + 1) & 7], acc[m & 3], 0, 0, 0);
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

template <int READS, int MODE, int DEPTH = 1>
void run(int iters) {
    static unsigned char* src = nullptr;
    const size_t total = (64ull << 20) + (size_t)3072 * (iters + 8) * 1024;
    if (!src) {
        (void)hipMalloc(&src, total);
        (void)hipMemset(src, 0x3c, total);
    }
    float* out;
    long long* clk;
    (void)hipMalloc(&out, 4096);
    (void)hipMalloc(&clk, 16);
    const int blocks = 256 * 3;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((loop<READS, MODE, DEPTH>), dim3(blocks), dim3(256), 0, 0, out, clk, iters, src);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((loop<READS, MODE, DEPTH>), dim3(blocks), dim3(256), 0, 0, out, clk, iters, src);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long hh[2];
    (void)hipMemcpy(hh, clk, 16, hipMemcpyDeviceToHost);
    const double n_mfma = (double)blocks * 4 * iters * 12;
    printf("mode %d depth %d  ds_read_b128 per 12 MFMAs: %2d   %7.3f ms  %7.1f TFLOP/s  %6.2f G MFMA/s  clock %.2f GHz  pipe busy %.2f\n", MODE, DEPTH, READS, ms,
           n_mfma * 32768.0 / ms / 1e9, n_mfma / ms / 1e6, (double)hh[0] / hh[1] * 0.1, n_mfma * 32 / 1024 / (ms * 1e-3 * (double)hh[0] / hh[1] * 1e8));
}

int main() {
    run<8, 0>(2000);
    run<8, 1>(2000);
    run<8, 2>(8000);
    run<8, 3>(8000);
    run<8, 3, 2>(8000);
    run<8, 3, 3>(8000);
    run<8, 2, 2>(8000);
    return 0;
}
