// HBM read-pattern micro-benchmark for the cost kernel's design (MI355X):  hipcc --offload-arch=gfx950 -O3 readpat.hip -o readpat
// Every wave sums "items" of 4 documents x 8 rows x 3072 B (98 304 B) with 8 x 16-byte loads per lane in flight, in three
// address patterns:
//   tile : pair_tile / pair_fused kernel's pattern -- one load instruction = 4 documents x 256 B (same row, same 256-B window),
//          12 stages x 8 rows
//   row  : one load instruction = 1 KB contiguous of ONE row (64 lanes x 16 B), 96 loads per item in groups of 8
//   flat : one load instruction = 1 KB contiguous, the item is one contiguous 96 KB block walked front to back
// and prints the achieved GB/s for several numbers of resident waves (items are claimed dynamically).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kRowF = 768;              // floats per row
constexpr int kItemRows = 32;

template <int PAT>
__global__ void __launch_bounds__(256) read_kernel(const float* __restrict__ base, unsigned n_items, unsigned* counter, float* out) {
    const int lane = threadIdx.x & 63;
    const unsigned n_waves = gridDim.x * 4;
    unsigned item = blockIdx.x * 4 + (threadIdx.x >> 6);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    while (item < n_items) {
        const float* it = base + (size_t)item * kItemRows * kRowF;
        if (PAT == 0) {
            const int g = lane >> 4, c = lane & 15;
            for (int st = 0; st < 12; ++st) {
                float4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(it + (size_t)(g * 8 + j) * kRowF + (st * 16 + c) * 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
            }
        } else if (PAT == 1) {
            for (int r0 = 0; r0 < kItemRows; r0 += 8)
                for (int t = 0; t < 3; ++t) {
                    float4 v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(it + (size_t)(r0 + j) * kRowF + t * 256 + lane * 4);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
                }
        } else {
            for (int k0 = 0; k0 < 96; k0 += 8) {
                float4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(it + (size_t)(k0 + j) * 256 + lane * 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
            }
        }
        unsigned nx = 0;
        if (lane == 0) nx = atomicAdd(counter, 1u);
        item = n_waves + __builtin_amdgcn_readfirstlane(nx);
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;      // keep the loads alive
}

int main() {
    const unsigned n_items = 5000;                       // 20 jobs x 1000 candidates / 4 = one bench call (492 MB)
    const size_t item_f = (size_t)kItemRows * kRowF;
    const int n_bufs = 4;                                // rotate 4 x 492 MB: nothing survives in the 256 MB L3
    float* buf;
    CK(hipMalloc(&buf, n_bufs * n_items * item_f * sizeof(float)));
    CK(hipMemset(buf, 0, n_bufs * n_items * item_f * sizeof(float)));
    unsigned* counter; float* out;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&out, 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char* names[3] = {"tile", "row", "flat"};
    for (int waves_per_cu : {4, 8, 12, 16, 24, 32}) {
        for (int pat = 0; pat < 3; ++pat) {
            const unsigned blocks = 256 * waves_per_cu / 4;
            float best = 1e9f;
            for (int rep = 0; rep < 12; ++rep) {
                CK(hipMemsetAsync(counter, 0, 4, 0));
                const float* src = buf + (size_t)(rep % n_bufs) * n_items * item_f;
                CK(hipEventRecord(a, 0));
                if (pat == 0) hipLaunchKernelGGL(read_kernel<0>, dim3(blocks), dim3(256), 0, 0, src, n_items, counter, out);
                else if (pat == 1) hipLaunchKernelGGL(read_kernel<1>, dim3(blocks), dim3(256), 0, 0, src, n_items, counter, out);
                else hipLaunchKernelGGL(read_kernel<2>, dim3(blocks), dim3(256), 0, 0, src, n_items, counter, out);
                CK(hipEventRecord(b, 0));
                CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (rep >= 2 && ms < best) best = ms;
            }
            printf("%2d waves/CU  %-4s  %7.1f us  %6.2f TB/s\n", waves_per_cu, names[pat], best * 1e3, n_items * item_f * 4 / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
