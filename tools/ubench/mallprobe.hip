// Does the 256 MB Infinity Cache (MALL) keep part of a stream that is larger than itself?  hipcc --offload-arch=gfx950 -O3 mallprobe.hip -o mallprobe
// A flat read kernel (every wave walks contiguous 96 KB items, 8 x 16-byte loads per lane in flight, 2048 waves) over
// buffers of 492 MB (one bench call) visited round-robin, n_bufs = 1, 2, 3, 4, 8: prints every launch's time by buffer.
// If the cache were LRU-like, every launch of every rotation would take the same (cold) time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr size_t kItemF = 32 * 768;

typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT>      // NT: the loads carry the non-temporal hint (__builtin_nontemporal_load)
__global__ void __launch_bounds__(256) read_kernel(const float* __restrict__ base, unsigned n_items, float* out) {
    const int lane = threadIdx.x & 63;
    const unsigned n_waves = gridDim.x * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (unsigned item = blockIdx.x * 4 + (threadIdx.x >> 6); item < n_items; item += n_waves) {
        const float* it = base + (size_t)item * kItemF;
        for (int k0 = 0; k0 < 96; k0 += 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float* src = it + (size_t)(k0 + j) * 256 + lane * 4;
                if constexpr (NT) {
                    const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src));
                    v[j] = make_float4(t.x, t.y, t.z, t.w);
                } else {
                    v[j] = *reinterpret_cast<const float4*>(src);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

__global__ void fill_kernel(float* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (float)((i * 2654435761u) & 0xFFFF) * 1e-4f - 3.f;
}

int main(int argc, char** argv) {
    const unsigned n_items = argc > 1 ? atoi(argv[1]) : 5000;      // 5000 x 96 KB = 492 MB
    const int max_bufs = 8;
    float* buf; float* out;
    const size_t buf_f = n_items * kItemF;
    CK(hipMalloc(&buf, max_bufs * buf_f * sizeof(float)));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, buf, max_bufs * buf_f);
    CK(hipMalloc(&out, 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    printf("buffer = %.0f MB\n", buf_f * 4 / 1e6);
    for (int n_bufs : {1, 2, 3, 4, 8}) {
        const int reps = 6 * n_bufs;
        std::vector<float> t(reps);
        for (int rep = 0; rep < reps; ++rep) {
            const float* src = buf + (size_t)(rep % n_bufs) * buf_f;
            CK(hipEventRecord(a, 0));
            hipLaunchKernelGGL(read_kernel<false>, dim3(512), dim3(256), 0, 0, src, n_items, out);
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&t[rep], a, b));
        }
        printf("%d buffers round-robin, us per launch (last 3 rounds):", n_bufs);
        for (int rep = reps - 3 * n_bufs; rep < reps; ++rep) printf("%s%.0f", rep % n_bufs == 0 ? " | " : " ", t[rep] * 1e3);
        printf("\n");
    }
    // back-to-back without host syncs (the bench's regime): 2 buffers alternating, 200 launches, events per launch
    for (int nt = 0; nt < 2; ++nt) {
        const int n = 200;
        std::vector<hipEvent_t> ev(n + 1);
        for (size_t i = 0; i < ev.size(); ++i) CK(hipEventCreate(&ev[i]));
        CK(hipEventRecord(ev[0], 0));
        for (int i = 0; i < n; ++i) {
            if (nt) hipLaunchKernelGGL(read_kernel<true>, dim3(512), dim3(256), 0, 0, buf + (size_t)(i % 2) * buf_f, n_items, out);
            else hipLaunchKernelGGL(read_kernel<false>, dim3(512), dim3(256), 0, 0, buf + (size_t)(i % 2) * buf_f, n_items, out);
            CK(hipEventRecord(ev[i + 1], 0));
        }
        CK(hipDeviceSynchronize());
        double s[2] = {0, 0};
        for (int i = 100; i < n; ++i) { float ms; CK(hipEventElapsedTime(&ms, ev[i], ev[i + 1])); s[i % 2] += ms; }
        printf("back-to-back, 2 buffers alternating, %s loads: buffer 0 %.1f us, buffer 1 %.1f us per launch (%.2f / %.2f TB/s)\n", nt ? "non-temporal" : "plain",
               s[0] / 50 * 1e3, s[1] / 50 * 1e3, buf_f * 4 / (s[0] / 50 * 1e-3) / 1e12, buf_f * 4 / (s[1] / 50 * 1e-3) / 1e12);
    }
    return 0;
}
