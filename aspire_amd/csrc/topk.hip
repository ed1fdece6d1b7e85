// A12: per-query top-k of a [Q, C] score matrix, descending, ties by ascending candidate index -- the
// order of Python's stable sorted(..., reverse=True) (src/evaluation/evaluate.py:76).
//
// Scores become 64-bit keys (monotone score bits << 32 | ~index) so one unsigned descending sort yields
// both rules.  Each workgroup bitonic-sorts a 4096-key chunk in LDS (32 KB) and keeps its top k; chunk
// winners are re-sorted by further passes until one chunk is left.
#include "common.h"

namespace aspire {
namespace {

constexpr int kChunk = 4096;
constexpr int kThreads = 256;

__device__ __forceinline__ uint32_t order_bits(float f) {
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unorder_bits(uint32_t u) {
    return __builtin_bit_cast(float, (u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
}

// in: either scores (first pass; index = position) or keys.  n_in per query; chunk c covers
// [c*kChunk, (c+1)*kChunk).  Writes kk = min(k, kChunk) keys per chunk, or the final outputs.
__global__ void __launch_bounds__(kThreads) topk_pass_kernel(const float* __restrict__ scores,
                                                             const uint64_t* __restrict__ keys_in, int64_t n_in,
                                                             int64_t in_stride, int64_t kk, uint64_t* __restrict__ keys_out,
                                                             int64_t out_stride, int64_t idx_base, int64_t k_final,
                                                             float* __restrict__ top_scores, int64_t* __restrict__ top_idx,
                                                             int sort_n) {
    __shared__ uint64_t key[kChunk];
    const int64_t q = blockIdx.x, chunk = blockIdx.y;
    const int64_t base = chunk * kChunk;
    // sort_n = power of two >= the keys in this chunk: a 1000-candidate pool sorts 1024 keys, not 4096
    for (int t = threadIdx.x; t < sort_n; t += kThreads) {
        const int64_t i = base + t;
        uint64_t kv = 0;  // pad: below every real key
        if (i < n_in) {
            if (scores) {
                kv = ((uint64_t)order_bits(scores[q * in_stride + i]) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)i);
            } else {
                kv = keys_in[q * in_stride + i];
            }
        }
        key[t] = kv;
    }
    __syncthreads();
    // bitonic sort, descending
    for (int size = 2; size <= sort_n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < sort_n / 2; t += kThreads) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const uint64_t a = key[lo], b = key[hi];
                if ((a < b) == desc) {
                    key[lo] = b;
                    key[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    if (top_scores == nullptr) {
        for (int t = threadIdx.x; t < kk; t += kThreads) keys_out[q * out_stride + chunk * kk + t] = key[t];
    } else {
        for (int t = threadIdx.x; t < k_final; t += kThreads) {
            const uint64_t kv = t < sort_n ? key[t] : 0;
            const bool real = kv != 0;
            top_scores[q * k_final + t] = real ? unorder_bits((uint32_t)(kv >> 32)) : -INFINITY;
            top_idx[q * k_final + t] = real ? idx_base + (int64_t)(0xFFFFFFFFu - (uint32_t)kv) : -1;
        }
    }
}

}  // namespace
}  // namespace aspire

using namespace aspire;

extern "C" size_t aspire_topk_workspace_bytes(int64_t Q, int64_t C, int64_t k) {
    if (Q <= 0 || C <= kChunk) return 0;
    const int64_t kk = k < kChunk ? k : kChunk;
    const int64_t n1 = (C + kChunk - 1) / kChunk * kk;
    const int64_t n2 = (n1 + kChunk - 1) / kChunk * kk;
    return (size_t)(Q * (n1 + n2)) * sizeof(uint64_t);
}

extern "C" int aspire_topk_desc_f32(const float* scores, int64_t Q, int64_t C, int64_t k, int64_t idx_base,
                                    float* top_scores, int64_t* top_idx, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    ASPIRE_REQUIRE(Q >= 0 && C >= 0 && k > 0, ASPIRE_ERR_INVALID_ARG, "bad shape Q=%lld C=%lld k=%lld", (long long)Q,
                   (long long)C, (long long)k);
    ASPIRE_REQUIRE(scores && top_scores && top_idx, ASPIRE_ERR_INVALID_ARG, "null pointer");
    ASPIRE_REQUIRE(C < (int64_t)0xFFFFFFFF, ASPIRE_ERR_UNSUPPORTED, "C too large for 32-bit local indices");
    ASPIRE_REQUIRE(C <= kChunk || k < kChunk, ASPIRE_ERR_UNSUPPORTED,
                   "k=%lld >= %d with C=%lld > %d: full sorts beyond one chunk are not built", (long long)k, kChunk,
                   (long long)C, kChunk);
    if (Q == 0) return ASPIRE_OK;
    ASPIRE_REQUIRE(workspace_bytes >= aspire_topk_workspace_bytes(Q, C, k), ASPIRE_ERR_INVALID_ARG,
                   "workspace too small: need %zu bytes", aspire_topk_workspace_bytes(Q, C, k));
    const int64_t kk = k < kChunk ? k : kChunk;
    int64_t n = C, in_stride = C;
    const float* sc = scores;
    const uint64_t* kin = nullptr;
    uint64_t* bufs[2];
    bufs[0] = (uint64_t*)workspace;
    bufs[1] = bufs[0] + (workspace ? Q * ((C + kChunk - 1) / kChunk * kk) : 0);
    int which = 0;
    for (;;) {
        const int64_t nch = n == 0 ? 1 : (n + kChunk - 1) / kChunk;
        const bool final_pass = nch == 1;
        const int64_t out_stride = nch * kk;
        int sort_n = 64;
        while (sort_n < kChunk && sort_n < n) sort_n <<= 1;
        hipLaunchKernelGGL(topk_pass_kernel, dim3((unsigned)Q, (unsigned)nch), dim3(kThreads), 0, (hipStream_t)stream, sc,
                           kin, n, in_stride, kk, final_pass ? nullptr : bufs[which], out_stride, idx_base, k,
                           final_pass ? top_scores : nullptr, final_pass ? top_idx : nullptr, sort_n);
        ASPIRE_LAUNCH_OK();
        if (final_pass) break;
        sc = nullptr;
        kin = bufs[which];
        n = out_stride;
        in_stride = out_stride;
        which ^= 1;
    }
    return ASPIRE_OK;
}
