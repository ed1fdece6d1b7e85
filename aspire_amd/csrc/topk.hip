// A12: per-query top-k of a [Q, C] score matrix, descending, ties by ascending candidate index -- the
// order of Python's stable sorted(..., reverse=True) (src/evaluation/evaluate.py:76).
//
// Scores become 64-bit keys (monotone score bits << 32 | ~index) so one unsigned descending sort yields
// both rules.  One workgroup (256 threads) bitonic-sorts a chunk of N = 1024 or 4096 keys held E = N/256 per
// thread IN REGISTERS: compare-exchange partners at distance < E are in the same thread, at distance
// < 64*E in the same wave (DPP / v_permlane*_swap lane exchange, no barrier, no LDS), and only the levels at
// distance >= 64*E cross waves through LDS -- 3 exchanges for 1024 keys instead of 55 barriers.  The network
// is fully unrolled at compile time so every exchange distance is a constant.  Chunk winners are re-sorted by further
// passes until one chunk is left.
#include "common.h"
#include "topk_device.h"

namespace aspire {
namespace {

constexpr int kThreads = kTopkThreads;
constexpr int kMaxChunk = kTopkMaxChunk;

template <int E>
__global__ void __launch_bounds__(kThreads) topk_pass_kernel(const float* __restrict__ scores,
                                                             const uint64_t* __restrict__ keys_in, int64_t n_in,
                                                             int64_t in_stride, int64_t kk, uint64_t* __restrict__ keys_out,
                                                             int64_t out_stride, int64_t idx_base, int64_t k_final,
                                                             float* __restrict__ top_scores, int64_t* __restrict__ top_idx,
                                                             uint64_t* __restrict__ keys_final, int64_t in_k) {
    __shared__ uint64_t lds[E * kThreads];
    topk_block_pass<E>(blockIdx.x, blockIdx.y, gridDim.x, lds, scores, keys_in, n_in, in_stride, kk, keys_out, out_stride, idx_base,
                       k_final, top_scores, top_idx, keys_final, in_k);
}

int chunk_for(int64_t n) { return n <= 1024 ? 1024 : kMaxChunk; }

}  // namespace
}  // namespace aspire

using namespace aspire;

extern "C" size_t aspire_topk_workspace_bytes(int64_t Q, int64_t C, int64_t k) {
    if (Q <= 0 || C <= kMaxChunk) return 0;
    const int64_t kk = k < kMaxChunk ? k : kMaxChunk;
    const int64_t n1 = (C + kMaxChunk - 1) / kMaxChunk * kk;
    const int64_t c2 = chunk_for(n1);
    const int64_t n2 = (n1 + c2 - 1) / c2 * kk;
    return (size_t)(Q * (n1 + n2)) * sizeof(uint64_t);
}

namespace aspire {
int topk_run(const float* scores, int64_t Q, int64_t C, int64_t k, int64_t idx_base, float* top_scores, int64_t* top_idx,
             uint64_t* keys_final, void* workspace, size_t workspace_bytes, void* stream) {
    ASPIRE_REQUIRE(Q >= 0 && C >= 0 && k > 0, ASPIRE_ERR_INVALID_ARG, "bad shape Q=%lld C=%lld k=%lld", (long long)Q,
                   (long long)C, (long long)k);
    ASPIRE_REQUIRE(scores && ((top_scores && top_idx) || keys_final), ASPIRE_ERR_INVALID_ARG, "null pointer");
    ASPIRE_REQUIRE(C < (int64_t)0xFFFFFFFF, ASPIRE_ERR_UNSUPPORTED, "C too large for 32-bit local indices");
    ASPIRE_REQUIRE(!keys_final || (idx_base >= 0 && idx_base + C < (int64_t)0xFFFFFFFF), ASPIRE_ERR_UNSUPPORTED,
                   "global candidate indices must fit 32 bits in key form");
    ASPIRE_REQUIRE(C <= kMaxChunk || k < 1024, ASPIRE_ERR_UNSUPPORTED,
                   "k=%lld >= 1024 with C=%lld > %d: full sorts beyond one chunk are not built", (long long)k,
                   (long long)C, kMaxChunk);
    if (Q == 0) return ASPIRE_OK;
    ASPIRE_REQUIRE(workspace_bytes >= aspire_topk_workspace_bytes(Q, C, k), ASPIRE_ERR_INVALID_ARG,
                   "workspace too small: need %zu bytes", aspire_topk_workspace_bytes(Q, C, k));
    const int64_t kk = k < kMaxChunk ? k : kMaxChunk;
    int64_t n = C, in_stride = C;
    const float* sc = scores;
    const uint64_t* kin = nullptr;
    uint64_t* bufs[2];
    bufs[0] = (uint64_t*)workspace;
    bufs[1] = bufs[0] + (workspace ? Q * ((C + kMaxChunk - 1) / kMaxChunk * kk) : 0);
    int which = 0;
    for (;;) {
        const int chunk = chunk_for(n);
        const int64_t nch = n == 0 ? 1 : (n + chunk - 1) / chunk;
        const bool final_pass = nch == 1;
        const int64_t out_stride = nch * kk;
        dim3 grid((unsigned)Q, (unsigned)nch);
        float* ts = final_pass && !keys_final ? top_scores : nullptr;
        int64_t* ti = final_pass && !keys_final ? top_idx : nullptr;
        uint64_t* kf = final_pass ? keys_final : nullptr;
        uint64_t* ko = final_pass ? nullptr : bufs[which];
        if (chunk == 1024) {
            hipLaunchKernelGGL(topk_pass_kernel<4>, grid, dim3(kThreads), 0, (hipStream_t)stream, sc, kin, n, in_stride, kk, ko,
                               out_stride, idx_base, k, ts, ti, kf, (int64_t)0);
        } else {
            hipLaunchKernelGGL(topk_pass_kernel<16>, grid, dim3(kThreads), 0, (hipStream_t)stream, sc, kin, n, in_stride, kk, ko,
                               out_stride, idx_base, k, ts, ti, kf, (int64_t)0);
        }
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
}  // namespace aspire

extern "C" int aspire_topk_desc_f32(const float* scores, int64_t Q, int64_t C, int64_t k, int64_t idx_base,
                                    float* top_scores, int64_t* top_idx, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    ASPIRE_REQUIRE(top_scores && top_idx, ASPIRE_ERR_INVALID_ARG, "null pointer");
    return topk_run(scores, Q, C, k, idx_base, top_scores, top_idx, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int aspire_topk_keys_f32(const float* scores, int64_t Q, int64_t C, int64_t k, int64_t idx_base,
                                    uint64_t* keys, void* workspace, size_t workspace_bytes, void* stream) {
    ASPIRE_REQUIRE(keys, ASPIRE_ERR_INVALID_ARG, "null pointer");
    return topk_run(scores, Q, C, k, idx_base, nullptr, nullptr, keys, workspace, workspace_bytes, stream);
}

extern "C" int aspire_topk_merge_keys(const uint64_t* keys, int64_t R, int64_t Q, int64_t k_in, int64_t k, float* top_scores,
                                      int64_t* top_idx, void* stream) {
    ASPIRE_REQUIRE(R > 0 && Q >= 0 && k_in > 0 && k > 0, ASPIRE_ERR_INVALID_ARG, "bad shape R=%lld Q=%lld k_in=%lld k=%lld",
                   (long long)R, (long long)Q, (long long)k_in, (long long)k);
    ASPIRE_REQUIRE(keys && top_scores && top_idx, ASPIRE_ERR_INVALID_ARG, "null pointer");
    ASPIRE_REQUIRE(R * k_in <= kMaxChunk, ASPIRE_ERR_UNSUPPORTED, "R * k_in = %lld keys per query exceed one %d-key chunk",
                   (long long)(R * k_in), kMaxChunk);
    if (Q == 0) return ASPIRE_OK;
    const int64_t n = R * k_in;
    dim3 grid((unsigned)Q, 1);
    if (n <= 1024) {
        hipLaunchKernelGGL(topk_pass_kernel<4>, grid, dim3(kThreads), 0, (hipStream_t)stream, (const float*)nullptr, keys, n, n, k,
                           (uint64_t*)nullptr, k, (int64_t)0, k, top_scores, top_idx, (uint64_t*)nullptr, k_in);
    } else {
        hipLaunchKernelGGL(topk_pass_kernel<16>, grid, dim3(kThreads), 0, (hipStream_t)stream, (const float*)nullptr, keys, n, n, k,
                           (uint64_t*)nullptr, k, (int64_t)0, k, top_scores, top_idx, (uint64_t*)nullptr, k_in);
    }
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}
