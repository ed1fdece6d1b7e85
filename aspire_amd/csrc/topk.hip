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

namespace aspire {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxChunk = 4096;

__device__ __forceinline__ uint32_t order_bits(float f) {
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unorder_bits(uint32_t u) {
    return __builtin_bit_cast(float, (u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
}

template <int M>
__device__ __forceinline__ uint64_t lane_xor_u64(uint64_t v) {
    // DPP / v_permlane*_swap forms (common.h): a few cycles each, where ds_bpermute costs ~150 per dependent hop
    const float lo = lane_xor<M>(__builtin_bit_cast(float, (uint32_t)v));
    const float hi = lane_xor<M>(__builtin_bit_cast(float, (uint32_t)(v >> 32)));
    return ((uint64_t)__builtin_bit_cast(uint32_t, hi) << 32) | __builtin_bit_cast(uint32_t, lo);
}

__device__ __forceinline__ uint64_t pick(uint64_t a, uint64_t b, bool take_max) {
    return take_max ? (a > b ? a : b) : (a < b ? a : b);
}

// One compare-exchange level of the bitonic network (partner = idx ^ STRIDE, direction from idx & SIZE).
// Element index of register r of thread t: idx = t * E + r.
template <int E, int SIZE, int STRIDE>
__device__ __forceinline__ void bitonic_level(uint64_t (&key)[E], uint64_t* lds, int tid) {
    if constexpr (STRIDE >= 64 * E) {
        // partner lives in another wave: exchange through LDS
        __syncthreads();
#pragma unroll
        for (int r = 0; r < E; ++r) lds[tid * E + r] = key[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const int idx = tid * E + r;
            key[r] = pick(key[r], lds[idx ^ STRIDE], ((idx & SIZE) == 0) == ((idx & STRIDE) == 0));
        }
    } else if constexpr (STRIDE >= E) {
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const int idx = tid * E + r;
            key[r] = pick(key[r], lane_xor_u64<STRIDE / E>(key[r]), ((idx & SIZE) == 0) == ((idx & STRIDE) == 0));
        }
    } else {
#pragma unroll
        for (int r = 0; r < E; ++r) {
            if ((r & STRIDE) == 0) {
                const int idx = tid * E + r;
                const bool desc = (idx & SIZE) == 0;
                const uint64_t a = key[r], b = key[r | STRIDE];
                const bool swap = desc ? (a < b) : (a > b);
                key[r] = swap ? b : a;
                key[r | STRIDE] = swap ? a : b;
            }
        }
    }
}

template <int E, int SIZE, int STRIDE>
__device__ __forceinline__ void bitonic_merge(uint64_t (&key)[E], uint64_t* lds, int tid) {
    bitonic_level<E, SIZE, STRIDE>(key, lds, tid);
    if constexpr (STRIDE > 1) bitonic_merge<E, SIZE, STRIDE / 2>(key, lds, tid);
}

template <int E, int SIZE>
__device__ __forceinline__ void bitonic_sort(uint64_t (&key)[E], uint64_t* lds, int tid) {
    if constexpr (SIZE > 2) bitonic_sort<E, SIZE / 2>(key, lds, tid);
    bitonic_merge<E, SIZE, SIZE / 2>(key, lds, tid);
}

// Sorts the block's E * 256 keys descending (fully unrolled network: every exchange distance is a constant).
template <int E>
__device__ __forceinline__ void block_bitonic_desc(uint64_t (&key)[E], uint64_t* lds, int tid) {
    bitonic_sort<E, E * kThreads>(key, lds, tid);
}

// in: either scores (first pass; index = position) or keys.  n_in per query; chunk c covers
// [c*N, (c+1)*N).  Writes kk = min(k, N) keys per chunk, or the final outputs.
template <int E>
__global__ void __launch_bounds__(kThreads) topk_pass_kernel(const float* __restrict__ scores,
                                                             const uint64_t* __restrict__ keys_in, int64_t n_in,
                                                             int64_t in_stride, int64_t kk, uint64_t* __restrict__ keys_out,
                                                             int64_t out_stride, int64_t idx_base, int64_t k_final,
                                                             float* __restrict__ top_scores, int64_t* __restrict__ top_idx,
                                                             uint64_t* __restrict__ keys_final, int64_t in_k) {
    constexpr int N = E * kThreads;
    __shared__ uint64_t lds[N];
    const int tid = threadIdx.x;
    const int64_t q = blockIdx.x, chunk = blockIdx.y;
    const int64_t base = chunk * N;
    uint64_t key[E];
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int64_t i = base + tid * E + r;
        uint64_t kv = 0;  // pad: below every real key
        if (i < n_in) {
            if (scores) {
                kv = ((uint64_t)order_bits(scores[q * in_stride + i]) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)i);
            } else if (in_k > 0) {
                // keys gathered from R ranks, laid out [rank][query][in_k]: element i of query q = (i / in_k, i % in_k)
                const int64_t r = i / in_k;
                kv = keys_in[(r * gridDim.x + q) * in_k + (i - r * in_k)];
            } else {
                kv = keys_in[q * in_stride + i];
            }
        }
        key[r] = kv;
    }
    block_bitonic_desc<E>(key, lds, tid);
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int t = tid * E + r;
        const uint64_t kv = key[r];
        if (keys_final != nullptr) {
            // final pass, key output: the low word carries the GLOBAL index (idx_base + position), so keys of
            // different shards compare like (score desc, global index asc); 0 = padding (C < k)
            if (t < k_final) {
                const uint32_t low = (uint32_t)kv;
                const uint32_t gl = in_k > 0 ? low : 0xFFFFFFFFu - (uint32_t)(idx_base + (int64_t)(0xFFFFFFFFu - low));
                keys_final[q * k_final + t] = kv != 0 ? ((kv & 0xFFFFFFFF00000000ull) | gl) : 0ull;
            }
        } else if (top_scores == nullptr) {
            if (t < kk) keys_out[q * out_stride + chunk * kk + t] = kv;
        } else if (t < k_final) {
            const bool real = kv != 0;
            top_scores[q * k_final + t] = real ? unorder_bits((uint32_t)(kv >> 32)) : -INFINITY;
            top_idx[q * k_final + t] = real ? idx_base + (int64_t)(0xFFFFFFFFu - (uint32_t)kv) : -1;
        }
    }
    if (keys_final != nullptr) {
        for (int64_t t = N + tid; t < k_final; t += kThreads) keys_final[q * k_final + t] = 0ull;
    } else if (top_scores != nullptr) {
        // k_final beyond the chunk (C < k): the tail is (-inf, -1)
        for (int64_t t = N + tid; t < k_final; t += kThreads) {
            top_scores[q * k_final + t] = -INFINITY;
            top_idx[q * k_final + t] = -1;
        }
    }
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

namespace {
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
}  // namespace

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
