// Device side of A12 (per-query top-k, see topk.hip), shared with the fused score + rank tail of score.hip:
// a register-resident bitonic network over 64-bit keys, one workgroup of 256 threads per (query, chunk).
#pragma once
#include "common.h"

namespace aspire {

constexpr int kTopkThreads = 256;
constexpr int kTopkMaxChunk = 4096;

__device__ __forceinline__ uint32_t order_bits(float f) {
    // -0.0 and +0.0 compare equal in the reference's sort (a tie, decided by pool order): give them one key
    const uint32_t u = __builtin_bit_cast(uint32_t, f + 0.0f);
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
    bitonic_sort<E, E * kTopkThreads>(key, lds, tid);
}

// Output of a sorted block (thread t holds ranks t*E .. t*E+E-1): chunk winners as keys, or the final
// (top_scores, top_idx) / key-form outputs.
template <int E>
__device__ __forceinline__ void topk_emit(const uint64_t (&key)[E], int64_t q, int64_t chunk, int64_t kk, uint64_t* __restrict__ keys_out,
                                          int64_t out_stride, int64_t idx_base, int64_t k_final, float* __restrict__ top_scores,
                                          int64_t* __restrict__ top_idx, uint64_t* __restrict__ keys_final, int64_t in_k) {
    constexpr int N = E * kTopkThreads;
    const int tid = threadIdx.x;
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
        for (int64_t t = N + tid; t < k_final; t += kTopkThreads) keys_final[q * k_final + t] = 0ull;
    } else if (top_scores != nullptr) {
        // k_final beyond the chunk (C < k): the tail is (-inf, -1)
        for (int64_t t = N + tid; t < k_final; t += kTopkThreads) {
            top_scores[q * k_final + t] = -INFINITY;
            top_idx[q * k_final + t] = -1;
        }
    }
}

// in: either scores (first pass; index = position) or keys.  n_in per query; chunk c covers
// [c*N, (c+1)*N).  Writes kk = min(k, N) keys per chunk, or the final outputs.
template <int E>
__device__ __forceinline__ void topk_block_pass(int64_t q, int64_t chunk, int64_t n_queries, uint64_t* lds, const float* __restrict__ scores,
                                                             const uint64_t* __restrict__ keys_in, int64_t n_in,
                                                             int64_t in_stride, int64_t kk, uint64_t* __restrict__ keys_out,
                                                             int64_t out_stride, int64_t idx_base, int64_t k_final,
                                                             float* __restrict__ top_scores, int64_t* __restrict__ top_idx,
                                                             uint64_t* __restrict__ keys_final, int64_t in_k,
                                                             const int32_t* __restrict__ seg_off = nullptr,
                                                             const int32_t* __restrict__ seg_base = nullptr) {
    constexpr int N = E * kTopkThreads;
    const int tid = threadIdx.x;
    const int64_t base = chunk * N;
    if (seg_off != nullptr) {      // segmented scores (batched jobs): query q owns scores[seg_off[q] .. seg_off[q + 1])
        scores += seg_off[q];
        n_in = seg_off[q + 1] - seg_off[q];
        in_stride = 0;
    }
    if (seg_base != nullptr) idx_base += seg_base[q];     // the job's first global candidate index (sharded pools)
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
                kv = keys_in[(r * n_queries + q) * in_k + (i - r * in_k)];
            } else {
                kv = keys_in[q * in_stride + i];
            }
        }
        key[r] = kv;
    }
    block_bitonic_desc<E>(key, lds, tid);
    topk_emit<E>(key, q, chunk, kk, keys_out, out_stride, idx_base, k_final, top_scores, top_idx, keys_final, in_k);
}

// host driver of the multi-pass rank (topk.hip); exactly one of (top_scores, top_idx) / keys_final is produced
// seg_off != NULL (device int32 [Q + 1]): query q's scores are scores[seg_off[q] .. seg_off[q + 1]) and C is a host-known
// upper bound of a segment's length.
// seg_base (device int32 [Q], optional): added to idx_base per query.
int topk_run(const float* scores, int64_t Q, int64_t C, int64_t k, int64_t idx_base, float* top_scores, int64_t* top_idx,
             uint64_t* keys_final, void* workspace, size_t workspace_bytes, void* stream, const int32_t* seg_off = nullptr,
             const int32_t* seg_base = nullptr);

}  // namespace aspire
