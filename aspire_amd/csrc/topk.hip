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
                                                             uint64_t* __restrict__ keys_final, int64_t in_k,
                                                             const int32_t* __restrict__ seg_off, const int32_t* __restrict__ seg_base) {
    __shared__ uint64_t lds[E * kThreads];
    topk_block_pass<E>(blockIdx.x, blockIdx.y, gridDim.x, lds, scores, keys_in, n_in, in_stride, kk, keys_out, out_stride, idx_base,
                       k_final, top_scores, top_idx, keys_final, in_k, seg_off, seg_base);
}

// Small pools (n <= 1024) ranked for a short list (k <= 128): select, then sort only the survivors.  The scores' order
// bits are bucketed monotonically into 256 bins between their minimum and maximum, a suffix count finds the bin that
// holds the k-th largest, every key in that bin or above survives (usually k plus a handful), and a 256-key network
// (one key per thread, 36 levels) replaces the 1024-key one (four keys per thread, 55 levels).  Exact: the bucket map
// is monotone, ties and order are decided by the full 64-bit keys of the survivors.  Falls back to the full sort in
// the same launch when the boundary bin is crowded (more than 256 survivors, e.g. all scores equal).
// E keys per thread: 4 (chunks of 1024) or 16 (chunks of 4096; pools beyond 4096 candidates take one workgroup per chunk
// and leave k keys each for the next pass -- the full 4096-key network of topk_pass_kernel<16> cost 16-44 us there).
template <int E>
__global__ void __launch_bounds__(kThreads) topk_select_kernel(const float* __restrict__ scores, int64_t n_in, int64_t in_stride,
                                                               int64_t idx_base, int64_t k_final, int64_t kk,
                                                               uint64_t* __restrict__ keys_out, int64_t out_stride,
                                                               float* __restrict__ top_scores, int64_t* __restrict__ top_idx,
                                                               uint64_t* __restrict__ keys_final, const int32_t* __restrict__ seg_off,
                                                               const int32_t* __restrict__ seg_base) {
    __builtin_amdgcn_s_setprio(3);     // few workgroups, latency only: issue ahead of co-resident throughput kernels
    if (seg_off != nullptr) {          // segmented scores (batched jobs): query q owns scores[seg_off[q] .. seg_off[q + 1])
        scores += seg_off[blockIdx.x];
        n_in = seg_off[blockIdx.x + 1] - seg_off[blockIdx.x];
        in_stride = 0;
    }
    if (seg_base != nullptr) idx_base += seg_base[blockIdx.x];
    __shared__ uint64_t lds[E * kThreads];
    __shared__ unsigned hist[256];
    __shared__ unsigned sm[16];      // [0..3] wave minima, [4..7] wave maxima, [8] boundary bin, [9] survivors, [10] cursor
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t q = blockIdx.x, chunk = blockIdx.y;
    const int64_t base = chunk * (E * kThreads);
    uint64_t key[E];
    unsigned mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int64_t i = base + tid * E + r;
        uint64_t kv = 0;
        if (i < n_in) {
            kv = ((uint64_t)order_bits(scores[q * in_stride + i]) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)i);
            const unsigned hi = (unsigned)(kv >> 32);
            mn = min(mn, hi);
            mx = max(mx, hi);
        }
        key[r] = kv;
    }
    hist[tid] = 0;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        mn = min(mn, (unsigned)__shfl_xor((int)mn, m));
        mx = max(mx, (unsigned)__shfl_xor((int)mx, m));
    }
    if (lane == 0) { sm[wave] = mn; sm[4 + wave] = mx; }
    if (tid == 0) { sm[8] = 0; sm[9] = 0xFFFFFFFFu; sm[10] = 0; }
    __syncthreads();
    mn = min(min(sm[0], sm[1]), min(sm[2], sm[3]));
    mx = max(max(sm[4], sm[5]), max(sm[6], sm[7]));
    const float scale = 255.0f / (float)(mx - mn);            // mx == mn: inf -> every bucket NaN/0 -> handled below
    auto bucket_of = [&](uint64_t kv) {
        const float t = (float)((unsigned)(kv >> 32) - mn) * scale;
        return (int)fminf(t, 255.0f);
    };
    const bool spread = mx > mn;
    if (spread) {
#pragma unroll
        for (int r = 0; r < E; ++r)
            if (key[r] != 0) atomicAdd(&hist[bucket_of(key[r])], 1u);
    }
    __syncthreads();
    if (spread && wave == 0) {
        // lane l owns bins 4l .. 4l+3; suffix counts from the top bin down
        unsigned h[4] = {hist[4 * lane], hist[4 * lane + 1], hist[4 * lane + 2], hist[4 * lane + 3]};
        unsigned incl = h[0] + h[1] + h[2] + h[3];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {                     // inclusive suffix sum over lanes (lane 63 first)
            const unsigned o = (unsigned)__shfl_down((int)incl, m);
            if (lane + m < 64) incl += o;
        }
        unsigned above = incl - (h[0] + h[1] + h[2] + h[3]);  // keys in bins of higher lanes
        const unsigned kq = (unsigned)kk;
#pragma unroll
        for (int j = 3; j >= 0; --j) {
            if (above < kq && above + h[j] >= kq) { sm[8] = 4 * lane + j; sm[9] = above + h[j]; }
            above += h[j];
        }
        if (lane == 0 && above < kq) { sm[8] = 0; sm[9] = above; }   // fewer than k keys in all: everything survives
    }
    __syncthreads();
    // a chunk with at most 256 keys (a small pool, or the tail chunk of a big one): every key survives -- without this
    // a one-key tail chunk (min == max) took the full 4096-key sort: 1 x 4097 cost 40 us more than 1 x 4096
    const int64_t n_valid = min((int64_t)(E * kThreads), n_in - base);
    const bool take_all = n_valid <= kThreads;
    const unsigned n_surv = take_all ? (unsigned)max(n_valid, (int64_t)0) : sm[9];
    if (take_all || (spread && n_surv <= (unsigned)kThreads)) {
        const int bmin = (int)sm[8];
#pragma unroll
        for (int r = 0; r < E; ++r)
            if (key[r] != 0 && (take_all || bucket_of(key[r]) >= bmin)) lds[atomicAdd(&sm[10], 1u)] = key[r];
        __syncthreads();
        uint64_t one[1] = {tid < (int)n_surv ? lds[tid] : 0ull};
        __syncthreads();                                      // the sort's exchanges reuse lds
        bitonic_sort<1, kThreads>(one, lds, tid);
        topk_emit<1>(one, q, chunk, kk, keys_out, out_stride, idx_base, k_final, top_scores, top_idx, keys_final, 0);
    } else {
        block_bitonic_desc<E>(key, lds, tid);
        topk_emit<E>(key, q, chunk, kk, keys_out, out_stride, idx_base, k_final, top_scores, top_idx, keys_final, 0);
    }
}

int chunk_for(int64_t n) { return n <= 1024 ? 1024 : kMaxChunk; }

// ---------------------------------------------------------------------------------------------
// Full stable sort of pools beyond one 4096-key chunk (k >= 1024: evaluate.py:76 and pp_gen_nearest.py:266,339 sort the
// WHOLE pool).  Pass 0 sorts 4096-key chunks with the register-resident network above; every further pass merges pairs of
// sorted runs: workgroup (query, tile) produces 4096 consecutive keys of a merged run -- a merge-path split (two
// binary searches over the runs) tells it which pieces of the two runs those are, the pieces are loaded as one bitonic
// sequence (run A's piece descending, zero padding, run B's piece reversed) and ONE bitonic merge (12 levels of the same
// unrolled network) sorts them.  Keys are unique (the index is part of the key), so the order is exactly Python's stable
// sort; zero pads (smaller than any real key) only ever sit at the tail of the last run.  With k < C a run is cut to the
// first `keep` = k rounded up to whole tiles keys after every pass.
// ---------------------------------------------------------------------------------------------
template <int E>
__global__ void __launch_bounds__(kThreads) topk_merge_pass_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                                   int64_t stride, int64_t run, int64_t keep, int tiles_per_pair) {
    constexpr int N = E * kThreads;
    __shared__ uint64_t lds[N];
    __shared__ int64_t split[2];
    const int tid = threadIdx.x;
    const int64_t q = blockIdx.x;
    const int64_t pair = blockIdx.y / tiles_per_pair, t = blockIdx.y % tiles_per_pair;
    const int64_t a_base = pair * 2 * run, b_base = a_base + run;
    int64_t len_a = stride - a_base, len_b = stride - b_base;
    len_a = len_a < 0 ? 0 : len_a < run ? len_a : run;
    len_b = len_b < 0 ? 0 : len_b < run ? len_b : run;
    len_a = len_a < keep ? len_a : keep;
    len_b = len_b < keep ? len_b : keep;
    const int64_t total = len_a + len_b < keep ? len_a + len_b : keep;
    const int64_t d0 = t * N;
    if (d0 >= total) return;
    const int64_t d1 = d0 + N < total ? d0 + N : total;
    const uint64_t* A = in + q * stride + a_base;
    const uint64_t* B = in + q * stride + b_base;
    if (tid == 0 || tid == 64) {
        // number of A keys among the first d keys of the merged run: the smallest a with A[a] <= B[d - 1 - a]
        const int64_t d = tid == 0 ? d0 : d1;
        int64_t lo = d - len_b > 0 ? d - len_b : 0, hi = d < len_a ? d : len_a;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (A[mid] > B[d - 1 - mid]) lo = mid + 1;
            else hi = mid;
        }
        split[tid == 0 ? 0 : 1] = lo;
    }
    __syncthreads();
    const int64_t a0 = split[0], a1 = split[1];
    const int64_t b0 = d0 - a0;
    const int na = (int)(a1 - a0), nb = (int)((d1 - d0) - (a1 - a0));
    uint64_t key[E];
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int i = tid * E + r;
        uint64_t kv = 0;
        if (i < na) kv = A[a0 + i];
        else if (i >= N - nb) kv = B[b0 + (N - 1 - i)];
        key[r] = kv;
    }
    bitonic_merge<E, N, N / 2>(key, lds, tid);
    uint64_t* o = out + q * stride + a_base + d0;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int i = tid * E + r;
        if (i < d1 - d0) o[i] = key[r];
    }
}

// The first k keys of every query's sorted buffer -> the final outputs (see topk_emit).
__global__ void __launch_bounds__(256) topk_emit_sorted_kernel(const uint64_t* __restrict__ keys, int64_t stride, int64_t k,
                                                               int64_t idx_base, float* __restrict__ top_scores,
                                                               int64_t* __restrict__ top_idx, uint64_t* __restrict__ keys_final,
                                                               const int32_t* __restrict__ seg_base) {
    const int64_t q = blockIdx.y;
    if (seg_base != nullptr) idx_base += seg_base[q];
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= k) return;
    const uint64_t kv = t < stride ? keys[q * stride + t] : 0ull;
    const bool real = kv != 0;
    const int64_t gidx = idx_base + (int64_t)(0xFFFFFFFFu - (uint32_t)kv);
    if (keys_final != nullptr) {
        keys_final[q * k + t] = real ? ((kv & 0xFFFFFFFF00000000ull) | (uint64_t)(0xFFFFFFFFu - (uint32_t)gidx)) : 0ull;
    } else {
        top_scores[q * k + t] = real ? unorder_bits((uint32_t)(kv >> 32)) : -INFINITY;
        top_idx[q * k + t] = real ? gidx : -1;
    }
}

bool full_sort_path(int64_t C, int64_t k) { return C > kMaxChunk && k >= 1024; }

}  // namespace
}  // namespace aspire

using namespace aspire;

extern "C" size_t aspire_topk_workspace_bytes(int64_t Q, int64_t C, int64_t k) {
    if (Q <= 0 || C <= kMaxChunk) return 0;
    if (full_sort_path(C, k)) {
        const int64_t stride = (C + kMaxChunk - 1) / kMaxChunk * kMaxChunk;
        return (size_t)(2 * Q * stride) * sizeof(uint64_t);
    }
    const int64_t kk = k < kMaxChunk ? k : kMaxChunk;
    const int64_t n1 = (C + kMaxChunk - 1) / kMaxChunk * kk;
    const int64_t c2 = chunk_for(n1);
    const int64_t n2 = (n1 + c2 - 1) / c2 * kk;
    return (size_t)(Q * (n1 + n2)) * sizeof(uint64_t);
}

namespace aspire {
int topk_run(const float* scores, int64_t Q, int64_t C, int64_t k, int64_t idx_base, float* top_scores, int64_t* top_idx,
             uint64_t* keys_final, void* workspace, size_t workspace_bytes, void* stream, const int32_t* seg_off,
             const int32_t* seg_base) {
    ASPIRE_REQUIRE(Q >= 0 && C >= 0 && k > 0, ASPIRE_ERR_INVALID_ARG, "bad shape Q=%lld C=%lld k=%lld", (long long)Q,
                   (long long)C, (long long)k);
    ASPIRE_REQUIRE(scores && ((top_scores && top_idx) || keys_final), ASPIRE_ERR_INVALID_ARG, "null pointer");
    ASPIRE_REQUIRE(C < (int64_t)0xFFFFFFFF, ASPIRE_ERR_UNSUPPORTED, "C too large for 32-bit local indices");
    ASPIRE_REQUIRE(!keys_final || (idx_base >= 0 && idx_base + C < (int64_t)0xFFFFFFFF), ASPIRE_ERR_UNSUPPORTED,
                   "global candidate indices must fit 32 bits in key form");
    if (Q == 0) return ASPIRE_OK;
    ASPIRE_REQUIRE(workspace_bytes >= aspire_topk_workspace_bytes(Q, C, k), ASPIRE_ERR_INVALID_ARG,
                   "workspace too small: need %zu bytes", aspire_topk_workspace_bytes(Q, C, k));
    ASPIRE_REQUIRE(workspace || aspire_topk_workspace_bytes(Q, C, k) == 0, ASPIRE_ERR_INVALID_ARG, "null workspace");
    hipStream_t st = (hipStream_t)stream;
    if (full_sort_path(C, k)) {
        // sorted 4096-key chunks, then log2(chunks) merge passes between two buffers
        const int64_t nch = (C + kMaxChunk - 1) / kMaxChunk, stride = nch * kMaxChunk;
        uint64_t* buf[2] = {(uint64_t*)workspace, (uint64_t*)workspace + Q * stride};
        hipLaunchKernelGGL(topk_pass_kernel<16>, dim3((unsigned)Q, (unsigned)nch), dim3(kThreads), 0, st, scores,
                           (const uint64_t*)nullptr, C, C, (int64_t)kMaxChunk, buf[0], stride, (int64_t)0, k, (float*)nullptr,
                           (int64_t*)nullptr, (uint64_t*)nullptr, (int64_t)0, seg_off, (const int32_t*)nullptr);
        ASPIRE_LAUNCH_OK();
        const int64_t keep = k >= stride ? stride : (k + kMaxChunk - 1) / kMaxChunk * kMaxChunk;
        int which = 0;
        for (int64_t run = kMaxChunk; run < stride; run *= 2) {
            const int64_t pairs = (stride + 2 * run - 1) / (2 * run);
            const int64_t merged = 2 * run < keep ? 2 * run : keep;
            const int tpp = (int)((merged + kMaxChunk - 1) / kMaxChunk);
            ASPIRE_REQUIRE(pairs * tpp <= 65535, ASPIRE_ERR_UNSUPPORTED, "pool of %lld candidates is too large for the full sort",
                           (long long)C);
            hipLaunchKernelGGL(topk_merge_pass_kernel<16>, dim3((unsigned)Q, (unsigned)(pairs * tpp)), dim3(kThreads), 0, st,
                               buf[which], buf[which ^ 1], stride, run, keep, tpp);
            ASPIRE_LAUNCH_OK();
            which ^= 1;
        }
        hipLaunchKernelGGL(topk_emit_sorted_kernel, dim3((unsigned)((k + 255) / 256), (unsigned)Q), dim3(256), 0, st, buf[which],
                           stride, k, idx_base, top_scores, top_idx, keys_final, seg_base);
        ASPIRE_LAUNCH_OK();
        return ASPIRE_OK;
    }
    const int64_t kk = k < kMaxChunk ? k : kMaxChunk;
    int64_t n = C, in_stride = C;
    const float* sc = scores;
    const uint64_t* kin = nullptr;
    uint64_t* bufs[2];
    bufs[0] = (uint64_t*)workspace;
    bufs[1] = bufs[0] + (workspace ? Q * ((C + kMaxChunk - 1) / kMaxChunk * kk) : 0);
    int which = 0;
    for (;;) {
        const bool select = sc != nullptr && k <= 128;       // first pass over scores, short list: select + 256-key sort per chunk
        const int chunk = select && n > 1024 && n <= 2048 ? 2048 : chunk_for(n);
        const int64_t nch = n == 0 ? 1 : (n + chunk - 1) / chunk;
        const bool final_pass = nch == 1;
        const int64_t out_stride = nch * kk;
        dim3 grid((unsigned)Q, (unsigned)nch);
        float* ts = final_pass && !keys_final ? top_scores : nullptr;
        int64_t* ti = final_pass && !keys_final ? top_idx : nullptr;
        uint64_t* kf = final_pass ? keys_final : nullptr;
        uint64_t* ko = final_pass ? nullptr : bufs[which];
        const int32_t* seg = sc != nullptr ? seg_off : nullptr;      // only the pass over the scores is segmented
        const int32_t* sb = final_pass ? seg_base : nullptr;          // the final outputs carry the global indices
        if (select) {
            // (final outputs if there is one chunk)
            if (chunk == 2048)
                hipLaunchKernelGGL(topk_select_kernel<8>, grid, dim3(kThreads), 0, st, sc, n, in_stride, idx_base, k, kk,
                                   ko, out_stride, ts, ti, kf, seg, sb);
            else if (chunk == 1024)
                hipLaunchKernelGGL(topk_select_kernel<4>, grid, dim3(kThreads), 0, st, sc, n, in_stride, idx_base, k, kk,
                                   ko, out_stride, ts, ti, kf, seg, sb);
            else
                hipLaunchKernelGGL(topk_select_kernel<16>, grid, dim3(kThreads), 0, st, sc, n, in_stride, idx_base, k, kk,
                                   ko, out_stride, ts, ti, kf, seg, sb);
        } else if (chunk == 1024) {
            hipLaunchKernelGGL(topk_pass_kernel<4>, grid, dim3(kThreads), 0, st, sc, kin, n, in_stride, kk, ko,
                               out_stride, idx_base, k, ts, ti, kf, (int64_t)0, seg, sb);
        } else {
            hipLaunchKernelGGL(topk_pass_kernel<16>, grid, dim3(kThreads), 0, st, sc, kin, n, in_stride, kk, ko,
                               out_stride, idx_base, k, ts, ti, kf, (int64_t)0, seg, sb);
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
                           (uint64_t*)nullptr, k, (int64_t)0, k, top_scores, top_idx, (uint64_t*)nullptr, k_in, (const int32_t*)nullptr,
                           (const int32_t*)nullptr);
    } else {
        hipLaunchKernelGGL(topk_pass_kernel<16>, grid, dim3(kThreads), 0, (hipStream_t)stream, (const float*)nullptr, keys, n, n, k,
                           (uint64_t*)nullptr, k, (int64_t)0, k, top_scores, top_idx, (uint64_t*)nullptr, k_in, (const int32_t*)nullptr,
                           (const int32_t*)nullptr);
    }
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}
