// A2 + A3: CLS read-out and span mean pooling.
// Reference: AspireConSent.consent_reps_bert, examples/ex_aspire_consent.py:75-100 -- one full
// [B, L, 768] mask-multiply-sum pass per sentence slot.  Here every token row is read at most once:
// one workgroup of 192 threads per (document, sentence slot); thread t owns the float4 at d = 4t, so a
// token row is a single coalesced 3 KB read; the slot's rows are summed in index order.
#include "common.h"

namespace aspire {
namespace {

__global__ void __launch_bounds__(192) span_mean_pool_kernel(const float* __restrict__ hidden, int64_t L,
                                                             const int32_t* __restrict__ tok_idx,
                                                             const int32_t* __restrict__ span_off, int64_t S,
                                                             float* __restrict__ sent_reps,
                                                             float* __restrict__ cls_reps,
                                                             const int32_t* __restrict__ out_row) {
    const int64_t slot = blockIdx.x;          // b * S + s
    const int64_t b = slot / S;
    const int d = threadIdx.x * 4;
    // out_row (rep-store form): slot (b, s) is row out_row[slot] of a rows + CSR store, < 0 = the document has no
    // sentence s (nothing is written: the store holds no padding rows)
    const int64_t orow = out_row != nullptr ? (int64_t)out_row[slot] : slot;
    if (cls_reps != nullptr && slot % S == 0) {
        *reinterpret_cast<float4*>(cls_reps + (size_t)b * kD + d) =
            *reinterpret_cast<const float4*>(hidden + (size_t)b * L * kD + d);
    }
    if (orow < 0) return;
    const float* doc = hidden + (size_t)b * L * kD;
    const int lo = span_off[slot], hi = span_off[slot + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = lo;
    // 4 rows in flight per thread to cover HBM latency.
    for (; k + 4 <= hi; k += 4) {
        const float4 v0 = ld4_stream(doc + (size_t)tok_idx[k] * kD + d);
        const float4 v1 = ld4_stream(doc + (size_t)tok_idx[k + 1] * kD + d);
        const float4 v2 = ld4_stream(doc + (size_t)tok_idx[k + 2] * kD + d);
        const float4 v3 = ld4_stream(doc + (size_t)tok_idx[k + 3] * kD + d);
        acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
        acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
        acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
        acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
    }
    for (; k < hi; ++k) {
        const float4 v = ld4_stream(doc + (size_t)tok_idx[k] * kD + d);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    // torch.count_nonzero(mask).clamp(min=1): an empty slot stays exactly zero.
    const float cnt = (float)max(hi - lo, 1);
    acc.x /= cnt; acc.y /= cnt; acc.z /= cnt; acc.w /= cnt;
    *reinterpret_cast<float4*>(sent_reps + (size_t)orow * kD + d) = acc;
}

// caching_score's document-level term: ||q_cls - c_cls + eps||_2 (torch.nn.functional.pairwise_distance), one wave per
// pair, 12 coordinates per lane.
__global__ void __launch_bounds__(256) cls_l2_kernel(const float* __restrict__ q_cls, int64_t Q, const float* __restrict__ c_cls,
                                                     int64_t C, int paired, float eps, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t P = paired ? C : Q * C;
    if (pair >= P) return;
    const int64_t qi = paired ? pair : pair / C, ci = paired ? pair : pair - qi * C;
    const float* x = q_cls + (size_t)qi * kD + lane * 4;
    const float* y = c_cls + (size_t)ci * kD + lane * 4;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float4 u = *reinterpret_cast<const float4*>(x + 256 * k), v = *reinterpret_cast<const float4*>(y + 256 * k);
        const float d0 = (u.x - v.x) + eps, d1 = (u.y - v.y) + eps, d2 = (u.z - v.z) + eps, d3 = (u.w - v.w) + eps;
        acc = fmaf(d3, d3, fmaf(d2, d2, fmaf(d1, d1, fmaf(d0, d0, acc))));
    }
    acc = wave_sum(acc);
    if (lane == 0) out[pair] = sqrtf(acc);
}

}  // namespace
}  // namespace aspire

using namespace aspire;

extern "C" int aspire_cls_l2_f32(const float* q_cls, int64_t Q, const float* c_cls, int64_t C, int64_t D, int pairing, double eps,
                                 float* dist, void* stream) {
    ASPIRE_REQUIRE(D == kD, ASPIRE_ERR_UNSUPPORTED, "encoding dim %lld unsupported (kernels are built for 768)", (long long)D);
    ASPIRE_REQUIRE(pairing == ASPIRE_PAIR_CROSS || pairing == ASPIRE_PAIR_PAIRED, ASPIRE_ERR_INVALID_ARG, "bad pairing %d", pairing);
    ASPIRE_REQUIRE(Q >= 0 && C >= 0 && (pairing != ASPIRE_PAIR_PAIRED || Q == C), ASPIRE_ERR_INVALID_ARG,
                   "paired distances need equal batch sizes (query %lld vs cand %lld)", (long long)Q, (long long)C);
    const int64_t P = pairing == ASPIRE_PAIR_PAIRED ? C : Q * C;
    if (P == 0) return ASPIRE_OK;
    ASPIRE_REQUIRE(q_cls && c_cls && dist, ASPIRE_ERR_INVALID_ARG, "null pointer");
    hipLaunchKernelGGL(cls_l2_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, (hipStream_t)stream, q_cls, Q, c_cls, C,
                       pairing == ASPIRE_PAIR_PAIRED ? 1 : 0, (float)eps, dist);
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}

extern "C" int aspire_span_mean_pool_f32(const float* hidden, int64_t B, int64_t L, int64_t D, const int32_t* tok_idx,
                                         const int32_t* span_off, int64_t S, float* sent_reps, float* cls_reps,
                                         void* stream) {
    ASPIRE_REQUIRE(D == kD, ASPIRE_ERR_UNSUPPORTED, "encoding dim %lld unsupported (kernels are built for 768)",
                   (long long)D);
    ASPIRE_REQUIRE(B >= 0 && L > 0 && S > 0, ASPIRE_ERR_INVALID_ARG, "bad shape B=%lld L=%lld S=%lld", (long long)B,
                   (long long)L, (long long)S);
    ASPIRE_REQUIRE(hidden && span_off && sent_reps, ASPIRE_ERR_INVALID_ARG, "null pointer");
    if (B == 0) return ASPIRE_OK;
    hipLaunchKernelGGL(span_mean_pool_kernel, dim3((unsigned)(B * S)), dim3(192), 0, (hipStream_t)stream, hidden, L,
                       tok_idx, span_off, S, sent_reps, cls_reps, (const int32_t*)nullptr);
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}

extern "C" int aspire_span_mean_pool_rows_f32(const float* hidden, int64_t B, int64_t L, int64_t D, const int32_t* tok_idx,
                                              const int32_t* span_off, int64_t S, const int32_t* out_row, float* rows,
                                              float* cls_reps, void* stream) {
    ASPIRE_REQUIRE(D == kD, ASPIRE_ERR_UNSUPPORTED, "encoding dim %lld unsupported (kernels are built for 768)",
                   (long long)D);
    ASPIRE_REQUIRE(B >= 0 && L > 0 && S > 0, ASPIRE_ERR_INVALID_ARG, "bad shape B=%lld L=%lld S=%lld", (long long)B,
                   (long long)L, (long long)S);
    ASPIRE_REQUIRE(hidden && span_off && out_row && rows, ASPIRE_ERR_INVALID_ARG, "null pointer");
    if (B == 0) return ASPIRE_OK;
    hipLaunchKernelGGL(span_mean_pool_kernel, dim3((unsigned)(B * S)), dim3(192), 0, (hipStream_t)stream, hidden, L,
                       tok_idx, span_off, S, rows, cls_reps, out_row);
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}
