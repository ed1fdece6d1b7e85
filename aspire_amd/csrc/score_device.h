// Small device helpers shared by the scoring kernels (score.hip, fused.hip).
#pragma once
#include "score_types.h"

namespace aspire {

// (v_pk_mul_f32 + v_pk_fma_f32 + add -- three issue slots instead of four -- measured no faster than this chain, alone
// or overlapped: 104-105 vs 106-108 M alignments/s in bench.py.)
__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}
__device__ __forceinline__ float sq4(const float4& a) { return dot4(a, a); }

// Per-coordinate box of a document's rows at one 16-byte chunk (doc = first row + the lane's chunk).  Eight rows per memory
// round trip (rows past the end repeat the last one): one row per trip made these latency-bound prologue kernels wait for
// eight dependent loads per 8-sentence document.
__device__ __forceinline__ void doc_box_chunk(const float* doc, int n, float4& mn, float4& mx) {
    mn = mx = ld4(doc);
    for (int r0 = 0; r0 < n; r0 += 8) {
        float4 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = ld4(doc + (size_t)min(r0 + r, n - 1) * kD);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            mn.x = fminf(mn.x, v[r].x); mn.y = fminf(mn.y, v[r].y); mn.z = fminf(mn.z, v[r].z); mn.w = fminf(mn.w, v[r].w);
            mx.x = fmaxf(mx.x, v[r].x); mx.y = fmaxf(mx.y, v[r].y); mx.z = fmaxf(mx.z, v[r].z); mx.w = fmaxf(mx.w, v[r].w);
        }
    }
}

// x / e with e's reciprocal r: one Newton step makes the quotient correctly rounded in all but
// pathological cases (what the v_div_* sequence does, minus its denormal scaling).
__device__ __forceinline__ float div_r(float x, float e, float r) {
    const float q = x * r;
    return fmaf(fmaf(-q, e, x), r, q);
}
__device__ __forceinline__ float rcp_refined(float e) {
    float r = __builtin_amdgcn_rcpf(e);
    return fmaf(fmaf(-e, r, 1.0f), r, r);
}

// Masked entries carry this instead of -inf so that fully masked (pad) lanes never form inf - inf.
constexpr float kNegBig = -1.0e30f;
constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;
// exp / log on the hardware transcendentals: v_exp_f32 / v_log_f32 are base 2, ~1 ulp.  The arguments
// met here are <= 0 (or within a few units of 0) for exp and in [2^-100, 2^100] for log: no denormal
// or range handling is needed, which is what makes libm's logf 12 instructions instead of 2.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * kLn2; }

// Length of geomloss's annealing schedule, n_mid = ceil((log blur - log diam) / log scaling) in float64 (numpy's
// arange) -- a discontinuous function of the diameter, so it has to be the float64 value.  The float64 logarithm of
// the diameter cost ~0.5 us of every solve's prologue: here the quotient is formed in fp32 (v_log_f32; the logs of blur
// and scaling come from the host) with a bound on its error, and only a quotient that close to an integer (a few
// pairs in 10^4) is redone in float64.
__device__ __forceinline__ int schedule_mid_steps(const ScoreArgs& a, float diam, float& log2_diam) {
    log2_diam = __builtin_amdgcn_logf(diam);
    const float x = (a.log2_blur - log2_diam) / a.log2_scaling;
    const float err = (fabsf(a.log2_blur) + fabsf(log2_diam) + 1.f) * 3e-7f / fabsf(a.log2_scaling) + fabsf(x) * 2e-7f;
    int n_mid;
    if (__builtin_expect(fabsf(x - rintf(x)) < 8.f * err, 0))
        n_mid = (int)ceil((a.log_blur - log((double)diam)) / a.log_scaling);
    else
        n_mid = (int)ceilf(x);
    return n_mid < 0 ? 0 : n_mid;
}

}  // namespace aspire
