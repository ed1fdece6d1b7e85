// The fused otAspire kernels' shared device code (fused.hip, split.hip): the staging geometry of an item (four candidates of one
// query, 64 coordinates per stage) and the Sinkhorn solve of a wave's four pairs in the 16-lanes-per-pair layout (A7 / A8;
// reference arithmetic: src/learning/facetid_models/pair_distances.py:57-92 + geomloss 0.2.4's sinkhorn_loop, restated).
#pragma once
#include <math.h>

#include "common.h"
#include "score_device.h"
#include "score_types.h"

namespace aspire {
namespace {

constexpr int kCh = 16;                                  // 16-byte chunks per row per stage (64 coordinates)
constexpr int kStages = kD / (4 * kCh);                  // 12
constexpr int kRowStride = 4 * kCh + 4;                  // floats; (kRowStride / 4) odd -> rows land on distinct bank slots
constexpr int kRows = 8 + 8 * 4;                         // staged rows: 8 query + 8 per candidate
constexpr int kNormLd = 68;
constexpr int kNormOfs = 256;                            // the norm table lives in the stage buffer, behind the 4 x 64 transposed entries
constexpr int kWaveLds = kRows * kRowStride;             // floats per wave (10.9 KB)
static_assert(kNormOfs + 16 * kNormLd <= kWaveLds, "norm table must fit the idle stage buffer");

typedef float mfma4_t __attribute__((ext_vector_type(4)));

// min / max as the bare instructions: fminf / fmaxf make the compiler canonicalise every loaded operand first (v_max_f32 x, x -- a
// third of the box's instructions); a NaN row poisons its norms, and through them the pair's score, either way
__device__ __forceinline__ float vmin1(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax1(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmin3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float4 vmin3_4(const float4& a, const float4& b, const float4& c) {
    return make_float4(vmin3(a.x, b.x, c.x), vmin3(a.y, b.y, c.y), vmin3(a.z, b.z, c.z), vmin3(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 vmax3_4(const float4& a, const float4& b, const float4& c) {
    return make_float4(vmax3(a.x, b.x, c.x), vmax3(a.y, b.y, c.y), vmax3(a.z, b.z, c.z), vmax3(a.w, b.w, c.w));
}

__device__ __forceinline__ float sum_lj(float v) {       // all-reduce over the 4 lanes that share li (lane bits 0, 1)
    v += lane_xor<1>(v);
    return v + lane_xor<2>(v);
}
__device__ __forceinline__ float sum_li(float v) {       // all-reduce over the 4 lanes that share lj (lane bits 2, 3)
    v += dpp_mov<0x124>(v, v);                           // row_ror:4
    return v + dpp_mov<0x128>(v, v);                     // row_ror:8
}
__device__ __forceinline__ float max_lj(float v) {
    v = fmaxf(v, lane_xor<1>(v));
    return fmaxf(v, lane_xor<2>(v));
}
__device__ __forceinline__ float max_li(float v) {
    v = fmaxf(v, dpp_mov<0x124>(v, v));
    return fmaxf(v, dpp_mov<0x128>(v, v));
}
__device__ __forceinline__ float sum16(float v) { return sum_li(sum_lj(v)); }
// CHUNK: a candidate of up to 32 rows occupies 1 .. 4 of the wave's 16-lane groups (one per 8 of its rows); an item's four groups
// hold candidates of [4], [3, 1], [2, 2], [2, 1, 1] or [1, 1, 1, 1] chunks (a 2-chunk candidate on groups 0, 1 or 2, 3; a 3-chunk
// one on 0 .. 2).  w (wave-uniform) = 4 for [4] and [3, 1], 2 for [2, 2] and [2, 1, 1], 1 otherwise; `wide` (per lane) = this
// lane's candidate spans the exchange (w == 2: two groups; w == 4: three or four).  All-reduce across a candidate's groups: one
// v_permlane{16,32}_swap + add per level (common.h: swap_add); a lane of a narrower candidate contributes the neutral element
// and keeps its own value.
template <bool XG>
__device__ __forceinline__ float xg_sum(float v, int w, bool wide) {
    if constexpr (XG) {
        if (w == 2) {
            const float s = swap_add<16>(v, v);
            v = wide ? s : v;
        } else if (w == 4) {
            float z = wide ? v : 0.f;
            z = swap_add<16>(z, z);
            z = swap_add<32>(z, z);
            v = wide ? z : v;
        }
    }
    return v;
}
template <bool XG>
__device__ __forceinline__ float xg_max(float v, int w, bool wide) {
    if constexpr (XG) {
        if (w == 2) {
            const float s = fmaxf(v, lane_xor<16>(v));
            v = wide ? s : v;
        } else if (w == 4) {
            float z = wide ? v : -__builtin_inff();
            z = fmaxf(z, lane_xor<16>(z));
            z = fmaxf(z, lane_xor<32>(z));
            v = wide ? z : v;
        }
    }
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// The Sinkhorn solve of the four pairs of a wave as a resumable state machine.  A solve is ~80 dependent epsilon steps of
// 31 issue slots each -- latency bound for a lone wave (two resident per SIMD here), ~8 us per item when run in one
// piece.  So the solve of item i is cut into slices that run INSIDE the cost stages of item i + 1, each slice in the
// shadow of that stage's HBM loads: the wave was going to wait there anyway.
// ---------------------------------------------------------------------------------------------------------------------
typedef float f2_t __attribute__((ext_vector_type(2)));     // a register pair: v_pk_{mul,add,fma}_f32 work on both halves at once

struct Solve {
    f2_t mc[2];            // masked cost, mc[x] = entries (x, 0), (x, 1): outside the valid block 0 (their weights are 0)
    f2_t neg[2];           // -cdist of the valid block (plan-weighted similarity output only)
    f2_t wa, wb;           // marginals (pair_distances.py:57-60)
    f2_t f, g;             // potentials
    float diam;            // the pair's epsilon_0 (the safe re-solve starts the schedule over)
    float r2, h;           // this step's log2e / eps and eps ln2 / 2: through the annealed part of the schedule the next step's
                           // follow by one multiply each (eps *= scaling) -- no transcendental for the constants
    int n_mid;             // annealed values between diam and blur; step k: 0 = diam, 1 .. n_mid, n_mid + 1 = blur, n_mid + 2 = final
    int k, max_steps;      // wave-uniform: next step, steps of the longest of the wave's four schedules
    int n_mid_lo;          // wave-uniform: the shortest of the four schedules -- steps 2 .. n_mid_lo anneal in all four pairs
    unsigned valid;        // bit x: row x valid, bit 2 + y: column y valid, bit 4: a document longer than the tile (poison)
    int w;                 // wave-uniform (CHUNK): widest exchange across lane groups the item needs (1, 2, 4), 1 otherwise
    bool wide, first;      // per lane (CHUNK): this lane's candidate takes part in the exchange; its group holds the candidate's first chunk
    int64_t out;           // index into scores, < 0 = nothing to store (clamped tail candidate; CHUNK: not the candidate's first group)
};

template <bool XG = false>
__device__ __forceinline__ void solve_begin(Solve& s, const ScoreArgs& a, const float (&cost)[2][2], const float (&neg)[2][2],
                                            const bool (&rv)[2], const bool (&cv)[2], float diam, int w = 1, bool wide = false,
                                            bool first = true) {
    s.w = w;
    s.wide = wide;
    s.first = first;
    s.diam = diam;
    // ---- marginals: soft-max over sentences of the best match / temp --------------------------------------------------
    const float temp = (float)a.temp;
    {
        float qm[2], cm[2];
#pragma unroll
        for (int x = 0; x < 2; ++x)
            qm[x] = xg_max<XG>(max_lj(fmaxf((rv[x] && cv[0]) ? neg[x][0] : kNegBig, (rv[x] && cv[1]) ? neg[x][1] : kNegBig)), w, wide) / temp;
#pragma unroll
        for (int y = 0; y < 2; ++y)
            cm[y] = max_li(fmaxf((rv[0] && cv[y]) ? neg[0][y] : kNegBig, (rv[1] && cv[y]) ? neg[1][y] : kNegBig)) / temp;
        const float mq = max_li(fmaxf(rv[0] ? qm[0] : kNegBig, rv[1] ? qm[1] : kNegBig));
        const float mc = xg_max<XG>(max_lj(fmaxf(cv[0] ? cm[0] : kNegBig, cv[1] ? cm[1] : kNegBig)), w, wide);
        const float sq = (rv[0] ? fast_exp(qm[0] - mq) : 0.f) + (rv[1] ? fast_exp(qm[1] - mq) : 0.f);
        const float sc = (cv[0] ? fast_exp(cm[0] - mc) : 0.f) + (cv[1] ? fast_exp(cm[1] - mc) : 0.f);
        const float lsq = fast_log(sum_li(sq)), lsc = fast_log(xg_sum<XG>(sum_lj(sc), w, wide));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            s.wa[t] = rv[t] ? fast_exp(qm[t] - mq - lsq) : 0.f;      // log_softmax(...).exp(); a zero weight is geomloss's
            s.wb[t] = cv[t] ? fast_exp(cm[t] - mc - lsc) : 0.f;      // log-weight -100000
        }
    }
    s.valid = (rv[0] ? 1u : 0u) | (rv[1] ? 2u : 0u) | (cv[0] ? 4u : 0u) | (cv[1] ? 8u : 0u);
    // ---- epsilon schedule ----------------------------------------------------------------------------------------------
    float ldf;
    s.n_mid = schedule_mid_steps(a, diam, ldf);
    int ms = s.n_mid, ml = s.n_mid;
    ms = max(ms, __shfl_xor(ms, 16));
    ms = max(ms, __shfl_xor(ms, 32));
    ml = min(ml, __shfl_xor(ml, 16));
    ml = min(ml, __shfl_xor(ml, 32));
    s.max_steps = __builtin_amdgcn_readfirstlane(ms) + 3;
    s.n_mid_lo = __builtin_amdgcn_readfirstlane(ml);
    s.k = 0;
    const float r2_first = kLog2e * rcp_refined(diam), h_first = 0.5f * kLn2 * diam;
    s.r2 = r2_first;
    s.h = h_first;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        s.mc[x] = f2_t{(rv[x] && cv[0]) ? cost[x][0] : 0.f, (rv[x] && cv[1]) ? cost[x][1] : 0.f};
        s.neg[x] = f2_t{(rv[x] && cv[0]) ? neg[x][0] : 0.f, (rv[x] && cv[1]) ? neg[x][1] : 0.f};
    }
    // ---- initialisation at eps = diam: softmin of the bare weights (no shift needed: the largest weight of a
    // probability vector over <= 8 atoms is >= 1/8 and C / diam <= ~1) -------------------------------------------------
    float rs[2] = {0.f, 0.f}, cs[2] = {0.f, 0.f};
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const float k0 = __builtin_amdgcn_exp2f(-s.mc[x][y] * r2_first);
            rs[x] = fmaf(s.wb[y], k0, rs[x]);
            cs[y] = fmaf(s.wa[x], k0, cs[y]);
        }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        s.f[t] = -2.f * h_first * __builtin_amdgcn_logf(xg_sum<XG>(sum_lj(rs[t]), w, wide));
        s.g[t] = -2.f * h_first * __builtin_amdgcn_logf(sum_li(cs[t]));
    }
}

// One epsilon step on the register pairs: K = 2^((f_x + g_y - C_xy) r2) as two packed rows, row sums / column sums with
// the weights as factors, f -= h log2(row sums), g -= h log2(column sums) -- 32 issue slots, 8 of them transcendental
// (as scalar code with per-step schedule selects it was 52: the solves are the kernel's largest VALU consumer and, with
// two waves per SIMD, VALU issue is what the HBM stream competes with).
// W: lane groups per candidate, a compile-time constant here -- with a run-time W the branches around the cross-group exchanges
// cut the step into basic blocks and the two row chains and the column chain, which otherwise interleave level by level, run
// one after the other (measured on the config-4 shape: every wave's last solve 26 us).
template <int W>
__device__ __forceinline__ float xg_sum_w(float v, bool wide) {
    if constexpr (W == 2) {
        const float s = swap_add<16>(v, v);
        v = wide ? s : v;
    } else if constexpr (W == 4) {
        float z = wide ? v : 0.f;
        z = swap_add<16>(z, z);
        z = swap_add<32>(z, z);
        v = wide ? z : v;
    }
    return v;
}
template <int W = 1>
__device__ __forceinline__ void solve_step(Solve& s, float r2, float h) {
    const f2_t fr = s.f * r2, gr = s.g * r2;
    const f2_t a0 = __builtin_elementwise_fma(s.mc[0], f2_t{-r2, -r2}, f2_t{fr.x, fr.x} + gr);
    const f2_t a1 = __builtin_elementwise_fma(s.mc[1], f2_t{-r2, -r2}, f2_t{fr.y, fr.y} + gr);
    const f2_t k0 = {__builtin_amdgcn_exp2f(a0.x), __builtin_amdgcn_exp2f(a0.y)};
    const f2_t k1 = {__builtin_amdgcn_exp2f(a1.x), __builtin_amdgcn_exp2f(a1.y)};
    const f2_t t0 = k0 * s.wb, t1 = k1 * s.wb;
    const f2_t cs = __builtin_elementwise_fma(k1, f2_t{s.wa.y, s.wa.y}, k0 * f2_t{s.wa.x, s.wa.x});
    const f2_t lr = {__builtin_amdgcn_logf(xg_sum_w<W>(sum_lj(t0.x + t0.y), s.wide)), __builtin_amdgcn_logf(xg_sum_w<W>(sum_lj(t1.x + t1.y), s.wide))};
    const f2_t lc = {__builtin_amdgcn_logf(sum_li(cs.x)), __builtin_amdgcn_logf(sum_li(cs.y))};
    s.f = __builtin_elementwise_fma(f2_t{-h, -h}, lr, s.f);
    s.g = __builtin_elementwise_fma(f2_t{-h, -h}, lc, s.g);
}

// up to `n` more annealing steps (all of the rest with n < 0)
template <int W>
__device__ __forceinline__ void solve_steps_w(Solve& s, const ScoreArgs& a, int n) {
    const float scal = (float)a.scaling, inv_scal = (float)(1.0 / a.scaling);
    const float eb = (float)a.blur;
    const float r2_blur = kLog2e * rcp_refined(eb), h_blur = 0.5f * kLn2 * eb;
    const int k_end = (n < 0 || s.k + n > s.max_steps) ? s.max_steps : s.k + n;
    int k = s.k;
    // eps_k: diam at k = 0 and 1, diam scaling^(k-1) up to k = n_mid, then blur (averaged), blur (final, h doubled), and
    // nothing (h = 0) while a wave mate with a longer schedule is still annealing.  Steps 2 .. n_mid_lo anneal in all four
    // pairs of the wave: no selects there.
    auto general = [&](int upto) {
#pragma unroll 1
        for (; k < upto; ++k) {
            const bool anneal = k >= 2 && k <= s.n_mid;
            float r2 = anneal ? s.r2 * inv_scal : s.r2;
            float h = anneal ? s.h * scal : s.h;
            if (k > s.n_mid) { r2 = r2_blur; h = k == s.n_mid + 1 ? h_blur : (k == s.n_mid + 2 ? 2.f * h_blur : 0.f); }
            s.r2 = r2;
            s.h = h;
            solve_step<W>(s, r2, h);
        }
    };
    general(min(k_end, 2));
    const int fast_end = min(k_end, s.n_mid_lo + 1);
#pragma unroll 1
    for (; k < fast_end; ++k) {
        s.r2 *= inv_scal;
        s.h *= scal;
        solve_step<W>(s, s.r2, s.h);
    }
    general(k_end);
    s.k = k_end;
}
template <bool XG = false>
__device__ __forceinline__ void solve_steps(Solve& s, const ScoreArgs& a, int n) {
    if constexpr (!XG) {
        solve_steps_w<1>(s, a, n);
    } else {
        if (s.w == 1) solve_steps_w<1>(s, a, n);
        else if (s.w == 2) solve_steps_w<2>(s, a, n);
        else solve_steps_w<4>(s, a, n);
    }
}

template <bool XG = false>
__device__ __forceinline__ float solve_output(const Solve& s, const ScoreArgs& a, const f2_t f, const f2_t g) {
    const int lp = threadIdx.x & 15, li = lp >> 2, lj = lp & 3;
    // CHUNK: the <a, f> terms are counted once per candidate, by the group of its first chunk
    const bool first_grp = !XG || s.first;
    const bool rv[2] = {(s.valid & 1u) != 0, (s.valid & 2u) != 0}, cv[2] = {(s.valid & 4u) != 0, (s.valid & 8u) != 0};
    float score;
    if (a.want != ASPIRE_OT_PLAN_SIM) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            acc += (lj == 0 && rv[t] && first_grp) ? s.wa[t] * f[t] : 0.f;
            acc += (li == 0 && cv[t]) ? s.wb[t] * g[t] : 0.f;
        }
        score = xg_sum<XG>(sum16(acc), s.w, s.wide);
        if (a.want == ASPIRE_OT_SIMILARITY) score = -score;
    } else {
        const float eb = (float)a.blur, rb = rcp_refined(eb);
        float acc = 0.f;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const bool valid = rv[x] && cv[y];
                const float outer = valid ? f[x] + g[y] : 0.f;
                acc += fast_exp(div_r(outer + s.neg[x][y], eb, rb)) * (s.wa[x] * s.wb[y]) * s.neg[x][y];
            }
        score = xg_sum<XG>(sum16(acc), s.w, s.wide);
    }
    return score;
}

// The fast solve shifts its log-sum-exps by the previous potential instead of a maximum: sums stay ~1 and six cross-lane steps leave
// the chain, but a sum CAN leave fp32 range -- scaling below ~0.03, or, at any scaling, a candidate that shares a sentence with
// the query on large vectors (a zero cost beside costs of ~80).  Such a pair used to be poisoned with NaN and re-solved by a
// launch behind every scoring launch (4.7 us of a 108 us call, nearly always for nothing).  Now the wave that finds a poisoned
// pair solves it again on the spot in geomloss's own form -- log-weights in the exponent (a weight of zero: -100000), every
// log-sum-exp shifted by its maximum, the schedule from the top: epsilon = diam twice, then x scaling down to blur, averaged
// updates, one simultaneous un-averaged update at blur (sinkhorn_loop of geomloss 0.2.4).  Rare and short.
template <bool XG>
__device__ __forceinline__ float solve_safe(Solve& s, const ScoreArgs& a) {
    const bool rv[2] = {(s.valid & 1u) != 0, (s.valid & 2u) != 0}, cv[2] = {(s.valid & 4u) != 0, (s.valid & 8u) != 0};
    float la[2], lb[2], f[2], g[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        la[t] = (rv[t] && s.wa[t] > 0.f) ? logf(s.wa[t]) : -100000.f;
        lb[t] = (cv[t] && s.wb[t] > 0.f) ? logf(s.wb[t]) : -100000.f;
        f[t] = g[t] = 0.f;
    }
    // one pair of softmins at eps from potentials (fi, gi): ft_x = -eps LSE_y(lb_y + (gi_y - C_xy) / eps), gt_y likewise
    auto softmins = [&](float eps, const float (&fi)[2], const float (&gi)[2], float (&ft)[2], float (&gt)[2]) {
        const float re = 1.0f / eps;
        float ar[2][2], ac[2][2];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                ar[x][y] = lb[y] + (gi[y] - s.mc[x][y]) * re;
                ac[x][y] = la[x] + (fi[x] - s.mc[x][y]) * re;
            }
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const float m = xg_max<XG>(max_lj(fmaxf(ar[x][0], ar[x][1])), s.w, s.wide);
            const float e = xg_sum<XG>(sum_lj(expf(ar[x][0] - m) + expf(ar[x][1] - m)), s.w, s.wide);
            ft[x] = -eps * (m + logf(e));
        }
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const float m = max_li(fmaxf(ac[0][y], ac[1][y]));
            const float e = sum_li(expf(ac[0][y] - m) + expf(ac[1][y] - m));
            gt[y] = -eps * (m + logf(e));
        }
    };
    // the wave's pairs run their own schedules; the loop goes to the longest (wave-uniform), a finished pair idles
    int n_max = s.n_mid;
    n_max = max(n_max, __shfl_xor(n_max, 16));
    n_max = max(n_max, __shfl_xor(n_max, 32));
    n_max = __builtin_amdgcn_readfirstlane(n_max);
    const float eb = (float)a.blur, scal = (float)a.scaling;
    float ft[2], gt[2];
    softmins(s.diam, f, g, ft, gt);          // initialisation at eps_0 = diam
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f[t] = ft[t];
        g[t] = gt[t];
    }
    float eps = s.diam;
#pragma unroll 1
    for (int k = 0; k <= n_max + 1; ++k) {   // k = 0: diam, 1 .. n_mid: diam scaling^(k-1) (the first of them diam again), n_mid + 1: blur
        const bool mine = k <= s.n_mid + 1;
        const float e = k > s.n_mid ? eb : eps;
        softmins(e, f, g, ft, gt);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f[t] = mine ? 0.5f * (f[t] + ft[t]) : f[t];
            g[t] = mine ? 0.5f * (g[t] + gt[t]) : g[t];
        }
        if (k >= 1) eps *= scal;
    }
    softmins(eb, f, g, ft, gt);              // the last extrapolation: both from the previous pair, not averaged
    return solve_output<XG>(s, a, f2_t{ft[0], ft[1]}, f2_t{gt[0], gt[1]});
}

// finish a solve: remaining steps, the score, the store
template <bool XG = false>
__device__ __forceinline__ void solve_finish(Solve& s, const ScoreArgs& a) {
    solve_steps<XG>(s, a, -1);
    float score = solve_output<XG>(s, a, s.f, s.g);
    // an overflowed / vanished sum has turned into inf / nan that sticks to the potentials and reaches the score; so does a clearly
    // NEGATIVE transport cost (an entropic OT value is >= 0 up to rounding): sums that left fp32 range on the way and came back
    // finite (seen with a sentence shared by query and candidate on large vectors)
    bool bad = !(fabsf(score) < 1e30f);
    if (a.want != ASPIRE_OT_PLAN_SIM && (a.want == ASPIRE_OT_SIMILARITY ? score : -score) > 1e-2f) bad = true;
    if (__any(bad && !(s.valid & 16u))) {
        const float again = solve_safe<XG>(s, a);
        if (bad) score = again;
    }
    // a document longer than the tile is never truncated silently: NaN, for the kernels queued behind this one (hybrid forms)
    if (s.valid & 16u) score = __builtin_nanf("");
    if (s.out >= 0 && (threadIdx.x & 15) == 0) a.scores[s.out] = score;
}

}  // namespace
}  // namespace aspire
