// otAspire cost stage for documents of 9 .. 16 sentence rows, few queries per candidate, CSR inputs (A5; reference arithmetic:
// src/learning/facetid_models/pair_distances.py:39-56 + geomloss 0.2.4's squared_distances, restated -- see score.hip).
//
// Real abstracts are often longer than the 8 sentences of the benchmark configurations.  Until now such pools went through
// the per-pair tile-loop kernel (pair_cost_kernel<2>: one workgroup per candidate, VALU difference sums: 20 x 1000 x 12
// pairs 540 us = 36 M pairs/s) or, for one big pool, the 32-column Gram tiles (1 x 20 000 x 12: 307 us).  This is the
// streaming phase of the fused kernel (fused.hip) widened to 16 x 16 pairs: HBM bound, every candidate row read once.
//
//   * item = TWO consecutive candidates of ONE query per wave.  A candidate occupies two 16-lane groups, one per half of its
//     (up to) 16 rows: lane group p = (candidate p >> 1, half p & 1) stages that half's 8 rows 64 coordinates at a time
//     (coalesced global_load_dwordx4 -> ds_write_b128 into padded rows) and query rows 4 p .. 4 p + 3; row norms and the
//     half's per-coordinate box fall out of the registers, the two halves' boxes are joined across the lane groups
//     (v_permlane16_swap) and with the query's precomputed box into geomloss's diameter term.
//   * dot products on the matrix pipe: v_mfma_f32_4x4x1, block (p, iq, jq) = query rows 4 iq .. + 3 x rows 4 jq .. + 3 of
//     half p -- issued twice per coordinate, for query rows 0 .. 7 and 8 .. 15.
//   * epilogue: |x|^2 - 2 x.y + |y|^2 -> the pair's 16 x 16 cost and -cdist slots of the workspace (entries where the
//     expansion cancels are redone with the direct formula by the whole wave, four per memory round trip) + the diameter
//     term; sinkhorn_block_kernel<2, ..> solves from there.
#include <math.h>

#include "common.h"
#include "score_device.h"
#include "score_types.h"
#include "tuning.h"

namespace aspire {
namespace {

constexpr int kCh = 16;                                  // 16-byte chunks per row per stage (64 coordinates)
constexpr int kStages = kD / (4 * kCh);                  // 12
constexpr int kRowStride = 4 * kCh + 4;                  // floats; (kRowStride / 4) odd -> rows land on distinct bank slots
constexpr int kRows = 16 + 2 * 16;                       // staged rows: 16 query + 16 per candidate
constexpr int kNormLd = 68;
constexpr int kWaveLds = kRows * kRowStride;             // floats per wave (13 KB); the norm table (12 x 68) shares it
static_assert(12 * kNormLd <= kWaveLds, "norm table must fit the idle stage buffer");

typedef float mfma4_t __attribute__((ext_vector_type(4)));
typedef float f2_t __attribute__((ext_vector_type(2)));

// L2MAX: tsAspire on the same streaming phase -- the score is the maximum of -cdist over the valid block
// (allpair_masked_dist_l2max, pair_distances.py:167-176); no boxes, nothing goes to the workspace.
// REC (batched jobs with documents of 17 .. 32 rows: whole abstracts on both sides, pp_settings.py:2-3): the items come from
// chunk16_prep_kernel's records -- a 16-row half of the query against two candidates of <= 16 rows or against the two halves of
// ONE candidate of 17 .. 32 (its box then joins across the wave's two candidate slots) -- and a pair's 16 x 16 blocks land in
// workspace slots of TW x TW tiles of 8 (TW = 3, 4), which sinkhorn_block_kernel<TW, ..> solves.  (Before: the VALU tile loop.)
template <bool L2MAX, int TW = 2, bool REC = false>
__global__ void __launch_bounds__(256, 2) pair_tile16_kernel(ScoreArgs a, PairWs<TW> ws, const float* __restrict__ qbox) {
    constexpr int LDW = 8 * TW, EW = 64 * TW * TW;            // workspace slot: row stride, entries
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const bool selective = gate_few_long(a);                // hybrid: only the pairs that hold a document of more than 8 rows
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* lds = lds_all + wave * kWaveLds;
    float* nscr = lds;                                      // [12][kNormLd]: norm partials, written once the stages are done
    const bool mapped = a.pairing == kPairMapped;           // batched jobs: items are pairs of candidates of jobs [job0, job1)
    const uint32_t nq = mapped ? 1u : (uint32_t)a.q.n;
    const uint32_t ncand = (uint32_t)(a.cand1 - a.cand0);
    // MAPPED: the job tables count groups of FOUR candidates (batch_prep_kernel); an item is half of one
    const uint32_t g4_lo = (mapped && !REC) ? (uint32_t)a.grp_off[a.job0] : 0u;
    const uint32_t n_items = REC ? (uint32_t)a.grp_off[0]        // the item counter chunk16_prep_kernel has left
                                 : mapped ? 2u * ((uint32_t)a.grp_off[a.job1] - g4_lo) : ((ncand + 1) / 2) * nq;   // CROSS: (pair of candidates, query), group-major
    const uint32_t n_waves = gridDim.x * 4;
    const bool own_diam = a.diameter == nullptr && !L2MAX;

    const int p = lane >> 4, lp = lane & 15, cs = p >> 1, hp = p & 1, sc = lp;
    const int mb = lane >> 2, mt = lane & 3, miq = (mb >> 1) & 1, mjq = mb & 1;

    struct Ctx {
        int64_t c_idx, q_idx, slot;
        int c_len, q_len, c_start, q_start;
        int qrow0, crow0;      // REC: first row of the query half / of this slot's candidate half
        bool my_c_real, wide;  // REC: wide = the wave's two slots are the halves of one candidate
    };
    auto load_ctx = [&](uint32_t item) {
        Ctx x;
        x.qrow0 = x.crow0 = 0;
        x.wide = false;
        uint32_t q_loc, c_loc0, c_end;
        if constexpr (REC) {
            const int32_t* rec = a.grp_rec + (size_t)item * 16;
            const int4 hd = *reinterpret_cast<const int4*>(rec);
            x.q_idx = hd.x;
            x.q_len = hd.y;
            x.q_start = hd.z;
            x.qrow0 = 16 * (hd.w & 1);
            x.wide = (hd.w >> 8) != 0;
            x.c_idx = rec[4 + cs];
            x.c_len = rec[6 + cs];
            x.c_start = rec[8 + cs];
            x.crow0 = rec[10 + cs];
            x.my_c_real = rec[12 + cs] != 0;
            x.slot = x.c_idx;
            return x;
        }
        if (mapped) {
            const uint32_t g4 = g4_lo + (item >> 1);
            q_loc = (uint32_t)a.grp_job[g4];
            c_loc0 = (uint32_t)a.job_off[q_loc] + (g4 - (uint32_t)a.grp_off[q_loc]) * 4 + (item & 1) * 2;
            c_end = (uint32_t)a.job_off[q_loc + 1];
        } else {
            const uint32_t cg = nq == 1 ? item : item / nq;
            q_loc = nq == 1 ? 0 : item - cg * nq;
            c_loc0 = cg * 2;
            c_end = ncand;
        }
        const uint32_t my_c_loc = min(c_loc0 + (uint32_t)cs, c_end - 1);     // tail items: clamp (duplicate work, not stored)
        x.my_c_real = c_loc0 + (uint32_t)cs < c_end;
        x.c_idx = a.cand0 + my_c_loc;
        x.q_idx = (int64_t)q_loc;
        x.slot = mapped ? (int64_t)my_c_loc : (int64_t)q_loc * ncand + my_c_loc;
        x.c_len = a.c.len[x.c_idx];
        x.q_len = a.q.len[x.q_idx];
        x.c_start = a.c.start[x.c_idx];
        x.q_start = a.q.start[x.q_idx];
        return x;
    };
    const uint32_t item_first = blockIdx.x * 4 + wave;
    Ctx next = load_ctx(item_first < n_items ? item_first : 0);

    for (uint32_t item = item_first; item < n_items; item += n_waves) {
        const Ctx cur = next;
        next = load_ctx(item + n_waves < n_items ? item + n_waves : item);      // one item ahead (fused.hip)
        const int64_t q_idx = cur.q_idx, slot = cur.slot;
        const int c_len = cur.c_len, q_len = cur.q_len, c_start = cur.c_start;
        const int qrow0 = cur.qrow0, crow0 = cur.crow0;
        const bool wide = cur.wide;
        if (selective && !__any(cur.my_c_real && (cur.q_len > 8 || cur.c_len > 8))) continue;      // the fused kernel scored this item
        const bool my_c_real = cur.my_c_real && (!selective || cur.q_len > 8 || cur.c_len > 8);
        const float* qdoc = a.q.rows + (size_t)cur.q_start * kD;
        const float* sy_doc = a.c.rows + (size_t)c_start * kD;
        // the query's per-coordinate box; with caller-supplied diameters the candidate's rows stand in (term unused): the
        // loads stay unconditional (fused.hip)
        const float* qb = (own_diam && !L2MAX) ? qbox + (size_t)q_idx * 2 * kD : sy_doc;
        const int qb_hi = own_diam ? kD : 0;

        mfma4_t macc[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int m = 0; m < 2; ++m) macc[s][m] = mfma4_t{0.f, 0.f, 0.f, 0.f};
        f2_t ny[8], nx[4], dsq = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) ny[k] = f2_t{0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) nx[k] = f2_t{0.f, 0.f};
        float4 vy[8], vx[4], qmn, qmx;
        auto issue_loads = [&](int st) {
            const int dofs = (st * kCh + sc) * 4;
#pragma unroll
            for (int j = 0; j < 8; ++j) vy[j] = ld4_stream(sy_doc + (size_t)min(crow0 + 8 * hp + j, c_len - 1) * kD + dofs);   // pad rows: copies of the last row
            if constexpr (!L2MAX) {
                qmn = ld4(qb + dofs);
                qmx = ld4(qb + qb_hi + dofs);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) vx[k] = ld4(qdoc + (size_t)min(qrow0 + 4 * p + k, q_len - 1) * kD + dofs);
        };
        auto sq_acc = [](f2_t acc, const float4& v) {
            acc = __builtin_elementwise_fma(f2_t{v.x, v.y}, f2_t{v.x, v.y}, acc);
            return __builtin_elementwise_fma(f2_t{v.z, v.w}, f2_t{v.z, v.w}, acc);
        };
        issue_loads(0);
#pragma unroll 1
        for (int st = 0; st < kStages; ++st) {
            // ---- stage: registers -> LDS with norm / box side products, THEN the next stage's loads into the same registers ----
            if (a.center) {
                // rows sharing a large common component (ASPIRE_OT_FLAG_CENTER, fused.hip): the mean of the sixteen staged query
                // rows comes off every row and off the query's box before anything is squared
                float4 mu = make_float4(vx[0].x + vx[1].x + vx[2].x + vx[3].x, vx[0].y + vx[1].y + vx[2].y + vx[3].y,
                                        vx[0].z + vx[1].z + vx[2].z + vx[3].z, vx[0].w + vx[1].w + vx[2].w + vx[3].w);
                mu.x += lane_xor<16>(mu.x); mu.y += lane_xor<16>(mu.y); mu.z += lane_xor<16>(mu.z); mu.w += lane_xor<16>(mu.w);
                mu.x += lane_xor<32>(mu.x); mu.y += lane_xor<32>(mu.y); mu.z += lane_xor<32>(mu.z); mu.w += lane_xor<32>(mu.w);
                mu.x *= 0.0625f; mu.y *= 0.0625f; mu.z *= 0.0625f; mu.w *= 0.0625f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    vy[j].x -= mu.x; vy[j].y -= mu.y; vy[j].z -= mu.z; vy[j].w -= mu.w;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    vx[k].x -= mu.x; vx[k].y -= mu.y; vx[k].z -= mu.z; vx[k].w -= mu.w;
                }
                if constexpr (!L2MAX) {
                    qmn.x -= mu.x; qmn.y -= mu.y; qmn.z -= mu.z; qmn.w -= mu.w;
                    qmx.x -= mu.x; qmx.y -= mu.y; qmx.z -= mu.z; qmx.w -= mu.w;
                }
            }
            {
                float4 mn = vy[0], mx = vy[0];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    ny[j] = sq_acc(ny[j], vy[j]);
                    if (j > 0 && !L2MAX) {
                        mn.x = fminf(mn.x, vy[j].x); mn.y = fminf(mn.y, vy[j].y); mn.z = fminf(mn.z, vy[j].z); mn.w = fminf(mn.w, vy[j].w);
                        mx.x = fmaxf(mx.x, vy[j].x); mx.y = fmaxf(mx.y, vy[j].y); mx.z = fmaxf(mx.z, vy[j].z); mx.w = fmaxf(mx.w, vy[j].w);
                    }
                    *reinterpret_cast<float4*>(lds + (16 + p * 8 + j) * kRowStride + sc * 4) = vy[j];
                }
                if constexpr (!L2MAX) {
                    // the candidate's other half sits in the neighbouring lane group (a half past the document's end repeats its
                    // last row: box neutral)
                    mn.x = fminf(mn.x, lane_xor<16>(mn.x)); mn.y = fminf(mn.y, lane_xor<16>(mn.y));
                    mn.z = fminf(mn.z, lane_xor<16>(mn.z)); mn.w = fminf(mn.w, lane_xor<16>(mn.w));
                    mx.x = fmaxf(mx.x, lane_xor<16>(mx.x)); mx.y = fmaxf(mx.y, lane_xor<16>(mx.y));
                    mx.z = fmaxf(mx.z, lane_xor<16>(mx.z)); mx.w = fmaxf(mx.w, lane_xor<16>(mx.w));
                    if constexpr (REC)
                        if (wide) {       // the candidate's other 16 rows sit in the wave's other slot
                            mn.x = fminf(mn.x, lane_xor<32>(mn.x)); mn.y = fminf(mn.y, lane_xor<32>(mn.y));
                            mn.z = fminf(mn.z, lane_xor<32>(mn.z)); mn.w = fminf(mn.w, lane_xor<32>(mn.w));
                            mx.x = fmaxf(mx.x, lane_xor<32>(mx.x)); mx.y = fmaxf(mx.y, lane_xor<32>(mx.y));
                            mx.z = fmaxf(mx.z, lane_xor<32>(mx.z)); mx.w = fmaxf(mx.w, lane_xor<32>(mx.w));
                        }
                    const f2_t dlo = {fmaxf(mx.x, qmx.x) - fminf(mn.x, qmn.x), fmaxf(mx.y, qmx.y) - fminf(mn.y, qmn.y)};
                    const f2_t dhi = {fmaxf(mx.z, qmx.z) - fminf(mn.z, qmn.z), fmaxf(mx.w, qmx.w) - fminf(mn.w, qmn.w)};
                    dsq = __builtin_elementwise_fma(dhi, dhi, __builtin_elementwise_fma(dlo, dlo, dsq));
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    nx[k] = sq_acc(nx[k], vx[k]);
                    *reinterpret_cast<float4*>(lds + (4 * p + k) * kRowStride + sc * 4) = vx[k];
                }
            }
            // pin the side products HERE (the optimiser otherwise sinks these loop-carried sums below the loads)
            asm volatile("" : "+v"(ny[0]), "+v"(ny[1]), "+v"(ny[2]), "+v"(ny[3]), "+v"(ny[4]), "+v"(ny[5]), "+v"(ny[6]), "+v"(ny[7]),
                              "+v"(nx[0]), "+v"(nx[1]), "+v"(nx[2]), "+v"(nx[3]), "+v"(dsq) : : "memory");
            if (st + 1 < kStages) issue_loads(st + 1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- accumulate: lane t of block (p, iq, jq) feeds query rows 4 iq + t and 8 + 4 iq + t as A, row 4 jq + t of half p as B
            {
                const float* xm = lds + (4 * miq + mt) * kRowStride;
                const float* ym = lds + (16 + p * 8 + 4 * mjq + mt) * kRowStride;
                if (q_len - qrow0 > 8) {
#pragma unroll 4
                    for (int c = 0; c < kCh; ++c) {
                        const float4 x0 = *reinterpret_cast<const float4*>(xm + c * 4);
                        const float4 x1 = *reinterpret_cast<const float4*>(xm + 8 * kRowStride + c * 4);
                        const float4 yb = *reinterpret_cast<const float4*>(ym + c * 4);
                        macc[0][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(x0.x, yb.x, macc[0][0], 0, 0, 0);
                        macc[1][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(x1.x, yb.x, macc[1][0], 0, 0, 0);
                        macc[0][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(x0.y, yb.y, macc[0][1], 0, 0, 0);
                        macc[1][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(x1.y, yb.y, macc[1][1], 0, 0, 0);
                        macc[0][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(x0.z, yb.z, macc[0][0], 0, 0, 0);
                        macc[1][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(x1.z, yb.z, macc[1][0], 0, 0, 0);
                        macc[0][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(x0.w, yb.w, macc[0][1], 0, 0, 0);
                        macc[1][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(x1.w, yb.w, macc[1][1], 0, 0, 0);
                    }
                } else {
                    // a short query (facet-selected rows) against longer candidates: query rows 8 .. 15 do not exist -- their
                    // entries are masked downstream and their sums stay zero
#pragma unroll 4
                    for (int c = 0; c < kCh; ++c) {
                        const float4 x0 = *reinterpret_cast<const float4*>(xm + c * 4);
                        const float4 yb = *reinterpret_cast<const float4*>(ym + c * 4);
                        macc[0][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(x0.x, yb.x, macc[0][0], 0, 0, 0);
                        macc[0][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(x0.y, yb.y, macc[0][1], 0, 0, 0);
                        macc[0][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(x0.z, yb.z, macc[0][0], 0, 0, 0);
                        macc[0][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(x0.w, yb.w, macc[0][1], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the stage buffer is rewritten next
            __builtin_amdgcn_wave_barrier();
        }

        // ---- norms: sum the staging lanes' partials through the table nscr[value][lane] (stage buffer, idle now) ----
        // value 0..7: |y|^2 partials of the lane's staged half rows; 8..11: |x|^2 partials of its four query rows
#pragma unroll
        for (int k = 0; k < 8; ++k) nscr[k * kNormLd + lane] = ny[k].x + ny[k].y;
#pragma unroll
        for (int k = 0; k < 4; ++k) nscr[(8 + k) * kNormLd + lane] = nx[k].x + nx[k].y;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        auto table_sum = [&](int value, int lane0) {
            const float4* src = reinterpret_cast<const float4*>(nscr + value * kNormLd + lane0);
            float t = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float4 u = src[m];
                t += (u.x + u.y) + (u.z + u.w);
            }
            return t;
        };
        const float yy = table_sum(4 * mjq + mt, p * 16);               // row 8 hp + 4 jq + t of the candidate: staged by group p
        float xx[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) xx[s][r] = table_sum(8 + r, (2 * s + miq) * 16);      // query row 8 s + 4 iq + r: group 2 s + iq
        float diam2 = dsq.x + dsq.y;
        diam2 += lane_xor<1>(diam2); diam2 += lane_xor<2>(diam2); diam2 += lane_xor<4>(diam2); diam2 += lane_xor<8>(diam2);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the table is the next item's stage buffer
        __builtin_amdgcn_wave_barrier();

        // ---- the pair's entries (i = 8 s + 4 iq + r, j = 8 hp + 4 jq + t) ---------------------------------------------
        // (round 6: a cancelling entry is redone from the exact sum whatever formula torch.cdist would pick -- also beyond 25 rows: include/aspire_hip.h, SHARED SENTENCES)
        const int j = crow0 + 8 * hp + 4 * mjq + mt;
        bool redo[2][4];
        float negv[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = qrow0 + 8 * s + 4 * miq + r;
                const float dot = macc[s][0][r] + macc[s][1][r];
                const float sq = fmaf(-2.f, dot, xx[s][r]) + yy;
                const float ns = xx[s][r] + yy;
                redo[s][r] = my_c_real && i < q_len && j < c_len && sq < 1e-4f * ns * ns;
                negv[s][r] = -sqrtf(fmaxf(sq, 0.f));
                if constexpr (!L2MAX)
                    if (my_c_real && (!REC || (i < LDW && j < LDW))) ws.cost[slot * EW + i * LDW + j] = sqrtf(fmaxf(sq, 1e-8f));
            }
        if constexpr (!L2MAX)
            if (my_c_real && own_diam && lp == 0 && hp == 0 && qrow0 == 0 && crow0 == 0) ws.diam2[slot] = diam2;
        // direct-formula redo, the whole wave on one entry (12 coordinates per lane), four entries per memory round trip
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                unsigned long long wm = __ballot(redo[s][r]);
                while (wm != 0) {
                    int owner[4];
                    float4 u[4][3], v[4][3];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        owner[e] = wm != 0 ? (int)__builtin_ctzll(wm) : -1;
                        wm = wm != 0 ? wm & (wm - 1) : 0;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {            // all 24 loads go out before the first is consumed
                        const int o = owner[e] >= 0 ? owner[e] : owner[0];
                        const int ob = o >> 2, oi = qrow0 + 8 * s + 4 * ((ob >> 1) & 1) + r;
                        const int oj = (REC ? __builtin_amdgcn_readlane(crow0, o) : 0) + 8 * ((ob >> 2) & 1) + 4 * (ob & 1) + (o & 3);
                        const int cs_e = __builtin_amdgcn_readlane(c_start, o);
                        const float* qrow = qdoc + (size_t)oi * kD + 4 * lane;
                        const float* crow = a.c.rows + ((size_t)cs_e + oj) * kD + 4 * lane;
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
                            u[e][t] = ld4(qrow + 256 * t);
                            v[e][t] = ld4(crow + 256 * t);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float part = 0.f;
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
                            const float d0 = u[e][t].x - v[e][t].x, d1 = u[e][t].y - v[e][t].y, d2 = u[e][t].z - v[e][t].z, d3 = u[e][t].w - v[e][t].w;
                            part = fmaf(d3, d3, fmaf(d2, d2, fmaf(d1, d1, fmaf(d0, d0, part))));
                        }
                        if (owner[e] >= 0) {
                            const float tot = wave_sum(part);
                            const int o = owner[e], ob = o >> 2;
                            const int oi = 8 * s + 4 * ((ob >> 1) & 1) + r, oj = 8 * ((ob >> 2) & 1) + 4 * (ob & 1) + (o & 3);
                            (void)oi; (void)oj;
                            if (lane == o) {
                                negv[s][r] = -sqrtf(tot);
                                if constexpr (!L2MAX) {      // geomloss's cost from the same exact sum (this lane wrote the expansion's value above)
                                    const int i = qrow0 + 8 * s + 4 * miq + r;
                                    if (!REC || (i < LDW && j < LDW)) ws.cost[slot * EW + i * LDW + j] = sqrtf(fmaxf(tot, 1e-8f));
                                }
                            }
                        }
                    }
                }
            }
        if constexpr (L2MAX) {
            float m = kNegBig;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, (qrow0 + 8 * s + 4 * miq + r < q_len && j < c_len) ? negv[s][r] : kNegBig);
            m = fmaxf(m, lane_xor<1>(m)); m = fmaxf(m, lane_xor<2>(m)); m = fmaxf(m, lane_xor<4>(m)); m = fmaxf(m, lane_xor<8>(m));
            m = fmaxf(m, lane_xor<16>(m));                                    // the candidate's two halves
            if constexpr (REC)
                if (wide) m = fmaxf(m, lane_xor<32>(m));                      // ... of each of its two slots
            if (q_len > 16 || c_len > (REC && wide ? 32 : 16)) m = __builtin_nanf("");          // longer than the tile: never truncated silently
            if (my_c_real && lp == 0 && hp == 0 && crow0 == 0) a.scores[mapped ? cur.c_idx : q_idx * a.c.n + cur.c_idx] = m;
        } else if (my_c_real) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = qrow0 + 8 * s + 4 * miq + r;
                    if (!REC || (i < LDW && j < LDW)) ws.neg[slot * EW + i * LDW + j] = negv[s][r];
                }
        }
    }
}

}  // namespace

// Documents of 9 .. 16 rows (both sides), CSR, CROSS or MAPPED pairing with the job tables built.
bool tile16_path_ok(const aspire_repset* q, const aspire_repset* c, int pairing) {
    if (q->ext != 0 || c->ext != 0 || (pairing != ASPIRE_PAIR_CROSS && pairing != kPairMapped)) return false;
    const int mq = q->max_len, mc = c->max_len;
    return mq > 0 && mc > 0 && mq <= 16 && mc <= 16 && (mq > 8 || mc > 8);
}

// items_bound: upper bound of the launch's items (pairs of candidates x queries)
int launch_pair_tile16(const ScoreArgs& a, float* cost, float* neg, float* diam2, int64_t items_bound, const float* qbox,
                       hipStream_t stream) {
    PairWs<2> ws{cost, neg, diam2};
    const int64_t waves = items_bound < 256 * 8 ? items_bound : 256 * 8;
    hipLaunchKernelGGL((pair_tile16_kernel<false>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 4 * kWaveLds * sizeof(float), stream, a, ws,
                       qbox);
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}

// tsAspire (max-sim) of every (query, candidate) pair, CROSS
int launch_pair_tile16_l2max(const ScoreArgs& a, int64_t items_bound, hipStream_t stream) {
    PairWs<2> ws{nullptr, nullptr, nullptr};
    const int64_t waves = items_bound < 256 * 8 ? items_bound : 256 * 8;
    hipLaunchKernelGGL((pair_tile16_kernel<true>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 4 * kWaveLds * sizeof(float), stream, a, ws,
                       (const float*)nullptr);
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}

// REC items (chunk16_prep_kernel's records in a.grp_rec, their count in a.grp_off[0]); T = the workspace slots' tile width (3, 4)
int launch_pair_tile16_rec(const ScoreArgs& a, int T, float* cost, float* neg, float* diam2, int64_t items_bound, const float* qbox,
                           hipStream_t stream) {
    const int64_t waves = items_bound < 256 * 8 ? items_bound : 256 * 8;
    const dim3 grid((unsigned)((waves + 3) / 4));
    if (T == 3) {
        PairWs<3> ws{cost, neg, diam2};
        hipLaunchKernelGGL((pair_tile16_kernel<false, 3, true>), grid, dim3(256), 4 * kWaveLds * sizeof(float), stream, a, ws, qbox);
    } else {
        PairWs<4> ws{cost, neg, diam2};
        hipLaunchKernelGGL((pair_tile16_kernel<false, 4, true>), grid, dim3(256), 4 * kWaveLds * sizeof(float), stream, a, ws, qbox);
    }
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}
int launch_pair_tile16_rec_l2max(const ScoreArgs& a, int64_t items_bound, hipStream_t stream) {
    PairWs<4> ws{nullptr, nullptr, nullptr};
    const int64_t waves = items_bound < 256 * 8 ? items_bound : 256 * 8;
    hipLaunchKernelGGL((pair_tile16_kernel<true, 4, true>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 4 * kWaveLds * sizeof(float), stream,
                       a, ws, (const float*)nullptr);
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}

}  // namespace aspire
