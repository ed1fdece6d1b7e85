// Scoring kernels: masked pairwise L2 (A5), soft-max marginals (A6), geomloss-0.2.4 Sinkhorn (A7/A8),
// max-sim (A9), batch bounding-box diameter.  Reference arithmetic:
//   src/learning/facetid_models/pair_distances.py:21-92, :138-186 (allenai/aspire)
//   geomloss==0.2.4 sinkhorn_tensorized / sinkhorn_loop (third party; restated, parity unpinned).
//
// Data layout and work decomposition (gfx950, wave = 64 lanes):
//   * one workgroup = 3 waves = one candidate document x a chunk of queries.  Wave w owns encoding
//     coordinates [256w, 256w+256): lane l holds the float4 at d = 256w + 4l of every sentence row, so a
//     768-float row is ONE global_load_dwordx4 per lane, perfectly coalesced, no LDS staging.
//   * sentence-pair sums are formed 8x8 rows at a time ("tile"): each lane accumulates the 64
//     (i,j) partial sums over its 4 coordinates, then a 63-exchange halving butterfly
//     (v_permlane32_swap / v_permlane16_swap / DPP) leaves lane l = 8*i + j holding the wave's sum for
//     (i,j).  The three waves' partials meet in LDS (6 KB per tile).
//   * the Sinkhorn solve of one (query, candidate) pair runs in ONE wave with lane (li,lj) = (l>>3,l&7)
//     holding the T x T entries (8a+li, 8b+lj): row log-sum-exps are DPP reductions over lane bits 0-2,
//     column ones over bits 3-5 (permlane swaps); potentials stay in registers for all ~70 eps-steps.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"
#include "tuning.h"
#include "score_types.h"
#include "score_device.h"
#include "topk_device.h"

namespace aspire {
namespace {

constexpr int kWaves = 3;
constexpr int kBlock = 64 * kWaves;
constexpr int kMaxT = 4;  // sentence rows per document <= 8 * kMaxT
// Groups of four candidates from which the fused streaming kernel (fused.hip) is ahead of the small-pool kernels, documents of
// <= 8 rows (tools/otbatchcross.py: one query x one pool 750 groups 39 against 46 us, 2000 groups 59 / 94; batched jobs 500
// groups 33 / 37, 1600 groups 45 / 79).  Below, a call is latency bound and the small-pool kernels' shorter chains win.
constexpr int64_t kStreamMinGroups1 = 640, kStreamMinGroupsBatch = 512;

// LDS carve (floats): red[kWaves][T*T][128] | rednorm[kWaves][T][16] | reddiam[4] | redo_mask (8 B) + pad | xpose[kWaves][32][68]
constexpr int kXpLd = 68;                 // row stride of the transpose scratch: 64 lanes + 4 (keeps b128 reads
constexpr int kXpWave = 32 * kXpLd;       // 16 B aligned and spreads the 16-lane read groups over all bank slots)
template <int T>
struct Lds {
    static constexpr int kRed = kWaves * T * T * 128;
    static constexpr int kNorm = kWaves * T * 16;
    static constexpr int kRedo = kRed + kNorm + 4;   // 64-bit mask of entries to redo (pair_cost1_body): a word no reduction scratch touches
    static constexpr int kXp = kRedo + 4;
    static constexpr int kTotal = kXp + kWaves * kXpWave;
};

// Sum N per-lane partials across the 64 lanes of a wave through LDS instead of cross-lane VALU ops: every lane
// stores its N values as a column (conflict-free ds_write_b32), then lane l reads back 64*N/64... = a contiguous
// piece of row (l * N / 64) as b128s and adds it up.  Element e ends up in the 64/N lanes e*64/N ...; returns it.
// On gfx950 a v_permlane*_swap costs ~22 issue cycles and a DPP op ~8 (build/dbg/thr.hip), so the 31-exchange
// register butterfly this replaces was 4x the cost of the 768 multiply-adds it served.
template <int N>
__device__ __forceinline__ float lds_wave_reduce(const float (&v)[N], float* xp, int lane) {
    static_assert(N == 32 || N == 16, "sizes used here");
#pragma unroll
    for (int k = 0; k < N; ++k) xp[k * kXpLd + lane] = v[k];
    // DS operations of one wave execute in order: the loads below see the stores above (other waves use
    // their own scratch).  The fence only stops the compiler from reordering them.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int kLanesPer = 64 / N;         // lanes sharing one element
    constexpr int kFloats = 64 / kLanesPer;   // floats each of them adds up
    const float4* row = reinterpret_cast<const float4*>(xp + (lane / kLanesPer) * kXpLd + (lane % kLanesPer) * kFloats);
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < kFloats / 4; ++m) {
        const float4 t = row[m];
        s += (t.x + t.y) + (t.z + t.w);
    }
    s += lane_xor<1>(s);
    if constexpr (kLanesPer == 4) s += lane_xor<2>(s);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the scratch is rewritten by the next call
    __builtin_amdgcn_wave_barrier();
    return s;
}

// Load N sentence rows (this lane's float4 slice) of one document; rows >= navail read as zero.
template <int N, bool BBOX, bool CENTER = false>
__device__ __forceinline__ void load_rows(float4 (&r)[N], const float* doc, int row0, int navail, int dofs, int nbox,
                                          float4& mn, float4& mx, const float4& mu) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int row = row0 + i;
        r[i] = row < navail ? ld4(doc + (size_t)row * kD + dofs) : make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (CENTER)                 // ASPIRE_OT_FLAG_CENTER (a row past the document stays a zero row: masked downstream)
            if (row < navail) { r[i].x -= mu.x; r[i].y -= mu.y; r[i].z -= mu.z; r[i].w -= mu.w; }
        if (BBOX && row < nbox) {
            mn.x = fminf(mn.x, r[i].x); mn.y = fminf(mn.y, r[i].y); mn.z = fminf(mn.z, r[i].z); mn.w = fminf(mn.w, r[i].w);
            mx.x = fmaxf(mx.x, r[i].x); mx.y = fmaxf(mx.y, r[i].y); mx.z = fmaxf(mx.z, r[i].z); mx.w = fmaxf(mx.w, r[i].w);
        }
    }
}

// Per-wave partial sums of half an 8x8 tile (4 query rows x 8 candidate rows) -> LDS.
// red layout: [tile][2][64] (0: x.y dot, 1: sum (x-y)^2), element 8*i + j.  Only 32 accumulators, 4 query
// rows and 8 candidate rows are live at a time (64 accumulators + both 8-row operand tiles cap the kernel
// at 2 waves/SIMD and a 1000-block grid then runs in two rounds).  lds_wave_reduce leaves element e in lanes
// 2e and 2e+1; even lanes write it.
template <bool NEED_G, bool NEED_D2>
__device__ __forceinline__ void half_tile_partials(const float4 (&x)[4], const float4 (&y)[8], float* red_half,
                                                   float* xp, int lane) {
    if constexpr (NEED_D2) {
        float acc[32];
#pragma unroll
        for (int j = 0; j < 8; ++j)      // candidate row outer: row j is needed only when its load has landed
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dx = x[i].x - y[j].x, dy = x[i].y - y[j].y, dz = x[i].z - y[j].z, dw = x[i].w - y[j].w;
                acc[i * 8 + j] = fmaf(dw, dw, fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
            }
        const float r = lds_wave_reduce<32>(acc, xp, lane);
        if ((lane & 1) == 0) red_half[64 + (lane >> 1)] = r;
    }
    __builtin_amdgcn_sched_barrier(0);  // do not overlap the passes: that doubles the live accumulators
    if constexpr (NEED_G) {
        float acc[32];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i * 8 + j] = dot4(x[i], y[j]);
        const float r = lds_wave_reduce<32>(acc, xp, lane);
        if ((lane & 1) == 0) red_half[(lane >> 1)] = r;
    }
    __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------------------------
// Phase 1: all three waves form the partial sums of (query doc, candidate doc) for every tile.
// ---------------------------------------------------------------------------------------------
template <int T, bool NEED_G, bool NEED_D2, bool BBOX, bool CENTER = false>
__device__ __forceinline__ void pair_partials(const float* qdoc, int q_avail, int q_box, const float* cdoc, int c_avail,
                                              int c_box, float* lds, int wave, int lane) {
    const int dofs = wave * 256 + lane * 4;
    // CENTER -- rows sharing a large common component (include/aspire_hip.h: ASPIRE_OT_FLAG_CENTER): the query's first row comes
    // off every row before anything is multiplied; distances and the bounding box's extent do not move, the expansion stops
    // cancelling.  A compile-time form: four more live registers push the plain kernels over their three-per-CU budget.
    const float4 mu = CENTER ? ld4(qdoc + dofs) : make_float4(0.f, 0.f, 0.f, 0.f);
    float* red = lds + wave * (T * T * 128);
    float* rednorm = lds + Lds<T>::kRed + wave * (T * 16);
    float* xp = lds + Lds<T>::kXp + wave * kXpWave;
    float4 mn = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    float4 mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if constexpr (T == 1) {
        // One tile: issue every load up front -- the query rows first (L2-resident, they land early), then the
        // candidate rows from HBM -- and let the accumulation start on y[0] while y[1..7] are still in flight
        // (the j-outer loops below wait per row with counted vmcnt).  Both query halves are resident, so the
        // second half starts without another exposed load latency.
        float4 x0[4], x1[4], y[8];
        load_rows<4, BBOX, CENTER>(x0, qdoc, 0, q_avail, dofs, q_box, mn, mx, mu);
        load_rows<4, BBOX, CENTER>(x1, qdoc, 4, q_avail, dofs, q_box, mn, mx, mu);
        load_rows<8, BBOX, CENTER>(y, cdoc, 0, c_avail, dofs, c_box, mn, mx, mu);
        half_tile_partials<NEED_G, NEED_D2>(x0, y, red, xp, lane);
        half_tile_partials<NEED_G, NEED_D2>(x1, y, red + 32, xp, lane);
        if (NEED_G) {
            float nrm[16];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                nrm[i] = sq4(x0[i]);
                nrm[4 + i] = sq4(x1[i]);
                nrm[8 + i] = sq4(y[i]);
                nrm[12 + i] = sq4(y[4 + i]);
            }
            const float r = lds_wave_reduce<16>(nrm, xp, lane);
            if ((lane & 3) == 0) rednorm[lane >> 2] = r;
        }
    } else {
    // whole 8 x 8 tiles past a document's rows are skipped (CSR documents: `avail` = the document's own length; consumers
    // never read those tiles' sums -- finish_pair, l2max_kernel).  Row norms: the query tile's with the first candidate tile,
    // the candidate tile's with the first query tile.
    const int Tq = min(T, max(1, (q_avail + 7) >> 3)), Tc = min(T, max(1, (c_avail + 7) >> 3));
#pragma unroll 1
    for (int tj = 0; tj < Tc; ++tj) {
        float4 y[8];
        load_rows<8, BBOX, CENTER>(y, cdoc, tj * 8, c_avail, dofs, c_box, mn, mx, mu);
#pragma unroll 1
        for (int ti = 0; ti < Tq; ++ti) {
            float nrm[16];  // |x_i|^2 of the 8 query rows, |y_j|^2 of the 8 candidate rows (first row / column of tiles only)
            const bool want_norms = NEED_G && (ti == 0 || tj == 0);
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                float4 x[4];
                load_rows<4, BBOX, CENTER>(x, qdoc, ti * 8 + half * 4, q_avail, dofs, q_box, mn, mx, mu);  // min/max idempotent
                half_tile_partials<NEED_G, NEED_D2>(x, y, red + (ti * T + tj) * 128 + half * 32, xp, lane);
                if (want_norms) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (half == 0) {
                            nrm[i] = sq4(x[i]);
                            nrm[8 + i] = sq4(y[i]);
                            nrm[12 + i] = sq4(y[4 + i]);
                        } else {
                            nrm[4 + i] = sq4(x[i]);
                        }
                    }
                }
            }
            if (want_norms) {
                const float r = lds_wave_reduce<16>(nrm, xp, lane);
                const int e = lane >> 2;                  // 0 .. 7: query rows of tile ti, 8 .. 15: candidate rows of tile tj
                if ((lane & 3) == 0 && (e < 8 ? tj == 0 : ti == 0)) rednorm[(e < 8 ? ti : tj) * 16 + e] = r;
            }
        }
    }
    }
    if (BBOX) {
        const float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z, dw = mx.w - mn.w;
        const float s = wave_sum(fmaf(dw, dw, fmaf(dz, dz, fmaf(dy, dy, dx * dx))));
        if (lane == 0) lds[Lds<T>::kRed + Lds<T>::kNorm + wave] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// max-sim kernel (A9)
// ---------------------------------------------------------------------------------------------
template <int T>
__global__ void __launch_bounds__(kBlock, 3) l2max_kernel(ScoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t c_idx = blockIdx.x;
    const int c_len = a.c.len[c_idx];
    const int c_avail = a.c.ext > 0 ? a.c.ext : c_len;
    const float* cdoc = a.c.rows + (size_t)a.c.start[c_idx] * kD;
    // one query per candidate (PAIRED: the candidate's own index; MAPPED: its job's) or a block of queries (CROSS)
    const bool one_q = a.pairing != ASPIRE_PAIR_CROSS;
    const int64_t q_begin = a.pairing == ASPIRE_PAIR_PAIRED ? c_idx : a.pairing == kPairMapped ? (int64_t)a.qmap[c_idx]
                                                                                                : (int64_t)blockIdx.y * a.q_per_block;
    const int64_t q_end = one_q ? q_begin + 1 : min(a.q.n, q_begin + a.q_per_block);
    const int li = lane >> 3, lj = lane & 7;
    for (int64_t q_idx = q_begin; q_idx < q_end; ++q_idx) {
        const int q_len = a.q.len[q_idx];
        const int q_avail = a.q.ext > 0 ? a.q.ext : q_len;
        const float* qdoc = a.q.rows + (size_t)a.q.start[q_idx] * kD;
        const bool mm = use_mm_formula(a.cdist_mode, q_avail, c_avail);
        if (mm) {
            pair_partials<T, true, false, false>(qdoc, q_avail, 0, cdoc, c_avail, 0, lds, wave, lane);
        } else {
            pair_partials<T, false, true, false>(qdoc, q_avail, 0, cdoc, c_avail, 0, lds, wave, lane);
        }
        __syncthreads();
        if (wave == 0) {
            const int64_t p = one_q ? c_idx : q_idx * a.c.n + c_idx;
            float negv[T][T];
            bool val[T][T];
#pragma unroll
            for (int ta = 0; ta < T; ++ta)
#pragma unroll
                for (int tb = 0; tb < T; ++tb) {
                    const int i = ta * 8 + li, j = tb * 8 + lj;
                    const float* r = lds + (ta * T + tb) * 128;
                    float d2;
                    if (T > 1 && (ta * 8 >= q_avail || tb * 8 >= c_avail)) {
                        d2 = 0.f;                       // a tile pair_partials skipped: no sums in LDS, no valid entry
                    } else if (mm) {
                        float g = 0.f, xx = 0.f, yy = 0.f;
#pragma unroll
                        for (int w = 0; w < kWaves; ++w) {
                            g += r[w * T * T * 128 + lane];
                            xx += lds[Lds<T>::kRed + w * T * 16 + ta * 16 + li];
                            yy += lds[Lds<T>::kRed + w * T * 16 + tb * 16 + 8 + lj];
                        }
                        const float sqv = fmaf(-2.f, g, xx) + yy, ns = xx + yy;
                        d2 = fmaxf(sqv, 0.f);
                        // (round 6: where the expansion cancels the entry comes from the exact sum under this formula too -- one rule in every kernel family:
                        // include/aspire_hip.h, SHARED SENTENCES.  Rare: the lane walks the two rows itself)
                        if (i < q_len && j < c_len && sqv < 1e-4f * ns * ns) {
                            const float* xr = qdoc + (size_t)i * kD;
                            const float* yr = cdoc + (size_t)j * kD;
                            float s0 = 0.f, s1 = 0.f;
                            for (int d = 0; d < kD; d += 8) {
                                const float4 u0 = ld4(xr + d), v0 = ld4(yr + d), u1 = ld4(xr + d + 4), v1 = ld4(yr + d + 4);
                                const float a0 = u0.x - v0.x, a1 = u0.y - v0.y, a2 = u0.z - v0.z, a3 = u0.w - v0.w;
                                const float b0 = u1.x - v1.x, b1 = u1.y - v1.y, b2 = u1.z - v1.z, b3 = u1.w - v1.w;
                                s0 = fmaf(a3, a3, fmaf(a2, a2, fmaf(a1, a1, fmaf(a0, a0, s0))));
                                s1 = fmaf(b3, b3, fmaf(b2, b2, fmaf(b1, b1, fmaf(b0, b0, s1))));
                            }
                            d2 = s0 + s1;
                        }
                    } else {
                        d2 = 0.f;
#pragma unroll
                        for (int w = 0; w < kWaves; ++w) d2 += r[w * T * T * 128 + 64 + lane];
                    }
                    negv[ta][tb] = -sqrtf(d2);
                    val[ta][tb] = i < q_len && j < c_len;
                    if (a.out_pairsims && i < a.q.ext && j < a.c.ext)
                        a.out_pairsims[(p * a.q.ext + i) * a.c.ext + j] =
                            negv[ta][tb] + ((val[ta][tb] || a.agg == ASPIRE_AGG_ATTENTION) ? 0.f : -10e8f);
                }
            float score;
            if (a.agg == ASPIRE_AGG_MAX) {
                float best = -INFINITY;
#pragma unroll
                for (int ta = 0; ta < T; ++ta)
#pragma unroll
                    for (int tb = 0; tb < T; ++tb)
                        if (val[ta][tb]) best = fmaxf(best, negv[ta][tb]);
                score = wave_max(best);
            } else if (a.agg == ASPIRE_AGG_TOP2) {
                // torch.topk(k=2) over the padded block: masked entries take part with -cdist - 10e8
                float m1 = -INFINITY, m2 = -INFINITY;
                auto push = [&](float v) {
                    m2 = fmaxf(m2, fminf(m1, v));
                    m1 = fmaxf(m1, v);
                };
#pragma unroll
                for (int ta = 0; ta < T; ++ta)
#pragma unroll
                    for (int tb = 0; tb < T; ++tb) {
                        const int i = ta * 8 + li, j = tb * 8 + lj;
                        if (val[ta][tb]) push(negv[ta][tb]);
                        else if (i < a.q.ext && j < a.c.ext) push(negv[ta][tb] + -10e8f);
                    }
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) {
                    const float o1 = __shfl_xor(m1, m), o2 = __shfl_xor(m2, m);
                    m2 = fmaxf(fminf(m1, o1), fmaxf(m2, o2));
                    m1 = fmaxf(m1, o1);
                }
                if (m2 == -INFINITY) m2 = -10e8f;   // no padded extent and a 1 x 1 pair
                score = m1 + m2;
            } else {
                // masked 2-D soft-max of -d / temp over the valid block, then sum p * (-d)
                const float temp = (float)a.temp;
                float mx = -INFINITY;
#pragma unroll
                for (int ta = 0; ta < T; ++ta)
#pragma unroll
                    for (int tb = 0; tb < T; ++tb)
                        if (val[ta][tb]) mx = fmaxf(mx, negv[ta][tb] / temp);
                mx = wave_max(mx);
                float e[T][T], se = 0.f, sn = 0.f;
#pragma unroll
                for (int ta = 0; ta < T; ++ta)
#pragma unroll
                    for (int tb = 0; tb < T; ++tb) {
                        e[ta][tb] = val[ta][tb] ? expf(negv[ta][tb] / temp - mx) : 0.f;
                        se += e[ta][tb];
                        sn = fmaf(e[ta][tb], negv[ta][tb], sn);
                    }
                se = wave_sum(se);
                sn = wave_sum(sn);
                score = sn / se;
                if (a.out_plan) {
#pragma unroll
                    for (int ta = 0; ta < T; ++ta)
#pragma unroll
                        for (int tb = 0; tb < T; ++tb) {
                            const int i = ta * 8 + li, j = tb * 8 + lj;
                            if (i < a.q.ext && j < a.c.ext) a.out_plan[(p * a.q.ext + i) * a.c.ext + j] = e[ta][tb] / se;
                        }
                }
            }
            if (lane == 0) a.scores[p] = score;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// otAspire kernel (A5-A8)
// ---------------------------------------------------------------------------------------------
template <int T>
struct PairState {
    float cost[T][T];  // geomloss cost: sqrt(max(|x|^2 - 2 x.y + |y|^2, 1e-8))
    float neg[T][T];   // -cdist (torch formula)
};

// After the three waves' partial sums of one pair met in LDS: all 192 threads finish the entries
// (sum of partials, both L2 formulas) and store them to the pair's workspace slot.
template <int T, bool DIRECT = true>
__device__ __forceinline__ void finish_pair(const float* lds, bool mm, bool want_diam, const PairWs<T>& ws, int64_t slot,
                                            const float* qdoc = nullptr, const float* cdoc = nullptr, int q_len = 0, int c_len = 0,
                                            int q_avail = 8 * T, int c_avail = 8 * T) {
    for (int e = threadIdx.x; e < 64 * T * T; e += kBlock) {
        const int tile = e >> 6, l = e & 63, ta = tile / T, tb = tile % T, li = l >> 3, lj = l & 7;
        const float* r = lds + tile * 128;
        if (T > 1 && (ta * 8 >= q_avail || tb * 8 >= c_avail)) {
            // pair_partials skipped this tile (no row of one side reaches it); the solvers mask it
            const int64_t o = slot * (64 * T * T) + (ta * 8 + li) * (8 * T) + tb * 8 + lj;
            ws.cost[o] = 1.f;
            ws.neg[o] = -1.f;
            continue;
        }
        float g = 0.f, d2 = 0.f, xx = 0.f, yy = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            g += r[w * T * T * 128 + l];
            if (DIRECT) d2 += r[w * T * T * 128 + 64 + l];
            xx += lds[Lds<T>::kRed + w * T * 16 + ta * 16 + li];
            yy += lds[Lds<T>::kRed + w * T * 16 + tb * 16 + 8 + lj];
        }
        const float sq = fmaf(-2.f, g, xx) + yy;
        const int64_t o = slot * (64 * T * T) + (ta * 8 + li) * (8 * T) + tb * 8 + lj;
        // geomloss's cost: its expansion -- except where that cancels (the test the streaming kernels use), where the exact sum
        // stands in: what the reference's own formula gives in float64 (in fp32 it returns the square root of rounding noise there)
        float costv = sqrtf(fmaxf(sq, 1e-8f));
        if constexpr (DIRECT) {
            const float ns = xx + yy;
            const bool cancels = sq < 1e-4f * ns * ns;
            if (cancels) costv = sqrtf(fmaxf(d2, 1e-8f));
            ws.cost[o] = costv;
            ws.neg[o] = (mm && !cancels) ? -sqrtf(fmaxf(sq, 0.f)) : -sqrtf(d2);      // (round 6: a cancelling entry from the exact sum under either formula)
        } else {
            // only x.y was accumulated (see pair_cost1_kernel): -cdist from the expansion, except where it cancels
            const int i = ta * 8 + li, j = tb * 8 + lj;
            const float ns = xx + yy;
            float negv = -sqrtf(fmaxf(sq, 0.f));
            if (i < q_len && j < c_len && sq < 1e-4f * ns * ns) {   // rare: this thread walks the two rows itself
                const float* xr = qdoc + (size_t)i * kD;
                const float* yr = cdoc + (size_t)j * kD;
                float s0 = 0.f, s1 = 0.f;
                for (int d = 0; d < kD; d += 8) {
                    const float4 u0 = ld4(xr + d), v0 = ld4(yr + d), u1 = ld4(xr + d + 4), v1 = ld4(yr + d + 4);
                    const float a0 = u0.x - v0.x, a1 = u0.y - v0.y, a2 = u0.z - v0.z, a3 = u0.w - v0.w;
                    const float b0 = u1.x - v1.x, b1 = u1.y - v1.y, b2 = u1.z - v1.z, b3 = u1.w - v1.w;
                    s0 = fmaf(a3, a3, fmaf(a2, a2, fmaf(a1, a1, fmaf(a0, a0, s0))));
                    s1 = fmaf(b3, b3, fmaf(b2, b2, fmaf(b1, b1, fmaf(b0, b0, s1))));
                }
                negv = -sqrtf(s0 + s1);
                costv = sqrtf(fmaxf(s0 + s1, 1e-8f));
            }
            ws.cost[o] = costv;
            ws.neg[o] = negv;
        }
    }
    if (want_diam && threadIdx.x == 0) {
        const float* dd = lds + Lds<T>::kRed + Lds<T>::kNorm;
        ws.diam2[slot] = dd[0] + dd[1] + dd[2];
    }
}

// Lane <-> entry map of the one-solve-per-wave kernel.  T > 1: (li, lj) = (lane >> 3, lane & 7).  T == 1 spreads the
// two cross-row lane bits over BOTH index directions -- lj = lane bits {0, 1, 4}, li = lane bits {2, 3, 5} -- so that
// each of the two reductions of a Sinkhorn step is two DPP levels plus ONE v_permlane*_swap, instead of three DPP
// levels for the rows and one DPP level plus two swaps (mov + swap + add each, the longest links of the dependent
// chain) for the columns.
template <int T>
__device__ __forceinline__ void lane_ij(int lane, int& li, int& lj) {
    if constexpr (T == 1) {
        lj = (lane & 3) | ((lane >> 2) & 4);
        li = ((lane >> 2) & 3) | ((lane >> 3) & 4);
    } else {
        li = lane >> 3;
        lj = lane & 7;
    }
}
template <int T>
__device__ __forceinline__ float rsum8(float v) {   // all-reduce over the 8 lanes that share li
    if constexpr (T == 1) {
        v += lane_xor<1>(v);
        v += lane_xor<2>(v);
        return swap_add<16>(v, v);
    } else {
        return row8_sum(v);
    }
}
template <int T>
__device__ __forceinline__ float csum8(float v) {   // all-reduce over the 8 lanes that share lj
    if constexpr (T == 1) {
        v += dpp_mov<0x124>(v, v);   // row_ror:4
        v += dpp_mov<0x128>(v, v);   // row_ror:8
        return swap_add<32>(v, v);
    } else {
        return col8_sum(v);
    }
}
template <int T>
__device__ __forceinline__ float rmax8(float v) {
    if constexpr (T == 1) {
        v = fmaxf(v, lane_xor<1>(v));
        v = fmaxf(v, lane_xor<2>(v));
        return fmaxf(v, lane_xor<16>(v));
    } else {
        return row8_max(v);
    }
}
template <int T>
__device__ __forceinline__ float cmax8(float v) {
    if constexpr (T == 1) {
        v = fmaxf(v, dpp_mov<0x124>(v, v));
        v = fmaxf(v, dpp_mov<0x128>(v, v));
        return fmaxf(v, lane_xor<32>(v));
    } else {
        return col8_max(v);
    }
}

template <int T>
__device__ __forceinline__ void load_pair(PairState<T>& s, const PairWs<T>& ws, int64_t slot, int lane, bool cost_from_neg = false) {
    int li, lj;
    lane_ij<T>(lane, li, lj);
#pragma unroll
    for (int ta = 0; ta < T; ++ta)
#pragma unroll
        for (int tb = 0; tb < T; ++tb) {
            const int64_t o = slot * (64 * T * T) + (ta * 8 + li) * (8 * T) + tb * 8 + lj;
            s.neg[ta][tb] = ws.neg[o];
            // (the matrix-pipe cost tiles store -cdist only: sqrt(max(sq, 1e-8)) = max(sqrt(max(sq, 0)), sqrt(1e-8)) bit for bit)
            s.cost[ta][tb] = cost_from_neg ? fmaxf(-s.neg[ta][tb], __builtin_sqrtf(1e-8f)) : ws.cost[o];
        }
}

#ifdef ASPIRE_PHASE_CLOCK
// debug build only (tools/k1phases.py): cycle stamps of one wave into the buffer set by aspire_debug_k1_buffer
static __device__ long long* g_k1dbg = nullptr;
#define PHASE_STAMP(k)                                                                              \
    do {                                                                                            \
        if (g_k1dbg && blockIdx.x == 3 && (threadIdx.x >> 6) == 1 && lane == 0 && stamp_ok)        \
            g_k1dbg[32 + (k)] = (long long)__builtin_readcyclecounter();                            \
    } while (0)
#else
#define PHASE_STAMP(k) \
    do {               \
    } while (0)
#endif

template <int T>
__device__ void sinkhorn_pair(const ScoreArgs& a, const PairState<T>& s, int q_len, int c_len, float diam, int64_t p,
                              int lane) {
    int li, lj;
    lane_ij<T>(lane, li, lj);
    const bool stamp_ok = true;
    (void)stamp_ok;
    PHASE_STAMP(3);
    bool rv[T], cv[T];  // row / column validity
#pragma unroll
    for (int t = 0; t < T; ++t) {
        rv[t] = t * 8 + li < q_len;
        cv[t] = t * 8 + lj < c_len;
    }
    // ---- marginals (pair_distances.py:57-60): softmax over sentences of the best match / temp -----
    const float temp = (float)a.temp;
    float la[T], lb[T], wa[T], wb[T];  // log-weights and weights
    {
        float qm[T], cm[T];
#pragma unroll
        for (int ta = 0; ta < T; ++ta) {
            float m = kNegBig;
#pragma unroll
            for (int tb = 0; tb < T; ++tb) m = fmaxf(m, (rv[ta] && cv[tb]) ? s.neg[ta][tb] : kNegBig);
            qm[ta] = rmax8<T>(m) / temp;
        }
#pragma unroll
        for (int tb = 0; tb < T; ++tb) {
            float m = kNegBig;
#pragma unroll
            for (int ta = 0; ta < T; ++ta) m = fmaxf(m, (rv[ta] && cv[tb]) ? s.neg[ta][tb] : kNegBig);
            cm[tb] = cmax8<T>(m) / temp;
        }
        float mq = kNegBig, mc = kNegBig;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            mq = fmaxf(mq, rv[t] ? qm[t] : kNegBig);
            mc = fmaxf(mc, cv[t] ? cm[t] : kNegBig);
        }
        mq = cmax8<T>(mq);  // rows are spread over lane bits 3-5
        mc = rmax8<T>(mc);  // columns over lane bits 0-2
        float sq = 0.f, sc = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            sq += rv[t] ? fast_exp(qm[t] - mq) : 0.f;
            sc += cv[t] ? fast_exp(cm[t] - mc) : 0.f;
        }
        const float lsq = fast_log(csum8<T>(sq)), lsc = fast_log(rsum8<T>(sc));
#pragma unroll
        for (int t = 0; t < T; ++t) {
            // log_softmax(...).exp(), then geomloss log_weights: log(a), a <= 0 -> -100000
            wa[t] = rv[t] ? fast_exp(qm[t] - mq - lsq) : 0.f;
            wb[t] = cv[t] ? fast_exp(cm[t] - mc - lsc) : 0.f;
            la[t] = wa[t] > 0.f ? fast_log(wa[t]) : -100000.f;
            lb[t] = wb[t] > 0.f ? fast_log(wb[t]) : -100000.f;
        }
    }
    PHASE_STAMP(4);
    // ---- epsilon schedule (geomloss epsilon_schedule, p = 1) --------------------------------------
    //   [diam] + [exp(e) for e in arange(log diam, log blur, log scaling)] + [blur]
    float ldf;                                    // log2 units
    const int n_mid = schedule_mid_steps(a, diam, ldf);
    const float lscf = a.log2_scaling;
    const float eps_last = (float)a.blur;
    PHASE_STAMP(5);

    float f[T], g[T];
    // out_i = -eps * logsumexp_j(hb_j - C_ij/eps) over valid j   (rows; `shift` = the caller's estimate of
    // -logsumexp, see step()).  With EXACT the shift is the true maximum, as torch.logsumexp does.
    auto lse_rows = [&](float eps, const float (&qc)[T][T], const float (&h)[T], const float (&shift)[T], bool exact,
                        float (&out)[T]) {
#pragma unroll
        for (int ta = 0; ta < T; ++ta) {
            float tv[T];
#pragma unroll
            for (int tb = 0; tb < T; ++tb) tv[tb] = cv[tb] ? h[tb] - qc[ta][tb] : kNegBig;
            float m = -shift[ta];
            if (exact) {
                m = tv[0];
#pragma unroll
                for (int tb = 1; tb < T; ++tb) m = fmaxf(m, tv[tb]);
                m = rmax8<T>(m);
            }
            float sum = 0.f;
#pragma unroll
            for (int tb = 0; tb < T; ++tb) sum += fast_exp(tv[tb] - m);
            out[ta] = -eps * (m + fast_log(rsum8<T>(sum)));
        }
    };
    auto lse_cols = [&](float eps, const float (&qc)[T][T], const float (&h)[T], const float (&shift)[T], bool exact,
                        float (&out)[T]) {
#pragma unroll
        for (int tb = 0; tb < T; ++tb) {
            float tv[T];
#pragma unroll
            for (int ta = 0; ta < T; ++ta) tv[ta] = rv[ta] ? h[ta] - qc[ta][tb] : kNegBig;
            float m = -shift[tb];
            if (exact) {
                m = tv[0];
#pragma unroll
                for (int ta = 1; ta < T; ++ta) m = fmaxf(m, tv[ta]);
                m = cmax8<T>(m);
            }
            float sum = 0.f;
#pragma unroll
            for (int ta = 0; ta < T; ++ta) sum += fast_exp(tv[ta] - m);
            out[tb] = -eps * (m + fast_log(csum8<T>(sum)));
        }
    };
    // One symmetric Sinkhorn update at `eps` (reps = 1/eps):
    //   gt_j = -eps*LSE_i(la_i + f_i/eps - C_ij/eps),  ft_i = -eps*LSE_j(lb_j + g_j/eps - C_ij/eps)
    // The log-sum-exps are stabilised by shifting with -g_j/eps resp. -f_i/eps -- the previous
    // potentials, which ARE (-eps times) the previous log-sum-exps -- instead of the running maximum:
    // mathematically identical, the sum then sits near 1, and six dependent cross-lane max steps leave
    // the critical path.  If a sum ever leaves [1e-30, 1e30] (it cannot while potentials move by less
    // than ~69*eps per step) the step is redone with the exact maximum.
    auto step = [&](float eps, float reps, bool averaged, bool exact) {
        float qc[T][T], qf[T], qg[T], ha[T], hb[T], ft[T], gt[T];
#pragma unroll
        for (int ta = 0; ta < T; ++ta)
#pragma unroll
            for (int tb = 0; tb < T; ++tb) qc[ta][tb] = div_r(s.cost[ta][tb], eps, reps);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            qf[t] = div_r(f[t], eps, reps);
            qg[t] = div_r(g[t], eps, reps);
            ha[t] = la[t] + qf[t];
            hb[t] = lb[t] + qg[t];
        }
        lse_cols(eps, qc, ha, qg, exact, gt);
        lse_rows(eps, qc, hb, qf, exact, ft);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            g[t] = averaged ? 0.5f * (g[t] + gt[t]) : gt[t];
            f[t] = averaged ? 0.5f * (f[t] + ft[t]) : ft[t];
        }
    };
    // The same update with the critical path cut to the bone (the kernel's time at ~1000 pairs IS 75 x this
    // chain).  Everything is in base 2 (r2 = log2(e)/eps, rounded once from float64, so v_exp_f32 / v_log_f32 need no
    // scaling multiplies) and the state carried from step to step is phi_ij = f_i + g_j - C_ij itself:
    //     sum_j b_j 2^(phi_ij r2) = exp((f_i - ft_i)/eps)        sum_i a_i 2^(phi_ij r2) = exp((g_j - gt_j)/eps)
    // (the log-sum-exps shifted by the previous potentials), so with LR_i, LC_j the log2 of those sums the averaged
    // update is  f_i -= h LR_i,  g_j -= h LC_j,  phi_ij -= h (LR_i + LC_j),  h = eps ln2 / 2  (eps ln2 for the final
    // extrapolation).  The dependent chain per step is fma - exp2 - reduce - log2 - add - fma; f and g are updated
    // off that chain.  phi's rounding matters only where |phi| is small (the transport plan's support), where its
    // ulp is far below the tolerance.  Measured against a float64 evaluation this is as accurate as the fp32 CPU
    // path (tools/oterr.py).
    float la2[T], lb2[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        la2[t] = la[t] * kLog2e;
        lb2[t] = lb[t] * kLog2e;
    }
    float phi[T][T];
    auto phi_init = [&]() {
#pragma unroll
        for (int ta = 0; ta < T; ++ta)
#pragma unroll
            for (int tb = 0; tb < T; ++tb)
                phi[ta][tb] = (rv[ta] && cv[tb]) ? (f[ta] + g[tb]) - s.cost[ta][tb] : -__builtin_inff();   // masked slots may hold stale bits
    };
    float pad1[T][T];
#pragma unroll
    for (int ta = 0; ta < T; ++ta)
#pragma unroll
        for (int tb = 0; tb < T; ++tb) pad1[ta][tb] = (rv[ta] && cv[tb]) ? 0.f : 1.f;
    // One exponential per entry and no potentials in the loop: E_ij = 2^(phi_ij r2) serves both sums with the marginal
    // weights as plain factors (sum_j b_j E_ij, sum_i a_i E_ij), and since sum a = sum b = 1 the result
    //     <a, f> + <b, g> = sum_ij a_i b_j (f_i + g_j) = sum_ij a_i b_j (phi_ij + C_ij)
    // needs phi alone -- f and g are never formed on this path.  With one entry per lane (T == 1) the update
    // h (LR_i + LC_j) = h log2(rowsum_i * colsum_j) is ONE logarithm: two transcendentals per entry and step instead
    // of four (they issue at quarter rate: the lone launch's dependent chain is unchanged, but overlapped queries share
    // the SIMDs' issue slots -- bench.py 110 -> 115 M alignments/s; 1 x 125 x 20 35.6 -> 31.4 us).  Masked entries carry phi = -inf (E = 0, out of every sum)
    // and a +1 on their own (empty) sums keeps their logarithm at 0.
    // A step at temperature eps:  E = 2^(phi r2),  phi -= h log2(rowsum colsum),  r2 = log2(e)/eps,  h = eps ln2 / 2
    // (eps ln2 for the final, un-averaged step).  Through the geometric part of the schedule the constants of the next
    // step follow from this one's by the factor scaling (r2 /= scaling, h *= scaling): two multiplies off the dependent
    // chain instead of two v_readlane broadcasts of a per-lane table, and nothing for the loop to index, so it
    // unrolls freely.  (Carrying psi = phi r2 instead saves one more multiply per step but rescales the state 77 times:
    // mean error against float64 8.6e-6 instead of 5.4e-6.)
    float r2v = 0.f, hv = 0.f;       // wave-uniform, kept in vector registers: gfx950 has no scalar float multiply
    auto step2 = [&](float r2_mul, float h_mul) {
        if constexpr (T == 1) {
            // One entry per lane.  The column chain and the row chain (two DPP levels and one v_permlane*_swap each,
            // see lane_ij) are independent; a single wave issues in order, so they are interleaved level by level
            // here and pinned with sched_barrier -- a cross-lane op costs 17-26 cycles of dependent latency
            // (tools: build/dbg/lat.hip), overlapped they cost it once, not twice.
            const float e = __builtin_amdgcn_exp2f(phi[0][0] * r2v);
            float sc = wa[0] * e;
            float sr = wb[0] * e;
            // opaque to the optimizer: left alone it contracts a * b + dpp(a * b) into mov_dpp + fma, two issue slots
            // per step more than mul + add_dpp
            asm volatile("" : "+v"(sc), "+v"(sr));
            __builtin_amdgcn_sched_barrier(0);
            sc += dpp_mov<0x124>(sc, sc);     // columns: lane bits 2, 3 (row_ror:4, row_ror:8), then bit 5
            sr += lane_xor<1>(sr);            // rows:    lane bits 0, 1 (quad_perm), then bit 4
            __builtin_amdgcn_sched_barrier(0);
            sc += dpp_mov<0x128>(sc, sc);
            sr += lane_xor<2>(sr);
            __builtin_amdgcn_sched_barrier(0);
            sc = swap_add<32>(sc, sc);
            sr = swap_add<16>(sr, sr);
            __builtin_amdgcn_sched_barrier(0);
            phi[0][0] = fmaf(-hv, __builtin_amdgcn_logf(fmaf(sc, sr, pad1[0][0])), phi[0][0]);
        } else {
            float e[T][T], lr[T], lc[T];
#pragma unroll
            for (int ta = 0; ta < T; ++ta)
#pragma unroll
                for (int tb = 0; tb < T; ++tb) e[ta][tb] = __builtin_amdgcn_exp2f(phi[ta][tb] * r2v);
#pragma unroll
            for (int tb = 0; tb < T; ++tb) {   // columns
                float sum = 0.f;
#pragma unroll
                for (int ta = 0; ta < T; ++ta) sum = fmaf(wa[ta], e[ta][tb], sum);
                lc[tb] = __builtin_amdgcn_logf(csum8<T>(sum) + (cv[tb] ? 0.f : 1.f));
            }
#pragma unroll
            for (int ta = 0; ta < T; ++ta) {   // rows
                float sum = 0.f;
#pragma unroll
                for (int tb = 0; tb < T; ++tb) sum = fmaf(wb[tb], e[ta][tb], sum);
                lr[ta] = __builtin_amdgcn_logf(rsum8<T>(sum) + (rv[ta] ? 0.f : 1.f));
            }
#pragma unroll
            for (int ta = 0; ta < T; ++ta)
#pragma unroll
                for (int tb = 0; tb < T; ++tb) phi[ta][tb] = fmaf(-hv, lr[ta] + lc[tb], phi[ta][tb]);
        }
        r2v *= r2_mul;      // the next step's constants
        hv *= h_mul;
    };
    // The whole annealing loop.  exact = false uses the shifted log-sum-exp; an overflowed / vanished
    // sum turns into inf / nan that then sticks to the potentials, so ONE finiteness test at the end
    // (instead of a compare + branch on every step's critical path) decides whether the solve has to be
    // repeated with exact maxima.
    auto solve = [&](bool exact) {
        if (exact) {   // initialisation at eps_s[0] = diam: softmin of the bare log-weights, exact maximum
            const float reps = rcp_refined(diam);
            float qc[T][T], zero[T];
#pragma unroll
            for (int ta = 0; ta < T; ++ta)
#pragma unroll
                for (int tb = 0; tb < T; ++tb) qc[ta][tb] = div_r(s.cost[ta][tb], diam, reps);
#pragma unroll
            for (int t = 0; t < T; ++t) zero[t] = 0.f;
            lse_cols(diam, qc, la, zero, true, g);
            lse_rows(diam, qc, lb, zero, true, f);
            step(diam, reps, true, true);
        } else {
            // the same initialisation without a max shift (the largest weight of a probability vector over <= 32
            // atoms is >= 1/32 and C/diam <= ~1, so the sums stay in range), weights as plain factors, then the
            // first averaged step at eps = diam in the phi form like all the others
            const float r2d = kLog2e * rcp_refined(diam), eln2d = diam * kLn2;
            float rs[T], cs[T];
#pragma unroll
            for (int t = 0; t < T; ++t) rs[t] = cs[t] = 0.f;
#pragma unroll
            for (int ta = 0; ta < T; ++ta)
#pragma unroll
                for (int tb = 0; tb < T; ++tb) {
                    const float k0 = (rv[ta] && cv[tb]) ? __builtin_amdgcn_exp2f(-s.cost[ta][tb] * r2d) : 0.f;
                    rs[ta] = fmaf(wb[tb], k0, rs[ta]);
                    cs[tb] = fmaf(wa[ta], k0, cs[tb]);
                }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                f[t] = -eln2d * __builtin_amdgcn_logf(rsum8<T>(rs[t]));
                g[t] = -eln2d * __builtin_amdgcn_logf(csum8<T>(cs[t]));
            }
            phi_init();
            // steps: eps = diam, then the n_mid annealed values diam scaling^k (k = 0 .. n_mid - 1; fp32 -- a relative
            // 1e-6 on an intermediate temperature moves the result by far less than the tolerance, and the float64 exp
            // cost as much as ten annealing steps), then blur, then the final un-averaged step at blur
            const float rho_s = __builtin_amdgcn_exp2f(-lscf), scal = __builtin_amdgcn_exp2f(lscf);     // 1 / scaling, scaling
            const float last_eps = n_mid > 0 ? __builtin_amdgcn_exp2f(fmaf((float)(n_mid - 1), lscf, ldf)) : diam;
            const float rho_b = last_eps * rcp_refined(eps_last), inv_rho_b = eps_last * rcp_refined(last_eps);
            const int n_s = __builtin_amdgcn_readfirstlane(n_mid);                          // wave-uniform: scalar loop control
            r2v = r2d;
            hv = 0.5f * eln2d;
            step2(n_s > 0 ? 1.f : rho_b, n_s > 0 ? 1.f : inv_rho_b);                        // at diam
            int k = 1;
            for (; k + 4 <= n_s; k += 4) {      // unrolled by hand (the pinned schedule inside step2 defeats #pragma unroll)
                step2(rho_s, scal);
                step2(rho_s, scal);
                step2(rho_s, scal);
                step2(rho_s, scal);
            }
            for (; k < n_s; ++k) step2(rho_s, scal);
            if (n_s > 0) step2(rho_b, inv_rho_b);                                            // the last annealed value -> blur
            // the two steps at blur with exactly rounded constants (drops the drift of the running products)
            r2v = kLog2e * rcp_refined(eps_last);
            hv = 0.5f * eps_last * kLn2;
            step2(1.f, 2.f);                                                                 // at blur, averaged
            step2(1.f, 1.f);                                                                 // at blur, final (h doubled)
            return;
        }

        // exact path only from here: float64 schedule exactly as numpy builds geomloss's, lane k of a chunk evaluates
        // eps_{base+k} and the per-step constants are broadcast with v_readlane
        const double ld = log((double)diam);
        for (int base = 0; base < n_mid; base += 64) {
            const float my_eps = (float)exp(ld + (double)(base + lane) * a.log_scaling);
            const float my_reps = rcp_refined(my_eps);
            const int cnt = min(64, n_mid - base);
            for (int k = 0; k < cnt; ++k) {
                const float eps = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_eps), k));
                const float reps = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_reps), k));
                step(eps, reps, true, true);
            }
        }
        const float rb = rcp_refined(eps_last);
        step(eps_last, rb, true, true);
        step(eps_last, rb, false, true);  // last extrapolation: simultaneous, not averaged
    };
    PHASE_STAMP(6);
    bool phi_live = true;
    solve(false);
    // <a, f> + <b, g> from phi alone (see step2); an overflowed / vanished sum anywhere has turned into inf / nan that
    // reaches this total, so its finiteness is the one test that decides whether the solve is repeated exactly.
    float fast_total;
    {
        float acc = 0.f;
#pragma unroll
        for (int ta = 0; ta < T; ++ta)
#pragma unroll
            for (int tb = 0; tb < T; ++tb)
                acc += (rv[ta] && cv[tb]) ? (wa[ta] * wb[tb]) * (phi[ta][tb] + s.cost[ta][tb]) : 0.f;
        fast_total = wave_sum(acc);
        if (__builtin_expect(!(fabsf(fast_total) < 1e30f), 0)) {
            solve(true);
            phi_live = false;
        }
    }
    const float rb = rcp_refined(eps_last);
    PHASE_STAMP(7);

    // ---- outputs ------------------------------------------------------------------------------
    float score;
    if (a.want != ASPIRE_OT_PLAN_SIM) {
        if (phi_live) {
            score = fast_total;
        } else {
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                acc += (lj == 0 && rv[t]) ? wa[t] * f[t] : 0.f;
                acc += (li == 0 && cv[t]) ? wb[t] * g[t] : 0.f;
            }
            score = wave_sum(acc);
        }
        if (a.want == ASPIRE_OT_SIMILARITY) score = -score;
    } else {
        score = 0.f;
    }
    const bool dump = a.out_plan != nullptr || a.out_pairsims != nullptr;
    if (a.want == ASPIRE_OT_PLAN_SIM || dump) {
        float acc = 0.f;
#pragma unroll
        for (int ta = 0; ta < T; ++ta)
#pragma unroll
            for (int tb = 0; tb < T; ++tb) {
                const bool valid = rv[ta] && cv[tb];
                const float negm = valid ? s.neg[ta][tb] : 0.f;
                // f_i + g_j - dist_ij: after the fast solve phi = f + g - C is at hand with the rounding of ITS
                // magnitude (small on the plan's support) rather than of f's and g's, and C - dist is an exact
                // difference of two nearby floats -- eps = 0.05 amplifies this exponent's error ~20x.
                const float expo = !valid ? 0.f : phi_live ? phi[ta][tb] + (s.cost[ta][tb] + negm) : (f[ta] + g[tb]) + negm;
                const float plan = fast_exp(div_r(expo, eps_last, rb)) * (wa[ta] * wb[tb]);
                acc += plan * negm;
                const int i = ta * 8 + li, j = tb * 8 + lj;
                if (dump && i < a.q.ext && j < a.c.ext) {
                    const int64_t o = (p * a.q.ext + i) * a.c.ext + j;
                    if (a.out_plan) a.out_plan[o] = plan;
                    if (a.out_pairsims) a.out_pairsims[o] = negm;
                }
            }
        if (a.want == ASPIRE_OT_PLAN_SIM) score = wave_sum(acc);
    }
    // a document longer than the launcher's tile bound would have been truncated silently: poison it
    if (q_len > 8 * T || c_len > 8 * T) score = __builtin_nanf("");
    if (lane == 0) a.scores[p] = score;
    if (a.out_qdistr) {
#pragma unroll
        for (int t = 0; t < T; ++t)
            if (lj == 0 && t * 8 + li < a.q.ext) a.out_qdistr[p * a.q.ext + t * 8 + li] = wa[t];
    }
    if (a.out_cdistr) {
#pragma unroll
        for (int t = 0; t < T; ++t)
            if (li == 0 && t * 8 + lj < a.c.ext) a.out_cdistr[p * a.c.ext + t * 8 + lj] = wb[t];
    }
    PHASE_STAMP(8);
}

// Kernel 1 of the otAspire path: pairwise sentence costs of the pairs of one chunk of candidates
// [a.cand0, a.cand1) -> workspace.  Streams every candidate row once; HBM bound for few queries.
// DIRECT: both L2 formulas accumulated (padded reference tensors: their pair matrices are compared at 1e-5).  !DIRECT
// (CSR inputs): x.y only, -cdist from the expansion with the cancelled entries redone -- half the arithmetic and
// half the cross-lane reductions of the T x T tile loop.
template <int T, bool DIRECT, bool CENTER = false>
__global__ void __launch_bounds__(kBlock, 3) pair_cost_kernel(ScoreArgs a, PairWs<T> ws) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool paired = a.pairing != ASPIRE_PAIR_CROSS;          // one query per candidate (PAIRED, MAPPED)
    const int64_t c_idx = a.cand0 + blockIdx.x;
    const int64_t ncand = a.cand1 - a.cand0;
    const int64_t q_own = a.pairing == kPairMapped ? (int64_t)a.qmap[c_idx] : c_idx;
    const int64_t q_begin = paired ? q_own : (int64_t)blockIdx.y * a.q_per_block;
    const int64_t q_end = paired ? q_own + 1 : min(a.q.n, q_begin + a.q_per_block);
    const bool own_diam = a.diameter == nullptr;
    const int c_len = a.c.len[c_idx];
    const int c_avail = a.c.ext > 0 ? a.c.ext : c_len;
    const float* cdoc = a.c.rows + (size_t)a.c.start[c_idx] * kD;
    for (int64_t q_idx = q_begin; q_idx < q_end; ++q_idx) {
        const int q_len = a.q.len[q_idx];
        const int q_avail = a.q.ext > 0 ? a.q.ext : q_len;
        const float* qdoc = a.q.rows + (size_t)a.q.start[q_idx] * kD;
        if (own_diam) {
            pair_partials<T, true, DIRECT, true, CENTER>(qdoc, q_avail, q_len, cdoc, c_avail, c_len, lds, wave, lane);
        } else {
            pair_partials<T, true, DIRECT, false, CENTER>(qdoc, q_avail, 0, cdoc, c_avail, 0, lds, wave, lane);
        }
        __syncthreads();
        const int64_t slot = paired ? (c_idx - a.cand0) : q_idx * ncand + (c_idx - a.cand0);
        finish_pair<T, DIRECT>(lds, use_mm_formula(a.cdist_mode, q_avail, c_avail), own_diam, ws, slot, qdoc, cdoc, q_len, c_len, q_avail,
                               c_avail);
        __syncthreads();
    }
}

// Kernel 1, single-tile (<= 8 sentence rows on both sides) persistent form: a fixed grid of workgroups walks the
// items (candidate-major pairs) with a stride of gridDim.x, and the NEXT item's 16 rows are already in flight
// (64 more VGPRs per lane) while the current item is accumulated, reduced and written -- the HBM latency that
// the one-item-per-workgroup form exposes at the head of every workgroup is paid once per workgroup instead.
struct RowSet {
    float4 x0[4], x1[4], y[8];
};
#ifdef ASPIRE_PHASE_CLOCK
// stamps of workgroup 7, wave 1, its second item
#define K1_STAMP(k)                                                                                         \
    do {                                                                                                    \
        if (g_k1dbg && blockIdx.x == 7 && wave == 1 && lane == 0 && item == 7 + gridDim.x)      \
            g_k1dbg[k] = (long long)__builtin_readcyclecounter();                                           \
    } while (0)
#else
#define K1_STAMP(k) \
    do {            \
    } while (0)
#endif

// Rows beyond a document's length are loaded as COPIES OF ITS LAST ROW (row index clamped): entries that involve
// them are masked downstream, and duplicates leave the bounding box unchanged, so the box needs no per-row
// predicate.  (Only used when ext == 0; padded tensors take the general kernel, which reads the real pad rows.)
// `item` = (sub-tile, pair): T * T sub-tiles of 8 x 8 entries per pair (1 for documents of <= 8 rows), the pair index
// fastest.  Sub-tile (ta, tb) takes query rows 8 ta .. and candidate rows 8 tb ...
// Plain global loads with per-row vector addresses (a buffer-descriptor form was measured: the scheduler spreads it
// over all 256 registers of its budget -- 197 here -- and a 256-register kernel shares a SIMD with nothing).
__device__ __forceinline__ void load_item(RowSet& r, const ScoreArgs& a, uint32_t item, uint32_t nq, int dofs,
                                          int& q_len, int& c_len, uint32_t T) {
    const uint32_t npairs = (uint32_t)(a.cand1 - a.cand0) * nq;
    const uint32_t tile = T == 1 ? 0 : item / npairs;        // pair index fastest: a pair's sub-tiles go to different workgroups
    const uint32_t pair = item - tile * npairs, ta = tile / T, tb = tile - ta * T;
    const uint32_t c_loc = nq == 1 ? pair : pair / nq;
    const int64_t c_idx = a.cand0 + c_loc;
    const int64_t q_idx = a.pairing == ASPIRE_PAIR_PAIRED ? c_idx
                          : a.pairing == kPairMapped      ? (int64_t)a.qmap[c_idx]
                                                          : (nq == 1 ? 0 : pair - c_loc * nq);
    const int i0 = 8 * ta, j0 = 8 * tb;
    c_len = a.c.len[c_idx];
    q_len = a.q.len[q_idx];
    const float* cdoc = a.c.rows + (size_t)a.c.start[c_idx] * kD + dofs;
    const float* qdoc = a.q.rows + (size_t)a.q.start[q_idx] * kD + dofs;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r.x0[i] = ld4(qdoc + (size_t)min(i0 + i, q_len - 1) * kD);
        r.x1[i] = ld4(qdoc + (size_t)min(i0 + 4 + i, q_len - 1) * kD);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) r.y[j] = ld4_stream(cdoc + (size_t)min(j0 + j, c_len - 1) * kD);
}

__device__ __forceinline__ float box_partial(const RowSet& r) {
    float4 mn = r.y[0], mx = r.y[0];
    auto upd = [&](const float4& v) {
        mn.x = fminf(mn.x, v.x); mn.y = fminf(mn.y, v.y); mn.z = fminf(mn.z, v.z); mn.w = fminf(mn.w, v.w);
        mx.x = fmaxf(mx.x, v.x); mx.y = fmaxf(mx.y, v.y); mx.z = fmaxf(mx.z, v.z); mx.w = fmaxf(mx.w, v.w);
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        upd(r.x0[i]);
        upd(r.x1[i]);
    }
#pragma unroll
    for (int j = 1; j < 8; ++j) upd(r.y[j]);
    const float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z, dw = mx.w - mn.w;
    return fmaf(dw, dw, fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
}

// The persistent, software-pipelined form: the next item's rows are in flight in a second register set while the
// current item is accumulated, reduced and written.
// SUB: documents of more than 8 rows, (sub-tile, pair) items; !SUB keeps the one-tile case free of the sub-tile
// arithmetic.  Register budgets decide how these kernels share a SIMD with OTHER launches (independent calls on other
// streams): at 197 registers two of these waves leave room for two 52-register Sinkhorn waves; at 256 nothing fits beside
// them and overlapped throughput fell from ~110 to ~70 M alignments/s with every kernel's own time unchanged.
// tests/test_abi_cpu.py pins the budgets.  (Superseded forms -- one register set with buffer loads, a matrix-core
// form, cost + solve fused per workgroup -- are described in NOTES.md "Tried and dropped".)
template <bool SUB>
__device__ __forceinline__ void pair_cost1_body(const ScoreArgs& a, const PairWs<1>& ws, uint32_t T_rt, float* lds) {
    const uint32_t T = SUB ? T_rt : 1u;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dofs = wave * 256 + lane * 4;
    const bool paired = a.pairing != ASPIRE_PAIR_CROSS;          // one query per candidate (PAIRED, MAPPED)
    // 32-bit item arithmetic: a chunk holds at most workspace / 516 B < 2^31 pairs, and 64-bit division costs
    // hundreds of cycles per item on this hardware.
    const uint32_t nq = paired ? 1u : (uint32_t)a.q.n;
    auto query_of = [&](int64_t c_idx, uint32_t q_loc) -> int64_t {
        return a.pairing == ASPIRE_PAIR_PAIRED ? c_idx : a.pairing == kPairMapped ? (int64_t)a.qmap[c_idx] : (int64_t)q_loc;
    };
    const uint32_t ncand = (uint32_t)(a.cand1 - a.cand0);
    const uint32_t tt = T * T, ld_e = 8 * T, n_ent = 64 * tt;      // sub-tiles per pair, row stride and entries of a pair's slot
    const uint32_t n_items = ncand * nq * tt;
    const bool own_diam = a.diameter == nullptr;
    float* red = lds + wave * 128;
    float* rednorm = lds + Lds<1>::kRed + wave * 16;
    float* xp = lds + Lds<1>::kXp + wave * kXpWave;

    // One item: accumulate, reduce, finish, hand over.  `r` is one of two register sets that take turns (the loop
    // below is unrolled by two so that the set being prefetched into is never copied).
    auto process = [&](RowSet& rs, int q_len, int c_len, uint32_t item) {
        if (a.center) {          // ASPIRE_OT_FLAG_CENTER: the tile's first query row comes off every row (see pair_partials)
            const float4 mu = rs.x0[0];
            auto sub = [&](float4& v) { v.x -= mu.x; v.y -= mu.y; v.z -= mu.z; v.w -= mu.w; };
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sub(rs.x0[i]);
                sub(rs.x1[i]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) sub(rs.y[j]);
        }
        // ---- accumulate + reduce the current item (register operands only).  Only the x.y sums are accumulated:
        // geomloss's cost is the expansion anyway, and torch.cdist's direct (x - y)^2 form (the marginals' -cdist)
        // is met by the same expansion to a few 1e-5 except where it cancels -- those entries (d^2 below 1e-4 of the
        // squared norm sum; none on unrelated vectors) are redone coordinate by coordinate below.  Dropping the
        // second set of 32 accumulators and its cross-lane reduction is 2/3 of this kernel's VALU work, which at
        // ~1000 pairs is what the kernel's time is made of.
        K1_STAMP(0);
        half_tile_partials<true, false>(rs.x0, rs.y, red, xp, lane);
        K1_STAMP(1);
        half_tile_partials<true, false>(rs.x1, rs.y, red + 32, xp, lane);
        K1_STAMP(2);
        {
            float nrm[16];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                nrm[i] = sq4(rs.x0[i]);
                nrm[4 + i] = sq4(rs.x1[i]);
                nrm[8 + i] = sq4(rs.y[i]);
                nrm[12 + i] = sq4(rs.y[4 + i]);
            }
            const float r = lds_wave_reduce<16>(nrm, xp, lane);
            if ((lane & 3) == 0) rednorm[lane >> 2] = r;
        }
        const uint32_t npairs = ncand * nq;
        const uint32_t tile = tt == 1 ? 0 : item / npairs;
        const uint32_t pair = item - tile * npairs, ta = tile / T, tb = tile - ta * T;
        const uint32_t c_loc = nq == 1 ? pair : pair / nq;
        const uint32_t q_loc = nq == 1 ? 0 : pair - c_loc * nq;
        if (own_diam && tile == 0) {
            float sbox;
            if (tt == 1) {
                sbox = wave_sum(box_partial(rs));
            } else {
                // long documents: the bounding box spans ALL rows of both documents; the pair's first sub-tile walks them
                const int64_t c_idx = a.cand0 + c_loc;
                const int64_t q_idx = query_of(c_idx, q_loc);
                const float* qd = a.q.rows + (size_t)a.q.start[q_idx] * kD + dofs;
                const float* cd = a.c.rows + (size_t)a.c.start[c_idx] * kD + dofs;
                float4 mn = ld4(qd), mx = mn;
                auto upd = [&](const float4& v) {
                    mn.x = fminf(mn.x, v.x); mn.y = fminf(mn.y, v.y); mn.z = fminf(mn.z, v.z); mn.w = fminf(mn.w, v.w);
                    mx.x = fmaxf(mx.x, v.x); mx.y = fmaxf(mx.y, v.y); mx.z = fmaxf(mx.z, v.z); mx.w = fmaxf(mx.w, v.w);
                };
                auto walk = [&](const float* doc, int n) {     // eight independent loads in flight (rows clamp: idempotent)
                    for (int r0 = 0; r0 < n; r0 += 8) {
                        float4 v[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = ld4(doc + (size_t)min(r0 + k, n - 1) * kD);
#pragma unroll
                        for (int k = 0; k < 8; ++k) upd(v[k]);
                    }
                };
                walk(qd, q_len);
                walk(cd, c_len);
                const float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z, dw = mx.w - mn.w;
                sbox = wave_sum(fmaf(dw, dw, fmaf(dz, dz, fmaf(dy, dy, dx * dx))));
            }
            if (lane == 0) lds[Lds<1>::kRed + Lds<1>::kNorm + wave] = sbox;
        }
        K1_STAMP(3);
        __syncthreads();
        K1_STAMP(4);
        const int64_t slot = paired ? (int64_t)c_loc : (int64_t)q_loc * ncand + c_loc;
        // A dedicated word: wave 0 rewrites it only after the NEXT item's first barrier, which every wave reaches only
        // after it has read this item's mask (it used to live in wave 0's reduction scratch, which wave 0 rewrites
        // at once when a workgroup walks several items).
        unsigned long long* redo_mask = reinterpret_cast<unsigned long long*>(lds + Lds<1>::kRedo);
        if (wave == 0) {
            const int li = lane >> 3, lj = lane & 7;
            float gsum = 0.f, xx = 0.f, yy = 0.f;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) {
                gsum += lds[w * 128 + lane];
                xx += lds[Lds<1>::kRed + w * 16 + li];
                yy += lds[Lds<1>::kRed + w * 16 + 8 + lj];
            }
            const float sq = fmaf(-2.f, gsum, xx) + yy;
            const float ns = xx + yy;
            // (round 6: a cancelling entry is redone from the exact sum whatever formula torch.cdist would pick -- also beyond 25 rows: include/aspire_hip.h, SHARED SENTENCES)
            const int gi = 8 * ta + li, gj = 8 * tb + lj;                  // entry of the pair's 8T x 8T slot
            const bool redo = gi < q_len && gj < c_len && sq < 1e-4f * ns * ns;
            const int64_t o = slot * n_ent + gi * ld_e + gj;
            if (!redo) {
                ws.cost[o] = sqrtf(fmaxf(sq, 1e-8f));
                ws.neg[o] = -sqrtf(fmaxf(sq, 0.f));
            }
            const unsigned long long m = __ballot(redo);
            if (lane == 0) {
                *redo_mask = m;
                if (own_diam && tile == 0) {
                    const float* dd = lds + Lds<1>::kRed + Lds<1>::kNorm;
                    ws.diam2[slot] = dd[0] + dd[1] + dd[2];
                }
            }
        }
        __syncthreads();
        {
            const unsigned long long todo = *redo_mask;     // workgroup-uniform
            if (__builtin_expect(todo != 0, 0)) {
                // 16 lanes (one DPP row) per flagged entry, 48 coordinates per lane, twelve entries at a time over
                // the three waves, no barriers: with real sentence vectors a few entries per pair can be this close
                const int64_t c_idx = a.cand0 + c_loc;
                const int64_t q_idx = query_of(c_idx, q_loc);
                const float* qdoc = a.q.rows + (size_t)a.q.start[q_idx] * kD;
                const float* cdoc = a.c.rows + (size_t)a.c.start[c_idx] * kD;
                const int n_flag = __builtin_popcountll(todo), l16 = lane & 15;
                for (int base = wave * 4; base < n_flag; base += 4 * kWaves) {
                    const int my = base + (lane >> 4);
                    const bool live = my < n_flag;
                    unsigned long long m = todo;
                    for (int t = 0; t < (live ? my : 0); ++t) m &= m - 1;      // drop the first `my` set bits
                    const int e = __builtin_ctzll(m);
                    const int gi = 8 * ta + (e >> 3), gj = 8 * tb + (e & 7);
                    const float* xr = qdoc + (size_t)gi * kD + 4 * l16;
                    const float* yr = cdoc + (size_t)gj * kD + 4 * l16;
                    float p0 = 0.f, p1 = 0.f;
#pragma unroll
                    for (int c = 0; c < 12; c += 2) {
                        const float4 u0 = ld4(xr + 64 * c), v0 = ld4(yr + 64 * c), u1 = ld4(xr + 64 * c + 64), v1 = ld4(yr + 64 * c + 64);
                        const float a0 = u0.x - v0.x, a1 = u0.y - v0.y, a2 = u0.z - v0.z, a3 = u0.w - v0.w;
                        const float b0 = u1.x - v1.x, b1 = u1.y - v1.y, b2 = u1.z - v1.z, b3 = u1.w - v1.w;
                        p0 = fmaf(a3, a3, fmaf(a2, a2, fmaf(a1, a1, fmaf(a0, a0, p0))));
                        p1 = fmaf(b3, b3, fmaf(b2, b2, fmaf(b1, b1, fmaf(b0, b0, p1))));
                    }
                    float part = p0 + p1;
                    part += lane_xor<1>(part);
                    part += lane_xor<2>(part);
                    part += lane_xor<4>(part);
                    part += lane_xor<8>(part);
                    if (live && l16 == 0) {       // geomloss's cost from the same exact sum (kCostFloor2: its clamp_min)
                        ws.neg[slot * n_ent + gi * ld_e + gj] = -sqrtf(part);
                        ws.cost[slot * n_ent + gi * ld_e + gj] = sqrtf(fmaxf(part, 1e-8f));
                    }
                }
            }
        }
        K1_STAMP(5);
    };
    RowSet ra, rb;
    int qa = 0, ca = 0, qb = 0, cb = 0;
    const uint32_t stride = gridDim.x;
    uint32_t item = blockIdx.x;
    if (item < n_items) load_item(ra, a, item, nq, dofs, qa, ca, T);
    while (item < n_items) {
        const uint32_t n1 = item + stride;
        if (n1 < n_items) load_item(rb, a, n1, nq, dofs, qb, cb, T);     // in flight under this item's arithmetic
        process(ra, qa, ca, item);
        if (n1 >= n_items) break;
        const uint32_t n2 = n1 + stride;
        if (n2 < n_items) load_item(ra, a, n2, nq, dofs, qa, ca, T);
        process(rb, qb, cb, n1);
        item = n2;
    }
}
__global__ void __launch_bounds__(kBlock, 2) pair_cost1_kernel(ScoreArgs a, PairWs<1> ws, uint32_t T_rt) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    pair_cost1_body<false>(a, ws, T_rt, lds);
}
// The sub-tile form, capped at 216 registers (amdgpu_num_vgpr counts HALF registers on gfx950: 108 -> 216; uncapped it
// takes 256 and no other launch's waves share a SIMD with it).
#ifndef SUB_CAP
#define SUB_CAP 108
#endif
__global__ void __launch_bounds__(kBlock, 2) __attribute__((amdgpu_num_vgpr(SUB_CAP)))
pair_cost1_sub_kernel(ScoreArgs a, PairWs<1> ws, uint32_t T_rt) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    pair_cost1_body<true>(a, ws, T_rt, lds);
}

// ---------------------------------------------------------------------------------------------
// Kernel 1, tiled form for documents of <= 8 sentence rows (T == 1, CSR inputs).
//
// The accumulate-then-reduce kernels above give every lane a slice of the 768 coordinates and all 64 (i,j)
// pairs, so each pair costs a 64-lane reduction of 128 accumulators -- on gfx950 that reduction (LDS transpose or
// permlane butterfly) costs several times the multiply-adds it serves.  Here the roles are swapped, GEMM style:
// a lane OWNS R x R entries (i,j) of a pair and walks all 768 coordinates itself, so its accumulators are finished
// sums and nothing is reduced across lanes.  The sentence rows are staged through LDS 16-byte chunk by chunk
// (coalesced global_load_dwordx4 -> ds_write_b128; row stride padded so the operand reads are conflict free) and
// re-read as ds_read_b128 broadcasts.  One wave handles NC = R*R candidates against one query:
//   R = 1: 1 candidate, lane (li,lj) = (l>>3, l&7) owns entry (li,lj); stages of 128 coordinates (lanes 0-31 stage
//          the query rows, lanes 32-63 the candidate rows)                                   -- lowest latency
//   R = 2: 4 candidates, 16 lanes each, lane owns the 2x2 block (2li+a, 2lj+b); stages of 64 coordinates (the 16
//          lanes of candidate p stage its 8 rows and query rows 2p, 2p+1)                    -- 3x fewer LDS reads
// While staging, the lane that holds all 8 rows of a candidate for one chunk also forms that chunk's bounding-box
// term (geomloss diameter) and the row norms, so those cost no extra pass either.
// ---------------------------------------------------------------------------------------------
template <int R>
struct TileCfg {
    static constexpr int kNC = R * R;              // candidates per wave
    static constexpr int kLanesPerCand = 64 / kNC;
    static constexpr int kGroups = R == 1 ? 2 : 4; // staging lane groups
    static constexpr int kCh = 64 / kGroups;       // 16-byte chunks per row per stage
    static constexpr int kStages = 192 / kCh;
    static constexpr int kRowStride = 4 * kCh + 4; // floats; (kRowStride / 4) is odd -> rows land on distinct bank slots
    static constexpr int kRows = 8 + 8 * kNC;      // staged rows: 8 query + 8 per candidate
    static constexpr int kNormLd = 68;
    static constexpr int kLdsFloats = kRows * kRowStride + 16 * kNormLd;   // + norm / box scratch
    static constexpr int kXRows = R == 1 ? 8 : 2;  // query rows staged by one lane
};

// per-coordinate bounding box of each query's valid rows: qbox[q][0][768] = min, qbox[q][1][768] = max
__global__ void __launch_bounds__(192) doc_box_kernel(RepSet d, float* __restrict__ box) {
    const int64_t k = blockIdx.x;
    const int n = d.len[k];
    const float* doc = d.rows + (size_t)d.start[k] * kD + threadIdx.x * 4;
    float4 mn, mx;
    doc_box_chunk(doc, n, mn, mx);
    *reinterpret_cast<float4*>(box + k * 2 * kD + threadIdx.x * 4) = mn;
    *reinterpret_cast<float4*>(box + k * 2 * kD + kD + threadIdx.x * 4) = mx;
}

// gate[0] += pairs of this launch that hold a document of more than 8 rows (MAPPED: candidate p against query qmap[p]; CROSS:
// one query).  The counter is zeroed on the stream in front of it.
__global__ void __launch_bounds__(256) long_pair_census_kernel(ScoreArgs a, int32_t* gate) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int is_long = 0;
    if (p < a.c.n) {
        const int64_t q_idx = a.pairing == kPairMapped ? (int64_t)a.qmap[p] : 0;
        is_long = a.c.len[p] > 8 || a.q.len[q_idx] > 8;
    }
    const int n = __popcll(__ballot(is_long));
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(gate, n);
}

// DS = 1: every wave of the 4-wave workgroup takes its own items (throughput form).  DS > 1: the DS waves of a
// workgroup share one item and each walks every DS-th stage, then wave 0 adds the partial results (latency form
// for small grids).
template <int R, int DS>
__global__ void __launch_bounds__(256) pair_tile_kernel(ScoreArgs a, PairWs<1> ws, const float* __restrict__ qbox) {
    using C = TileCfg<R>;
    constexpr int kAcc = 2 * R * R;
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* lds = lds_all + wave * C::kLdsFloats;
    float* nscr = lds + C::kRows * C::kRowStride;       // [16][kNormLd]: norm partials, then (DS = 4) accumulators
    const bool paired = a.pairing == ASPIRE_PAIR_PAIRED;
    const bool mapped = a.pairing == kPairMapped;                // batched jobs: items are the groups of four of jobs [job0, job1)
    const bool own_diam = a.diameter == nullptr;
    const uint32_t nq = (paired || mapped) ? 1u : (uint32_t)a.q.n;
    const uint32_t ncand = (uint32_t)(a.cand1 - a.cand0);
    const uint32_t ngroups = (ncand + C::kNC - 1) / C::kNC;
    const uint32_t item_lo = mapped ? (uint32_t)a.grp_off[a.job0] : 0u;
    const uint32_t n_items = mapped ? (uint32_t)a.grp_off[a.job1] : ngroups * nq;   // item = (candidate group, query), group-major
    const uint32_t first = item_lo + (DS == 1 ? blockIdx.x * 4 + wave : blockIdx.x);
    const uint32_t stride = DS == 1 ? gridDim.x * 4 : gridDim.x;

    // lane roles ------------------------------------------------------------------------------------------
    const int p = lane / C::kLanesPerCand;                       // candidate of this lane (compute AND staging, R = 2)
    const int lp = lane % C::kLanesPerCand;
    const int li = lp / (8 / R), lj = lp % (8 / R);
    const int sg = lane / C::kCh, sc = lane % C::kCh;            // staging group, staging chunk
    // what this lane stages: R = 1: group 0 -> the 8 query rows, group 1 -> the 8 candidate rows;
    //                        R = 2: group g -> the 8 rows of candidate g and query rows 2g, 2g+1.
    const bool stages_y = R == 2 || sg == 1;
    const bool stages_x = R == 2 || sg == 0;

    for (uint32_t item = first; item < n_items; item += stride) {
        const uint32_t cg = nq == 1 ? item : item / nq;
        uint32_t q_loc = nq == 1 ? 0 : item - cg * nq;
        uint32_t c_loc0 = cg * C::kNC;                                         // first candidate of the group
        uint32_t c_end = ncand;                                                // candidates of the group stop here
        if (mapped) {
            q_loc = (uint32_t)a.grp_job[item];
            c_loc0 = (uint32_t)a.job_off[q_loc] + (item - (uint32_t)a.grp_off[q_loc]) * C::kNC;
            c_end = (uint32_t)a.job_off[q_loc + 1];
        }
        const uint32_t my_c_loc = min(c_loc0 + (R == 1 ? 0u : (uint32_t)p), c_end - 1);   // tail groups: clamp (duplicate work, not stored)
        const bool my_c_real = c_loc0 + (R == 1 ? 0u : (uint32_t)p) < c_end;
        const int64_t c_idx = a.cand0 + my_c_loc;
        const int64_t q_idx = paired ? c_idx : (int64_t)q_loc;
        const int c_len = a.c.len[c_idx], q_len = a.q.len[q_idx];
        const float* qdoc = a.q.rows + (size_t)a.q.start[q_idx] * kD;
        // staging source of this lane (pad rows clamp to the last valid row: masked downstream, box-neutral)
        const int64_t sy_idx = a.cand0 + (R == 1 ? my_c_loc : min(c_loc0 + (uint32_t)sg, c_end - 1));
        const int sy_len = a.c.len[sy_idx];
        const float* sy_doc = a.c.rows + (size_t)a.c.start[sy_idx] * kD;

        float accg[R][R];
#pragma unroll
        for (int x = 0; x < R; ++x)
#pragma unroll
            for (int y = 0; y < R; ++y) accg[x][y] = 0.f;
        float ny[8], nx[C::kXRows], dsq = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) ny[k] = 0.f;
#pragma unroll
        for (int k = 0; k < C::kXRows; ++k) nx[k] = 0.f;

        float4 vy[8], vx[C::kXRows], qmn, qmx;
        const float* qb = own_diam ? qbox + (size_t)q_idx * 2 * kD : sy_doc;
        const int qb_hi = own_diam ? kD : 0;
        auto issue_loads = [&](int st) {
            const int dofs = (st * C::kCh + sc) * 4;
            if (stages_y) {
#pragma unroll
                for (int j = 0; j < 8; ++j) vy[j] = ld4_stream(sy_doc + (size_t)min(j, sy_len - 1) * kD + dofs);
                // UNCONDITIONAL (with caller-supplied diameters the candidate's first row stands in and the box term
                // is unused): a branch around these two loads made the compiler wait for the row loads just issued
                // at the join -- every stage's HBM latency in series with its arithmetic (see fused.hip)
                qmn = ld4(qb + dofs);
                qmx = ld4(qb + qb_hi + dofs);
            }
            if (stages_x) {
#pragma unroll
                for (int k = 0; k < C::kXRows; ++k)
                    vx[k] = ld4(qdoc + (size_t)min(R == 1 ? k : 2 * sg + k, q_len - 1) * kD + dofs);
            }
        };
        const int st0 = DS == 1 ? 0 : wave;
        if (st0 < C::kStages) issue_loads(st0);
#pragma unroll 1
        for (int st = st0; st < C::kStages; st += DS) {
            // ---- stage: registers -> LDS, with box / norm side products; then the NEXT stage's loads go out so
            // that they fly under this stage's arithmetic (no extra registers: the rows were just consumed) ----
            if (stages_y) {
                float4 mn = vy[0], mx = vy[0];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    ny[j] += sq4(vy[j]);
                    if (j > 0) {
                        mn.x = fminf(mn.x, vy[j].x); mn.y = fminf(mn.y, vy[j].y); mn.z = fminf(mn.z, vy[j].z); mn.w = fminf(mn.w, vy[j].w);
                        mx.x = fmaxf(mx.x, vy[j].x); mx.y = fmaxf(mx.y, vy[j].y); mx.z = fmaxf(mx.z, vy[j].z); mx.w = fmaxf(mx.w, vy[j].w);
                    }
                    const int row = 8 + (R == 1 ? 0 : sg) * 8 + j;
                    *reinterpret_cast<float4*>(lds + row * C::kRowStride + sc * 4) = vy[j];
                }
                if (own_diam) {
                    const float dx = fmaxf(mx.x, qmx.x) - fminf(mn.x, qmn.x), dy = fmaxf(mx.y, qmx.y) - fminf(mn.y, qmn.y);
                    const float dz = fmaxf(mx.z, qmx.z) - fminf(mn.z, qmn.z), dw = fmaxf(mx.w, qmx.w) - fminf(mn.w, qmn.w);
                    dsq += fmaf(dw, dw, fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
                }
            }
            if (stages_x) {
#pragma unroll
                for (int k = 0; k < C::kXRows; ++k) {
                    nx[k] += sq4(vx[k]);
                    *reinterpret_cast<float4*>(lds + (R == 1 ? k : 2 * sg + k) * C::kRowStride + sc * 4) = vx[k];
                }
            }
            if (st + DS < C::kStages) issue_loads(st + DS);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- accumulate: every lane walks the staged chunks for its own R x R entries ----------------------
            const float* xr = lds + (R * li) * C::kRowStride;
            const float* yr = lds + (8 + p * 8 + R * lj) * C::kRowStride;
#pragma unroll 1
            for (int c = 0; c < C::kCh; c += 2) {
              // two chunks per trip: the second chunk's LDS reads are in flight under the first chunk's arithmetic
#pragma unroll
              for (int cc = 0; cc < 2; ++cc) {
                float4 xv[R], yv[R];
#pragma unroll
                for (int x = 0; x < R; ++x) xv[x] = *reinterpret_cast<const float4*>(xr + x * C::kRowStride + (c + cc) * 4);
#pragma unroll
                for (int y = 0; y < R; ++y) yv[y] = *reinterpret_cast<const float4*>(yr + y * C::kRowStride + (c + cc) * 4);
#pragma unroll
                for (int x = 0; x < R; ++x)
#pragma unroll
                    for (int y = 0; y < R; ++y) {
                        accg[x][y] = fmaf(xv[x].w, yv[y].w, fmaf(xv[x].z, yv[y].z, fmaf(xv[x].y, yv[y].y, fmaf(xv[x].x, yv[y].x, accg[x][y]))));
                    }
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the stage buffer is rewritten next
            __builtin_amdgcn_wave_barrier();
        }

        // ---- norms and box: sum the staging lanes' partials through the scratch table nscr[value][lane] --------
        // value 0..7: |y_j|^2 partials of the lane's staged candidate; 8..8+kXRows-1: |x|^2 partials of its query rows
#pragma unroll
        for (int k = 0; k < 8; ++k) nscr[k * C::kNormLd + lane] = stages_y ? ny[k] : 0.f;
#pragma unroll
        for (int k = 0; k < C::kXRows; ++k) nscr[(8 + k) * C::kNormLd + lane] = stages_x ? nx[k] : 0.f;
        if constexpr (DS == 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            // the other waves' accumulators and box terms travel through their stage buffers (free by now)
#pragma unroll
            for (int x = 0; x < R; ++x)
#pragma unroll
                for (int y = 0; y < R; ++y) {
                    lds[((x * R + y) * 2 + 0) * 64 + lane] = accg[x][y];
                }
            lds[kAcc * 64 + lane] = dsq;
            __syncthreads();
            if (wave != 0) {
                __syncthreads();   // matches the end-of-item barrier below
                continue;
            }
#pragma unroll
            for (int w = 1; w < DS; ++w) {
                const float* o = lds_all + w * C::kLdsFloats;
#pragma unroll
                for (int x = 0; x < R; ++x)
#pragma unroll
                    for (int y = 0; y < R; ++y) {
                        accg[x][y] += o[((x * R + y) * 2 + 0) * 64 + lane];
                    }
                dsq += o[kAcc * 64 + lane];
            }
        }
        auto table_sum = [&](int value, int lane0, int nlanes) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < DS; ++w) {
                const float4* src = reinterpret_cast<const float4*>(lds_all + (DS == 1 ? wave : w) * C::kLdsFloats +
                                                                    C::kRows * C::kRowStride + value * C::kNormLd + lane0);
                for (int m = 0; m < nlanes / 4; ++m) {
                    const float4 u = src[m];
                    t += (u.x + u.y) + (u.z + u.w);
                }
            }
            return t;
        };
        float xx[R], yy[R];
        if constexpr (R == 1) {
            // x partials live in lanes 0..31 (staging group 0), y partials in lanes 32..63
            yy[0] = table_sum(lj, 32, 32);
            xx[0] = table_sum(8 + li, 0, 32);
        } else {
#pragma unroll
            for (int y = 0; y < R; ++y) yy[y] = table_sum(R * lj + y, p * 16, 16);
#pragma unroll
            for (int x = 0; x < R; ++x) xx[x] = table_sum(8 + ((R * li + x) & 1), ((R * li + x) >> 1) * 16, 16);
        }
        float diam2 = 0.f;
        if (own_diam) {
            // box terms were formed by the lanes that staged candidate rows: sum them over that candidate's lanes
            if constexpr (R == 1) {
                diam2 = wave_sum(stages_y ? dsq : 0.f);
            } else {
                float t = dsq;                       // 16 staging lanes of candidate sg == this lane's p (same grouping)
                t += lane_xor<1>(t); t += lane_xor<2>(t); t += lane_xor<4>(t); t += lane_xor<8>(t);
                diam2 = t;
            }
        }

        // ---- finish the entries and hand them to the Sinkhorn kernel -----------------------------------------
        // Only x.y was accumulated: -cdist comes from the same expansion as the cost, and the entries where it cancels
        // (torch.cdist's direct formula differs there) are redone below.  See pair_cost1_kernel.
        // (round 6: a cancelling entry is redone from the exact sum whatever formula torch.cdist would pick -- also beyond 25 rows: include/aspire_hip.h, SHARED SENTENCES)
        const int64_t slot = (paired || mapped) ? (int64_t)my_c_loc : (int64_t)q_loc * ncand + my_c_loc;
        bool redo[R][R];
#pragma unroll
        for (int x = 0; x < R; ++x)
#pragma unroll
            for (int y = 0; y < R; ++y) {
                const int i = R * li + x, j = R * lj + y;
                const float sq = fmaf(-2.f, accg[x][y], xx[x]) + yy[y];
                const float ns = xx[x] + yy[y];
                redo[x][y] = my_c_real && i < q_len && j < c_len && sq < 1e-4f * ns * ns;
                if (my_c_real && !redo[x][y]) {
                    ws.cost[slot * 64 + i * 8 + j] = sqrtf(fmaxf(sq, 1e-8f));
                    ws.neg[slot * 64 + i * 8 + j] = -sqrtf(fmaxf(sq, 0.f));
                }
            }
        if (my_c_real && own_diam && lp == 0) ws.diam2[slot] = diam2;
        if constexpr (R == 2 && DS == 1) {
            // direct-formula redo, the whole wave on one entry (12 coordinates per lane), four entries per memory round trip
            // (see pair_fused_kernel: one entry per trip makes a wave with a duplicate document fall behind by 8 trips)
            const int c_start_v = a.c.start[c_idx];
#pragma unroll
            for (int x = 0; x < R; ++x)
#pragma unroll
                for (int y = 0; y < R; ++y) {
                    unsigned long long wm = __ballot(redo[x][y]);
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
                            const int ol = o & 15, i = R * (ol >> 2) + x, j = R * (ol & 3) + y;
                            const int cs_e = __builtin_amdgcn_readlane(c_start_v, o);
                            const float* qrow = qdoc + (size_t)i * kD + 4 * lane;
                            const float* crow = a.c.rows + ((size_t)cs_e + j) * kD + 4 * lane;
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
                                const int ol = owner[e] & 15, i = R * (ol >> 2) + x, j = R * (ol & 3) + y;
                                if (lane == owner[e]) {
                                    ws.neg[slot * 64 + i * 8 + j] = -sqrtf(tot);
                                    ws.cost[slot * 64 + i * 8 + j] = sqrtf(fmaxf(tot, 1e-8f));      // geomloss's cost from the same exact sum
                                }
                            }
                        }
                    }
                }
        } else {
#pragma unroll
            for (int x = 0; x < R; ++x)
#pragma unroll
                for (int y = 0; y < R; ++y)
                    if (redo[x][y]) {   // other layouts (not instantiated for production): lane-local direct sum
                        const int i = R * li + x, j = R * lj + y;
                        const float* xr = qdoc + (size_t)i * kD;
                        const float* yr = a.c.rows + ((size_t)a.c.start[c_idx] + j) * kD;
                        float d2s = 0.f;
                        for (int d = 0; d < kD; d += 4) {
                            const float4 u = ld4(xr + d), v = ld4(yr + d);
                            const float d0 = u.x - v.x, d1 = u.y - v.y, d2 = u.z - v.z, d3 = u.w - v.w;
                            d2s = fmaf(d3, d3, fmaf(d2, d2, fmaf(d1, d1, fmaf(d0, d0, d2s))));
                        }
                        ws.neg[slot * 64 + i * 8 + j] = -sqrtf(d2s);
                        ws.cost[slot * 64 + i * 8 + j] = sqrtf(fmaxf(d2s, 1e-8f));
                    }
        }
        if constexpr (DS == 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // scratch and stage buffers are reused by the next item
            __builtin_amdgcn_wave_barrier();
        } else {
            __syncthreads();
        }
    }
}

// Kernel 2: one wave = one Sinkhorn solve, four pairs per workgroup; only registers and cross-lane ops.
// ~50 VGPRs at T = 1, so up to 8 solves share a SIMD and hide each other's cross-lane / transcendental
// latencies.
template <int T>
__global__ void __launch_bounds__(256) sinkhorn_kernel(ScoreArgs a, PairWs<T> ws, int64_t n_slots) {
    // this kernel is ONE long dependent chain per wave: when it shares a SIMD with throughput work of another launch
    // (a cost kernel of the next query), its instructions should issue first
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t slot = (int64_t)blockIdx.x * 4 + wave;
    if (a.pairing == kPairMapped) {              // the slots of jobs [job0, job1); the grid is sized from an upper bound
        slot += a.job_off[a.job0];
        n_slots = a.job_off[a.job1];
    }
    if (slot < n_slots) {
        const PairIdx ix = pair_of_slot(a, slot);
        PairState<T> st;
        load_pair<T>(st, ws, slot, lane, a.cost_from_neg != 0);
        const float diam = a.diameter == nullptr ? fmaxf(sqrtf(ws.diam2[slot]), kMinDiameter) : group_diameter_of(a, ix);
        sinkhorn_pair<T>(a, st, a.q.len[ix.q_idx], a.c.len[ix.c_idx], diam, ix.p, lane);
    }
}

// ---------------------------------------------------------------------------------------------
// One wave = one PAIR, costs and solve in ONE launch (documents of <= 8 rows, CSR, a few dozen to a few thousand pairs: the
// per-query call of evaluate.py:58-76 -- one query against its pool of ~10^2 .. 10^3 candidates).  The two-launch form
// (pair_cost1_kernel: three waves per pair + sinkhorn_kernel<1>) costs a workspace round trip and a dependent launch: 12.1 + 7.5 us
// of kernels and ~3.5 us between them at 1 x 1000.  Here a wave
//   * issues ALL 24 loads of its candidate's rows at once (one HBM round trip; lane l owns coordinates 4 l + 256 s, s = 0 .. 2, of
//     every row), reads the query's rows (L2) stage by stage,
//   * accumulates the 64 dot products as 64 per-lane partial sums (each lane: all 8 x 8 pairs of rows over ITS twelve coordinates),
//     the 16 squared norms and the joint box's extent (all sixteen rows of a coordinate sit in one lane),
//   * folds them across the wave with the halving butterfly (common.h: butterfly_sum) so that lane l ends up with entry
//     lane_ij<1>(l) -- the layout sinkhorn_pair<1> solves in -- and goes straight on to the solve.
// Entries where the expansion cancels take -cdist and geomloss's cost from the exact sum, as everywhere (round 5).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pair_one_kernel(ScoreArgs a, int64_t n_slots) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t slot = (int64_t)blockIdx.x * 4 + wave;
    if (a.pairing == kPairMapped) {
        slot += a.job_off[a.job0];
        n_slots = a.job_off[a.job1];
    }
    if (slot < n_slots) {
    const PairIdx ix = pair_of_slot(a, slot);
    const int q_len = a.q.len[ix.q_idx], c_len = a.c.len[ix.c_idx];
    const float* qdoc = a.q.rows + (size_t)a.q.start[ix.q_idx] * kD;
    const float* cdoc = a.c.rows + (size_t)a.c.start[ix.c_idx] * kD;
    float4 y[3][8];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int r = 0; r < 8; ++r) y[s][r] = ld4_stream(cdoc + (size_t)min(r, c_len - 1) * kD + 4 * lane + 256 * s);   // pad rows: copies of the last
    float acc[64];
#pragma unroll
    for (int e = 0; e < 64; ++e) acc[e] = 0.f;
    float nrm[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) nrm[e] = 0.f;
    float dsq = 0.f;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        float4 x[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = ld4(qdoc + (size_t)min(r, q_len - 1) * kD + 4 * lane + 256 * s);
        if (a.center) {
            // rows that share a large common component: the mean of the (padded) query rows comes off every row (fused.hip)
            float4 mu = x[0];
#pragma unroll
            for (int r = 1; r < 8; ++r) { mu.x += x[r].x; mu.y += x[r].y; mu.z += x[r].z; mu.w += x[r].w; }
            mu.x *= 0.125f; mu.y *= 0.125f; mu.z *= 0.125f; mu.w *= 0.125f;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                x[r].x -= mu.x; x[r].y -= mu.y; x[r].z -= mu.z; x[r].w -= mu.w;
                y[s][r].x -= mu.x; y[s][r].y -= mu.y; y[s][r].z -= mu.z; y[s][r].w -= mu.w;
            }
        }
        float4 mn = x[0], mx = x[0];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            nrm[r] += sq4(x[r]);
            nrm[8 + r] += sq4(y[s][r]);
            const float4 u = x[r], v = y[s][r];
            mn.x = fminf(mn.x, fminf(u.x, v.x)); mn.y = fminf(mn.y, fminf(u.y, v.y)); mn.z = fminf(mn.z, fminf(u.z, v.z)); mn.w = fminf(mn.w, fminf(u.w, v.w));
            mx.x = fmaxf(mx.x, fmaxf(u.x, v.x)); mx.y = fmaxf(mx.y, fmaxf(u.y, v.y)); mx.z = fmaxf(mx.z, fmaxf(u.z, v.z)); mx.w = fmaxf(mx.w, fmaxf(u.w, v.w));
        }
        const float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z, dw = mx.w - mn.w;
        dsq = fmaf(dw, dw, fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, dsq))));
#pragma unroll
        for (int e = 0; e < 64; ++e) {
            // element e of the butterfly = the entry lane e will own: lane_ij<1>
            const int ej = (e & 3) | ((e >> 2) & 4), ei = ((e >> 2) & 3) | ((e >> 3) & 4);
            acc[e] = fmaf(x[ei].w, y[s][ej].w, fmaf(x[ei].z, y[s][ej].z, fmaf(x[ei].y, y[s][ej].y, fmaf(x[ei].x, y[s][ej].x, acc[e]))));
        }
    }
    const float dot = butterfly_sum<64>(acc, lane);
    const float nsum = butterfly_sum<16>(nrm, lane);              // lane l: element l >> 2 (0 .. 7 = |x_i|^2, 8 .. 15 = |y_j|^2)
    const float diam2 = wave_sum(dsq);
    int li, lj;
    lane_ij<1>(lane, li, lj);
    const float xx = __shfl(nsum, 4 * li), yy = __shfl(nsum, 32 + 4 * lj);
    const float sq = fmaf(-2.f, dot, xx) + yy, ns = xx + yy;
    PairState<1> st;
    st.cost[0][0] = sqrtf(fmaxf(sq, 1e-8f));
    st.neg[0][0] = -sqrtf(fmaxf(sq, 0.f));
    // (round 6: a cancelling entry is redone from the exact sum whatever formula torch.cdist would pick -- also beyond 25 rows: include/aspire_hip.h, SHARED SENTENCES)
    unsigned long long todo = __ballot(li < q_len && lj < c_len && sq < 1e-4f * ns * ns);
    while (todo != 0) {        // rare: the whole wave on one entry, from the rows as they are in memory (a common shift drops out)
        const int o = (int)__builtin_ctzll(todo);
        todo &= todo - 1;
        int oi, oj;
        lane_ij<1>(o, oi, oj);
        float part = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const float4 u = ld4(qdoc + (size_t)oi * kD + 4 * lane + 256 * t), v = ld4(cdoc + (size_t)oj * kD + 4 * lane + 256 * t);
            const float d0 = u.x - v.x, d1 = u.y - v.y, d2 = u.z - v.z, d3 = u.w - v.w;
            part = fmaf(d3, d3, fmaf(d2, d2, fmaf(d1, d1, fmaf(d0, d0, part))));
        }
        const float tot = wave_sum(part);
        if (lane == o) {
            st.neg[0][0] = -sqrtf(tot);
            st.cost[0][0] = sqrtf(fmaxf(tot, 1e-8f));
        }
    }
    const float diam = a.diameter == nullptr ? fmaxf(sqrtf(diam2), kMinDiameter) : group_diameter_of(a, ix);
    sinkhorn_pair<1>(a, st, q_len, c_len, diam, ix.p, lane);
    }
}


// ---------------------------------------------------------------------------------------------
// Kernel 2, block form (throughput form for big grids, any T): a pair occupies LD x LD lanes of one DPP row and
// lane (li, lj) owns the R x R block of entries (R li + x, R lj + y), LD * R = 8 T:
//     T = 1: LD 2, R 4 (16 solves per wave) or LD 1, R 8 (64);  T = 2: LD 4, R 3 / 4 (4 solves per wave) or LD 2, R 6 / 8 (16);
//     T = 3, 4: LD 4, R 5 .. 8 (4 solves per wave)
// so most of a reduction is in-register adds and the cross-lane part is 1-2 DPP levels per direction.  The update
// is written around ONE exponential per entry, K_ij = exp2((f_i + g_j - C_ij) * log2(e)/eps):
//     sum_j b_j K_ij = exp((f_i - ft_i)/eps)   =>   ft_i = f_i - eps ln2 log2(sum_j b_j K_ij)
//     sum_i a_i K_ij = exp((g_j - gt_j)/eps)   =>   gt_j = g_j - eps ln2 log2(sum_i a_i K_ij)
// (the log-sum-exp of sinkhorn_pair::step2 shifted by the previous potential, with the marginal weights a, b as
// plain factors instead of log-weights inside the exponent), and the averaged update collapses to one FMA,
// f_i <- f_i - h log2(.), h = eps ln2 / 2 (eps ln2 for the final extrapolation, 0 once a pair has run out of
// steps while its wave mates have not).  Per step that is R^2 exp2 + 2R log2 per lane against 2 R^2 exp2 before.
// Every pair follows its own epsilon schedule; the per-step constants are two exp2 of an affine function of the
// step index (fp32: a relative 1e-7 on an intermediate temperature is far below the tolerance), no table.
// A sum that leaves fp32 range (extreme scaling) poisons the score with NaN; sinkhorn_repair_kernel then redoes
// such pairs with the max-shifted solver.
// ---------------------------------------------------------------------------------------------
typedef float f2v __attribute__((ext_vector_type(2)));     // a register pair for the packed fp32 instructions

template <int LD>
__device__ __forceinline__ float blk_sum_j(float v) {   // all-reduce over the LD lanes that share li
    if constexpr (LD >= 2) v += lane_xor<1>(v);
    if constexpr (LD == 4) v += lane_xor<2>(v);
    return v;
}
template <int LD>
__device__ __forceinline__ float blk_max_j(float v) {
    if constexpr (LD >= 2) v = fmaxf(v, lane_xor<1>(v));
    if constexpr (LD == 4) v = fmaxf(v, lane_xor<2>(v));
    return v;
}
template <int LD>
__device__ __forceinline__ float blk_sum_i(float v) {   // all-reduce over the LD lanes that share lj
    if constexpr (LD == 1) {
        return v;
    } else if constexpr (LD == 2) {
        return v + lane_xor<2>(v);
    } else {
        v += dpp_mov<0x124>(v, v);       // row_ror:4
        return v + dpp_mov<0x128>(v, v); // row_ror:8
    }
}
template <int LD>
__device__ __forceinline__ float blk_max_i(float v) {
    if constexpr (LD == 1) {
        return v;
    } else if constexpr (LD == 2) {
        return fmaxf(v, lane_xor<2>(v));
    } else {
        v = fmaxf(v, dpp_mov<0x124>(v, v));
        return fmaxf(v, dpp_mov<0x128>(v, v));
    }
}

template <int T, int LD, int R>
__global__ void __launch_bounds__(256) sinkhorn_block_kernel(ScoreArgs a, PairWs<T> ws, int64_t n_slots) {
    static_assert(LD * R <= 8 * T && LD * R > 8 * (T - 1), "block layout must fit the 8T x 8T slot");
    constexpr int NL = LD * LD, PPW = 64 / NL, E = 64 * T * T, LDS_ = 8 * T;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pp = lane / NL, lp = lane % NL, li = lp / LD, lj = lp % LD;
    int64_t slot0 = ((int64_t)blockIdx.x * 4 + wave) * PPW;
    if (a.pairing == kPairMapped) {              // the slots of jobs [job0, job1); the grid is sized from an upper bound
        slot0 += a.job_off[a.job0];
        n_slots = a.job_off[a.job1];
    }
    if (slot0 >= n_slots) return;
    bool real = slot0 + pp < n_slots;                        // tail wave: surplus groups redo the last pair, store nothing
    int64_t slot = real ? slot0 + pp : n_slots - 1;
    PairIdx ix = pair_of_slot(a, slot);
    int q_len = a.q.len[ix.q_idx], c_len = a.c.len[ix.c_idx];
    if (gate_few_long(a)) {
        // hybrid (score_types.h): only the pairs the fused kernel left to the 16-row kernels have slots.  The other lane
        // groups of the wave mirror its first such pair (their own slots hold nothing: a garbage diameter could mean any
        // number of steps) and store nothing.
        real = real && (q_len > 8 || c_len > 8);
        const unsigned long long todo = __ballot(real);
        if (todo == 0) return;
        const int lead = (int)__builtin_ctzll(todo);
        const int lo = __builtin_amdgcn_readlane((int)(uint32_t)slot, lead), hi = __builtin_amdgcn_readlane((int)(slot >> 32), lead);
        if (!real) slot = ((int64_t)hi << 32) | (uint32_t)lo;
        ix = pair_of_slot(a, slot);
        q_len = a.q.len[ix.q_idx];
        c_len = a.c.len[ix.c_idx];
    }
    const int64_t p = ix.p;

    float cost[R][R];
    bool rv[R], cv[R];
#pragma unroll
    for (int t = 0; t < R; ++t) {
        rv[t] = R * li + t < q_len;
        cv[t] = R * lj + t < c_len;
    }
    // Entries outside the pair's q_len x c_len rectangle are never written by some producers (and are masked by
    // zero weights here): read them as 0 so that no stale inf / nan can reach a sum through 0 * x.
    auto load_block = [&](const float* base, float (&dst)[R][R]) {
#pragma unroll
        for (int x = 0; x < R; ++x) {
            const float* row = base + slot * E + (R * li + x) * LDS_ + R * lj;
            if constexpr (R % 4 == 0) {
#pragma unroll
                for (int y = 0; y < R; y += 4) {
                    const float4 v = ld4(row + y);
                    dst[x][y] = v.x; dst[x][y + 1] = v.y; dst[x][y + 2] = v.z; dst[x][y + 3] = v.w;
                }
            } else if constexpr (R % 2 == 0) {
#pragma unroll
                for (int y = 0; y < R; y += 2) {
                    const float2 v = *reinterpret_cast<const float2*>(row + y);
                    dst[x][y] = v.x; dst[x][y + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int y = 0; y < R; ++y) dst[x][y] = row[y];
            }
#pragma unroll
            for (int y = 0; y < R; ++y) dst[x][y] = (rv[x] && cv[y]) ? dst[x][y] : 0.f;
        }
    };
    // ---- marginals (pair_distances.py:57-60) from -cdist; only the weights survive this scope -----------------
    const float temp = (float)a.temp;
    float wa[R], wb[R];
    {
        load_block(ws.neg, cost);   // borrowed: holds -cdist here
        float qm[R], cm[R];
#pragma unroll
        for (int x = 0; x < R; ++x) {
            float m = kNegBig;
#pragma unroll
            for (int y = 0; y < R; ++y) m = fmaxf(m, (rv[x] && cv[y]) ? cost[x][y] : kNegBig);
            qm[x] = blk_max_j<LD>(m) / temp;
        }
#pragma unroll
        for (int y = 0; y < R; ++y) {
            float m = kNegBig;
#pragma unroll
            for (int x = 0; x < R; ++x) m = fmaxf(m, (rv[x] && cv[y]) ? cost[x][y] : kNegBig);
            cm[y] = blk_max_i<LD>(m) / temp;
        }
        float mq = kNegBig, mc = kNegBig;
#pragma unroll
        for (int t = 0; t < R; ++t) {
            mq = fmaxf(mq, rv[t] ? qm[t] : kNegBig);
            mc = fmaxf(mc, cv[t] ? cm[t] : kNegBig);
        }
        mq = blk_max_i<LD>(mq);
        mc = blk_max_j<LD>(mc);
        float sq = 0.f, sc = 0.f;
#pragma unroll
        for (int t = 0; t < R; ++t) {
            sq += rv[t] ? fast_exp(qm[t] - mq) : 0.f;
            sc += cv[t] ? fast_exp(cm[t] - mc) : 0.f;
        }
        const float lsq = fast_log(blk_sum_i<LD>(sq)), lsc = fast_log(blk_sum_j<LD>(sc));
#pragma unroll
        for (int t = 0; t < R; ++t) {
            wa[t] = rv[t] ? fast_exp(qm[t] - mq - lsq) : 0.f;   // log_softmax(...).exp(); zero weight == geomloss's
            wb[t] = cv[t] ? fast_exp(cm[t] - mc - lsc) : 0.f;   // log-weight -100000
        }
    }
    if (a.cost_from_neg) {      // `cost` still holds the pair's -cdist block (zeros outside its rectangle)
#pragma unroll
        for (int x = 0; x < R; ++x)
#pragma unroll
            for (int y = 0; y < R; ++y) cost[x][y] = (rv[x] && cv[y]) ? fmaxf(-cost[x][y], __builtin_sqrtf(1e-8f)) : 0.f;
    } else {
        load_block(ws.cost, cost);
    }
    const float diam = a.diameter == nullptr ? fmaxf(sqrtf(ws.diam2[slot]), kMinDiameter) : group_diameter_of(a, ix);
    // ---- epsilon schedule: step 0 = diam, 1 .. n_mid = exp(ld + (k-1) lsc), n_mid+1 = blur, n_mid+2 = blur (final)
    float ldf;
    const int n_mid = schedule_mid_steps(a, diam, ldf);
    const int n_steps = n_mid + 3;
    int max_steps = n_steps;
#pragma unroll
    for (int m = NL; m < 64; m <<= 1) max_steps = max(max_steps, __shfl_xor(max_steps, m));
    const float r2_first = kLog2e * rcp_refined(diam), h_first = 0.5f * kLn2 * diam;
    const float eb = (float)a.blur;
    const float r2_blur = kLog2e * rcp_refined(eb), h_blur = 0.5f * kLn2 * eb;

    // ---- initialisation at eps = diam: softmin of the bare weights.  No shift is needed: the largest weight of a
    // probability vector over <= 32 atoms is >= 1/32 and C/diam <= ~1, so the sums stay in range. --------------
    float f[R], g[R];
    {
        float rs[R], cs[R];
#pragma unroll
        for (int t = 0; t < R; ++t) rs[t] = cs[t] = 0.f;
#pragma unroll
        for (int x = 0; x < R; ++x)
#pragma unroll
            for (int y = 0; y < R; ++y) {
                const float k0 = __builtin_amdgcn_exp2f(-cost[x][y] * r2_first);
                rs[x] = fmaf(wb[y], k0, rs[x]);
                cs[y] = fmaf(wa[x], k0, cs[y]);
            }
#pragma unroll
        for (int t = 0; t < R; ++t) {
            f[t] = -2.f * h_first * __builtin_amdgcn_logf(blk_sum_j<LD>(rs[t]));
            g[t] = -2.f * h_first * __builtin_amdgcn_logf(blk_sum_i<LD>(cs[t]));
        }
    }
    // ---- the annealing loop ---------------------------------------------------------------------------------
    // One step on register PAIRS (v_pk_mul / v_pk_fma_f32 work on two entries at once): the entries of a row as R / 2 column
    // pairs (+ a single for odd R).  R = 4: 76 issue slots per step instead of 122, R = 3: 62 instead of 90 -- the kernel
    // runs at its VALU-issue roof (profiles/sinkhorn_roofline.json), so only fewer instructions make it faster.  The
    // per-step constants follow from the previous step's by one multiply each through the annealed part of the schedule
    // (steps 2 .. n_mid: eps *= scaling), with a select-free loop while all of the wave's pairs anneal.
    constexpr int RP = R / 2;
    constexpr bool ODD = (R & 1) != 0;
    f2v cp[R][RP > 0 ? RP : 1], wbp[RP > 0 ? RP : 1], gp[RP > 0 ? RP : 1];
#pragma unroll
    for (int j = 0; j < RP; ++j) {
        wbp[j] = f2v{wb[2 * j], wb[2 * j + 1]};
        gp[j] = f2v{g[2 * j], g[2 * j + 1]};
#pragma unroll
        for (int x = 0; x < R; ++x) cp[x][j] = f2v{cost[x][2 * j], cost[x][2 * j + 1]};
    }
    float go = ODD ? g[R - 1] : 0.f;
    auto step = [&](float r2, float h) {
        f2v g2p[RP > 0 ? RP : 1], csp[RP > 0 ? RP : 1];
        float rs[R], cso = 0.f;
        const float g2o = go * r2;
#pragma unroll
        for (int j = 0; j < RP; ++j) {
            g2p[j] = gp[j] * r2;
            csp[j] = f2v{0.f, 0.f};
        }
#pragma unroll
        for (int x = 0; x < R; ++x) {
            const float fx = f[x] * r2;
            f2v racc = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                const f2v arg = __builtin_elementwise_fma(cp[x][j], f2v{-r2, -r2}, f2v{fx, fx} + g2p[j]);
                const f2v kk = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
                racc = __builtin_elementwise_fma(kk, wbp[j], racc);
                csp[j] = __builtin_elementwise_fma(kk, f2v{wa[x], wa[x]}, csp[j]);
            }
            rs[x] = racc.x + racc.y;
            if constexpr (ODD) {
                const float ko = __builtin_amdgcn_exp2f(fmaf(-cost[x][R - 1], r2, fx + g2o));
                rs[x] = fmaf(wb[R - 1], ko, rs[x]);
                cso = fmaf(wa[x], ko, cso);
            }
        }
#pragma unroll
        for (int x = 0; x < R; ++x) f[x] = fmaf(-h, __builtin_amdgcn_logf(blk_sum_j<LD>(rs[x])), f[x]);
#pragma unroll
        for (int j = 0; j < RP; ++j) {
            const f2v lc = {__builtin_amdgcn_logf(blk_sum_i<LD>(csp[j].x)), __builtin_amdgcn_logf(blk_sum_i<LD>(csp[j].y))};
            gp[j] = __builtin_elementwise_fma(f2v{-h, -h}, lc, gp[j]);
        }
        if constexpr (ODD) go = fmaf(-h, __builtin_amdgcn_logf(blk_sum_i<LD>(cso)), go);
    };
    {
        const float scal = (float)a.scaling, inv_scal = (float)(1.0 / a.scaling);
        int n_mid_lo = n_mid;
#pragma unroll
        for (int m = NL; m < 64; m <<= 1) n_mid_lo = min(n_mid_lo, __shfl_xor(n_mid_lo, m));
        n_mid_lo = __builtin_amdgcn_readfirstlane(n_mid_lo);
        max_steps = __builtin_amdgcn_readfirstlane(max_steps);
        float r2 = r2_first, h = h_first;
        int k = 0;
        // eps_k: diam at k = 0 and 1, diam scaling^(k-1) up to k = n_mid, then blur (averaged), blur (final, h doubled), and
        // nothing (h = 0) while a wave mate with a longer schedule is still annealing
        auto general = [&](int upto) {
#pragma unroll 1
            for (; k < upto; ++k) {
                const bool anneal = k >= 2 && k <= n_mid;
                r2 = anneal ? r2 * inv_scal : r2;
                h = anneal ? h * scal : h;
                if (k > n_mid) { r2 = r2_blur; h = k == n_mid + 1 ? h_blur : (k == n_mid + 2 ? 2.f * h_blur : 0.f); }
                step(r2, h);
            }
        };
        general(min(max_steps, 2));
        const int fast_end = min(max_steps, n_mid_lo + 1);
#pragma unroll 1
        for (; k < fast_end; ++k) {
            r2 *= inv_scal;
            h *= scal;
            step(r2, h);
        }
        general(max_steps);
    }
#pragma unroll
    for (int j = 0; j < RP; ++j) {
        g[2 * j] = gp[j].x;
        g[2 * j + 1] = gp[j].y;
    }
    if constexpr (ODD) g[R - 1] = go;
    // ---- outputs ---------------------------------------------------------------------------------------------
    float score;
    if (a.want != ASPIRE_OT_PLAN_SIM) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < R; ++t) {
            acc += (lj == 0 && rv[t]) ? wa[t] * f[t] : 0.f;
            acc += (li == 0 && cv[t]) ? wb[t] * g[t] : 0.f;
        }
        score = blk_sum_i<LD>(blk_sum_j<LD>(acc));
        if (a.want == ASPIRE_OT_SIMILARITY) score = -score;
    } else {
        const float rb = rcp_refined(eb);
        load_block(ws.neg, cost);
        float acc = 0.f;
#pragma unroll
        for (int x = 0; x < R; ++x)
#pragma unroll
            for (int y = 0; y < R; ++y) {
                const bool valid = rv[x] && cv[y];
                const float negm = valid ? cost[x][y] : 0.f;
                const float outer = valid ? f[x] + g[y] : 0.f;
                acc += fast_exp(div_r(outer + negm, eb, rb)) * (wa[x] * wb[y]) * negm;
            }
        score = blk_sum_i<LD>(blk_sum_j<LD>(acc));
    }
    // an overflowed / vanished sum sticks to the potentials as inf / nan: poison the pair (sinkhorn_repair_kernel re-solves it)
    if (!(fabsf(score) < 1e30f) || q_len > LD * R || c_len > LD * R) score = __builtin_nanf("");
    if (real && lp == 0) a.scores[p] = score;
}

// Pairs the block form poisoned (NaN score) are solved again, one wave each, by the max-shifted solver.
template <int T>
__global__ void __launch_bounds__(256) sinkhorn_repair_kernel(ScoreArgs a, PairWs<T> ws, int64_t n_slots) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 64;
    if (a.pairing == kPairMapped) {
        base += a.job_off[a.job0];
        n_slots = a.job_off[a.job1];
    }
    if (base >= n_slots) return;
    bool bad = false;
    if (base + lane < n_slots) {
        const PairIdx ix = pair_of_slot(a, base + lane);
        const float s = a.scores[ix.p];
        bad = !(fabsf(s) < 1e30f);
        // hybrid, few long pairs: only those have slots in the workspace (the fused kernel re-solves its own overflowed pairs)
        if (gate_few_long(a) && a.q.len[ix.q_idx] <= 8 && a.c.len[ix.c_idx] <= 8) bad = false;
    }
    unsigned long long todo = __ballot(bad);
    while (todo) {
        const int k = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int64_t slot = base + k;
        const PairIdx ix = pair_of_slot(a, slot);
        PairState<T> st;
        load_pair<T>(st, ws, slot, lane, a.cost_from_neg != 0);
        const float diam = a.diameter == nullptr ? fmaxf(sqrtf(ws.diam2[slot]), kMinDiameter) : group_diameter_of(a, ix);
        sinkhorn_pair<T>(a, st, a.q.len[ix.q_idx], a.c.len[ix.c_idx], diam, ix.p, lane);
    }
}

// ---------------------------------------------------------------------------------------------
// Batch bounding-box diameter (geomloss max_diameter over the call's x and y tensors)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) diameter_kernel(ScoreArgs a, int64_t group, int64_t ngroups, float* out) {
    __shared__ float part[kWaves];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int dofs = threadIdx.x * 4;
    const bool paired = a.pairing == ASPIRE_PAIR_PAIRED;
    // CROSS: block = (query, group) folded into grid.x (grid.y stops at 65535 queries)
    const int64_t qy = paired ? 0 : (int64_t)(blockIdx.x / (uint32_t)ngroups);
    const int64_t g = paired ? (int64_t)blockIdx.x : (int64_t)blockIdx.x - qy * ngroups;
    const int64_t c_lo = g * group, c_hi = min(a.c.n, c_lo + group);
    float4 mn = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    float4 mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    auto add_rows = [&](const RepSet& s, int64_t k, bool use_ext) {
        const int n = (use_ext && s.ext > 0) ? s.ext : s.len[k];
        const float* doc = s.rows + (size_t)s.start[k] * kD;
        for (int r = 0; r < n; ++r) {
            const float4 v = ld4(doc + (size_t)r * kD + dofs);
            mn.x = fminf(mn.x, v.x); mn.y = fminf(mn.y, v.y); mn.z = fminf(mn.z, v.z); mn.w = fminf(mn.w, v.w);
            mx.x = fmaxf(mx.x, v.x); mx.y = fmaxf(mx.y, v.y); mx.z = fmaxf(mx.z, v.z); mx.w = fmaxf(mx.w, v.w);
        }
    };
    bool zero_row = false;
    if (paired) {
        for (int64_t k = c_lo; k < c_hi; ++k) {
            add_rows(a.q, k, true);
            add_rows(a.c, k, true);
        }
    } else {
        add_rows(a.q, qy, false);
        int lmin = 1 << 30, lmax = 0;
        for (int64_t k = c_lo; k < c_hi; ++k) {
            add_rows(a.c, k, false);
            lmin = min(lmin, a.c.len[k]);
            lmax = max(lmax, a.c.len[k]);
        }
        zero_row = lmin != lmax;  // caching_score zero-pads shorter candidates to the group max
    }
    if (zero_row) {
        mn.x = fminf(mn.x, 0.f); mn.y = fminf(mn.y, 0.f); mn.z = fminf(mn.z, 0.f); mn.w = fminf(mn.w, 0.f);
        mx.x = fmaxf(mx.x, 0.f); mx.y = fmaxf(mx.y, 0.f); mx.z = fmaxf(mx.z, 0.f); mx.w = fmaxf(mx.w, 0.f);
    }
    const float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z, dw = mx.w - mn.w;
    const float s = wave_sum(fmaf(dw, dw, fmaf(dz, dz, fmaf(dy, dy, dx * dx))));
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x] = sqrtf(part[0] + part[1] + part[2]);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int check_repsets(const aspire_repset* q, const aspire_repset* c, int64_t D, int pairing) {
    ASPIRE_REQUIRE(q && c, ASPIRE_ERR_INVALID_ARG, "null repset");
    ASPIRE_REQUIRE(D == kD, ASPIRE_ERR_UNSUPPORTED, "encoding dim %lld unsupported (kernels are built for 768)",
                   (long long)D);
    ASPIRE_REQUIRE(pairing == ASPIRE_PAIR_CROSS || pairing == ASPIRE_PAIR_PAIRED, ASPIRE_ERR_INVALID_ARG,
                   "bad pairing %d", pairing);
    ASPIRE_REQUIRE(q->n >= 0 && c->n >= 0, ASPIRE_ERR_INVALID_ARG, "negative document count");
    // pair_distances.py:46  assert (qef_batch_size == cef_batch_size)
    ASPIRE_REQUIRE(pairing != ASPIRE_PAIR_PAIRED || q->n == c->n, ASPIRE_ERR_INVALID_ARG,
                   "paired scoring needs equal batch sizes (query %lld vs cand %lld)", (long long)q->n, (long long)c->n);
    ASPIRE_REQUIRE(q->ext >= 0 && c->ext >= 0, ASPIRE_ERR_INVALID_ARG, "negative ext");
    return ASPIRE_OK;
}

RepSet to_dev(const aspire_repset* s) { return RepSet{s->rows, s->start, s->len, s->n, s->ext}; }

int max_rows_of(const aspire_repset* q, const aspire_repset* c) {
    const int mq = q->ext > 0 ? q->ext : q->max_len;
    const int mc = c->ext > 0 ? c->ext : c->max_len;
    return mq > mc ? mq : mc;
}

template <typename F>
int dispatch_T(int max_rows, F&& f) {
    if (max_rows <= 8) return f(std::integral_constant<int, 1>{});
    if (max_rows <= 16) return f(std::integral_constant<int, 2>{});
    if (max_rows <= 24) return f(std::integral_constant<int, 3>{});
    if (max_rows <= 32) return f(std::integral_constant<int, 4>{});
    set_error("documents with more than %d sentence rows are not supported (got %d)", 8 * kMaxT, max_rows);
    return ASPIRE_ERR_UNSUPPORTED;
}

// Query chunking.  CROSS: grid.x = candidates, grid.y = query chunks; queries are split over grid.y only while
// the grid is too small to fill 256 CUs several times over (each block re-reads its candidate from L2 per
// query chunk).
int query_chunks(ScoreArgs& a) {
    if (a.pairing == ASPIRE_PAIR_PAIRED) {
        a.q_per_block = 1;
        return 1;
    }
    int64_t chunks = 1;
    const int64_t target_blocks = 256 * 8;
    while (a.c.n * chunks < target_blocks && chunks < a.q.n) chunks *= 2;
    int64_t qpb = (a.q.n + chunks - 1) / chunks;
    chunks = (a.q.n + qpb - 1) / qpb;
    a.q_per_block = (int)qpb;
    return (int)chunks;
}

}  // namespace
}  // namespace aspire

using namespace aspire;

#ifdef ASPIRE_PHASE_CLOCK
extern "C" void aspire_debug_k1_buffer(void* p) {
    long long* q = (long long*)p;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_k1dbg), &q, sizeof(q));
}
#endif

extern "C" int aspire_max_sents(void) { return generic_max_rows(); }

extern "C" int aspire_l2agg_scores_f32(const aspire_repset* q, const aspire_repset* c, int64_t D, int pairing,
                                       int cdist_mode, int agg, double temp, float* scores, float* pair_sims,
                                       float* pair_softmax, void* stream) {
    if (int rc = check_repsets(q, c, D, pairing)) return rc;
    ASPIRE_REQUIRE(agg == ASPIRE_AGG_MAX || agg == ASPIRE_AGG_TOP2 || agg == ASPIRE_AGG_ATTENTION, ASPIRE_ERR_INVALID_ARG,
                   "bad aggregation %d", agg);
    ASPIRE_REQUIRE(agg != ASPIRE_AGG_ATTENTION || temp > 0, ASPIRE_ERR_INVALID_ARG, "attention temperature must be positive");
    ASPIRE_REQUIRE(!pair_softmax || agg == ASPIRE_AGG_ATTENTION, ASPIRE_ERR_INVALID_ARG,
                   "pair_softmax is an output of the attention aggregation only");
    if (q->n == 0 || c->n == 0) return ASPIRE_OK;   // nothing to score (an empty pool has no buffers either)
    ASPIRE_REQUIRE(scores, ASPIRE_ERR_INVALID_ARG, "scores is null");
    ASPIRE_REQUIRE((!pair_sims && !pair_softmax) || (q->ext > 0 && c->ext > 0), ASPIRE_ERR_INVALID_ARG,
                   "pair outputs need padded extents (ext > 0)");
    const bool one_form = (cdist_mode & ASPIRE_CDIST_ONE_FORM) != 0, center = (cdist_mode & ASPIRE_CDIST_CENTER) != 0;
    cdist_mode &= ~(ASPIRE_CDIST_ONE_FORM | ASPIRE_CDIST_CENTER);
    ASPIRE_REQUIRE(!one_form || agg == ASPIRE_AGG_MAX, ASPIRE_ERR_UNSUPPORTED, "ASPIRE_CDIST_ONE_FORM is built for the max-sim score only");
    ScoreArgs a{};
    a.q = to_dev(q);
    a.c = to_dev(c);
    a.q_planes = q->planes;
    a.c_planes = c->planes;
    a.c_box = c->doc_box;
    a.pairing = pairing;
    a.cdist_mode = cdist_mode;
    a.agg = agg;
    a.temp = temp;
    a.scores = scores;
    a.out_pairsims = pair_sims;
    a.out_plan = pair_softmax;
    a.center = center;
    if (one_form)      // every pair through the long-form kernel, whatever the size of the call (include/aspire_hip.h)
        return launch_pair_generic(a, 1, 0, q->ext > 0 ? q->ext : q->max_len, c->ext > 0 ? c->ext : c->max_len, (hipStream_t)stream);
    // both sides carry fp16 planes: the matrix-pipe tiles (gramp.hip) whatever the number of queries
    if (agg == ASPIRE_AGG_MAX && !pair_sims && gram_planes_wanted_l2max(q, c, pairing))
        return launch_pair_gram_l2max(a, q->max_len, c->max_len, (hipStream_t)stream);
    // ONE query against a big pool of 9 .. 16-row documents: the streaming kernel's max-sim form (tile16.hip)
    if (agg == ASPIRE_AGG_MAX && !pair_sims && q->n == 1 && c->n >= 4096 && tile16_path_ok(q, c, pairing) &&
        pairing == ASPIRE_PAIR_CROSS && tuning().cost_path != 1 && tuning().ot_form != 1) {
        a.cand0 = 0;
        a.cand1 = c->n;
        return launch_pair_tile16_l2max(a, (c->n + 1) / 2, (hipStream_t)stream);
    }
    if (agg == ASPIRE_AGG_MAX && !pair_sims && gram_path_wanted(q, c, pairing))
        return launch_pair_gram_l2max(a, q->max_len, c->max_len, (hipStream_t)stream);
    // Few queries against a big pool of short documents (CSR): the fused kernel's streaming phase with a max epilogue --
    // four candidates per wave, dot products on the matrix pipe (l2max_kernel<1> is one workgroup per candidate with VALU
    // difference sums: 114 us at 1 x 20 000 against the stream's 88)
    {
        const int form_t = tuning().ot_form;
        const int64_t groups4 = (c->n + 3) / 4 * q->n;
        if (agg == ASPIRE_AGG_MAX && !pair_sims && pairing == ASPIRE_PAIR_CROSS && fused_path_ok(q, c) &&
            (form_t == 3 || (form_t == 0 && groups4 >= (q->n == 1 ? kStreamMinGroups1 : 2048)))) {
            a.cand0 = 0;
            a.cand1 = c->n;
            return launch_pair_fused_l2max(a, groups4, (hipStream_t)stream);
        }
    }
    // Documents beyond the tile kernels' 32 rows (max-sim only): padded tensors that wide go through the one-workgroup-
    // per-pair kernel for every pair; CSR pools run the tile kernel on 32-row tiles first (it covers every pair of short
    // documents) and the long-document kernel then rewrites the pairs that hold a longer one.
    const int rows_q = q->ext > 0 ? q->ext : q->max_len, rows_c = c->ext > 0 ? c->ext : c->max_len;
    const bool long_docs = max_rows_of(q, c) > 8 * kMaxT;
    if (long_docs && (q->ext > 0 || c->ext > 0)) return launch_pair_generic(a, 1, 0, rows_q, rows_c, (hipStream_t)stream);
    dim3 grid;
    grid = dim3((unsigned)a.c.n, (unsigned)query_chunks(a), 1);
    const int tile_rows = long_docs ? 8 * kMaxT : max_rows_of(q, c);
    const int rc_tiles = dispatch_T(tile_rows, [&](auto tc) -> int {
        constexpr int T = decltype(tc)::value;
        hipLaunchKernelGGL(l2max_kernel<T>, grid, dim3(kBlock), Lds<T>::kTotal * sizeof(float), (hipStream_t)stream, a);
        ASPIRE_LAUNCH_OK();
        return (int)ASPIRE_OK;
    });
    if (rc_tiles || !long_docs) return rc_tiles;
    return launch_pair_generic(a, 1, 8 * kMaxT, rows_q, rows_c, (hipStream_t)stream);
}

extern "C" int aspire_l2max_scores_f32(const aspire_repset* q, const aspire_repset* c, int64_t D, int pairing,
                                       int cdist_mode, float* scores, float* pair_sims, void* stream) {
    return aspire_l2agg_scores_f32(q, c, D, pairing, cdist_mode, ASPIRE_AGG_MAX, 1.0, scores, pair_sims, nullptr, stream);
}

namespace {
// bytes of workspace per pair slot at tile size T
size_t slot_bytes(int max_rows) {
    const int T = (max_rows + 7) / 8;
    return (size_t)(2 * 64 * T * T + 1) * sizeof(float);
}
size_t qbox_bytes(const aspire_repset* q) { return (size_t)q->n * 2 * kD * sizeof(float); }
constexpr size_t kWsCap = (size_t)1 << 30;  // suggested workspace is capped at 1 GiB; larger jobs run in chunks
constexpr size_t kWsSlack = 48;             // alignment of the box tables behind the pair slots
size_t align16(size_t n) { return (n + 15) & ~(size_t)15; }
// workspace bytes one candidate of a chunk needs: its pair slots (+ its bounding box on the matrix-core path)
size_t per_cand_bytes(const aspire_repset* q, const aspire_repset* c, int pairing) {
    return slot_bytes(max_rows_of(q, c)) * (pairing == ASPIRE_PAIR_CROSS ? (size_t)q->n : 1) +
           (gram_path_wanted(q, c, pairing) ? gram_extra_bytes_per_cand() : 0);
}
}  // namespace

// (a multiple of 16 bytes: the rank scratch of aspire_ot_rank_f32 -- 64-bit keys -- sits right behind it)
extern "C" size_t aspire_ot_workspace_bytes(const aspire_repset* q, const aspire_repset* c, int pairing) {
    if (!q || !c || q->n <= 0 || c->n <= 0) return 0;
    const size_t per_cand = per_cand_bytes(q, c, pairing);
    const size_t full = per_cand * (size_t)c->n;
    if (full <= kWsCap) return align16(full + qbox_bytes(q) + kWsSlack);
    return align16((per_cand > kWsCap ? per_cand : kWsCap / per_cand * per_cand) + qbox_bytes(q) + kWsSlack);
}

namespace {
// rank request of aspire_ot_rank_f32 (k == 0: scores only)
struct RankReq {
    int64_t k, idx_base;
    float* top_scores;
    int64_t* top_idx;
    uint64_t* keys;
};

int check_ot_params(const aspire_ot_params* prm, int want) {
    ASPIRE_REQUIRE(prm, ASPIRE_ERR_INVALID_ARG, "null params");
    ASPIRE_REQUIRE(want == ASPIRE_OT_DISTANCE || want == ASPIRE_OT_PLAN_SIM || want == ASPIRE_OT_SIMILARITY, ASPIRE_ERR_INVALID_ARG,
                   "bad want %d", want);
    ASPIRE_REQUIRE(prm->blur > 0 && prm->scaling > 0 && prm->scaling < 1 && prm->sent_sm_temp > 0,
                   ASPIRE_ERR_INVALID_ARG, "need blur > 0, 0 < scaling < 1, temp > 0");
    return ASPIRE_OK;
}

void fill_ot_args(ScoreArgs& a, const aspire_repset* q, const aspire_repset* c, int pairing, const aspire_ot_params* prm,
                  const float* diameter, int64_t diam_group, int want, float* scores) {
    a.q = to_dev(q);
    a.c = to_dev(c);
    a.q_planes = q->planes;
    a.c_planes = c->planes;
    a.c_box = c->doc_box;
    a.pairing = pairing;
    a.cdist_mode = prm->cdist_mode;
    a.blur = prm->blur;
    a.scaling = prm->scaling;
    a.temp = prm->sent_sm_temp;
    a.log_blur = std::log(prm->blur);
    a.log_scaling = std::log(prm->scaling);
    a.log2_blur = (float)(a.log_blur * 1.4426950408889634);
    a.log2_scaling = (float)(a.log_scaling * 1.4426950408889634);
    a.diameter = diameter;
    a.diam_group = diameter ? diam_group : 1;
    a.n_groups = diameter ? (c->n + diam_group - 1) / diam_group : 0;
    a.want = want;
    a.scores = scores;
    a.center = (prm->flags & ASPIRE_OT_FLAG_CENTER) != 0;
}

// The hybrid forms (fused kernel in front of the 16-row kernels, ScoreArgs::gate) need a solve stage that leaves the short pairs'
// scores alone: the block kernels and their repair kernel check the gate, the one-solve-per-wave kernel (SINKHORN=wave) does not.
bool sinkhorn_form_honours_gate() {
    const int f = tuning().sinkhorn_form;
    return f == 0 || f == 3 || f == 5 || f == 6 || f == 7;
}

// ---- stage 1 of an otAspire pass: pairwise costs of the chunk [a.cand0, a.cand1) -> workspace slots ---------------------
// (MAPPED pairing: the whole batch, or with `tile_blocks` > 0 the jobs [a.job0, a.job1) on the throughput kernel.)
template <int T>
int launch_cost_stage(const ScoreArgs& a, const aspire_repset* q, const aspire_repset* c, const PairWs<T>& ws, int64_t n_slots,
                      int qchunks, bool gram, float* qbox, float* cbox, bool first_chunk, hipStream_t stream) {
    const bool csr = q->ext == 0 && c->ext == 0;
    if (gram) {
        // many queries or long documents: Gram tiles on the matrix cores (gram.hip)
        return launch_pair_gram_ot(a, T, q->max_len, c->max_len, ws.cost, ws.neg, a.diameter ? nullptr : ws.diam2, qbox, cbox, stream);
    }
    if (T == 1 && csr) {
        PairWs<1> ws1{ws.cost, ws.neg, ws.diam2};
        const int64_t ncand = a.cand1 - a.cand0;
        // groups of four candidates x queries (MAPPED: an upper bound; the kernel reads the exact range from grp_off)
        const int64_t groups4 = a.pairing == kPairMapped ? (int64_t)(a.job1 - a.job0) * a.max_job_groups
                                                         : (ncand + 3) / 4 * (a.pairing == ASPIRE_PAIR_CROSS ? q->n : 0);
        const int form_t = tuning().ot_form;
        const bool tile = a.pairing == kPairMapped ? a.tile_form
                                                   : (a.pairing == ASPIRE_PAIR_CROSS && (form_t == 2 || (form_t != 1 && groups4 >= 2048)));
        if (tile) {
            // enough groups of 4 candidates to fill the chip: tiled form (lanes own finished (i,j) sums), one group per
            // wave (measured 4.7 TB/s algorithmic at 1 x 20 000 against 1.8 TB/s for the accumulate-then-reduce kernel)
            if (!a.diameter && first_chunk && a.pairing != kPairMapped) {   // per-coordinate boxes of the queries, once per call
                hipLaunchKernelGGL(doc_box_kernel, dim3((unsigned)q->n), dim3(192), 0, stream, a.q, qbox);
                ASPIRE_LAUNCH_OK();
            }
            const int64_t waves = groups4 < 256 * 8 ? groups4 : 256 * 8;
            hipLaunchKernelGGL((pair_tile_kernel<2, 1>), dim3((unsigned)((waves + 3) / 4)), dim3(256),
                               4 * TileCfg<2>::kLdsFloats * sizeof(float), stream, a, ws1, qbox);
        } else {
            // small grids are latency bound: three waves per pair (a third of the coordinates each), persistent and
            // software pipelined (measured 15.6 us per launch at 50-250 pairs against 20-23 us for the tiled form with its
            // stages split over three or four waves).  One pair per workgroup up to 2048 pairs (at ~1000 pairs it beats
            // 512 persistent workgroups with two each, alone and beside other launches); beyond, 512 persistent
            // workgroups = what is resident at two per CU.
            const int cap_t = tuning().cost1_blocks;
            const int64_t cap = cap_t > 0 ? cap_t : (n_slots <= 2048 ? 2048 : 512);
            const int64_t blocks = n_slots < cap ? n_slots : cap;
            hipLaunchKernelGGL(pair_cost1_kernel, dim3((unsigned)blocks), dim3(kBlock), Lds<1>::kTotal * sizeof(float), stream, a,
                               ws1, 1u);
        }
    } else if (T == 2 && tile16_path_ok(q, c, a.pairing) && (a.pairing != kPairMapped || a.grp_off != nullptr) &&
               tuning().ot_form != 1 &&
               (tuning().ot_form == 2 || (a.pairing == kPairMapped ? 2 * (int64_t)(a.job1 - a.job0) * a.max_job_groups
                                                                      : (a.cand1 - a.cand0 + 1) / 2 * q->n) >= 2048)) {
        // documents of 9 .. 16 rows, enough pairs of candidates to fill the chip: the streaming kernel (tile16.hip)
        if (!a.diameter && first_chunk && a.pairing != kPairMapped) {   // per-coordinate boxes of the queries, once per call
            hipLaunchKernelGGL(doc_box_kernel, dim3((unsigned)q->n), dim3(192), 0, stream, a.q, qbox);
            ASPIRE_LAUNCH_OK();
        }
        const int64_t items = a.pairing == kPairMapped ? 2 * (int64_t)(a.job1 - a.job0) * a.max_job_groups
                                                       : (a.cand1 - a.cand0 + 1) / 2 * q->n;
        return launch_pair_tile16(a, ws.cost, ws.neg, ws.diam2, items, qbox, stream);
    } else if (csr && n_slots < 512) {
        // CSR documents of more than 8 rows: every 8 x 8 sub-tile of every pair is an item of the small-pool kernel
        // while the pairs alone would not fill the chip (1 x 125 x 20: 125 workgroups walking 9 tiles each -> 1024
        // side by side, 54 -> 36 us; from ~1000 pairs the per-pair kernel is ahead again)
        PairWs<1> ws1{ws.cost, ws.neg, ws.diam2};
        const int64_t items = n_slots * T * T;
        hipLaunchKernelGGL(pair_cost1_sub_kernel, dim3((unsigned)(items < 1024 ? items : 1024)), dim3(kBlock),
                           Lds<1>::kTotal * sizeof(float), stream, a, ws1, (uint32_t)T);
    } else if (csr && a.center) {        // rows with a large common component: the same kernel on centred rows
        hipLaunchKernelGGL((pair_cost_kernel<T, false, true>), dim3((unsigned)(a.cand1 - a.cand0), (unsigned)qchunks, 1), dim3(kBlock),
                           Lds<T>::kTotal * sizeof(float), stream, a, ws);
    } else if (csr) {
        hipLaunchKernelGGL((pair_cost_kernel<T, false>), dim3((unsigned)(a.cand1 - a.cand0), (unsigned)qchunks, 1), dim3(kBlock),
                           Lds<T>::kTotal * sizeof(float), stream, a, ws);
    } else {
        hipLaunchKernelGGL((pair_cost_kernel<T, true>), dim3((unsigned)(a.cand1 - a.cand0), (unsigned)qchunks, 1), dim3(kBlock),
                           Lds<T>::kTotal * sizeof(float), stream, a, ws);
    }
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}

// ---- stage 2: one Sinkhorn solve per workspace slot.  n_slots: the slots of this launch (MAPPED: an upper bound, the
// kernels read the exact range of jobs [a.job0, a.job1) from job_off). --------------------------------------------------
// One solve per wave has the lowest latency (15.4 vs 26.7 us per call at 50 pairs), the block forms several times the
// throughput; measured crossovers (1 x N x 8, cost + solve, us): N = 5000: wave 58 / 16-lane block 64 / 4-lane block 68;
// 8000: 88 / 83 / 89; 12000: 119 / 109 / 106.  S = 12 / 20: even at 2000 / ~2500, block ahead at 3000.
template <int T>
int launch_sinkhorn_stage(const ScoreArgs& a, const PairWs<T>& ws, int64_t n_slots, int max_rows, bool extra, int form_hint,
                          hipStream_t stream) {
    const int pinned = tuning().sinkhorn_form;
    const int form = pinned ? pinned : form_hint ? form_hint
                     : T == 1 ? (n_slots < 7000 ? 1 : n_slots < 10000 ? 5 : 3)
                              : (n_slots >= 2500 ? 3 : 1);
    if (form >= 3 && !extra) {
        // lanes per pair side LD and entries per lane side R: the smallest block grid that covers max_rows
        auto launch_block = [&](auto ldc, auto rc) {
            constexpr int LD = decltype(ldc)::value, R = decltype(rc)::value, PPB = 4 * 64 / (LD * LD);
            if constexpr (LD * R <= 8 * T && LD * R > 8 * (T - 1)) {
                hipLaunchKernelGGL((sinkhorn_block_kernel<T, LD, R>), dim3((unsigned)((n_slots + PPB - 1) / PPB)),
                                   dim3(256), 0, stream, a, ws, n_slots);
            }
        };
        using I2 = std::integral_constant<int, 2>;
        using I4 = std::integral_constant<int, 4>;
        const int r4 = (max_rows + 3) / 4;
        // Lanes per pair.  The dense layouts (documents of <= 8 rows: ONE lane per pair, 8 x 8 entries, no cross-lane step at
        // all; 9 .. 16 rows: 2 x 2 lanes of 6 x 6 / 8 x 8 entries) need a third fewer issue slots per pair than the wide ones
        // (2 x 2 lanes of 4 x 4; 4 x 4 lanes of 3 x 3 / 4 x 4): 32 x 50 000 x 8 0.90 -> 0.74 ms per launch, 128 x 8192 x 12
        // 1.10 -> 0.76, x 16 1.59 -> 1.27 -- but hold 64 / 16 pairs per wave, so only grids that still fill the chip take them.
        const bool dense = form == 6 || (form != 7 && n_slots >= (T == 1 ? 196608 : 49152));
        if (T == 1 && form == 5) launch_block(I4{}, I2{});
        else if (T == 1 && dense) launch_block(std::integral_constant<int, 1>{}, std::integral_constant<int, 8>{});
        else if (T == 1) launch_block(I2{}, I4{});
        else if (r4 == 3 && dense) launch_block(I2{}, std::integral_constant<int, 6>{});
        else if (r4 == 4 && dense) launch_block(I2{}, std::integral_constant<int, 8>{});
        else if (r4 == 3) launch_block(I4{}, std::integral_constant<int, 3>{});
        else if (r4 == 4) launch_block(I4{}, I4{});
        else if (r4 == 5) launch_block(I4{}, std::integral_constant<int, 5>{});
        else if (r4 == 6) launch_block(I4{}, std::integral_constant<int, 6>{});
        else if (r4 == 7) launch_block(I4{}, std::integral_constant<int, 7>{});
        else launch_block(I4{}, std::integral_constant<int, 8>{});
        ASPIRE_LAUNCH_OK();
        if (form != 4)      // pairs whose sums left fp32 range (NaN score) are solved again with the max-shifted solver
            hipLaunchKernelGGL(sinkhorn_repair_kernel<T>, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, stream, a, ws,
                               n_slots);
    } else {
        hipLaunchKernelGGL(sinkhorn_kernel<T>, dim3((unsigned)((n_slots + 3) / 4)), dim3(256), 0, stream, a, ws, n_slots);
    }
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}

int ot_run_tiles(const aspire_repset* q, const aspire_repset* c, int64_t D, int pairing, const aspire_ot_params* prm,
                 const float* diameter, int64_t diam_group, int want, float* scores, float* out_qdistr, float* out_cdistr,
                 float* out_pairsims, float* out_plan, void* workspace, size_t workspace_bytes, void* stream, const RankReq& rank,
                 bool cost_only);
}  // namespace
namespace aspire {
namespace {
// (defined with the batched entry points below)
__global__ void chunk_prep_kernel(RepSet q, RepSet c, const int32_t* __restrict__ job_off, float* __restrict__ qbox, int32_t* __restrict__ cand_job,
                                  int32_t* __restrict__ counter, int32_t* __restrict__ grp_rec, int region_cap);
// CHUNK items without a counter (ScoreArgs::chunk_regions): slices = J * parts <= 64; a slice's region holds min(384, max_job) records
inline int chunk_regions_of(int64_t J, int64_t max_job);
inline int chunk_region_cap_of(int64_t max_job) { return (int)(max_job < 384 ? (max_job > 0 ? max_job : 1) : 384); }
int64_t chunk_parts(int64_t max_job);
int64_t chunk_items_bound(int64_t J, int64_t C, int64_t max_job);
// smallest pool / batch (candidates) that takes the CHUNK / REC forms (below: the small-batch kernels; tools/csfbench.py sweeps)
constexpr int64_t kChunkMinCands = 256;
// pairs of short documents per single-pool call that take pair_one_kernel (tools/experiments/singlejob.py sweeps)
constexpr int64_t kOneMinPairs = 1, kOneMaxPairs = 8192;        // (ONE form for every small grid: a pair scored alone, in a subset or in a shard gets the same bits)
}  // namespace
}  // namespace aspire
namespace {

// Documents beyond the tile kernels' 32 rows: padded tensors that wide go through the one-workgroup-per-pair kernel
// (generic.hip) for every pair; CSR pools run the tile kernels with their documents' bound clamped to 32 rows (every pair
// of short documents is scored there, a pair that holds a longer one gets NaN) and the long-document kernel then rewrites
// exactly those pairs.  The rank follows.
int ot_run(const aspire_repset* q, const aspire_repset* c, int64_t D, int pairing, const aspire_ot_params* prm,
           const float* diameter, int64_t diam_group, int want, float* scores, float* out_qdistr, float* out_cdistr,
           float* out_pairsims, float* out_plan, void* workspace, size_t workspace_bytes, void* stream, const RankReq& rank,
           bool cost_only = false) {
    if (int rc = check_repsets(q, c, D, pairing)) return rc;
    const int tile_max = 8 * kMaxT;
    // ONE_FORM (include/aspire_hip.h): every pair through the long-form kernel, whatever the size of the call
    const bool one_form = prm && (prm->flags & ASPIRE_OT_FLAG_ONE_FORM) && !cost_only;
    if (q->n == 0 || c->n == 0 || (max_rows_of(q, c) <= tile_max && !one_form))
        return ot_run_tiles(q, c, D, pairing, prm, diameter, diam_group, want, scores, out_qdistr, out_cdistr, out_pairsims, out_plan,
                            workspace, workspace_bytes, stream, rank, cost_only);
    if (int rc = check_ot_params(prm, want)) return rc;
    ASPIRE_REQUIRE(scores, ASPIRE_ERR_INVALID_ARG, "null scores");
    ASPIRE_REQUIRE(!diameter || diam_group > 0, ASPIRE_ERR_INVALID_ARG, "diam_group must be positive");
    const bool extra = out_qdistr || out_cdistr || out_pairsims || out_plan;
    ASPIRE_REQUIRE(!extra || (q->ext > 0 && c->ext > 0), ASPIRE_ERR_INVALID_ARG, "pair outputs need padded extents (ext > 0)");
    const int rows_q = q->ext > 0 ? q->ext : q->max_len, rows_c = c->ext > 0 ? c->ext : c->max_len;
    ScoreArgs a{};
    fill_ot_args(a, q, c, pairing, prm, diameter, diam_group, want, scores);
    a.out_qdistr = out_qdistr;
    a.out_cdistr = out_cdistr;
    a.out_pairsims = out_pairsims;
    a.out_plan = out_plan;
    int skip = 0;
    if (q->ext == 0 && c->ext == 0 && !one_form) {
        aspire_repset q32 = *q, c32 = *c;
        q32.max_len = q->max_len < tile_max ? q->max_len : tile_max;
        c32.max_len = c->max_len < tile_max ? c->max_len : tile_max;
        if (int rc = ot_run_tiles(&q32, &c32, D, pairing, prm, diameter, diam_group, want, scores, nullptr, nullptr, nullptr, nullptr,
                                  workspace, workspace_bytes, stream, RankReq{0, 0, nullptr, nullptr, nullptr}, cost_only))
            return rc;
        skip = tile_max;
    }
    if (cost_only) return ASPIRE_OK;
    if (int rc = launch_pair_generic(a, 0, skip, rows_q, rows_c, (hipStream_t)stream)) return rc;
    if (rank.k > 0) {
        const size_t need = aspire_topk_workspace_bytes(q->n, c->n, rank.k);
        void* tws = need ? (char*)workspace + workspace_bytes : nullptr;
        return topk_run(scores, q->n, c->n, rank.k, rank.idx_base, rank.top_scores, rank.top_idx, rank.keys, tws, need, stream);
    }
    return ASPIRE_OK;
}

int ot_run_tiles(const aspire_repset* q, const aspire_repset* c, int64_t D, int pairing, const aspire_ot_params* prm,
                 const float* diameter, int64_t diam_group, int want, float* scores, float* out_qdistr, float* out_cdistr,
                 float* out_pairsims, float* out_plan, void* workspace, size_t workspace_bytes, void* stream, const RankReq& rank,
                 bool cost_only) {
    if (int rc = check_repsets(q, c, D, pairing)) return rc;
    if (q->n == 0 || c->n == 0) return ASPIRE_OK;   // nothing to score (an empty pool has no buffers either)
    if (int rc = check_ot_params(prm, want)) return rc;
    ASPIRE_REQUIRE(scores, ASPIRE_ERR_INVALID_ARG, "null scores");
    const bool extra = out_qdistr || out_cdistr || out_pairsims || out_plan;
    ASPIRE_REQUIRE(!extra || (q->ext > 0 && c->ext > 0), ASPIRE_ERR_INVALID_ARG,
                   "pair outputs need padded extents (ext > 0)");
    ASPIRE_REQUIRE(!diameter || diam_group > 0, ASPIRE_ERR_INVALID_ARG, "diam_group must be positive");
    const int max_rows = max_rows_of(q, c);
    const size_t per_cand = per_cand_bytes(q, c, pairing);
    // ONE query against a big pool of 9 .. 16-row documents: the streaming kernel (tile16.hip) beats the 32-column Gram tiles,
    // whose 12 .. 16 real query rows fill a third to a half of the MFMA tile (1 x 20 000 x 12 otAspire: 247 vs 363 us; at two
    // queries they tie, from three the Gram tiles win: 623 vs 565 us)
    // both sides on fp16 planes, the pool's boxes cached: the plane tiles whatever the number of queries (gram.hip)
    const bool planes_ot = !extra && gram_planes_wanted_ot(q, c, pairing, diameter != nullptr) && tuning().ot_form == 0;
    const bool stream16 = !planes_ot && q->n == 1 && c->n >= 4096 && tile16_path_ok(q, c, pairing) && tuning().cost_path != 1 &&
                          tuning().ot_form != 1;
    const int form_t = tuning().ot_form;
    // ONE short query (facet-selected rows) against a pool of abstracts of up to 32 rows: the fused kernel's CHUNK form, as in
    // ot_rank_batch (1 x 20 000 x (3..20): 674 us on the Gram tiles + block Sinkhorn before) -- the item records and their counter
    // take the (unused) front of the pair-slot workspace.
    const bool chunk1 = pairing == ASPIRE_PAIR_CROSS && q->n == 1 && q->ext == 0 && c->ext == 0 && q->max_len <= 8 && c->max_len > 8 &&
                        c->max_len <= 8 * kMaxT && c->n >= kChunkMinCands && c->n < ((int64_t)1 << 30) && !extra && !cost_only && !diameter &&
                        (form_t == 0 || form_t == 4) && prm->scaling >= 0.25 && !tuning().fused_nosolve && !tuning().fused_valu &&
                        tuning().cost_path == 0 && workspace &&
                        (size_t)(chunk_items_bound(1, c->n, c->n) + 2) * 64 + 256 + qbox_bytes(q) + 64 <= workspace_bytes;
    const bool gram = (gram_path_wanted(q, c, pairing) || planes_ot) && !stream16 && !chunk1;
    ASPIRE_REQUIRE(workspace && workspace_bytes >= per_cand + qbox_bytes(q) + kWsSlack, ASPIRE_ERR_INVALID_ARG,
                   "workspace too small: %zu bytes given, at least %zu needed (aspire_ot_workspace_bytes suggests %zu)",
                   workspace_bytes, per_cand, aspire_ot_workspace_bytes(q, c, pairing));
    ScoreArgs a{};
    fill_ot_args(a, q, c, pairing, prm, diameter, diam_group, want, scores);
    a.out_qdistr = out_qdistr;
    a.out_cdistr = out_cdistr;
    a.out_pairsims = out_pairsims;
    a.out_plan = out_plan;
    // the matrix-pipe cost tiles derive -cdist and geomloss's cost from ONE distance: they store -cdist only and the solve stage takes
    // cost = max(cdist, 1e-4) from it -- half the tile bytes written and read (the debug cost stage keeps both buffers)
    a.cost_from_neg = gram && !cost_only;
    const int qchunks = query_chunks(a);
    const int64_t cand_per_chunk = (int64_t)((((workspace_bytes - qbox_bytes(q)) & ~(size_t)15) - 32) / per_cand);
    const int64_t pairs_per_cand = pairing == ASPIRE_PAIR_PAIRED ? 1 : q->n;
    const size_t ot_bytes = workspace_bytes;
    // query boxes sit at a fixed place (the tail of the workspace) so that every candidate chunk finds them
    float* qbox = (float*)((char*)workspace + ((workspace_bytes - qbox_bytes(q)) & ~(size_t)15));
    // Few queries against a big pool of short documents: costs and solves in ONE launch, no workspace slots, no candidate
    // chunks (fused.hip).
    const int64_t groups4_all = (c->n + 3) / 4 * q->n;
    const bool fused = pairing == ASPIRE_PAIR_CROSS && !extra && !gram && !cost_only && fused_path_ok(q, c) &&
                       (form_t == 3 || (form_t == 0 && groups4_all >= (q->n == 1 ? kStreamMinGroups1 : 2048)));
    if (fused) {
        a.cand0 = 0;
        a.cand1 = c->n;
        const bool inbox = fused_inbox_ok(q, diameter);      // ONE query: the kernel forms its box itself
        if (!diameter && !inbox) {   // per-coordinate boxes of the queries (the kernel adds each candidate's rows)
            hipLaunchKernelGGL(doc_box_kernel, dim3((unsigned)q->n), dim3(192), 0, (hipStream_t)stream, a.q, qbox);
            ASPIRE_LAUNCH_OK();
        }
#ifdef ASPIRE_EXPERIMENT_SPLIT      // round 5's role-split kernel: an experiment that lost, built only by tools/experiments/split/build.sh
        if (inbox && split_path_ok(groups4_all, prm) && c->n < ((int64_t)1 << 31) - 8) {
            if (int rc = launch_pair_split(a, (hipStream_t)stream)) return rc;
        } else
#endif
        if (int rc = launch_pair_fused(a, groups4_all, inbox ? nullptr : qbox, (hipStream_t)stream)) return rc;
    }
    if (chunk1) {
        a.cand0 = 0;
        a.cand1 = c->n;
        int32_t* counter = (int32_t*)workspace;
        int32_t* recs = (int32_t*)((char*)workspace + 256);
        a.grp_off = counter;
        a.grp_rec = recs;
        a.chunk_regions = chunk_regions_of(1, c->n);
        a.chunk_region_cap = chunk_region_cap_of(c->n);
        if (a.chunk_regions == 0) ASPIRE_HIP_OK(hipMemsetAsync(counter, 0, sizeof(int32_t), (hipStream_t)stream));
        hipLaunchKernelGGL(chunk_prep_kernel, dim3(1, (unsigned)chunk_parts(c->n) + 1), dim3(192), 0, (hipStream_t)stream, a.q, a.c,
                           (const int32_t*)nullptr, qbox, (int32_t*)nullptr, counter, recs, a.chunk_regions > 0 ? a.chunk_region_cap : 0);
        ASPIRE_LAUNCH_OK();
        if (int rc = launch_pair_fused_chunk(a, chunk_items_bound(1, c->n, c->n), qbox, (hipStream_t)stream)) return rc;
    }
    const int rc_run = (fused || chunk1) ? (int)ASPIRE_OK : dispatch_T(max_rows, [&](auto tc) -> int {
        constexpr int T = decltype(tc)::value;
        for (int64_t c0 = 0; c0 < c->n; c0 += cand_per_chunk) {
            a.cand0 = c0;
            a.cand1 = c0 + cand_per_chunk < c->n ? c0 + cand_per_chunk : c->n;
            const int64_t n_slots = (a.cand1 - a.cand0) * pairs_per_cand;
            PairWs<T> ws;
            ws.cost = (float*)workspace;
            ws.neg = ws.cost + n_slots * PairWs<T>::kEntries;
            ws.diam2 = ws.neg + n_slots * PairWs<T>::kEntries;
            float* cbox = (float*)(((uintptr_t)(ws.diam2 + n_slots) + 15) & ~(uintptr_t)15);
            if constexpr (T == 1) {
                // a small pool of short documents (the per-query call of evaluate.py:58-76): one wave per pair, costs and solve in ONE
                // launch (pair_one_kernel) instead of the cost launch + the Sinkhorn launch and the workspace between them
                const int64_t groups4 = (a.cand1 - a.cand0 + 3) / 4 * (pairing == ASPIRE_PAIR_CROSS ? q->n : 1);
                const bool small_grid = !(pairing == ASPIRE_PAIR_CROSS && groups4 >= 2048);        // (beyond: the tiled cost kernel's grid)
                if (q->ext == 0 && c->ext == 0 && !gram && !cost_only && small_grid && n_slots >= kOneMinPairs && n_slots <= kOneMaxPairs &&
                    (form_t == 5 || (form_t == 0 && tuning().sinkhorn_form == 0 && tuning().cost_path == 0 && tuning().cost1_blocks == 0))) {
                    hipLaunchKernelGGL(pair_one_kernel, dim3((unsigned)((n_slots + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, n_slots);
                    ASPIRE_LAUNCH_OK();
                    continue;
                }
            }
            if (int rc = launch_cost_stage<T>(a, q, c, ws, n_slots, qchunks, gram, qbox, cbox, c0 == 0, (hipStream_t)stream)) return rc;
            if (cost_only) continue;
            if (int rc = launch_sinkhorn_stage<T>(a, ws, n_slots, max_rows, extra, 0, (hipStream_t)stream)) return rc;
        }
        return (int)ASPIRE_OK;
    });
    if (rc_run) return rc_run;
    if (rank.k > 0) {
        // the rank kernels follow the scores on the same stream (their scratch sits behind the OT workspace proper,
        // which aspire_ot_workspace_bytes keeps a multiple of 16 bytes)
        const size_t need = aspire_topk_workspace_bytes(q->n, c->n, rank.k);
        void* tws = need ? (char*)workspace + ot_bytes : nullptr;
        return topk_run(scores, q->n, c->n, rank.k, rank.idx_base, rank.top_scores, rank.top_idx, rank.keys, tws, need, stream);
    }
    return ASPIRE_OK;
}
}  // namespace

extern "C" int aspire_ot_sinkhorn_f32(const aspire_repset* q, const aspire_repset* c, int64_t D, int pairing,
                                      const aspire_ot_params* prm, const float* diameter, int64_t diam_group, int want,
                                      float* scores, float* out_qdistr, float* out_cdistr, float* out_pairsims,
                                      float* out_plan, void* workspace, size_t workspace_bytes, void* stream) {
    return ot_run(q, c, D, pairing, prm, diameter, diam_group, want, scores, out_qdistr, out_cdistr, out_pairsims, out_plan,
                  workspace, workspace_bytes, stream, RankReq{0, 0, nullptr, nullptr, nullptr});
}

// Diagnostics: the cost stage of aspire_ot_sinkhorn_f32 alone (bench.py times the HBM-bound kernel of a pass this way).
extern "C" int aspire_debug_ot_cost_stage_f32(const aspire_repset* q, const aspire_repset* c, int64_t D, int pairing,
                                              const aspire_ot_params* prm, float* scores, void* workspace,
                                              size_t workspace_bytes, void* stream) {
    return ot_run(q, c, D, pairing, prm, nullptr, 0, ASPIRE_OT_DISTANCE, scores, nullptr, nullptr, nullptr, nullptr, workspace,
                  workspace_bytes, stream, RankReq{0, 0, nullptr, nullptr, nullptr}, true);
}

extern "C" size_t aspire_ot_rank_workspace_bytes(const aspire_repset* q, const aspire_repset* c, int64_t k) {
    if (!q || !c || q->n <= 0 || c->n <= 0) return 0;
    return aspire_ot_workspace_bytes(q, c, ASPIRE_PAIR_CROSS) + aspire_topk_workspace_bytes(q->n, c->n, k);
}

extern "C" int aspire_ot_rank_f32(const aspire_repset* q, const aspire_repset* c, int64_t D, const aspire_ot_params* prm,
                                  const float* diameter, int64_t diam_group, int want, float* scores, int64_t k,
                                  int64_t idx_base, float* top_scores, int64_t* top_idx, uint64_t* keys, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    ASPIRE_REQUIRE(k > 0, ASPIRE_ERR_INVALID_ARG, "k must be positive");
    ASPIRE_REQUIRE((top_scores && top_idx) || keys, ASPIRE_ERR_INVALID_ARG, "need (top_scores, top_idx) or keys");
    ASPIRE_REQUIRE(q && c, ASPIRE_ERR_INVALID_ARG, "null repset");
    const size_t tneed = aspire_topk_workspace_bytes(q->n, c->n, k);
    ASPIRE_REQUIRE(workspace_bytes >= tneed, ASPIRE_ERR_INVALID_ARG, "workspace too small for the rank scratch");
    // the OT part is what is left, rounded down to 16 bytes so that the 64-bit rank scratch behind it stays aligned
    return ot_run(q, c, D, ASPIRE_PAIR_CROSS, prm, diameter, diam_group, want, scores, nullptr, nullptr, nullptr, nullptr, workspace,
                  (workspace_bytes - tneed) & ~(size_t)15, stream, RankReq{k, idx_base, top_scores, top_idx, keys});
}

// ---------------------------------------------------------------------------------------------
// Batched jobs: J independent (query, pool) re-ranks in one call
// ---------------------------------------------------------------------------------------------
namespace aspire {
namespace {

// One workgroup per job j: the per-coordinate box of query j (the cost kernel adds each candidate's rows to it), the
// job's first group of four (groups never straddle jobs, so a wave of the cost kernel serves ONE query), and the
// candidate -> job / group -> job tables the kernels index.
__global__ void __launch_bounds__(192) batch_prep_kernel(RepSet q, RepSet c, const int32_t* __restrict__ job_off, int J,
                                                         float* __restrict__ qbox, int32_t* __restrict__ cand_job,
                                                         int32_t* __restrict__ grp_off, int32_t* __restrict__ grp_job,
                                                         int32_t* __restrict__ grp_rec) {
    // grid = (J, parts + 1): block (j, parts) forms the query's box and nothing else -- its chain of dependent loads (length,
    // start -> rows -> store) runs beside the table blocks' chain instead of in front of it; every other part of a job derives
    // the job's first group itself (a block-wide sum over the earlier jobs' group counts) and then takes its share of the
    // job's candidates / groups.  (One block per job made 20 blocks walk 250 groups each with dependent gathers: 20 us for
    // a 20 x 1000 batch.)
    __shared__ int part[3];
    const int j = blockIdx.x, tid = threadIdx.x;
    if (blockIdx.y == gridDim.y - 1) {
        const int n = q.len[j];
        const float* doc = q.rows + (size_t)q.start[j] * kD + tid * 4;
        float4 mn, mx;
        doc_box_chunk(doc, n, mn, mx);
        *reinterpret_cast<float4*>(qbox + (size_t)j * 2 * kD + tid * 4) = mn;
        *reinterpret_cast<float4*>(qbox + (size_t)j * 2 * kD + kD + tid * 4) = mx;
        return;
    }
    const int sub = blockIdx.y * 192 + tid, nsub = (gridDim.y - 1) * 192;
    int g = 0;
    for (int i = tid; i < j; i += 192) g += (job_off[i + 1] - job_off[i] + 3) >> 2;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) g += __shfl_xor(g, m);
    if ((tid & 63) == 0) part[tid >> 6] = g;
    __syncthreads();
    const int g0 = part[0] + part[1] + part[2];
    const int c0 = job_off[j], c1 = job_off[j + 1], ng = (c1 - c0 + 3) >> 2;
    if (blockIdx.y == 0 && tid == 0) {
        grp_off[j] = g0;
        if (j == J - 1) grp_off[J] = g0 + ng;
    }
    for (int cc = c0 + sub; cc < c1; cc += nsub) cand_job[cc] = j;
    for (int k = sub; k < ng; k += nsub) grp_job[g0 + k] = j;
    // the per-group records of the fused kernel (see ScoreArgs::grp_rec): thread = (group, field)
    const int q_len = q.len[j], q_start = q.start[j];
    for (int e = sub; e < ng * 16; e += nsub) {
        const int k = e >> 4, f = e & 15;
        const int first = c0 + 4 * k;
        const int cand = min(first + (f & 3), c1 - 1);
        int v;
        if (f == 0) v = j;
        else if (f == 1) v = q_len;
        else if (f == 2) v = q_start;
        else if (f == 3) v = min(4, c1 - first);
        else if (f < 8) v = cand;
        else if (f < 12) v = c.len[cand];
        else v = c.start[cand];
        grp_rec[(size_t)(g0 + k) * 16 + f] = v;
    }
}

// Items of the fused kernel's CHUNK form (fused.hip) for batched jobs whose candidates reach 9 .. 32 rows: an item = four 8-row
// chunk slots holding candidates of ONE job with [4], [3, 1], [2, 2], [2, 1, 1] or [1, 1, 1, 1] chunks (a 2-chunk candidate on
// slots 0, 1 or 2, 3; a 3-chunk one on 0 .. 2 with a 1-chunk candidate beside it: on the config-4 shape 3300 -> 2950 items, so
// that no SIMD of the scoring launch holds two waves of two items each).  Block (j, part) counts its slice of job j's candidates
// by chunk count (LDS counters), reserves its items with ONE atomicAdd on the launch's item counter (items need not be
// contiguous per job: a score is stored by candidate index), gives every candidate its place by its rank within its class, and
// writes the 64-byte item records: [0] query, [1] its len, [2] its first row, [3] widest exchange across lane groups the item
// needs (1, 2, 4), [4..7] the slots' candidates, [8..11] per slot: len | first slot of the candidate << 8 | its slots << 12 |
// real << 16, [12..15] the slots' first rows.  Slots that stay empty repeat the item's first candidate as a one-chunk
// candidate (scored, never stored).  Block (j, last) forms the query's box, as in batch_prep_kernel.
constexpr int kChunkPrepPart = 384;      // candidates per classification block
// job_off == nullptr: ONE query against the pool [0, c.n) (the single-pool entry points); cand_job may be null then.
// region_cap > 0 (at most 64 slices): no counter -- slice s = j * parts + part leaves its item count in counter[s] and its records in
// records [s * region_cap, ..) (ScoreArgs::chunk_regions).
__global__ void __launch_bounds__(192) chunk_prep_kernel(RepSet q, RepSet c, const int32_t* __restrict__ job_off, float* __restrict__ qbox,
                                                         int32_t* __restrict__ cand_job, int32_t* __restrict__ counter,
                                                         int32_t* __restrict__ grp_rec, int region_cap) {
    __shared__ int cnt[4], pos[4], base_s;
    const int slice = blockIdx.x * (gridDim.y - 1) + blockIdx.y;
    const int j = blockIdx.x, tid = threadIdx.x;
    if (blockIdx.y == gridDim.y - 1) {
        const int n = q.len[j];
        const float* doc = q.rows + (size_t)q.start[j] * kD + tid * 4;
        float4 mn, mx;
        doc_box_chunk(doc, n, mn, mx);
        *reinterpret_cast<float4*>(qbox + (size_t)j * 2 * kD + tid * 4) = mn;
        *reinterpret_cast<float4*>(qbox + (size_t)j * 2 * kD + kD + tid * 4) = mx;
        return;
    }
    const int c0 = (job_off ? job_off[j] : 0) + blockIdx.y * kChunkPrepPart, c1 = min(job_off ? job_off[j + 1] : (int)c.n, c0 + kChunkPrepPart);
    if (c0 >= c1) {
        if (region_cap > 0 && tid == 0) counter[slice] = 0;
        return;
    }
    if (tid < 4) cnt[tid] = pos[tid] = 0;
    __syncthreads();
    int len[2], start[2], nch[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int cc = c0 + tid + 192 * r;
        len[r] = cc < c1 ? c.len[cc] : 0;
        start[r] = cc < c1 ? c.start[cc] : 0;
        nch[r] = min(4, max(1, (len[r] + 7) >> 3));          // chunks (a longer document is poisoned by the kernel)
        if (cc < c1) {
            atomicAdd(&cnt[nch[r] - 1], 1);
            if (cand_job) cand_job[cc] = j;
        }
    }
    __syncthreads();
    // the block's items, in this order: [4] x n4, [3, 1] x n3, [2, 2] x n2 / 2, one [2, 1, 1] if n2 is odd, [1, 1, 1, 1] for the
    // singles the [3, 1] and [2, 1, 1] items have left
    const int n1 = cnt[0], n2 = cnt[1], n3 = cnt[2], n4 = cnt[3];
    const int s3 = min(n1, n3);                               // singles beside 3-chunk candidates
    const int odd2 = n2 & 1, s2 = odd2 ? min(n1 - s3, 2) : 0; // singles beside the odd 2-chunk candidate
    const int n1r = n1 - s3 - s2, items1 = (n1r + 3) >> 2;
    if (tid == 0) {
        const int items = n4 + n3 + (n2 >> 1) + odd2 + items1;
        if (region_cap > 0) {
            counter[slice] = items;
            base_s = slice * region_cap;
        } else {
            base_s = atomicAdd(counter, items);
        }
    }
    __syncthreads();
    const int b4 = base_s, b3 = b4 + n4, b2 = b3 + n3, bo = b2 + (n2 >> 1), b1 = bo + odd2;
    const int q_len = q.len[j], q_start = q.start[j];
    auto put = [&](int item, int slot, int cc, int ln, int st, int g0, int gsz, int real) {
        int32_t* rec = grp_rec + (size_t)item * 16;
        rec[4 + slot] = cc;
        rec[8 + slot] = ln | (g0 << 8) | (gsz << 12) | (real << 16);
        rec[12 + slot] = st;
    };
    auto head = [&](int item, int w) {
        int32_t* rec = grp_rec + (size_t)item * 16;
        rec[0] = j;
        rec[1] = q_len;
        rec[2] = q_start;
        rec[3] = w;
    };
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int cc = c0 + tid + 192 * r;
        if (cc >= c1) continue;
        const int k = nch[r], ps = atomicAdd(&pos[k - 1], 1), ln = len[r], st = start[r];
        if (k == 4) {
            for (int t = 0; t < 4; ++t) put(b4 + ps, t, cc, ln, st, 0, 4, 1);
            head(b4 + ps, 4);
        } else if (k == 3) {
            for (int t = 0; t < 3; ++t) put(b3 + ps, t, cc, ln, st, 0, 3, 1);
            if (ps >= s3) put(b3 + ps, 3, cc, ln, st, 3, 1, 0);             // no single left for this item
            head(b3 + ps, 4);
        } else if (k == 2) {
            const bool last_odd = odd2 && ps == n2 - 1;
            const int item = last_odd ? bo : b2 + (ps >> 1), s0 = last_odd ? 0 : 2 * (ps & 1);
            put(item, s0, cc, ln, st, s0, 2, 1);
            put(item, s0 + 1, cc, ln, st, s0, 2, 1);
            if (s0 == 0) head(item, 2);
            if (last_odd)
                for (int t = 2 + s2; t < 4; ++t) put(item, t, cc, ln, st, t, 1, 0);
        } else if (ps < s3) {
            put(b3 + ps, 3, cc, ln, st, 3, 1, 1);
        } else if (ps < s3 + s2) {
            put(bo, 2 + (ps - s3), cc, ln, st, 2 + (ps - s3), 1, 1);
        } else {
            const int rr = ps - s3 - s2, item = b1 + (rr >> 2), slot = rr & 3;
            put(item, slot, cc, ln, st, slot, 1, 1);
            if (slot == 0) {
                head(item, 1);
                for (int t = min(4, n1r - (rr & ~3)); t < 4; ++t) put(item, t, cc, ln, st, t, 1, 0);
            }
        }
    }
}
// Items of the 16-row streaming kernel's REC form (tile16.hip) for batched jobs whose queries AND candidates can have 9 .. 32 rows:
// an item = a 16-row half of the query against two candidate slots of 16 rows -- two candidates of <= 16 rows, or the two halves
// of one candidate of 17 .. 32.  Same scheme as chunk_prep_kernel (counts in LDS, one atomicAdd on the launch's item counter per
// block, a candidate's place by its rank within its class); a query of more than 16 rows gets every item twice, once per half.
// Record: [0] query, [1] its len, [2] its first row, [3] query half | wide << 8, [4,5] the slots' candidates, [6,7] their lens,
// [8,9] their first rows, [10,11] first row of the slot's half (0 / 16), [12,13] real.
__global__ void __launch_bounds__(192) chunk16_prep_kernel(RepSet q, RepSet c, const int32_t* __restrict__ job_off, float* __restrict__ qbox,
                                                           int32_t* __restrict__ cand_job, int32_t* __restrict__ counter,
                                                           int32_t* __restrict__ grp_rec) {
    __shared__ int cnt[2], pos[2], base_s;
    const int j = blockIdx.x, tid = threadIdx.x;
    if (blockIdx.y == gridDim.y - 1) {
        const int n = q.len[j];
        const float* doc = q.rows + (size_t)q.start[j] * kD + tid * 4;
        float4 mn, mx;
        doc_box_chunk(doc, n, mn, mx);
        *reinterpret_cast<float4*>(qbox + (size_t)j * 2 * kD + tid * 4) = mn;
        *reinterpret_cast<float4*>(qbox + (size_t)j * 2 * kD + kD + tid * 4) = mx;
        return;
    }
    const int c0 = job_off[j] + blockIdx.y * kChunkPrepPart, c1 = min(job_off[j + 1], c0 + kChunkPrepPart);
    if (c0 >= c1) return;
    if (tid < 2) cnt[tid] = pos[tid] = 0;
    __syncthreads();
    int len[2], start[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int cc = c0 + tid + 192 * r;
        len[r] = cc < c1 ? c.len[cc] : 0;
        start[r] = cc < c1 ? c.start[cc] : 0;
        if (cc < c1) {
            atomicAdd(&cnt[len[r] > 16 ? 1 : 0], 1);
            cand_job[cc] = j;
        }
    }
    __syncthreads();
    const int q_len = q.len[j], q_start = q.start[j], nqh = q_len > 16 ? 2 : 1;
    const int n_narrow = cnt[0], n_wide = cnt[1], per_half = ((n_narrow + 1) >> 1) + n_wide;
    if (tid == 0) base_s = atomicAdd(counter, nqh * per_half);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int cc = c0 + tid + 192 * r;
        if (cc >= c1) continue;
        const bool wide = len[r] > 16;
        const int ps = atomicAdd(&pos[wide ? 1 : 0], 1);
        const int local = wide ? ((n_narrow + 1) >> 1) + ps : ps >> 1, slot = wide ? 0 : ps & 1;
        const bool alone = !wide && slot == 0 && ps == n_narrow - 1;      // an odd narrow candidate: its item's second slot repeats it
        for (int qh = 0; qh < nqh; ++qh) {
            int32_t* rec = grp_rec + (size_t)(base_s + qh * per_half + local) * 16;
            if (slot == 0) {
                rec[0] = j;
                rec[1] = q_len;
                rec[2] = q_start;
                rec[3] = qh | (wide ? 256 : 0);
            }
            for (int t = slot; t < (wide || alone ? 2 : slot + 1); ++t) {
                rec[4 + t] = cc;
                rec[6 + t] = len[r];
                rec[8 + t] = start[r];
                rec[10 + t] = wide ? 16 * t : 0;
                rec[12 + t] = (wide || t == slot) ? 1 : 0;
            }
        }
    }
}
// parts (classification blocks) per job, and the bound on the items the launch can make
int64_t chunk_parts(int64_t max_job) { return max_job > 0 ? (max_job + kChunkPrepPart - 1) / kChunkPrepPart : 1; }
inline int chunk_regions_of(int64_t J, int64_t max_job) {
    const int64_t n = J * chunk_parts(max_job);
    return n <= 64 ? (int)n : 0;
}
int64_t chunk_items_bound(int64_t J, int64_t C, int64_t max_job) {
    const int64_t by_count = C + 3 * J * chunk_parts(max_job);
    const int64_t by_region = (int64_t)chunk_regions_of(J, max_job) * chunk_region_cap_of(max_job);      // (regions mode: every slice its own region)
    return by_count > by_region ? by_count : by_region;
}

struct BatchLayout {
    size_t slots, qbox, cand_job, grp_job, grp_off, grp_rec, gate, topk, total;
};
BatchLayout batch_layout(int64_t J, int64_t C, int max_rows, int64_t max_job, int64_t k) {
    BatchLayout L{};
    size_t o = 0;
    L.slots = o; o = align16(o + slot_bytes(max_rows) * (size_t)C);
    L.qbox = o; o = align16(o + (size_t)J * 2 * kD * sizeof(float));
    L.cand_job = o; o = align16(o + (size_t)C * sizeof(int32_t));
    L.grp_job = o; o = align16(o + (size_t)(C / 4 + J + 1) * sizeof(int32_t));
    L.grp_off = o; o = align16(o + (size_t)(J + 1 > 64 ? J + 1 : 64) * sizeof(int32_t));
    // (documents of more than 8 rows: room for the CHUNK form's items, up to one per candidate)
    const size_t n_rec = max_rows > 16 ? 2 * (size_t)chunk_items_bound(J, C, max_job) + 1       // (16-row items, per query half)
                         : max_rows > 8 ? (size_t)chunk_items_bound(J, C, max_job) + 1 : (size_t)(C / 4 + J + 1);
    L.grp_rec = o; o = align16(o + n_rec * 16 * sizeof(int32_t));
    L.gate = o; o = align16(o + 16);
    L.topk = o; o = align16(o + aspire_topk_workspace_bytes(J, max_job, k));
    L.total = o;
    return L;
}
}  // namespace
}  // namespace aspire

extern "C" size_t aspire_ot_rank_batch_workspace_bytes(const aspire_repset* q, const aspire_repset* c, int64_t max_job, int64_t k) {
    if (!q || !c || q->n <= 0 || c->n <= 0) return 0;
    const int mr = max_rows_of(q, c);
    return batch_layout(q->n, c->n, mr < 8 * kMaxT ? mr : 8 * kMaxT, max_job, k).total;
}

namespace {
constexpr int kStagePrep = 1, kStageCost = 2, kStageSolve = 4, kStageRank = 8, kStageAll = 15;
int ot_rank_batch(const aspire_repset* q, const aspire_repset* c, int64_t D, const int32_t* job_off, int64_t max_job,
                  const aspire_ot_params* prm, int want, float* scores, int64_t k, const int32_t* job_base, float* top_scores,
                  int64_t* top_idx, uint64_t* keys, void* workspace, size_t workspace_bytes, void* stream, int stages) {
    if (int rc = check_repsets(q, c, D, ASPIRE_PAIR_CROSS)) return rc;
    if (int rc = check_ot_params(prm, want)) return rc;
    const int64_t J = q->n, C = c->n;
    ASPIRE_REQUIRE(q->ext == 0 && c->ext == 0, ASPIRE_ERR_INVALID_ARG, "batched jobs take CSR rep sets (ext == 0)");
    ASPIRE_REQUIRE(k >= 0 && (k == 0 || (top_scores && top_idx) || keys), ASPIRE_ERR_INVALID_ARG,
                   "k > 0 needs (top_scores, top_idx) or keys");
    if (J == 0) return ASPIRE_OK;
    ASPIRE_REQUIRE(job_off && max_job >= 0 && max_job <= C, ASPIRE_ERR_INVALID_ARG, "need job_off and 0 <= max_job <= C");
    ASPIRE_REQUIRE(J < ((int64_t)1 << 30) && C < ((int64_t)1 << 31) - 8, ASPIRE_ERR_UNSUPPORTED, "batch too large for 32-bit offsets");
    hipStream_t s0 = (hipStream_t)stream;
    if (C == 0) {
        // every pool is empty: the lists are all padding
        const float* unread = reinterpret_cast<const float*>(job_off);     // every segment is empty: never dereferenced
        if (k > 0) return topk_run(unread, J, 0, k, 0, top_scores, top_idx, keys, nullptr, 0, stream, job_off, job_base);
        return ASPIRE_OK;
    }
    ASPIRE_REQUIRE(scores, ASPIRE_ERR_INVALID_ARG, "null scores");
    // Documents beyond the tile kernels' 32 rows (AspireNER's appended entity rows, models.py:224-233), as in ot_run: the tile
    // kernels run with their documents' bound clamped to 32 rows (a pair that holds a longer document gets NaN there) and the
    // long-document kernel (generic.hip) then rewrites exactly those pairs, in front of the rank.
    const int max_rows_all = max_rows_of(q, c);
    ASPIRE_REQUIRE(max_rows_all <= generic_max_rows(), ASPIRE_ERR_UNSUPPORTED,
                   "documents with more than %d sentence rows are not supported (got %d)", generic_max_rows(), max_rows_all);
    const int max_rows = max_rows_all < 8 * kMaxT ? max_rows_all : 8 * kMaxT;
    const BatchLayout L = batch_layout(J, C, max_rows, max_job, k);
    ASPIRE_REQUIRE(workspace && workspace_bytes >= L.total, ASPIRE_ERR_INVALID_ARG,
                   "workspace too small: %zu bytes given, aspire_ot_rank_batch_workspace_bytes says %zu", workspace_bytes, L.total);
    ASPIRE_REQUIRE(((uintptr_t)workspace & 15) == 0, ASPIRE_ERR_INVALID_ARG, "workspace must be 16-byte aligned");
    char* wsb = (char*)workspace;
    float* qbox = (float*)(wsb + L.qbox);
    int32_t* cand_job = (int32_t*)(wsb + L.cand_job);
    int32_t* grp_job = (int32_t*)(wsb + L.grp_job);
    int32_t* grp_off = (int32_t*)(wsb + L.grp_off);
    int32_t* grp_rec = (int32_t*)(wsb + L.grp_rec);
    ScoreArgs a{};
    fill_ot_args(a, q, c, kPairMapped, prm, nullptr, 0, want, scores);
    a.q_per_block = 1;
    a.cand0 = 0;
    a.cand1 = C;
    a.qmap = cand_job;
    a.job_off = job_off;
    a.grp_off = grp_off;
    a.grp_job = grp_job;
    a.grp_rec = grp_rec;
    a.job0 = 0;
    a.job1 = (int32_t)J;
    a.max_job_groups = (int32_t)((max_job + 3) / 4);
    // Forms.  Small batches are latency bound and take the kernels the single-pool entry points use.  Once the batch fills
    // the chip (documents of <= 8 rows): the fused kernel -- four candidates of a job per wave, costs and solves in one
    // launch (fused.hip) -- or, pinned for A/B tests, the same cost kernel + the block Sinkhorn kernel as two launches.
    // (Chunks of jobs on two streams, Sinkhorn of chunk i beside the cost kernel of chunk i + 1, were measured and dropped:
    // 20 x 1000: 169 us on one stream, 213 / 266 / 403 us in 2 / 4 / 8 chunks -- cross-stream waits cost more than they hide.)
    if (prm->flags & ASPIRE_OT_FLAG_ONE_FORM) {
        // every pair through the long-form kernel (one workgroup per pair; it finds a candidate's job by searching job_off)
        a.qmap = nullptr;
        if (stages & kStageSolve)
            if (int rc = launch_pair_generic(a, 0, 0, q->max_len, c->max_len, s0)) return rc;
        const size_t need = aspire_topk_workspace_bytes(J, max_job, k);
        if (k > 0 && (stages & kStageRank))
            return topk_run(scores, J, max_job, k, 0, keys ? nullptr : top_scores, keys ? nullptr : top_idx, keys,
                            need ? wsb + L.topk : nullptr, need, stream, job_off, job_base);
        return ASPIRE_OK;
    }
    const int64_t groups_bound = J * ((max_job + 3) / 4);
    const int form_t = tuning().ot_form;
    const size_t topk_need = aspire_topk_workspace_bytes(J, max_job, k);
    const bool big = max_rows <= 8 && groups_bound >= kStreamMinGroupsBatch;
    const bool fused = max_rows <= 8 && (form_t == 3 || (form_t == 0 && big));
    a.tile_form = max_rows <= 8 && (form_t == 2 || fused);
    // Documents of up to 16 rows in a batch that fills the chip: usually pools of mostly short abstracts with a few longer
    // ones.  The fused kernel is queued in front of the 16-row streaming kernel + block Sinkhorn, and a census of the long
    // pairs, taken on the device, decides how they work (score_types.h: ScoreArgs::gate): few long pairs -- the fused kernel
    // scores the short ones, the 16-row kernels only the long ones; many -- the fused kernel returns at once.  No host round
    // trip, and ONE long document no longer moves 20 000 pairs onto kernels 2.5 x slower.
    const bool hybrid = max_rows > 8 && max_rows <= 16 && form_t == 0 && groups_bound >= 2048 && C >= 6000 && stages == kStageAll &&
                        prm->scaling >= 0.25 && want != ASPIRE_OT_PLAN_SIM && !tuning().fused_nosolve && !tuning().fused_valu &&
                        sinkhorn_form_honours_gate();
    // Short queries (facet-selected rows, models.py:127-163) against abstracts of up to 32 rows -- config 4's shape: the fused
    // kernel's CHUNK form, costs and solves of every pair in one launch behind a launch that sorts the candidates into items.
    const bool chunked = q->max_len <= 8 && max_rows > 8 && max_rows_all <= 8 * kMaxT && (form_t == 4 || (form_t == 0 && C >= kChunkMinCands)) &&
                         stages == kStageAll && prm->scaling >= 0.25 && !tuning().fused_valu;
    if (chunked) {
        a.chunk_regions = chunk_regions_of(J, max_job);
        a.chunk_region_cap = chunk_region_cap_of(max_job);
        if (a.chunk_regions == 0) ASPIRE_HIP_OK(hipMemsetAsync(grp_off, 0, sizeof(int32_t), s0));
        hipLaunchKernelGGL(chunk_prep_kernel, dim3((unsigned)J, (unsigned)chunk_parts(max_job) + 1), dim3(192), 0, s0, a.q, a.c, job_off, qbox,
                           cand_job, grp_off, grp_rec, a.chunk_regions > 0 ? a.chunk_region_cap : 0);
        ASPIRE_LAUNCH_OK();
        if (int rc = launch_pair_fused_chunk(a, chunk_items_bound(J, C, max_job), qbox, s0)) return rc;
        if (k > 0)
            return topk_run(scores, J, max_job, k, 0, keys ? nullptr : top_scores, keys ? nullptr : top_idx, keys,
                            topk_need ? wsb + L.topk : nullptr, topk_need, stream, job_off, job_base);
        return ASPIRE_OK;
    }
    // Whole abstracts on both sides, documents of 17 .. 32 rows among them (un-faceted queries: pp_settings.py:2-3): the 16-row
    // streaming kernel on record items (a query half against two candidate halves) + the block Sinkhorn kernel on 24- / 32-row
    // workspace slots.  (Before: the VALU tile loop, one workgroup per candidate.)
    const bool rec16 = !chunked && max_rows > 16 && max_rows_all <= 8 * kMaxT && (form_t == 4 || (form_t == 0 && C >= kChunkMinCands)) &&
                       stages == kStageAll && tuning().cost_path != 2;
    if (rec16) {
        ASPIRE_HIP_OK(hipMemsetAsync(grp_off, 0, sizeof(int32_t), s0));
        hipLaunchKernelGGL(chunk16_prep_kernel, dim3((unsigned)J, (unsigned)chunk_parts(max_job) + 1), dim3(192), 0, s0, a.q, a.c, job_off, qbox,
                           cand_job, grp_off, grp_rec);
        ASPIRE_LAUNCH_OK();
        const int rc_run = dispatch_T(max_rows, [&](auto tc) -> int {
            constexpr int T = decltype(tc)::value;
            if constexpr (T >= 3) {
                PairWs<T> ws;
                ws.cost = (float*)(wsb + L.slots);
                ws.neg = ws.cost + C * PairWs<T>::kEntries;
                ws.diam2 = ws.neg + C * PairWs<T>::kEntries;
                if (int rc = launch_pair_tile16_rec(a, T, ws.cost, ws.neg, ws.diam2, 2 * chunk_items_bound(J, C, max_job), qbox, s0)) return rc;
                return launch_sinkhorn_stage<T>(a, ws, C, max_rows, false, 3, s0);
            }
            return (int)ASPIRE_ERR_UNSUPPORTED;
        });
        if (rc_run) return rc_run;
        if (k > 0)
            return topk_run(scores, J, max_job, k, 0, keys ? nullptr : top_scores, keys ? nullptr : top_idx, keys,
                            topk_need ? wsb + L.topk : nullptr, topk_need, stream, job_off, job_base);
        return ASPIRE_OK;
    }
    // batches of <= 64 jobs on the fused kernel need no tables launch: the kernel's waves derive them (fused.hip, SELF)
    const bool self = fused && fused_self_ok(J, prm);
    if ((stages & kStagePrep) && !self) {
        // parts per job: enough blocks that a job's groups take a couple of trips each
        const int64_t work = ((max_job + 3) / 4) * 16;
        int64_t parts = (work + 2 * 192 - 1) / (2 * 192);
        parts = parts < 1 ? 1 : parts > 64 ? 64 : parts;
        while (parts > 1 && J * parts > 4096) parts /= 2;
        hipLaunchKernelGGL(batch_prep_kernel, dim3((unsigned)J, (unsigned)parts + 1), dim3(192), 0, s0, a.q, a.c, job_off, (int)J, qbox, cand_job,
                           grp_off, grp_job, grp_rec);
        ASPIRE_LAUNCH_OK();
    }
    if (hybrid) {
        int32_t* gate = (int32_t*)(wsb + L.gate);
        ASPIRE_HIP_OK(hipMemsetAsync(gate, 0, sizeof(int32_t), s0));
        hipLaunchKernelGGL(long_pair_census_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s0, a, gate);
        ASPIRE_LAUNCH_OK();
        a.gate = gate;
        a.gate_limit = (int32_t)(C / 24);     // up to ~4 % long pairs (measured crossover at 20 x 1000: 5 %): fused kernel + the 16-row kernels on the long pairs only
        if (int rc = launch_pair_fused(a, groups_bound, qbox, s0)) return rc;
    }
    if (fused) {
        if (stages & (kStageCost | kStageSolve))
        {
#ifdef ASPIRE_EXPERIMENT_SPLIT
            if (self && stages == kStageAll && split_path_ok(groups_bound, prm)) {
                if (int rc = launch_pair_split(a, s0)) return rc;
            } else
#endif
            if (int rc = launch_pair_fused(a, groups_bound, self ? nullptr : qbox, s0)) return rc;
        }
    } else {
        const int rc_run = dispatch_T(max_rows, [&](auto tc) -> int {
            constexpr int T = decltype(tc)::value;
            if constexpr (T == 1) {
                // a small batch of short documents: the single-pool calls' one-launch form (pair_one_kernel: one wave per pair) -- the
                // same kernel whether a pair is scored in a batch or in a call of its own: the same bits
                if (stages == kStageAll && !a.tile_form && C <= kOneMaxPairs &&
                    (form_t == 5 || (form_t == 0 && tuning().sinkhorn_form == 0 && tuning().cost_path == 0 && tuning().cost1_blocks == 0))) {
                    hipLaunchKernelGGL(pair_one_kernel, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, s0, a, C);
                    ASPIRE_LAUNCH_OK();
                    return (int)ASPIRE_OK;
                }
            }
            PairWs<T> ws;
            ws.cost = (float*)(wsb + L.slots);
            ws.neg = ws.cost + C * PairWs<T>::kEntries;
            ws.diam2 = ws.neg + C * PairWs<T>::kEntries;
            if (stages & kStageCost)
                if (int rc = launch_cost_stage<T>(a, q, c, ws, C, 1, false, qbox, nullptr, true, s0)) return rc;
            if (stages & kStageSolve)
                if (int rc = launch_sinkhorn_stage<T>(a, ws, C, max_rows, false, a.tile_form ? 3 : 0, s0)) return rc;
            return (int)ASPIRE_OK;
        });
        if (rc_run) return rc_run;
        if (max_rows_all > max_rows && (stages & kStageSolve))
            if (int rc = launch_pair_generic(a, 0, max_rows, q->max_len, c->max_len, s0)) return rc;
    }
    if (k > 0 && (stages & kStageRank))
        return topk_run(scores, J, max_job, k, 0, keys ? nullptr : top_scores, keys ? nullptr : top_idx, keys,
                        topk_need ? wsb + L.topk : nullptr, topk_need, stream, job_off, job_base);
    return ASPIRE_OK;
}
}  // namespace

extern "C" int aspire_ot_rank_batch_f32(const aspire_repset* q, const aspire_repset* c, int64_t D, const int32_t* job_off,
                                        int64_t max_job, const aspire_ot_params* prm, int want, float* scores, int64_t k,
                                        const int32_t* job_base, float* top_scores, int64_t* top_idx, uint64_t* keys,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    return ot_rank_batch(q, c, D, job_off, max_job, prm, want, scores, k, job_base, top_scores, top_idx, keys, workspace,
                         workspace_bytes, stream, kStageAll);
}

// ---- tsAspire over batched jobs ------------------------------------------------------------------------------------------
namespace {
// groups of four candidates from which aspire_l2max_rank_batch_f32 takes the streaming kernels (measured, tools/l2batchbench.py)
constexpr int64_t kL2StreamMinGroups = 384;
struct L2BatchLayout {
    size_t cand_job, grp_job, grp_off, grp_rec, qbox, gate, topk, total;
};
L2BatchLayout l2_batch_layout(int64_t J, int64_t C, int64_t max_job, int64_t k) {
    L2BatchLayout L{};
    size_t o = 0;
    L.cand_job = o; o = align16(o + (size_t)C * sizeof(int32_t));
    L.grp_job = o; o = align16(o + (size_t)(C / 4 + J + 1) * sizeof(int32_t));
    L.grp_off = o; o = align16(o + (size_t)(J + 1 > 64 ? J + 1 : 64) * sizeof(int32_t));
    L.grp_rec = o; o = align16(o + (2 * (size_t)chunk_items_bound(J, C, max_job) + 1) * 16 * sizeof(int32_t));     // (room for the CHUNK / REC forms' items)
    L.qbox = o; o = align16(o + (size_t)J * 2 * kD * sizeof(float));      // (written by the tables kernel, unused by max-sim)
    L.gate = o; o = align16(o + 16);
    L.topk = o; o = align16(o + aspire_topk_workspace_bytes(J, max_job, k));
    L.total = o;
    return L;
}
}  // namespace

extern "C" size_t aspire_l2max_rank_batch_workspace_bytes(const aspire_repset* q, const aspire_repset* c, int64_t max_job, int64_t k) {
    if (!q || !c || q->n <= 0 || c->n <= 0) return 0;
    return l2_batch_layout(q->n, c->n, max_job, k).total;
}

extern "C" int aspire_l2max_rank_batch_f32(const aspire_repset* q, const aspire_repset* c, int64_t D, const int32_t* job_off,
                                           int64_t max_job, int cdist_mode, float* scores, int64_t k, const int32_t* job_base,
                                           float* top_scores, int64_t* top_idx, uint64_t* keys, void* workspace,
                                           size_t workspace_bytes, void* stream) {
    if (int rc = check_repsets(q, c, D, ASPIRE_PAIR_CROSS)) return rc;
    const int64_t J = q->n, C = c->n;
    ASPIRE_REQUIRE(q->ext == 0 && c->ext == 0, ASPIRE_ERR_INVALID_ARG, "batched jobs take CSR rep sets (ext == 0)");
    ASPIRE_REQUIRE(k >= 0 && (k == 0 || (top_scores && top_idx) || keys), ASPIRE_ERR_INVALID_ARG,
                   "k > 0 needs (top_scores, top_idx) or keys");
    if (J == 0) return ASPIRE_OK;
    ASPIRE_REQUIRE(job_off && max_job >= 0 && max_job <= C, ASPIRE_ERR_INVALID_ARG, "need job_off and 0 <= max_job <= C");
    ASPIRE_REQUIRE(J < ((int64_t)1 << 30) && C < ((int64_t)1 << 31) - 8, ASPIRE_ERR_UNSUPPORTED, "batch too large for 32-bit offsets");
    hipStream_t s0 = (hipStream_t)stream;
    if (C == 0) {
        const float* unread = reinterpret_cast<const float*>(job_off);     // every segment is empty: never dereferenced
        if (k > 0) return topk_run(unread, J, 0, k, 0, top_scores, top_idx, keys, nullptr, 0, stream, job_off, job_base);
        return ASPIRE_OK;
    }
    ASPIRE_REQUIRE(scores, ASPIRE_ERR_INVALID_ARG, "null scores");
    const int max_rows = max_rows_of(q, c);
    ASPIRE_REQUIRE(max_rows <= generic_max_rows(), ASPIRE_ERR_UNSUPPORTED, "documents with more than %d sentence rows are not supported (got %d)",
                   generic_max_rows(), max_rows);
    const L2BatchLayout L = l2_batch_layout(J, C, max_job, k);
    ASPIRE_REQUIRE(workspace && workspace_bytes >= L.total, ASPIRE_ERR_INVALID_ARG,
                   "workspace too small: %zu bytes given, aspire_l2max_rank_batch_workspace_bytes says %zu", workspace_bytes, L.total);
    ASPIRE_REQUIRE(((uintptr_t)workspace & 15) == 0, ASPIRE_ERR_INVALID_ARG, "workspace must be 16-byte aligned");
    char* wsb = (char*)workspace;
    const bool one_form = (cdist_mode & ASPIRE_CDIST_ONE_FORM) != 0, center = (cdist_mode & ASPIRE_CDIST_CENTER) != 0;
    cdist_mode &= ~(ASPIRE_CDIST_ONE_FORM | ASPIRE_CDIST_CENTER);
    ScoreArgs a{};
    a.q = to_dev(q);
    a.c = to_dev(c);
    a.q_planes = q->planes;
    a.c_planes = c->planes;
    a.c_box = c->doc_box;
    a.pairing = kPairMapped;
    a.cdist_mode = cdist_mode;
    a.center = center;
    a.agg = ASPIRE_AGG_MAX;
    a.temp = 1.0;
    a.scores = scores;
    a.cand0 = 0;
    a.cand1 = C;
    a.qmap = (int32_t*)(wsb + L.cand_job);
    a.job_off = job_off;
    a.grp_off = (int32_t*)(wsb + L.grp_off);
    a.grp_job = (int32_t*)(wsb + L.grp_job);
    a.grp_rec = (int32_t*)(wsb + L.grp_rec);
    a.job0 = 0;
    a.job1 = (int32_t)J;
    a.max_job_groups = (int32_t)((max_job + 3) / 4);
    const int64_t groups_bound = J * ((max_job + 3) / 4);
    const int form_t = tuning().ot_form;
    const bool big = groups_bound >= 2048 && C >= 6000 && form_t != 1;
    // (small batches too: a wave walks an item's twelve stages in ~15 us, what a one-workgroup-per-pair launch takes anyway)
    // batches of <= 64 jobs of short documents: the streaming kernel's waves derive the tables themselves (fused.hip, SELF) --
    // one launch in front of the rank, at any size (2 x 20: 19 us either way)
    // short queries against abstracts of up to 32 rows (config 4's shape): the CHUNK items of ot_rank_batch, max epilogue
    if (q->max_len <= 8 && max_rows > 8 && max_rows <= 8 * kMaxT && (form_t == 4 || (form_t == 0 && C >= kChunkMinCands)) && !one_form) {
        a.chunk_regions = chunk_regions_of(J, max_job);
        a.chunk_region_cap = chunk_region_cap_of(max_job);
        if (a.chunk_regions == 0) ASPIRE_HIP_OK(hipMemsetAsync((int32_t*)(wsb + L.grp_off), 0, sizeof(int32_t), s0));
        hipLaunchKernelGGL(chunk_prep_kernel, dim3((unsigned)J, (unsigned)chunk_parts(max_job) + 1), dim3(192), 0, s0, a.q, a.c, job_off,
                           (float*)(wsb + L.qbox), (int32_t*)(wsb + L.cand_job), (int32_t*)(wsb + L.grp_off), (int32_t*)(wsb + L.grp_rec),
                           a.chunk_regions > 0 ? a.chunk_region_cap : 0);
        ASPIRE_LAUNCH_OK();
        if (int rc = launch_pair_fused_chunk_l2max(a, chunk_items_bound(J, C, max_job), s0)) return rc;
        const size_t need = aspire_topk_workspace_bytes(J, max_job, k);
        if (k > 0)
            return topk_run(scores, J, max_job, k, 0, keys ? nullptr : top_scores, keys ? nullptr : top_idx, keys,
                            need ? wsb + L.topk : nullptr, need, stream, job_off, job_base);
        return ASPIRE_OK;
    }
    // whole abstracts of up to 32 rows against queries of 9 .. 16: the 16-row streaming kernel on record items, max epilogue (a
    // query of more than 16 rows would need its two halves' maxima joined across items: the per-candidate kernel keeps those)
    if (q->max_len <= 16 && max_rows > 16 && max_rows <= 8 * kMaxT && (form_t == 4 || (form_t == 0 && C >= kChunkMinCands)) && !one_form) {
        ASPIRE_HIP_OK(hipMemsetAsync((int32_t*)(wsb + L.grp_off), 0, sizeof(int32_t), s0));
        hipLaunchKernelGGL(chunk16_prep_kernel, dim3((unsigned)J, (unsigned)chunk_parts(max_job) + 1), dim3(192), 0, s0, a.q, a.c, job_off,
                           (float*)(wsb + L.qbox), (int32_t*)(wsb + L.cand_job), (int32_t*)(wsb + L.grp_off), (int32_t*)(wsb + L.grp_rec));
        ASPIRE_LAUNCH_OK();
        if (int rc = launch_pair_tile16_rec_l2max(a, 2 * chunk_items_bound(J, C, max_job), s0)) return rc;
        const size_t need = aspire_topk_workspace_bytes(J, max_job, k);
        if (k > 0)
            return topk_run(scores, J, max_job, k, 0, keys ? nullptr : top_scores, keys ? nullptr : top_idx, keys,
                            need ? wsb + L.topk : nullptr, need, stream, job_off, job_base);
        return ASPIRE_OK;
    }
    const bool self = form_t != 1 && max_rows <= 8 && J <= 64 && !tuning().fused_noself && !one_form;
    const bool streaming = form_t != 1 && (self || big || form_t >= 2 || groups_bound >= kL2StreamMinGroups) && !one_form;
    if (!self) {
        const int64_t work = ((max_job + 3) / 4) * 16;
        int64_t parts = (work + 2 * 192 - 1) / (2 * 192);
        parts = parts < 1 ? 1 : parts > 64 ? 64 : parts;
        while (parts > 1 && J * parts > 4096) parts /= 2;
        hipLaunchKernelGGL(batch_prep_kernel, dim3((unsigned)J, (unsigned)parts + 1), dim3(192), 0, s0, a.q, a.c, job_off, (int)J,
                           (float*)(wsb + L.qbox), (int32_t*)(wsb + L.cand_job), (int32_t*)(wsb + L.grp_off), (int32_t*)(wsb + L.grp_job),
                           (int32_t*)(wsb + L.grp_rec));
        ASPIRE_LAUNCH_OK();
    }
    // Forms: the streaming kernels once the batch fills the chip (documents of <= 8 rows: fused.hip's max-sim form, four
    // candidates of a job per wave; 9 .. 16 rows: tile16.hip, two), else -- small batches, longer documents -- the
    // one-workgroup-per-pair kernel (generic.hip).
    if (streaming && max_rows <= 8) {
        if (int rc = launch_pair_fused_l2max(a, groups_bound, s0, self)) return rc;
    } else if (streaming && max_rows <= 16) {
        // mostly short documents with a few of 9 .. 16 rows: the hybrid of ot_rank_batch (ScoreArgs::gate)
        if (big && form_t == 0) {
            int32_t* gate = (int32_t*)(wsb + L.gate);
            ASPIRE_HIP_OK(hipMemsetAsync(gate, 0, sizeof(int32_t), s0));
            hipLaunchKernelGGL(long_pair_census_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s0, a, gate);
            ASPIRE_LAUNCH_OK();
            a.gate = gate;
            a.gate_limit = (int32_t)(C / 24);
            if (int rc = launch_pair_fused_l2max(a, groups_bound, s0)) return rc;
        }
        if (int rc = launch_pair_tile16_l2max(a, 2 * groups_bound, s0)) return rc;
    } else if (max_rows <= 8 * kMaxT && !one_form) {
        // one workgroup per candidate against its job's query (small batches, documents of 17 .. 32 rows)
        const int rc_tiles = dispatch_T(max_rows, [&](auto tc) -> int {
            constexpr int T = decltype(tc)::value;
            hipLaunchKernelGGL(l2max_kernel<T>, dim3((unsigned)C, 1, 1), dim3(kBlock), Lds<T>::kTotal * sizeof(float), s0, a);
            ASPIRE_LAUNCH_OK();
            return (int)ASPIRE_OK;
        });
        if (rc_tiles) return rc_tiles;
    } else {
        if (int rc = launch_pair_generic(a, 1, 0, q->max_len, c->max_len, s0)) return rc;
    }
    const size_t topk_need = aspire_topk_workspace_bytes(J, max_job, k);
    if (k > 0)
        return topk_run(scores, J, max_job, k, 0, keys ? nullptr : top_scores, keys ? nullptr : top_idx, keys,
                        topk_need ? wsb + L.topk : nullptr, topk_need, stream, job_off, job_base);
    return ASPIRE_OK;
}

// Diagnostics: chosen stages of aspire_ot_rank_batch_f32 on the caller's stream alone (1 tables + query boxes, 2 cost
// kernel, 4 Sinkhorn kernel, 8 rank) -- bench.py times each stage of a pass this way, after a full call has filled the
// workspace.
extern "C" int aspire_debug_ot_rank_batch_stages_f32(const aspire_repset* q, const aspire_repset* c, int64_t D,
                                                     const int32_t* job_off, int64_t max_job, const aspire_ot_params* prm,
                                                     int want, float* scores, int64_t k, float* top_scores, int64_t* top_idx,
                                                     void* workspace, size_t workspace_bytes, void* stream, int stages) {
    ASPIRE_REQUIRE(stages > 0 && stages <= kStageAll, ASPIRE_ERR_INVALID_ARG, "bad stage mask %d", stages);
    return ot_rank_batch(q, c, D, job_off, max_job, prm, want, scores, k, nullptr, top_scores, top_idx, nullptr, workspace,
                         workspace_bytes, stream, stages);
}

extern "C" int aspire_group_diameter_f32(const aspire_repset* q, const aspire_repset* c, int64_t D, int pairing,
                                         int64_t group, float* diameter, void* stream) {
    if (int rc = check_repsets(q, c, D, pairing)) return rc;
    ASPIRE_REQUIRE(group > 0 && diameter, ASPIRE_ERR_INVALID_ARG, "group must be positive, diameter non-null");
    if (q->n == 0 || c->n == 0) return ASPIRE_OK;
    ScoreArgs a{};
    a.q = to_dev(q);
    a.c = to_dev(c);
    a.q_planes = q->planes;
    a.c_planes = c->planes;
    a.c_box = c->doc_box;
    a.pairing = pairing;
    const int64_t ngroups = (c->n + group - 1) / group;
    const int64_t blocks = pairing == ASPIRE_PAIR_PAIRED ? ngroups : ngroups * q->n;     // (query, group) folded into grid.x
    ASPIRE_REQUIRE(blocks < ((int64_t)1 << 31), ASPIRE_ERR_UNSUPPORTED, "too many (query, group) boxes: %lld", (long long)blocks);
    hipLaunchKernelGGL(diameter_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, a, group, ngroups, diameter);
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}
