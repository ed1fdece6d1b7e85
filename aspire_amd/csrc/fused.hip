// otAspire throughput kernel: pairwise sentence costs AND the Sinkhorn solve of a pair in ONE launch, nothing handed over
// through HBM (A5-A8; reference arithmetic: src/learning/facetid_models/pair_distances.py:21-92 + geomloss 0.2.4's
// sinkhorn_tensorized, restated -- see score.hip).
//
// Why fused.  For few queries per candidate the cost stage is HBM bound (24.6 KB of candidate rows per pair, 98 K flop)
// and the Sinkhorn stage is VALU / transcendental bound (~80 epsilon steps on an 8 x 8 problem).  As two kernels they
// run one after the other (1 x 20 000: 110 + 50 us), and running them side by side on two streams costs more in
// cross-stream waits than it hides (measured: 20 jobs x 1000 candidates 169 us on one stream, 213 / 266 / 403 us in 2 / 4 /
// 8 chunks over two streams).  Here a wave that has finished the costs of its four candidates solves those four pairs at
// once from its own registers while the other resident waves keep the memory system busy: the solve costs issue slots the
// streaming phase leaves idle, and the 516 B / pair workspace round trip disappears.
//
// Work decomposition (gfx950, wave = 64 lanes), documents of <= 8 sentence rows, CSR inputs:
//   * item = four consecutive candidates of ONE query (groups never straddle two jobs of a batch); a wave owns an item and
//     walks the items with a static stride (SELF: a run of consecutive items).  (Claiming items dynamically -- one atomicAdd
//     on a shared counter per item -- was measured and dropped: 2048 waves finishing an item together queue 2048
//     same-address atomics, ~40 us per round; and it buys nothing here: the kernel is bandwidth bound, so a last partial
//     round of fewer waves simply runs each of them faster.)
//   * cost phase: 16 lanes per candidate stage its 8 rows 64 coordinates at a time (coalesced global_load_dwordx4 ->
//     ds_write_b128 into padded rows), the row norms and the bounding-box term (geomloss's diameter) fall out of the
//     registers before they go to LDS, the NEXT stage's loads then go out into the same registers; the dot products run on
//     the matrix pipe (v_mfma_f32_4x4x1: 16 blocks of 4 x 4 = the 64 entries of each of the wave's four pairs; conflict-free
//     ds_read_b128 operands) and are transposed through LDS into the solve's layout: lane (li, lj) of a candidate's 16
//     lanes owns the 2 x 2 entries (2 li + x, 2 lj + y).
//   * solve phase: the same 16 lanes x (2 x 2) layout IS a Sinkhorn layout: row sums are two quad_perm DPP adds, column sums
//     two row_ror DPP adds, both inside a DPP row of 16 lanes; four solves per wave, every pair on its own epsilon schedule
//     (the loop runs to the longest of the four, finished pairs idle with h = 0).  One exponential per entry and step,
//     K_ij = 2^((f_i + g_j - C_ij) log2e / eps), weights as plain factors, f_i -= h log2(sum_j b_j K_ij) (see
//     sinkhorn_block_kernel in score.hip for the derivation).  A sum that leaves fp32 range (extreme scaling, never at the
//     reference's hyper-parameters) poisons the pair's score with NaN; with such hyper-parameters the launcher follows up
//     with the long-form kernel (generic.hip: max-shifted log-sum-exps), which re-solves exactly the NaN pairs.
#include <math.h>

#include <mutex>

#include "common.h"
#include "score_device.h"
#include "score_types.h"
#include "fused_solve.h"
#include "tuning.h"

namespace aspire {
namespace {

#ifdef ASPIRE_PHASE_CLOCK
// debug build only (tools/chunkphases.py): per-wave time stamps (100 MHz wall clock) into the buffer set by
// aspire_debug_fused_buffer: [wave][8] = start, then per item (end of its streaming phase, its solve set up), ..., [6] the
// wave's last solve begins, [7] end
static __device__ long long* g_fdbg = nullptr;
#define F_STAMP(k)                                                                                                   \
    do {                                                                                                             \
        if (g_fdbg && lane == 0 && (k) < 8) g_fdbg[(size_t)(blockIdx.x * 4 + wave) * 8 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define F_STAMP(k) \
    do {           \
    } while (0)
#endif


// MFMA: the dot products on the matrix pipe (v_mfma_f32_4x4x1_16B_f32: 16 blocks of 4 x 4 = the 64 entries of each of the
// wave's four pairs, one coordinate per instruction, exact fp32 multiply-adds), which leaves the VALU issue slots to the
// staging side products and to the solve slices -- with two waves per SIMD the kernel is otherwise VALU-issue bound
// (20 x 1000 pairs: 146 -> 130 us; 100 x 1000: 661 -> 566 us).  !MFMA: the same sums as VALU FMAs (kept for A/B and parity
// tests).  SOLVE = false (diagnostics): the cost phase alone, diam^2 as the score.
// SELF (batches of <= 64 jobs): no tables, no query boxes from a launch in front of this one -- every wave derives an item's
// documents from job_off itself (a 64-lane scan, once) and forms the query's box per stage from the query rows it has just
// staged (+32 min/max and 8 LDS reads per stage, against ~7 us for the extra launch in a 115 us call).
// L2MAX (with SOLVE = false): tsAspire on the same streaming phase -- the score is the maximum of -cdist over the valid
// block (allpair_masked_dist_l2max, pair_distances.py:167-176); no boxes, no solve.
// QBOX (one query against a big pool): the in-wave query box of SELF without its tables -- every item has the same query, so
// the box is formed during a wave's first item and read from the LDS cache afterwards; no doc_box launch in front.
// CHUNK (batched jobs whose candidates reach 9 .. 32 rows, queries of <= 8: config 4's facet-selected queries against whole
// abstracts -- pp_settings.py:2-3, models.py:127-163): an item's four 16-lane groups hold four 8-row CHUNKS instead of four
// candidates -- four candidates of <= 8 rows, two of 9 .. 16 (two chunks each) or one of 17 .. 32 (four); chunk_prep_kernel
// (score.hip) sorts a job's candidates into such items.  The streaming phase is the same (a group stages rows row0 .. row0 + 7
// of its document; the candidate's per-coordinate box is joined across its groups), the solve's row sums and the marginals'
// normalisations cross the candidate's groups (xg_sum / xg_max).
// (-DASPIRE_FUSED_WAVES3: the experiment of NOTES.md -- the same kernel built for three workgroups per CU, 168 registers: the
// compiler spills the kernel-invariant values and the launch gets slower; profiles/r03_fused_3waves_* hold its counters)
#ifndef ASPIRE_FUSED_MFMA_UNROLL
#define ASPIRE_FUSED_MFMA_UNROLL 4
#endif
constexpr int kMfmaUnroll = ASPIRE_FUSED_MFMA_UNROLL;      // coordinates chunks per iteration of a stage's MFMA loop
#ifdef ASPIRE_FUSED_WAVES3
#define ASPIRE_FUSED_MIN_WAVES 3
#else
#define ASPIRE_FUSED_MIN_WAVES 2
#endif
template <bool MFMA, bool SOLVE = true, bool SELF = false, bool L2MAX = false, bool QBOX = false, bool CHUNK = false>
__global__ void __launch_bounds__(256, ASPIRE_FUSED_MIN_WAVES) pair_fused_kernel(ScoreArgs a, const float* __restrict__ qbox) {
    static_assert(!CHUNK || (MFMA && !SELF && !QBOX), "CHUNK: table-driven items only");
    constexpr bool INBOX = (SELF || QBOX) && !L2MAX;        // the query's box comes from the staged query rows (max-sim needs none)
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    if (a.gate != nullptr && !gate_few_long(a)) return;      // hybrid (score_types.h): mostly long pairs -- the 16-row kernels take them all
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* lds = lds_all + wave * kWaveLds;
    float* nscr = lds + kNormOfs;                           // [16][kNormLd]: norm partials, written once the stages are done
    const bool mapped = a.pairing == kPairMapped;           // batched jobs: items are the groups of four of jobs [job0, job1)
    const uint32_t nq = mapped ? 1u : (uint32_t)a.q.n;
    const uint32_t ncand = (uint32_t)(a.cand1 - a.cand0);
    // SELF: lane l holds job job0 + l's candidate range and the groups of four up to and including it
    int jo0 = 0, jo1 = 0, gend = 0;
    if constexpr (SELF) {
        if (lane < a.job1 - a.job0) {
            jo0 = a.job_off[a.job0 + lane];
            jo1 = a.job_off[a.job0 + lane + 1];
        }
        gend = (jo1 - jo0 + 3) >> 2;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const int t = __shfl_up(gend, m);
            if (lane >= m) gend += t;
        }
    }
    // CHUNK with regions: lane l holds the items of slices 0 .. l (the slices' counts, prefix-summed)
    int cend = 0;
    if constexpr (CHUNK) {
        if (a.chunk_regions > 0) {
            cend = lane < a.chunk_regions ? a.grp_off[lane] : 0;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                const int t = __shfl_up(cend, m);
                if (lane >= m) cend += t;
            }
        }
    }
    const uint32_t item_lo = (SELF || CHUNK) ? 0u : mapped ? (uint32_t)a.grp_off[a.job0] : 0u;
    const uint32_t n_items = SELF ? (uint32_t)__builtin_amdgcn_readlane(gend, 63)
                                  : CHUNK ? (a.chunk_regions > 0 ? (uint32_t)__builtin_amdgcn_readlane(cend, 63)
                                                                 : (uint32_t)a.grp_off[0])          // the item counter chunk_prep_kernel has left
                                  : mapped ? (uint32_t)a.grp_off[a.job1] : ((ncand + 3) / 4) * nq;   // item = (candidate group, query), group-major
    const uint32_t n_waves = gridDim.x * 4;
    const bool own_diam = a.diameter == nullptr;            // else: the caller's per-group diameters (caching_score's batches)
    constexpr bool with_solve = SOLVE;      // false (diagnostics): the cost phase alone, diam^2 as the score

    // lane roles: p = candidate of this lane (compute AND staging); (li, lj) = its 2 x 2 block of the 8 x 8 entries;
    // staging: the 16 lanes of group sg stage the 8 rows of candidate sg and query rows 2 sg, 2 sg + 1, chunk sc each
    const int p = lane >> 4, lp = lane & 15, li = lp >> 2, lj = lp & 3;
    const int sg = p, sc = lp;

    Solve pend;                    // the previous item's solve, advanced inside this item's cost stages
    bool have_pend = false;
    int slice = 0;                 // steps of the pending solve per cost stage

    // an item's documents: one memory round trip per item -- fetched one item AHEAD, so that the trip (several us under a
    // saturated HBM stream) runs under the current item's stages instead of in front of the next item's first loads
    struct Ctx {
        int64_t c_idx, q_idx;
        int c_len, q_len, c_start, q_start;
        int row0, w;           // CHUNK: first row of this group's chunk; widest exchange the item needs (wave-uniform: 1, 2, 4)
        int gsz;               // CHUNK: lane groups of this group's candidate
        bool my_c_real;
    };
    auto load_ctx = [&](uint32_t item) {
        Ctx x;
        x.row0 = 0;
        x.w = 1;
        x.gsz = 1;
        if constexpr (SELF) {
            const int job = __popcll(__ballot(gend <= (int)item));          // jobs that end at or before this item (empty ones included)
            const int g0 = job > 0 ? __builtin_amdgcn_readlane(gend, job - 1) : 0;
            const int cj0 = __builtin_amdgcn_readlane(jo0, job), cj1 = __builtin_amdgcn_readlane(jo1, job);
            const int first = cj0 + 4 * ((int)item - g0);
            const int cand = min(first + p, cj1 - 1);
            x.q_idx = a.job0 + job;
            x.q_len = a.q.len[x.q_idx];
            x.q_start = a.q.start[x.q_idx];
            x.my_c_real = first + p < cj1;
            x.c_idx = cand;
            x.c_len = a.c.len[cand];
            x.c_start = a.c.start[cand];
        } else if (mapped || CHUNK) {
            // the group's record (batch_prep_kernel): ONE memory round trip where the job tables + document tables take four
            size_t rec_i = item;
            if constexpr (CHUNK) {
                if (a.chunk_regions > 0) {      // item -> (slice, place in the slice's region)
                    const int sl = __popcll(__ballot(cend <= (int)item));
                    const int g0 = sl > 0 ? __builtin_amdgcn_readlane(cend, sl - 1) : 0;
                    rec_i = (size_t)sl * a.chunk_region_cap + (item - g0);
                }
            }
            const int32_t* rec = a.grp_rec + rec_i * 16;
            const int4 hd = *reinterpret_cast<const int4*>(rec);
            x.q_idx = hd.x;
            x.q_len = hd.y;
            x.q_start = hd.z;
            x.c_idx = rec[4 + p];
            x.c_len = rec[8 + p];
            x.c_start = rec[12 + p];
            if constexpr (CHUNK) {
                // slot word (chunk_prep_kernel): length | first group of the candidate << 8 | its groups << 12 | real << 16
                x.w = __builtin_amdgcn_readfirstlane(hd.w);
                x.row0 = 8 * (p - ((x.c_len >> 8) & 3));
                x.gsz = (x.c_len >> 12) & 7;
                x.my_c_real = ((x.c_len >> 16) & 1) != 0;
                x.c_len &= 0xff;
            } else {
                x.my_c_real = p < hd.w;
            }
        } else {
            const uint32_t cg = nq == 1 ? item : item / nq;
            const uint32_t q_loc = nq == 1 ? 0 : item - cg * nq;
            const uint32_t c_loc0 = cg * 4;
            const uint32_t my_c_loc = min(c_loc0 + (uint32_t)p, ncand - 1);    // tail groups: clamp (duplicate work, not stored)
            x.my_c_real = c_loc0 + (uint32_t)p < ncand;
            x.c_idx = a.cand0 + my_c_loc;
            x.q_idx = (int64_t)q_loc;
            x.c_len = a.c.len[x.c_idx];
            x.q_len = a.q.len[x.q_idx];
            x.c_start = a.c.start[x.c_idx];
            x.q_start = a.q.start[x.q_idx];
        }
        return x;
    };
    // a wave's items: every n_waves-th one -- or, SELF, a run of consecutive ones: those are mostly groups of ONE job, and the
    // query's box, formed during the first of them, is kept in LDS for the others
    uint32_t item_first = item_lo + blockIdx.x * 4 + wave, item_end = n_items, item_step = n_waves;
    if constexpr (SELF) {
        const uint32_t w = blockIdx.x * 4 + wave, base = n_items / n_waves, rem = n_items % n_waves;
        item_first = w * base + min(w, rem);
        item_end = item_first + base + (w < rem ? 1u : 0u);
        item_step = 1;
    }
    float* qcache = lds_all + 4 * kWaveLds + wave * 2 * kD;      // INBOX: [2][768] box of query cached_q
    int64_t cached_q = -1;
    Ctx next = load_ctx(item_first < item_end ? item_first : item_lo);
    F_STAMP(0);
    int n_done = 0;
    (void)n_done;

    for (uint32_t item = item_first; item < item_end; item += item_step) {
        const Ctx cur = next;
        next = load_ctx(item + item_step < item_end ? item + item_step : item);      // (the last item fetches itself again)
        const int64_t c_idx = cur.c_idx, q_idx = cur.q_idx;
        const int c_len = cur.c_len, q_len = cur.q_len, c_start = cur.c_start, q_start = cur.q_start;
        const bool my_c_real = cur.my_c_real;
        const int row0 = cur.row0, cw = cur.w;
        const bool cwide = CHUNK && (cw == 2 ? cur.gsz == 2 : cur.gsz >= 3);       // this lane's candidate crosses lane groups
        const float* qdoc = a.q.rows + (size_t)q_start * kD;
        const float* sy_doc = a.c.rows + (size_t)c_start * kD;                 // staging group == compute group
        // the query's per-coordinate box; with caller-supplied diameters any readable row stands in (the box term is then
        // unused) -- the loads stay UNCONDITIONAL: a branch around them makes the compiler wait for the just-issued row
        // loads at the join (a register copy of the conditionally defined value), which serialises every stage's HBM
        // latency with its arithmetic (measured: 160 instead of 110 us for the cost phase of 20 x 1000 pairs)
        const bool have_box = INBOX && q_idx == cached_q;       // wave-uniform
        const float* qb = (own_diam && !INBOX) ? qbox + (size_t)q_idx * 2 * kD : sy_doc;
        const int qb_hi = (own_diam && !INBOX) ? kD : 0;

        float accg[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        mfma4_t macc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) macc[m] = mfma4_t{0.f, 0.f, 0.f, 0.f};
        // squared-norm partials as register pairs ({x^2 + .., y^2 + ..} of the lane's chunks: two packed FMAs per row and
        // stage), halves added once per item; the box term likewise
        f2_t ny[8], nx[2], dsq = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) ny[k] = f2_t{0.f, 0.f};
        nx[0] = nx[1] = f2_t{0.f, 0.f};
        float4 vy[8], vx[2], qmn, qmx;
        auto issue_loads = [&](int st) {
            const int dofs = (st * kCh + sc) * 4;
#pragma unroll
            for (int j = 0; j < 8; ++j) vy[j] = ld4_stream(sy_doc + (size_t)min(row0 + j, c_len - 1) * kD + dofs);   // pad rows: copies of the last row
            if constexpr (!INBOX && !L2MAX) {
                qmn = ld4(qb + dofs);
                qmx = ld4(qb + qb_hi + dofs);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) vx[k] = ld4(qdoc + (size_t)min(2 * sg + k, q_len - 1) * kD + dofs);
        };
        auto sq_acc = [](f2_t acc, const float4& v) {
            acc = __builtin_elementwise_fma(f2_t{v.x, v.y}, f2_t{v.x, v.y}, acc);
            return __builtin_elementwise_fma(f2_t{v.z, v.w}, f2_t{v.z, v.w}, acc);
        };
        issue_loads(0);
#pragma unroll 1
        for (int st = 0; st < kStages; ++st) {
            // ---- stage: registers -> LDS, with box / norm side products; only THEN the next stage's loads go out, into
            // the registers just consumed (hoisted above the side products they need a second set of 48 registers and a
            // copy of all of them per stage); they fly under this stage's arithmetic ----
            if (a.center) {
                // centre of the rows at this lane's chunk: the mean of the eight staged query rows (rows past the query's end repeat
                // its last one: any vector near the data will do) -- two rows per lane group, summed across the four groups
                float4 mu = make_float4(vx[0].x + vx[1].x, vx[0].y + vx[1].y, vx[0].z + vx[1].z, vx[0].w + vx[1].w);
                mu.x += lane_xor<16>(mu.x); mu.y += lane_xor<16>(mu.y); mu.z += lane_xor<16>(mu.z); mu.w += lane_xor<16>(mu.w);
                mu.x += lane_xor<32>(mu.x); mu.y += lane_xor<32>(mu.y); mu.z += lane_xor<32>(mu.z); mu.w += lane_xor<32>(mu.w);
                mu.x *= 0.125f; mu.y *= 0.125f; mu.z *= 0.125f; mu.w *= 0.125f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    vy[j].x -= mu.x; vy[j].y -= mu.y; vy[j].z -= mu.z; vy[j].w -= mu.w;
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    vx[k].x -= mu.x; vx[k].y -= mu.y; vx[k].z -= mu.z; vx[k].w -= mu.w;
                }
                if constexpr (!INBOX && !L2MAX) {       // the query's box (from the tables launch) moves with its rows
                    qmn.x -= mu.x; qmn.y -= mu.y; qmn.z -= mu.z; qmn.w -= mu.w;
                    qmx.x -= mu.x; qmx.y -= mu.y; qmx.z -= mu.z; qmx.w -= mu.w;
                }
            }
            float4 mn = vy[0], mx = vy[0];
            {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    ny[j] = sq_acc(ny[j], vy[j]);
                    *reinterpret_cast<float4*>(lds + (8 + sg * 8 + j) * kRowStride + sc * 4) = vy[j];
                }
                if constexpr (!L2MAX) {
                    // the box over the eight rows as bare v_min3 / v_max3 (fused_solve.h): four instructions per component and side;
                    // fminf / fmaxf cost a canonicalising v_max x, x per loaded operand on top
                    mn = vmin3_4(vy[0], vy[1], vy[2]); mx = vmax3_4(vy[0], vy[1], vy[2]);
                    mn = vmin3_4(mn, vy[3], vy[4]); mx = vmax3_4(mx, vy[3], vy[4]);
                    mn = vmin3_4(mn, vy[5], vy[6]); mx = vmax3_4(mx, vy[5], vy[6]);
                    // the last link takes the query's box along where it is at hand: from the tables launch, or (INBOX) from the wave's cache
                    if constexpr (!INBOX && !CHUNK) {
                        mn = vmin3_4(mn, vy[7], qmn); mx = vmax3_4(mx, vy[7], qmx);
                    } else if constexpr (INBOX) {
                        if (have_box) {
                            const float* qc = qcache + (st * kCh + sc) * 4;
                            mn = vmin3_4(mn, vy[7], *reinterpret_cast<const float4*>(qc));
                            mx = vmax3_4(mx, vy[7], *reinterpret_cast<const float4*>(qc + kD));
                        } else {
                            mn = vmin3_4(mn, vy[7], vy[7]); mx = vmax3_4(mx, vy[7], vy[7]);
                        }
                    } else {
                        mn = vmin3_4(mn, vy[7], vy[7]); mx = vmax3_4(mx, vy[7], vy[7]);
                    }
                }
                if constexpr (CHUNK && !L2MAX) {
                    // the candidate's other chunks sit in the neighbouring lane groups (a chunk past the document's end repeats
                    // its last row: box neutral)
                    if (cw >= 2) {
                        const float inf = __builtin_inff();
                        float4 zn = mn, zx = mx;
                        if (!cwide) {           // a narrower candidate in the item: neutral towards its neighbours, keeps its own box
                            zn = make_float4(inf, inf, inf, inf);
                            zx = make_float4(-inf, -inf, -inf, -inf);
                        }
                        zn.x = vmin1(zn.x, lane_xor<16>(zn.x)); zn.y = vmin1(zn.y, lane_xor<16>(zn.y));
                        zn.z = vmin1(zn.z, lane_xor<16>(zn.z)); zn.w = vmin1(zn.w, lane_xor<16>(zn.w));
                        zx.x = vmax1(zx.x, lane_xor<16>(zx.x)); zx.y = vmax1(zx.y, lane_xor<16>(zx.y));
                        zx.z = vmax1(zx.z, lane_xor<16>(zx.z)); zx.w = vmax1(zx.w, lane_xor<16>(zx.w));
                        if (cw == 4) {
                            zn.x = vmin1(zn.x, lane_xor<32>(zn.x)); zn.y = vmin1(zn.y, lane_xor<32>(zn.y));
                            zn.z = vmin1(zn.z, lane_xor<32>(zn.z)); zn.w = vmin1(zn.w, lane_xor<32>(zn.w));
                            zx.x = vmax1(zx.x, lane_xor<32>(zx.x)); zx.y = vmax1(zx.y, lane_xor<32>(zx.y));
                            zx.z = vmax1(zx.z, lane_xor<32>(zx.z)); zx.w = vmax1(zx.w, lane_xor<32>(zx.w));
                        }
                        if (cwide) {
                            mn = zn;
                            mx = zx;
                        }
                    }
                }
                if constexpr (!INBOX && !L2MAX && CHUNK) {
                    const f2_t dlo = {vmax1(mx.x, qmx.x) - vmin1(mn.x, qmn.x), vmax1(mx.y, qmx.y) - vmin1(mn.y, qmn.y)};
                    const f2_t dhi = {vmax1(mx.z, qmx.z) - vmin1(mn.z, qmn.z), vmax1(mx.w, qmx.w) - vmin1(mn.w, qmn.w)};
                    dsq = __builtin_elementwise_fma(dhi, dhi, __builtin_elementwise_fma(dlo, dlo, dsq));
                } else if constexpr (!INBOX && !L2MAX) {
                    const f2_t dlo = {mx.x - mn.x, mx.y - mn.y}, dhi = {mx.z - mn.z, mx.w - mn.w};
                    dsq = __builtin_elementwise_fma(dhi, dhi, __builtin_elementwise_fma(dlo, dlo, dsq));
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    nx[k] = sq_acc(nx[k], vx[k]);
                    *reinterpret_cast<float4*>(lds + (2 * sg + k) * kRowStride + sc * 4) = vx[k];
                }
            }
            // pin the side products HERE (the optimiser otherwise sinks these loop-carried sums below the loads)
            if constexpr (INBOX)
                asm volatile("" : "+v"(ny[0]), "+v"(ny[1]), "+v"(ny[2]), "+v"(ny[3]), "+v"(ny[4]), "+v"(ny[5]), "+v"(ny[6]), "+v"(ny[7]),
                                  "+v"(nx[0]), "+v"(nx[1]), "+v"(mn.x), "+v"(mn.y), "+v"(mn.z), "+v"(mn.w), "+v"(mx.x), "+v"(mx.y),
                                  "+v"(mx.z), "+v"(mx.w) : : "memory");
            else
                asm volatile("" : "+v"(ny[0]), "+v"(ny[1]), "+v"(ny[2]), "+v"(ny[3]), "+v"(ny[4]), "+v"(ny[5]), "+v"(ny[6]), "+v"(ny[7]),
                                  "+v"(nx[0]), "+v"(nx[1]), "+v"(dsq) : : "memory");
            if (st + 1 < kStages) issue_loads(st + 1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if constexpr (INBOX) {
                // the query's box at this lane's chunk: from the eight staged query rows (rows past the document's end are
                // copies of its last row) and into the wave's cache, or from the cache; joined with the candidate's
                f2_t dlo = {mx.x - mn.x, mx.y - mn.y}, dhi = {mx.z - mn.z, mx.w - mn.w};      // have_box: the cached box went into mn / mx above
                if (!have_box) {
                    float4 qn, qx;
                    float* qc = qcache + (st * kCh + sc) * 4;
                    qn = qx = *reinterpret_cast<const float4*>(lds + sc * 4);
#pragma unroll
                    for (int r = 1; r < 8; ++r) {
                        const float4 qv = *reinterpret_cast<const float4*>(lds + r * kRowStride + sc * 4);
                        qn.x = vmin1(qn.x, qv.x); qn.y = vmin1(qn.y, qv.y); qn.z = vmin1(qn.z, qv.z); qn.w = vmin1(qn.w, qv.w);
                        qx.x = vmax1(qx.x, qv.x); qx.y = vmax1(qx.y, qv.y); qx.z = vmax1(qx.z, qv.z); qx.w = vmax1(qx.w, qv.w);
                    }
                    if (sg == 0) {
                        *reinterpret_cast<float4*>(qc) = qn;
                        *reinterpret_cast<float4*>(qc + kD) = qx;
                    }
                    dlo = f2_t{vmax1(mx.x, qx.x) - vmin1(mn.x, qn.x), vmax1(mx.y, qx.y) - vmin1(mn.y, qn.y)};
                    dhi = f2_t{vmax1(mx.z, qx.z) - vmin1(mn.z, qn.z), vmax1(mx.w, qx.w) - vmin1(mn.w, qn.w)};
                }
                dsq = __builtin_elementwise_fma(dhi, dhi, __builtin_elementwise_fma(dlo, dlo, dsq));
            }
            // ---- accumulate: every lane walks the staged chunks for its own 2 x 2 entries ----------------------------
            if constexpr (MFMA) {
                // matrix-pipe form: block b = lane >> 2 = (p, iq, jq) is the 4 x 4 sub-block (rows 4 iq .., columns 4 jq ..) of
                // pair p; lane t = lane & 3 feeds query row 4 iq + t as A and candidate row 4 jq + t as B, one coordinate per
                // instruction; it ends up with column 4 jq + t of the block (4 accumulator registers = rows 4 iq .. + 3)
                const int mb = lane >> 2, mt = lane & 3, miq = (mb >> 1) & 1, mjq = mb & 1;
                const float* xm = lds + (4 * miq + mt) * kRowStride;
                const float* ym = lds + (8 + p * 8 + 4 * mjq + mt) * kRowStride;
#pragma unroll kMfmaUnroll
                for (int c = 0; c < kCh; ++c) {
                    const float4 xa = *reinterpret_cast<const float4*>(xm + c * 4);
                    const float4 yb = *reinterpret_cast<const float4*>(ym + c * 4);
                    macc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa.x, yb.x, macc[0], 0, 0, 0);
                    macc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa.y, yb.y, macc[1], 0, 0, 0);
                    macc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa.z, yb.z, macc[2], 0, 0, 0);
                    macc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa.w, yb.w, macc[3], 0, 0, 0);
                }
            } else {
            const float* xr = lds + (2 * li) * kRowStride;
            const float* yr = lds + (8 + p * 8 + 2 * lj) * kRowStride;
#pragma unroll 1
            for (int c = 0; c < kCh; ++c) {
                float4 xv[2], yv[2];
#pragma unroll
                for (int x = 0; x < 2; ++x) xv[x] = *reinterpret_cast<const float4*>(xr + x * kRowStride + c * 4);
#pragma unroll
                for (int y = 0; y < 2; ++y) yv[y] = *reinterpret_cast<const float4*>(yr + y * kRowStride + c * 4);
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y)
                        accg[x][y] = fmaf(xv[x].w, yv[y].w, fmaf(xv[x].z, yv[y].z, fmaf(xv[x].y, yv[y].y, fmaf(xv[x].x, yv[y].x, accg[x][y]))));
            }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the stage buffer is rewritten next
            __builtin_amdgcn_wave_barrier();
            // ---- a slice of the PREVIOUS item's solve, in the shadow of the loads just issued ------------------------
            if constexpr (SOLVE)
                if (have_pend) solve_steps<CHUNK>(pend, a, slice);
        }

        if constexpr (MFMA) {
            // block columns -> the solve's 2 x 2 layout, through the (idle) stage buffer: pair p's 8 x 8 entries row-major
            const int mb = lane >> 2, mt = lane & 3, miq = (mb >> 1) & 1, mjq = mb & 1;
            float* tr = lds + p * 64;
#pragma unroll
            for (int r = 0; r < 4; ++r) tr[(4 * miq + r) * 8 + 4 * mjq + mt] = (macc[0][r] + macc[1][r]) + (macc[2][r] + macc[3][r]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) accg[x][y] = tr[(2 * li + x) * 8 + 2 * lj + y];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // ---- norms: sum the staging lanes' partials through the scratch table nscr[value][lane] ----------------------
        // value 0..7: |y_j|^2 partials of the lane's staged candidate; 8, 9: |x|^2 partials of its two query rows
#pragma unroll
        for (int k = 0; k < 8; ++k) nscr[k * kNormLd + lane] = ny[k].x + ny[k].y;
#pragma unroll
        for (int k = 0; k < 2; ++k) nscr[(8 + k) * kNormLd + lane] = nx[k].x + nx[k].y;
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
        float xx[2], yy[2];
#pragma unroll
        for (int y = 0; y < 2; ++y) yy[y] = table_sum(2 * lj + y, p * 16);
#pragma unroll
        for (int x = 0; x < 2; ++x) xx[x] = table_sum(8 + ((2 * li + x) & 1), ((2 * li + x) >> 1) * 16);
        // box terms were formed by the lanes that staged the candidate's rows = this candidate's 16 lanes
        float diam2 = dsq.x + dsq.y;
        diam2 += lane_xor<1>(diam2); diam2 += lane_xor<2>(diam2); diam2 += lane_xor<4>(diam2); diam2 += lane_xor<8>(diam2);

        // ---- finish the entries.  Only x.y was accumulated: -cdist comes from the same expansion as geomloss's cost, and
        // the entries where it cancels (torch.cdist's direct formula differs there) are redone below ----------------------
        // (round 6: a cancelling entry is redone from the exact sum whatever formula torch.cdist would pick -- also beyond 25 rows: include/aspire_hip.h, SHARED SENTENCES)
        float cost[2][2], neg[2][2];
        bool redo[2][2];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const int i = 2 * li + x, j = row0 + 2 * lj + y;
                const float sq = fmaf(-2.f, accg[x][y], xx[x]) + yy[y];
                const float ns = xx[x] + yy[y];
                redo[x][y] = i < q_len && j < c_len && sq < 1e-4f * ns * ns;
                cost[x][y] = sqrtf(fmaxf(sq, 1e-8f));
                neg[x][y] = -sqrtf(fmaxf(sq, 0.f));
            }
        {
            // direct-formula redo, the whole wave on one entry (12 coordinates per lane), FOUR entries per memory round trip:
            // under a saturated HBM stream a dependent round trip is ~5 us, and a wave that falls behind by n of them
            // finishes the launch n x 5 us late (one duplicate document = 8 entries in a 20 000-pair call, one entry per
            // round trip: 145 instead of 101 us)
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) {
                    unsigned long long wm = __ballot(redo[x][y]);
                    while (wm != 0) {
                        int owner[4];
                        float part[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            owner[e] = wm != 0 ? (int)__builtin_ctzll(wm) : -1;
                            wm = wm != 0 ? wm & (wm - 1) : 0;
                        }
                        // all 24 loads go out before the first is consumed: no branch around them (an empty slot
                        // repeats the first entry's rows)
                        float4 u[4][3], v[4][3];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int o = owner[e] >= 0 ? owner[e] : owner[0];
                            const int ol = o & 15, i = 2 * (ol >> 2) + x;
                            const int j = (CHUNK ? __builtin_amdgcn_readlane(row0, o) : 0) + 2 * (ol & 3) + y;
                            const int cs_e = __builtin_amdgcn_readlane(c_start, o);
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
                            part[e] = 0.f;
#pragma unroll
                            for (int t = 0; t < 3; ++t) {
                                const float d0 = u[e][t].x - v[e][t].x, d1 = u[e][t].y - v[e][t].y, d2 = u[e][t].z - v[e][t].z, d3 = u[e][t].w - v[e][t].w;
                                part[e] = fmaf(d3, d3, fmaf(d2, d2, fmaf(d1, d1, fmaf(d0, d0, part[e]))));
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (owner[e] >= 0) {
                                const float tot = wave_sum(part[e]);
                                if (lane == owner[e]) {
                                    neg[x][y] = -sqrtf(tot);
                                    cost[x][y] = sqrtf(fmaxf(tot, 1e-8f));      // geomloss's cost from the same exact sum (see the header of this block)
                                }
                            }
                    }
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the scratch table is rewritten by the next item
        __builtin_amdgcn_wave_barrier();
        cached_q = q_idx;
        F_STAMP(1 + 2 * n_done);

        // ---- the previous item's solve ends here (its last steps, score, store); this item's begins ---------------------
        if constexpr (SOLVE)
            if (have_pend) solve_finish<CHUNK>(pend, a);
        if constexpr (with_solve) {
            bool rv[2], cv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                rv[t] = 2 * li + t < q_len;
                cv[t] = row0 + 2 * lj + t < c_len;
            }
            const float diam = own_diam ? fmaxf(sqrtf(diam2), kMinDiameter) : a.diameter[q_idx * a.n_groups + c_idx / a.diam_group];      // CROSS only
            solve_begin<CHUNK>(pend, a, cost, neg, rv, cv, diam, cw, cwide, row0 == 0);
            pend.out = (my_c_real && row0 == 0) ? (mapped ? c_idx : q_idx * a.c.n + c_idx) : (int64_t)-1;
            if (q_len > 8 || c_len > 8 * cur.gsz) pend.valid |= 16u;
            slice = (pend.max_steps + kStages - 1) / kStages;
            have_pend = true;
            F_STAMP(2 + 2 * n_done);
            ++n_done;
        } else if constexpr (L2MAX) {
            float m = kNegBig;
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) m = fmaxf(m, (2 * li + x < q_len && row0 + 2 * lj + y < c_len) ? neg[x][y] : kNegBig);
            m = xg_max<CHUNK>(max_li(max_lj(m)), cw, cwide);
            if (q_len > 8 || c_len > 8 * cur.gsz) m = __builtin_nanf("");          // longer than the tile: never truncated silently
            if (my_c_real && lp == 0 && row0 == 0) a.scores[mapped ? c_idx : q_idx * a.c.n + c_idx] = m;
        } else if (my_c_real && lp == 0) {
            a.scores[mapped ? c_idx : q_idx * a.c.n + c_idx] = diam2;
        }
    }
    F_STAMP(6);
    // (CHUNK batches are a round and a half of items: a wave in its last solve shares its SIMD with a wave that still streams an
    // item and paces the launch.  Lowering the solving wave's priority -- s_setprio 2 for the item loop, 0 here -- measured
    // nothing: 97.6 against 96.7 us on the config-4 shape.)
    if constexpr (SOLVE)
        if (have_pend && !a.skip_tail) solve_finish<CHUNK>(pend, a);   // the wave's last item: nothing left to hide it behind
    F_STAMP(7);
}

}  // namespace

// batched jobs: few enough jobs for the in-wave tables (one lane per job), and hyper-parameters at which overflowed sums (each
// re-solved by the wave that finds it) stay the exception
bool fused_self_ok(int64_t jobs, const aspire_ot_params* prm) {
    return jobs <= 64 && prm->scaling >= 0.25 && tuning().fused_nosolve != 1 && !tuning().fused_valu && !tuning().fused_noself;
}

// one query against a pool with its own per-pair diameters: the box comes from the staged query rows (no box launch)
bool fused_inbox_ok(const aspire_repset* q, const float* diameter) {
    return q->n == 1 && diameter == nullptr && tuning().fused_nosolve != 1 && !tuning().fused_valu && !tuning().fused_noself;
}

bool fused_path_ok(const aspire_repset* q, const aspire_repset* c) {
    const int mq = q->max_len, mc = c->max_len;
    return q->ext == 0 && c->ext == 0 && mq > 0 && mc > 0 && mq <= 8 && mc <= 8;
}

// tsAspire (max-sim) of every (query, candidate) pair, CROSS, documents of <= 8 rows: the fused kernel's streaming phase
// self (batched jobs, <= 64 of them): no tables in front of the launch -- the waves derive an item's job from job_off (SELF)
int launch_pair_fused_l2max(const ScoreArgs& a, int64_t groups_bound, hipStream_t stream, bool self) {
    const int64_t waves = groups_bound < 256 * 8 ? groups_bound : 256 * 8;
    const dim3 grid((unsigned)((waves + 3) / 4));
    if (self)
        hipLaunchKernelGGL((pair_fused_kernel<true, false, true, true>), grid, dim3(256), 4 * kWaveLds * sizeof(float), stream, a,
                           (const float*)nullptr);
    else
        hipLaunchKernelGGL((pair_fused_kernel<true, false, false, true>), grid, dim3(256), 4 * kWaveLds * sizeof(float), stream, a,
                           (const float*)nullptr);
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}

// groups_bound: upper bound of the launch's items (groups of four candidates x queries)
int launch_pair_fused(const ScoreArgs& a_in, int64_t groups_bound, const float* qbox, hipStream_t stream) {
    // two 4-wave workgroups per CU are resident.  (Built for three -- 168 registers, the kernel-invariant values spilled,
    // the norm table already shares the stage buffer so the LDS fits -- the 20 x 1000 call went from 120 to 144 us.)
    ScoreArgs a = a_in;
    a.skip_tail = tuning().fused_nosolve == 2;
    const int64_t cap = tuning().fused_waves > 0 ? tuning().fused_waves : 256 * 8;
    const int64_t waves = groups_bound < cap ? groups_bound : cap;
    const dim3 grid((unsigned)((waves + 3) / 4));
    const bool self = qbox == nullptr && a.pairing == kPairMapped;       // batched jobs without the tables launch (fused_self_ok)
    const bool inbox1 = qbox == nullptr && a.pairing == ASPIRE_PAIR_CROSS;      // one query, box in-wave (fused_inbox_ok)
    if ((self || inbox1) && (tuning().fused_nosolve == 1 || tuning().fused_valu)) return ASPIRE_ERR_INVALID_ARG;
    const size_t lds = 4 * (kWaveLds + (self || inbox1 ? 2 * kD : 0)) * sizeof(float);      // + a query-box cache per wave
    if (tuning().fused_nosolve == 1) hipLaunchKernelGGL((pair_fused_kernel<true, false>), grid, dim3(256), lds, stream, a, qbox);
    else if (tuning().fused_valu) hipLaunchKernelGGL((pair_fused_kernel<false, true>), grid, dim3(256), lds, stream, a, qbox);
    else if (self) {
        // 67.6 KB of dynamic LDS: above the default limit of 64 KB (raised once per process; two workgroups still fit a CU)
        static std::once_flag raised;
        static hipError_t raise_rc = hipSuccess;
        std::call_once(raised, [] {
            raise_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(pair_fused_kernel<true, true, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        });
        ASPIRE_HIP_OK(raise_rc);
        hipLaunchKernelGGL((pair_fused_kernel<true, true, true>), grid, dim3(256), lds, stream, a, qbox);
    }
    else if (inbox1) {
        static std::once_flag raised1;
        static hipError_t raise1_rc = hipSuccess;
        std::call_once(raised1, [] {
            raise1_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(pair_fused_kernel<true, true, false, false, true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        });
        ASPIRE_HIP_OK(raise1_rc);
        hipLaunchKernelGGL((pair_fused_kernel<true, true, false, false, true>), grid, dim3(256), lds, stream, a, qbox);
    }
    else hipLaunchKernelGGL((pair_fused_kernel<true, true>), grid, dim3(256), lds, stream, a, qbox);
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}

}  // namespace aspire
#ifdef ASPIRE_PHASE_CLOCK
extern "C" void aspire_debug_fused_buffer(void* p) {
    long long* q = (long long*)p;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(aspire::g_fdbg), &q, sizeof(q));
}
#endif
namespace aspire {
// tsAspire on the same items: the streaming phase with the max epilogue
int launch_pair_fused_chunk_l2max(const ScoreArgs& a, int64_t items_bound, hipStream_t stream) {
    const int64_t cap = tuning().fused_waves > 0 ? tuning().fused_waves : 256 * 8;
    const int64_t waves = items_bound < cap ? items_bound : cap;
    hipLaunchKernelGGL((pair_fused_kernel<true, false, false, true, false, true>), dim3((unsigned)((waves + 3) / 4)), dim3(256),
                       4 * kWaveLds * sizeof(float), stream, a, (const float*)nullptr);
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}

// batched jobs with candidates of up to 32 rows against queries of <= 8 (the CHUNK form; items from chunk_prep_kernel)
int launch_pair_fused_chunk(const ScoreArgs& a_in, int64_t items_bound, const float* qbox, hipStream_t stream) {
    ScoreArgs a = a_in;
    a.skip_tail = tuning().fused_nosolve == 2;
    const int64_t cap = tuning().fused_waves > 0 ? tuning().fused_waves : 256 * 8;
    const int64_t waves = items_bound < cap ? items_bound : cap;
    const dim3 grid((unsigned)((waves + 3) / 4));
    if (tuning().fused_nosolve == 1)      // timing experiments: the streaming phase alone
        hipLaunchKernelGGL((pair_fused_kernel<true, false, false, false, false, true>), grid, dim3(256), 4 * kWaveLds * sizeof(float), stream, a, qbox);
    else
        hipLaunchKernelGGL((pair_fused_kernel<true, true, false, false, false, true>), grid, dim3(256), 4 * kWaveLds * sizeof(float), stream, a, qbox);
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}

}  // namespace aspire
