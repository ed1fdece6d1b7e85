// Many-query cost tiles on pre-split fp16 planes: the x.y term of A5's pairwise distances as a GEMM on
// v_mfma_f32_32x32x16_f16, three exact products per term, operands moved HBM/L2 -> LDS by LDS-DMA.
//
// Reference arithmetic: src/learning/facetid_models/pair_distances.py:48-55 and :167-176 (torch.cdist of every query
// sentence against every candidate sentence, then the mask and the max / the Sinkhorn costs) and geomloss 0.2.4's
// squared_distances (|x|^2 - 2 x.y + |y|^2).
//
// gram.hip's 128-column form splits every fp32 tile into three bf16 planes while it is staged (VALU work in the main loop,
// SIX matrix products per term).  Here the split is done ONCE per resident store (include/aspire_hip.h: aspire_rep_planes --
// per-row power-of-two scale, the store's common vector mu taken off first, h + l fp16 planes, |x - mu|^2 per row) and the
// kernel is the encoder's gemm_p_kernel loop (encoder.hip) with the Gram epilogue of gram.hip:
//   * tile = whole documents in slots of mr rows, 128 candidate rows (MFMA M side: a lane's four consecutive accumulator
//     rows are four consecutive candidate sentences) x 128 query rows; every (query, candidate) pair lives in one workgroup;
//   * a stage = one 16-coordinate k block: 8 KB of candidate rows + 8 KB of query rows, gathered row by row (64 B per row
//     and k block, 4 lanes x 16 B) by global_load_lds_dwordx4 with a scalar base (+ k block pitch per stage) and a per-lane
//     offset that is computed once per tile -- no VALU, no vector registers in the staging path; rows a document does not have
//     read the store's zero row; the piece a lane fetches is XOR-swizzled by its LDS row so that the fragment reads
//     (ds_read_b128, lane = (row, k half)) are conflict free;
//   * three-stage ring (48 KB) + 4 KB of tables: three workgroups per CU; per stage and wave 4 DMA instructions, 8 fragment
//     reads, 12 MFMAs, one barrier;
//   * epilogue: x.y = acc / (s_row s_col), d^2 = |x|^2 + |y|^2 - 2 x.y with the stores' precomputed norms; entries where the
//     expansion cancels are redone from the fp32 rows with the direct formula (work list, 16 lanes per entry) exactly as in
//     gram.hip; tsAspire reduces each pair's maximum in LDS, otAspire writes the pair's cost / -cdist slots.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <utility>

#include "tuning.h"
#include "score_types.h"
#include "score_device.h"

namespace aspire {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

constexpr int kKBlocks = kD / 16;          // 48
constexpr int kRowB = 64;                  // one row of one k block: 2 planes x 2 k halves x 16 bytes
constexpr float kDirectTau = 1e-4f;        // as gram.hip: redo (x-y)^2 directly when d^2 < tau * (|x|^2 + |y|^2)^2
constexpr int kMuSample = 4096;
#ifndef GRAMP_SPREAD
#define GRAMP_SPREAD 1   // 1: the LDS-DMA pieces of a stage go out one at a time between runs of MFMAs, 0: together behind the barrier
#endif
#ifndef GRAMP_GRP_SHIFT
#define GRAMP_GRP_SHIFT 2   // ping-pong groups: bit of the wave index that tells the two waves of a SIMD apart
#endif
#ifndef GRAMP_PROBE
#define GRAMP_PROBE 0   // timing probes of the ping-pong loop (wrong results): 1 no MFMAs, 2 no reads / LDS-DMA, 3 no LDS-DMA, 4 no reads
#endif
#ifndef GRAMP_PRE
#define GRAMP_PRE 4      // MFMAs of a stage issued in front of the next stage's barrier
#endif

struct PlaneView {
    const unsigned char* planes;
    const float* nrm;
    const float* iscale;
    uint32_t plane_rows;
    uint32_t zero_row;
};

struct GramPArgs {
    RepSet q, c;
    PlaneView qp, cp;
    int64_t cand0;
    uint32_t ncand, nq;
    int mr_q, mr_c, dpt_q, dpt_c, n_qt, n_ct;
    int E, ld;
    int cdist_mode;
    float* cost;
    float* neg;
    float* scores;
    int skip_cost;       // the 128 x 128 tiles leave geomloss's cost to the solve stage (ScoreArgs::cost_from_neg); the wide tiles always store it
};

// The 1 KB pieces of one k block that a wave moves (two of the candidate tile, NB of the query tile) in one statement; M0
// (compiler-reserved) saved and restored inside.  An instruction's `offset:` moves the global AND the LDS address: the bases arrive
// (NB - 1) KB low and piece p's lane offsets (NB - 1 - p) KB high, so that `offset:1024 p` lands piece p p KB further in LDS at
// its own rows' addresses.  hipcc does not count these loads: the caller waits with s_waitcnt vmcnt(N).
template <int NB>
__device__ __forceinline__ void glds_kblock_gather(uint64_t a_base, uint64_t b_base, const uint32_t (&va)[2], const uint32_t (&vb)[NB],
                                                   uint32_t a_dst, uint32_t b_dst) {
    uint32_t keep;
    if constexpr (NB == 1)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %4 offset:0\n\tglobal_load_lds_dwordx4 %2, %4 offset:1024\n\t"
                     "s_mov_b32 m0, %7\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %3, %5 offset:0\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(va[0]), "v"(va[1]), "v"(vb[0]), "s"(a_base), "s"(b_base), "s"(a_dst), "s"(b_dst)
                     : "memory");
    else if constexpr (NB == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %5 offset:0\n\tglobal_load_lds_dwordx4 %2, %5 offset:1024\n\t"
                     "s_mov_b32 m0, %8\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %3, %6 offset:0\n\tglobal_load_lds_dwordx4 %4, %6 offset:1024\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(va[0]), "v"(va[1]), "v"(vb[0]), "v"(vb[1]), "s"(a_base), "s"(b_base), "s"(a_dst), "s"(b_dst)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %9\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %7 offset:0\n\tglobal_load_lds_dwordx4 %2, %7 offset:1024\n\t"
                     "s_mov_b32 m0, %10\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %3, %8 offset:0\n\tglobal_load_lds_dwordx4 %4, %8 offset:1024\n\t"
                     "global_load_lds_dwordx4 %5, %8 offset:2048\n\tglobal_load_lds_dwordx4 %6, %8 offset:3072\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(va[0]), "v"(va[1]), "v"(vb[0]), "v"(vb[1]), "v"(vb[2]), "v"(vb[3]), "s"(a_base), "s"(b_base), "s"(a_dst),
                       "s"(b_dst)
                     : "memory");
}

// One 1 KB piece: 64 lanes x 16 B from global [base + voff(lane) + IMM] to LDS [dst + IMM + 16 lane] (M0 saved and restored inside)
template <int IMM>
__device__ __forceinline__ void glds_piece(uint64_t base, uint32_t voff, uint32_t dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(base), "s"(dst), "n"(IMM)
                 : "memory");
}

// step(t + S, ring slot S % RING, fragment set S & 1) for S = 0 .. N - 1
template <class F, int... P>
__device__ __forceinline__ void unrolled_pieces(F&& f, std::integer_sequence<int, P...>) {
    (f(std::integral_constant<int, P>{}), ...);
}
template <class F, int... S, int RING>
__device__ __forceinline__ void unrolled_steps(F& step, int t, std::integer_sequence<int, S...>, std::integral_constant<int, RING>) {
    (step(t + S, std::integral_constant<int, S % RING>{}, std::integral_constant<int, S & 1>{}), ...);
}

__device__ __forceinline__ float4 ldrow4(const float* rows, int32_t row, int ofs) {
    return *reinterpret_cast<const float4*>(rows + (size_t)row * kD + ofs);
}

// Tile = BM candidate rows x BN query rows, BM / 32 waves of 64 x BN / 2 (wave = (wr, wc): candidate rows 64 wr .., query rows
// (BN / 2) wc ..), a ring of RING stages:
//   128 x 128: four waves of 64 x 64, three workgroups per CU (the encoder GEMM's geometry)
//   128 x 256: four waves of 64 x 128, two per CU
//   256 x 128: eight waves of 64 x 64 (the encoder's gemm_p_w8_kernel geometry, which wins there at 118 registers = two workgroups per CU, four
//              waves per SIMD).  HERE the loop needs 141 registers (gathered DMA offsets, two fragment sets): one workgroup per CU, 539 us
//              against 496 - 502 at 32 x 50 000 x 8; capped at 128 it spills inside the loop (2 189 us).  Pinned form only (GRAM_TILE=256128).
//   256 x 256: eight waves of 64 x 128, one per CU -- a row is moved into LDS once per 256 rows of the other side: half the LDS-DMA
//              bytes per product of the 128 x 128 form, which is what that form runs out of (16 KB per workgroup and stage through
//              the CU's vector-memory path for 384 matrix-pipe cycles per SIMD), and two thirds of its fragment reads
// PP (256 x 256 only): ping-pong.  The eight waves are two groups of four (waves w and w + 4 share a SIMD); every k block is two
// segments with a workgroup barrier between segments, and group 1 runs ONE segment behind group 0 (it passes one extra barrier up
// front): while one wave of a SIMD is in its load segment -- fragment reads of stage t out of LDS, its LDS-DMA pieces of stage
// t + RING - 1 into the slot stage t - 1 has left, wait for both and for its pieces of stage t + 1 -- the other is in its MFMA
// segment (24 MFMAs at raised priority, nothing else): the matrix pipe of a SIMD always has exactly one wave feeding it, without a
// second fragment set or MFMAs threaded between loads.  (All waves in step -- the non-PP form of this tile -- leaves both waves of a
// SIMD loading at the same time, then both competing for the pipe: 45 % busy.)
template <int BM, int BN, int RING, bool L2MAX, bool PP = false>
// (the second launch bound is WAVES PER SIMD on this toolchain: 3 for the 128 x 128 form's three workgroups per CU; the 256 x 128 form would
// want 4 (two workgroups of eight waves per CU) but its loop needs 141 registers -- bounded to 4 it spills inside the loop (2 189 us) -- so it
// says 3 and runs ONE workgroup per CU: the measured 539 us of NOTES.md round 5)
__global__ void __launch_bounds__(2 * BM, (BM == 128 && BN == 128) ? 3 : (BM == 256 && BN == 128) ? 3 : BM == 128 ? 2 : 1) pair_gram_p_kernel(GramPArgs g) {
    static_assert(!PP || (BM == 256 && BN == 256), "ping-pong: two groups of four waves");
    constexpr int NT = 2 * BM;                      // threads
    constexpr int TN = BN / 64;                     // 32-column blocks per wave
    constexpr int NB = 2 * BN / BM;                 // 1 KB pieces of the query tile per wave and k block (the candidate tile's: 2)
    constexpr int kATileB = BM * kRowB, kBTileB = BN * kRowB;
    constexpr int kStageB = kATileB + kBTileB;      // [candidate rows][query rows]
    constexpr int kPerWave = 2 + NB;                // LDS-DMA instructions per wave and stage
    constexpr int NMMA = 3 * 2 * TN;                // MFMAs per wave and stage
    static_assert(NB == 1 || NB == 2 || NB == 4, "pieces per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
    __shared__ int32_t c_row[BM], q_row[BN];        // fp32 row of the tile row, -1 = none
    __shared__ int32_t c_di[BM], q_di[BN];          // document slot | row in the document << 8 | (len > 25) << 16, -1 = no document
    __shared__ __attribute__((aligned(16))) float c_nrm[BM];
    __shared__ __attribute__((aligned(16))) float c_is[BM];
    __shared__ float q_nrm[BN], q_is[BN];
    __shared__ uint32_t wl_count;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, lr = lane & 31, lk = lane >> 5;
    // (a run-time guard around the cost stores costs the wide tiles 576 bytes of scratch: they keep storing it, the solve stage ignores it)
    const bool put_cost = (BM == 128 && BN == 128) ? !g.skip_cost : true;
    (void)put_cost;
    // workgroups that share a candidate tile sit next to each other in an XCD's launch order (gram.hip)
    uint32_t L;
    {
        const uint32_t nb = gridDim.x, b = blockIdx.x, x = b & 7, q8 = nb >> 3, r8 = nb & 7;
        L = x * q8 + (x < r8 ? x : r8) + (b >> 3);
    }
    const uint32_t ct = L / (uint32_t)g.n_qt, qt = L - ct * (uint32_t)g.n_qt;

    // ---- tile tables -----------------------------------------------------------------------------------
    for (int rr = tid; rr < BM + BN; rr += NT) {
        if (rr < BM) {
            const int r = rr;
            const int d = r / g.mr_c, i = r - d * g.mr_c;
            const uint32_t c_loc = ct * g.dpt_c + d;
            const bool doc_ok = d < g.dpt_c && c_loc < g.ncand;
            int len = 0, start = 0;
            if (doc_ok) {
                len = g.c.len[g.cand0 + c_loc];
                start = g.c.start[g.cand0 + c_loc];
            }
            const bool row_ok = doc_ok && i < len;
            const int32_t row = row_ok ? start + i : -1;
            c_row[r] = row;
            c_di[r] = doc_ok ? (d | (i << 8) | ((len > 25) << 16)) : -1;
            c_nrm[r] = row_ok ? g.cp.nrm[row] : (L2MAX ? INFINITY : 0.f);      // max-sim: a row that is not there never is the minimum
            c_is[r] = row_ok ? g.cp.iscale[row] : 0.f;
        } else {
            const int r = rr - BM;
            const int d = r / g.mr_q, i = r - d * g.mr_q;
            const uint32_t q_loc = qt * g.dpt_q + d;
            const bool doc_ok = d < g.dpt_q && q_loc < g.nq;
            int len = 0, start = 0;
            if (doc_ok) {
                len = g.q.len[q_loc];
                start = g.q.start[q_loc];
            }
            const bool row_ok = doc_ok && i < len;
            const int32_t row = row_ok ? start + i : -1;
            q_row[r] = row;
            q_di[r] = doc_ok ? (d | (i << 8) | ((len > 25) << 16)) : -1;
            q_nrm[r] = row_ok ? g.qp.nrm[row] : (L2MAX ? INFINITY : 0.f);
            q_is[r] = row_ok ? g.qp.iscale[row] : 0.f;
        }
    }
    if (tid == 0) wl_count = 0;
    __syncthreads();

    // ---- LDS-DMA addressing: wave w moves rows [32 w, 32 w + 32) of the candidate tile and [16 NB w, 16 NB (w + 1)) of the query
    // tile, lane = (row, 16-byte slot) -----------------------------------------------------------------------------------------
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)ring;
    uint32_t va[2], vb[NB];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = 32 * wave + 16 * p + (lane >> 2);
        const uint32_t piece = (uint32_t)((lane & 3) ^ ((r >> 2) & 3));      // LDS slot j of row r holds piece j ^ r[2..3]
        const int32_t cr = c_row[r];
        va[p] = (cr < 0 ? g.cp.zero_row : (uint32_t)cr) * kRowB + 16 * piece + 1024u * (1 - p);
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        const int r = 16 * NB * wave + 16 * p + (lane >> 2);
        const uint32_t piece = (uint32_t)((lane & 3) ^ ((r >> 2) & 3));
        const int32_t qr = q_row[r];
        vb[p] = (qr < 0 ? g.qp.zero_row : (uint32_t)qr) * kRowB + 16 * piece + 1024u * (NB - 1 - p);
    }
    const uint64_t a_src = (uint64_t)(uintptr_t)g.cp.planes - 1024, b_src = (uint64_t)(uintptr_t)g.qp.planes - 1024 * (NB - 1);
    const uint64_t a_step = (uint64_t)g.cp.plane_rows * kRowB, b_step = (uint64_t)g.qp.plane_rows * kRowB;
    auto issue = [&](int slot, int kb) {
        const uint32_t dst = lds0 + slot * kStageB;
        glds_kblock_gather<NB>(a_src + (uint64_t)kb * a_step, b_src + (uint64_t)kb * b_step, va, vb, dst + (2 * wave) * 1024,
                               dst + kATileB + (NB * wave) * 1024);
    };
    // piece p of the wave's kPerWave pieces of a stage on its own (the main loop spreads them between its MFMAs: issued back to
    // back behind the barrier by every wave at once they queue up in front of the CU's vector-memory path, and a wave waiting
    // to issue one issues no MFMA either)
    auto issue_piece = [&](int slot, int kb, auto pc) {
        constexpr int p = decltype(pc)::value;
        const uint32_t dst = lds0 + slot * kStageB;
        if constexpr (p < 2)
            glds_piece<1024 * p>(a_src + (uint64_t)kb * a_step, va[p], dst + (2 * wave) * 1024);
        else
            glds_piece<1024 * (p - 2)>(b_src + (uint64_t)kb * b_step, vb[p - 2], dst + kATileB + (NB * wave) * 1024);
    };
    const uint32_t frag0 = 16 * (lk ^ ((lr >> 2) & 3));
    const unsigned char* a_rd = ring + (wr * 64 + lr) * kRowB;
    const unsigned char* b_rd = ring + kATileB + (wc * 32 * TN + lr) * kRowB;

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Software pipeline.  The fragments of stage t sit in registers (two sets taking turns) while stage t + 1 is read out of LDS
    // and stages t + 2, t + 3 are in flight: a step is
    //     [first kPre MFMAs of stage t]  wait: own pieces of stage t + 1 landed, own reads of stage t done;  barrier;
    //     LDS-DMA of stage t + 3 into the slot stage t has just left;  fragment reads of stage t + 1;  [the other MFMAs of stage t]
    // -- behind the barrier a wave has MFMAs to issue at once (nothing it reads from LDS is needed before the NEXT step), in front of
    // it the matrix pipe still holds the kPre it issued.  (Left to the compiler -- reads, sched_barrier, MFMAs per step -- the next
    // step's wait + barrier were hoisted above MFMAs whose operands were still landing: a wave crossed the barrier with three
    // ds_reads of the slot in flight that its neighbours then refilled.  One tile in ~3000 came back with one k block's low plane
    // stale, an error of 1e-4 in a handful of scores, run-dependent; tools/planeerr.py.)
    struct Frags {
        f16x8_t a[2][2], b[TN][2];      // [block][plane]
    };
    Frags F[2];
    auto read_frags = [&](Frags& f, int slot) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const uint32_t fo = frag0 ^ (32 * pl);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                f.a[i][pl] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(a_rd + slot * kStageB + i * 32 * kRowB + fo));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                f.b[j][pl] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(b_rd + slot * kStageB + j * 32 * kRowB + fo));
        }
    };
    auto mma = [&](const Frags& f, int first, int last) {
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};        // the small products first
#pragma unroll
        for (int m = 0; m < NMMA; ++m) {
            if (m < first || m >= last) continue;
            const int term = m / (2 * TN), i = (m / TN) & 1, j = m % TN;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i][PA[term]], f.b[j][PB[term]], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (PP) {
        const int grp = (wave >> GRAMP_GRP_SHIFT) & 1;
        Frags& f = F[0];
#pragma unroll
        for (int st = 0; st < RING - 1; ++st) issue(st, st);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * kPerWave) : "memory");
        __builtin_amdgcn_s_barrier();                  // everybody's pieces of stage 0 have landed
        if (grp == 1) __builtin_amdgcn_s_barrier();    // the stagger: group 1 starts one segment late
        asm volatile("" ::: "memory");
        auto pp_step = [&](int t, auto slotc) {
            constexpr int slot = decltype(slotc)::value;
            // ---- load segment: stage t -> registers; the wave's pieces of stage t + RING - 1 into the slot of stage t - 1 (both
            // groups read that one at least a segment ago and waited for the reads before their barrier)
            __builtin_amdgcn_sched_barrier(0);
            if (GRAMP_PROBE != 2 && GRAMP_PROBE != 4) read_frags(f, slot);
            if (GRAMP_PROBE != 2 && GRAMP_PROBE != 3 && t + RING - 1 < kKBlocks) issue((slot + RING - 1) % RING, t + RING - 1);
            // reads done (the slot may be refilled by whoever passes the next barrier), own pieces of stage t + 1 landed: all
            // but the younger stages' (RING - 2 of them, fewer at the end)
            const int younger = kKBlocks - 2 - t < RING - 2 ? (kKBlocks - 2 - t < 0 ? 0 : kKBlocks - 2 - t) : RING - 2;
            if (RING >= 4 && younger == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * kPerWave) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kPerWave) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // ---- MFMA segment
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            if (GRAMP_PROBE != 1) mma(f, 0, NMMA);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        static_assert(kKBlocks % RING == 0, "ring depth");
#pragma unroll 1
        for (int t = 0; t < kKBlocks; t += RING) {
            pp_step(t, std::integral_constant<int, 0>{});
            pp_step(t + 1, std::integral_constant<int, 1>{});
            pp_step(t + 2, std::integral_constant<int, 2>{});
            if constexpr (RING == 4) pp_step(t + 3, std::integral_constant<int, 3>{});
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();    // as many barriers as group 1
    } else {
    constexpr int kPre = GRAMP_PRE;
#pragma unroll
    for (int st = 0; st < RING; ++st) issue(st, st);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 1) * kPerWave) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_frags(F[0], 0);
    auto step = [&](int t, auto slotc, auto parc) {
        constexpr int slot = decltype(slotc)::value, par = decltype(parc)::value;     // stage t lives in ring slot `slot`, fragment set `par`
        __builtin_amdgcn_sched_barrier(0);
        if (GRAMP_PROBE != 6) mma(F[par], 0, kPre);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < kKBlocks) {
            // own pieces of stage t + 1: everything but the younger stages' (RING - 2 of them, fewer at the end of the loop)
            const int younger = kKBlocks - 2 - t < RING - 2 ? kKBlocks - 2 - t : RING - 2;
            if (RING >= 4 && younger == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * kPerWave) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kPerWave) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            read_frags(F[par ^ 1], (slot + 1) % RING);
            if (GRAMP_SPREAD == 0 && t + RING < kKBlocks) issue(slot, t + RING);
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (GRAMP_SPREAD == 0) {
            mma(F[par], kPre, NMMA);
        } else {
            // the stage's remaining MFMAs in kPerWave + 1 runs with one LDS-DMA piece of stage t + RING between two runs
            constexpr int kRun = (NMMA - kPre) / (kPerWave + 1);
            const bool more = t + RING < kKBlocks;
            unrolled_pieces([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                if (GRAMP_PROBE != 6) mma(F[par], kPre + p * kRun, kPre + (p + 1) * kRun);
                __builtin_amdgcn_sched_barrier(0);
                if (more && GRAMP_PROBE != 7) issue_piece(slot, t + RING, pc);
                __builtin_amdgcn_sched_barrier(0);
            }, std::make_integer_sequence<int, kPerWave>{});
            if (GRAMP_PROBE != 6) mma(F[par], kPre + kPerWave * kRun, NMMA);
        }
    };
    constexpr int kUnroll = RING % 2 ? 2 * RING : RING;      // ring slots x fragment sets
    static_assert(RING >= 3 && RING <= 4 && kKBlocks % kUnroll == 0, "ring depth");
#pragma unroll 1
    for (int t = 0; t < kKBlocks; t += kUnroll) {
        unrolled_steps(step, t, std::make_integer_sequence<int, kUnroll>{}, std::integral_constant<int, RING>{});
    }
    }
    if (GRAMP_PROBE == 5 && acc[0][0][0] != 12345.678f) return;      // timing probe: no epilogue
    __syncthreads();      // every wave is done with the ring: the epilogue's scratch lives there

    // ---- epilogue (gram.hip's, on the stores' precomputed norms).  C/D layout of the 32x32 MFMA: col = lane & 31 (query row),
    // row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (candidate row)
    constexpr int kPairs = (BM / 8) * (BN / 8);                            // [BM / 8 candidate documents][BN / 8 query documents]
    uint32_t* pairmin = reinterpret_cast<uint32_t*>(ring);
    uint32_t* wlist = reinterpret_cast<uint32_t*>(ring) + kPairs;
    constexpr int kCap = (RING * kStageB) / 4 - kPairs;
    if constexpr (L2MAX) {
        // max over the pair's entries of -sqrt(d^2) = -sqrt(min d^2) (the square root is monotone): the tile works on squared
        // distances -- eight VALU instructions per entry instead of ~50 -- and takes one square root per pair at the end.  d^2 >= 0
        // orders like its bit pattern; rows that do not exist carry |.|^2 = +inf and never win.
#pragma unroll
        for (int e = tid; e < kPairs; e += NT) pairmin[e] = 0x7F800000u;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = wc * 32 * TN + 32 * j + lr;
                const int32_t qdi = q_di[n];
                const float xx = q_nrm[n], qis2 = -2.f * q_is[n];
                const int qd = qdi & 255;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int m0 = wr * 64 + 32 * i + 8 * g4 + 4 * lk;
                    const int32_t cdi = c_di[m0];
                    if (qdi < 0 || cdi < 0) continue;
                    const float4 yy4 = *reinterpret_cast<const float4*>(&c_nrm[m0]);
                    const float4 is4 = *reinterpret_cast<const float4*>(&c_is[m0]);
                    const float yy[4] = {yy4.x, yy4.y, yy4.z, yy4.w};
                    const float cis[4] = {is4.x, is4.y, is4.z, is4.w};
                    float best = INFINITY;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float ns = xx + yy[k];
                        const float sq = fmaf(acc[i][j][4 * g4 + k], qis2 * cis[k], ns);       // the scales are powers of two: exact
                        const bool redo = sq < kDirectTau * ns * ns;
                        if (__builtin_expect(redo, 0)) {
                            const uint32_t slot = atomicAdd(&wl_count, 1u);
                            if (slot < (uint32_t)kCap) {
                                wlist[slot] = ((uint32_t)(m0 + k) << 16) | (uint32_t)n;
                            } else {       // list full: lane-local
                                const float* x = g.q.rows + (size_t)q_row[n] * kD;
                                const float* y = g.c.rows + (size_t)c_row[m0 + k] * kD;
                                float s0 = 0.f;
                                for (int e = 0; e < kD; ++e) {
                                    const float df = x[e] - y[e];
                                    s0 = fmaf(df, df, s0);
                                }
                                best = fminf(best, s0);
                            }
                        } else {
                            best = fminf(best, fmaxf(sq, 0.f));
                        }
                    }
                    if (best < INFINITY) atomicMin(&pairmin[(cdi & 255) * (BN / 8) + qd], __builtin_bit_cast(uint32_t, best));
                }
            }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = wc * 32 * TN + 32 * j + lr;
                const int32_t qdi = q_di[n];
                const float xx = q_nrm[n], qis = q_is[n];
                const int32_t qrow = q_row[n];
                const int qd = qdi & 255, qi = (qdi >> 8) & 255;
                const long long qo = ((long long)(qt * g.dpt_q + qd) * g.ncand) * g.E + (long long)qi * g.ld;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int m0 = wr * 64 + 32 * i + 8 * g4 + 4 * lk;
                    const int32_t cdi = c_di[m0];
                    if (qdi < 0 || cdi < 0) continue;
                    const int cd = cdi & 255, ci = (cdi >> 8) & 255;
                    const float4 yy4 = *reinterpret_cast<const float4*>(&c_nrm[m0]);
                    const float4 is4 = *reinterpret_cast<const float4*>(&c_is[m0]);
                    const float yy[4] = {yy4.x, yy4.y, yy4.z, yy4.w};
                    const float cis[4] = {is4.x, is4.y, is4.z, is4.w};
                    float cost[4], neg[4];
                    bool redo[4];
                    bool any_redo = false;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float dot = acc[i][j][4 * g4 + k] * qis * cis[k];       // powers of two: exact
                        const float sq = fmaf(-2.f, dot, xx) + yy[k];
                        const float d = sqrtf(fmaxf(sq, 0.f));
                        cost[k] = fmaxf(d, __builtin_sqrtf(1e-8f));                   // = sqrt(max(sq, 1e-8)): geomloss's clamp
                        neg[k] = -d;
                        const float ns = xx + yy[k];
                        // (otAspire: ALSO under torch.cdist's matmul formula -- a side beyond 25 rows -- where the reference's own -cdist is the expansion: geomloss's
                        // cost of the entry is derived from this tile (cost_from_neg), and the rule for a cancelling entry is the exact sum for both:
                        // include/aspire_hip.h, SHARED SENTENCES; round 6, tools/fuzz_parity.py 24 412 planes)
                        redo[k] = sq < kDirectTau * ns * ns && qrow >= 0 && c_row[m0 + k] >= 0;
                        if (redo[k]) {
                            const uint32_t slot = atomicAdd(&wl_count, 1u);
                            if (slot < (uint32_t)kCap) {
                                wlist[slot] = ((uint32_t)(m0 + k) << 16) | (uint32_t)n;
                            } else {       // list full: lane-local
                                const float* x = g.q.rows + (size_t)qrow * kD;
                                const float* y = g.c.rows + (size_t)c_row[m0 + k] * kD;
                                float s0 = 0.f;
                                for (int e = 0; e < kD; ++e) {
                                    const float df = x[e] - y[e];
                                    s0 = fmaf(df, df, s0);
                                }
                                neg[k] = -sqrtf(s0);
                                cost[k] = sqrtf(fmaxf(s0, 1e-8f));
                                redo[k] = false;
                            }
                        }
                        any_redo |= redo[k];
                    }
                    const long long co = (long long)(ct * g.dpt_c + cd) * g.E + ci;
                    if (!any_redo) {
                        if (put_cost) *reinterpret_cast<float4*>(g.cost + qo + co) = make_float4(cost[0], cost[1], cost[2], cost[3]);
                        *reinterpret_cast<float4*>(g.neg + qo + co) = make_float4(neg[0], neg[1], neg[2], neg[3]);
                    } else {   // the work-list pass writes the flagged ones (-cdist AND geomloss's cost, from the same exact sum): no address is stored twice
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (!redo[k]) {
                                if (put_cost) g.cost[qo + co + k] = cost[k];
                                g.neg[qo + co + k] = neg[k];
                            }
                    }
                }
            }
    }
    __syncthreads();
    {
        // 16 lanes (one DPP row) per flagged entry, 48 coordinates per lane, from the fp32 rows
        const uint32_t n_redo = min(wl_count, (uint32_t)kCap);
        const int grp = tid >> 4, l16 = tid & 15;
        for (uint32_t e0 = 0; e0 < n_redo; e0 += NT / 16) {
            const uint32_t e = e0 + grp;
            const bool live = e < n_redo;
            const uint32_t mn = wlist[live ? e : 0];
            const int m = (int)(mn >> 16), n = (int)(mn & 0xFFFFu);
            const int32_t xr = q_row[n], yr = c_row[m];
            float p0 = 0.f, p1 = 0.f;
#pragma unroll
            for (int c = 0; c < 12; c += 2) {
                const float4 u0 = ldrow4(g.q.rows, xr, 4 * l16 + 64 * c), v0 = ldrow4(g.c.rows, yr, 4 * l16 + 64 * c);
                const float4 u1 = ldrow4(g.q.rows, xr, 4 * l16 + 64 * c + 64), v1 = ldrow4(g.c.rows, yr, 4 * l16 + 64 * c + 64);
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
            if (live && l16 == 0) {
                const int32_t cdi = c_di[m], qdi = q_di[n];
                if constexpr (L2MAX) {
                    atomicMin(&pairmin[(cdi & 255) * (BN / 8) + (qdi & 255)], __builtin_bit_cast(uint32_t, part));
                } else {
                    const float negd = -sqrtf(part);
                    const long long qo = ((long long)(qt * g.dpt_q + (qdi & 255)) * g.ncand) * g.E + (long long)((qdi >> 8) & 255) * g.ld;
                    const long long co = (long long)(ct * g.dpt_c + (cdi & 255)) * g.E + ((cdi >> 8) & 255);
                    g.neg[qo + co] = negd;
                    if (put_cost) g.cost[qo + co] = sqrtf(fmaxf(part, 1e-8f));
                }
            }
        }
    }
    if constexpr (L2MAX) {
        __syncthreads();
#pragma unroll
        for (int e = tid; e < kPairs; e += NT) {
            const int cd = e / (BN / 8), qd = e % (BN / 8);
            const uint32_t c_loc = ct * g.dpt_c + cd, q_loc = qt * g.dpt_q + qd;
            if (cd < g.dpt_c && qd < g.dpt_q && c_loc < g.ncand && q_loc < g.nq)
                g.scores[(int64_t)q_loc * g.c.n + g.cand0 + c_loc] = -sqrtf(__builtin_bit_cast(float, pairmin[e]));
        }
    }
}

// ---- preparation ---------------------------------------------------------------------------------------------------------
// mu[64 b .. 64 b + 63] = mean over rows k * stride, k < nsample
__global__ void __launch_bounds__(256) rows_mean_kernel(const float* __restrict__ rows, int64_t nsample, int64_t stride,
                                                        float* __restrict__ mu) {
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
    float s = 0.f;
    for (int64_t k = grp; k < nsample; k += 4) s += rows[(size_t)(k * stride) * kD + blockIdx.x * 64 + c];
    part[grp][c] = s;
    __syncthreads();
    if (grp == 0) mu[blockIdx.x * 64 + c] = ((part[0][c] + part[1][c]) + (part[2][c] + part[3][c])) / (float)nsample;
}

// One wave per row (a workgroup = 16 consecutive rows = one 1 KB run per k block): v = rows[r] - mu, |v|^2, the row's
// power-of-two scale, the two fp16 planes.  Rows [total_rows, plane_rows) come out as zeros.
__global__ void __launch_bounds__(256) rows_to_planes_kernel(const float* __restrict__ rows, int64_t total_rows, uint32_t plane_rows,
                                                             const float* __restrict__ mu, unsigned char* __restrict__ planes,
                                                             float* __restrict__ nrm, float* __restrict__ iscale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 m[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) m[c] = *reinterpret_cast<const float4*>(mu + 4 * lane + 256 * c);
    for (int rr = 0; rr < 4; ++rr) {
        const int64_t r = (int64_t)blockIdx.x * 16 + wave * 4 + rr;
        if (r >= plane_rows) return;
        float4 v[3];
        const bool real = r < total_rows;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (real) {
                const float4 x = ld4_stream(rows + (size_t)r * kD + 4 * lane + 256 * c);
                v[c] = make_float4(x.x - m[c].x, x.y - m[c].y, x.z - m[c].z, x.w - m[c].w);
            }
        }
        float amax = 0.f, sq = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[c].x), fabsf(v[c].y)), fmaxf(fabsf(v[c].z), fabsf(v[c].w))));
            sq += sq4(v[c]);
        }
        amax = wave_max(amax);
        sq = wave_sum(sq);
        // s = 2^(14 - floor(log2 amax)): amax * s in [2^14, 2^15); exponent field clamped so that s and 1 / s are normal numbers
        // (an all-zero row, a row of denormals or a non-finite one gets whatever the clamp gives: its planes are 0 / inf as its
        // fp32 rows are)
        int se = 268 - (int)((__builtin_bit_cast(uint32_t, amax) >> 23) & 255u);
        se = se < 1 ? 1 : se > 253 ? 253 : se;
        const float s = __builtin_bit_cast(float, (uint32_t)se << 23);
        const float is = __builtin_bit_cast(float, (uint32_t)(254 - se) << 23);
        if (lane == 0) {
            nrm[r] = real ? sq : 0.f;
            iscale[r] = real ? is : 0.f;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int k = 4 * lane + 256 * c;
            const float x0 = v[c].x * s, x1 = v[c].y * s, x2 = v[c].z * s, x3 = v[c].w * s;
            const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1, h2 = (_Float16)x2, h3 = (_Float16)x3;
            const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1), l2 = (_Float16)(x2 - (float)h2),
                           l3 = (_Float16)(x3 - (float)h3);
            auto pack = [](_Float16 a, _Float16 b) {
                return (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
            };
            unsigned char* dst = planes + ((size_t)(k >> 4) * plane_rows + (size_t)r) * kRowB + ((k >> 3) & 1) * 16 + ((k >> 2) & 1) * 8;
            *reinterpret_cast<uint2*>(dst) = make_uint2(pack(h0, h1), pack(h2, h3));
            *reinterpret_cast<uint2*>(dst + 32) = make_uint2(pack(l0, l1), pack(l2, l3));
        }
    }
}

constexpr size_t kMuBytes = 4096;
uint32_t plane_rows_of(int64_t total_rows) { return (uint32_t)((total_rows + 1 + 15) / 16 * 16); }

PlaneView view_of(const aspire_rep_planes* p) {
    return PlaneView{(const unsigned char*)p->planes, p->row_nrm, p->row_iscale, (uint32_t)p->plane_rows, (uint32_t)p->total_rows};
}

}  // namespace

bool gram_planes_ok(const ScoreArgs& a) {
    const aspire_rep_planes *q = a.q_planes, *c = a.c_planes;
    if (!q || !c || !q->planes || !c->planes || q->mu != c->mu) return false;
    if (tuning().gemm_form == 1 || tuning().gemm_form == 2) return false;      // ASPIRE_HIP_GEMM=f32 | bf16x3: the forms that read the fp32 rows
    // per-lane byte offsets of the LDS-DMA are 32 bits wide
    return q->plane_rows * kRowB + 2048 < ((int64_t)1 << 32) && c->plane_rows * kRowB + 2048 < ((int64_t)1 << 32);
}

int launch_pair_gram_planes(const ScoreArgs& a, const GramGeometry& geo, bool l2max, float* cost, float* neg, hipStream_t stream) {
    GramPArgs g{};
    g.q = a.q;
    g.c = a.c;
    g.qp = view_of(a.q_planes);
    g.cp = view_of(a.c_planes);
    g.cand0 = a.cand0;
    g.ncand = (uint32_t)(a.cand1 - a.cand0);
    g.nq = (uint32_t)a.q.n;
    g.mr_q = geo.mr_q;
    g.mr_c = geo.mr_c;
    g.dpt_q = geo.dpt_q;
    g.dpt_c = geo.dpt_c;
    g.n_qt = geo.n_qt;
    g.n_ct = geo.n_ct;
    g.E = geo.E;
    g.ld = geo.ld;
    g.cdist_mode = a.cdist_mode;
    g.cost = cost;
    g.neg = neg;
    g.scores = a.scores;
    g.skip_cost = a.cost_from_neg;
    const dim3 grid((unsigned)(g.n_ct * g.n_qt));
    auto launch = [&](auto kern, int bm, int bn, int ring) {
        const int lds = ring * (bm + bn) * kRowB;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            (void)e;
        }
        hipLaunchKernelGGL(kern, grid, dim3(2 * bm), lds, stream, g);
    };
    const int form = geo.bm * 1000 + geo.bn;
    if (form == 256256) {
        if (tuning().gram_pp && tuning().gram_ring == 4) {
            if (l2max) launch(pair_gram_p_kernel<256, 256, 4, true, true>, 256, 256, 4);
            else launch(pair_gram_p_kernel<256, 256, 4, false, true>, 256, 256, 4);
        } else if (tuning().gram_pp) {
            if (l2max) launch(pair_gram_p_kernel<256, 256, 3, true, true>, 256, 256, 3);
            else launch(pair_gram_p_kernel<256, 256, 3, false, true>, 256, 256, 3);
        } else if (tuning().gram_ring == 4) {
            if (l2max) launch(pair_gram_p_kernel<256, 256, 4, true>, 256, 256, 4);
            else launch(pair_gram_p_kernel<256, 256, 4, false>, 256, 256, 4);
        } else {
            if (l2max) launch(pair_gram_p_kernel<256, 256, 3, true>, 256, 256, 3);
            else launch(pair_gram_p_kernel<256, 256, 3, false>, 256, 256, 3);
        }
    } else if (form == 256128) {
        if (l2max) launch(pair_gram_p_kernel<256, 128, 3, true>, 256, 128, 3);
        else launch(pair_gram_p_kernel<256, 128, 3, false>, 256, 128, 3);
    } else if (form == 128256) {
        if (l2max) launch(pair_gram_p_kernel<128, 256, 3, true>, 128, 256, 3);
        else launch(pair_gram_p_kernel<128, 256, 3, false>, 128, 256, 3);
    } else {
        ASPIRE_REQUIRE(form == 128128, ASPIRE_ERR_INVALID_ARG, "no %d x %d plane tiles", geo.bm, geo.bn);
        if (l2max) launch(pair_gram_p_kernel<128, 128, 3, true>, 128, 128, 3);
        else launch(pair_gram_p_kernel<128, 128, 3, false>, 128, 128, 3);
    }
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}

}  // namespace aspire

using namespace aspire;

extern "C" size_t aspire_rep_planes_bytes(int64_t total_rows) {
    if (total_rows < 0) return 0;
    const size_t pr = plane_rows_of(total_rows);
    return kMuBytes + 2 * pr * sizeof(float) + pr * (size_t)kD * 4;
}

extern "C" int aspire_rep_planes_prepare(const float* rows, int64_t total_rows, int64_t D, const float* mu, void* blob, size_t blob_bytes,
                                         aspire_rep_planes* out_host, void* stream) {
    ASPIRE_REQUIRE(D == kD, ASPIRE_ERR_UNSUPPORTED, "encoding dim must be %d", kD);
    ASPIRE_REQUIRE(total_rows >= 0 && (rows || total_rows == 0) && blob && out_host, ASPIRE_ERR_INVALID_ARG, "null argument");
    ASPIRE_REQUIRE(total_rows + 17 < ((int64_t)1 << 32) / kRowB, ASPIRE_ERR_UNSUPPORTED, "more rows than the plane tiles address (%lld)",
                   (long long)total_rows);
    ASPIRE_REQUIRE(blob_bytes >= aspire_rep_planes_bytes(total_rows), ASPIRE_ERR_INVALID_ARG, "blob too small");
    ASPIRE_REQUIRE(((uintptr_t)blob & 15) == 0, ASPIRE_ERR_INVALID_ARG, "blob must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const uint32_t pr = plane_rows_of(total_rows);
    unsigned char* b = (unsigned char*)blob;
    float* mu_own = (float*)b;
    float* nrm = (float*)(b + kMuBytes);
    float* is = nrm + pr;
    unsigned char* planes = (unsigned char*)(is + pr);
    if (!mu) {
        if (total_rows > 0) {
            const int64_t ns = total_rows < kMuSample ? total_rows : kMuSample;
            hipLaunchKernelGGL(rows_mean_kernel, dim3(kD / 64), dim3(256), 0, st, rows, ns, total_rows / ns, mu_own);
            ASPIRE_LAUNCH_OK();
        } else {
            ASPIRE_HIP_OK(hipMemsetAsync(mu_own, 0, kD * sizeof(float), st));
        }
        mu = mu_own;
    }
    hipLaunchKernelGGL(rows_to_planes_kernel, dim3(pr / 16), dim3(256), 0, st, rows, total_rows, pr, mu, planes, nrm, is);
    ASPIRE_LAUNCH_OK();
    out_host->planes = planes;
    out_host->row_nrm = nrm;
    out_host->row_iscale = is;
    out_host->mu = mu;
    out_host->total_rows = total_rows;
    out_host->plane_rows = pr;
    return ASPIRE_OK;
}
