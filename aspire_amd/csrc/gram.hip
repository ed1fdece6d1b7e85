// Pairwise sentence costs on the matrix cores, for the many-query / long-document regimes of A5.
//
// Reference arithmetic: src/learning/facetid_models/pair_distances.py:48-55 (torch.cdist of every query
// sentence against every candidate sentence) and geomloss 0.2.4's squared_distances (|x|^2 - 2 x.y + |y|^2).
//
// With one query the cost stage streams every candidate row once and is HBM bound (score.hip: pair_tile_kernel).
// With Q queries each candidate row meets Q * S_q query rows and the stage is 2 * 768 flops per (row, row) entry:
// at 32 queries x 8 sentences that is 128 flop per candidate byte, far on the compute side of the ridge, and the
// VALU kernels of score.hip top out near 20 T entries*coords/s.  The x.y term IS a GEMM (fp32 in, fp32 out), so
// here it runs on v_mfma_f32_32x32x2_f32 -- exact fp32 multiply-adds, 4x the VALU rate -- and the epilogue turns the
// Gram tile into the two distance forms.  torch.cdist's direct (x - y)^2 formula (used by the reference below 26
// rows) is reproduced to ~1e-5 by the expansion except where x ~ y (cancellation); those entries -- squared distance
// below 1e-4 of the squared norm sum -- are recomputed coordinate by coordinate.
//
// Tile = whole documents: a 128-row candidate tile holds floor(128 / mr_c) documents in slots of mr_c rows
// (mr = the set's longest document, rounded up to 4), a 32/64/128-row query tile likewise; rows beyond a
// document's length are zero and their entries are masked downstream exactly like the other kernels' pad entries.
// Every (query doc, candidate doc) pair therefore lives in exactly one workgroup: the otAspire epilogue writes the
// pair's cost / -cdist slot (16-byte stores, the 8T x 8T slot of a pair is contiguous), the tsAspire epilogue
// reduces the pair's maximum in LDS and stores the score -- no global atomics, no second pass.
//
// Candidates are the MFMA M side (a lane's 4 consecutive accumulator rows = 4 consecutive candidate sentences = one
// float4 of the pair's row-major cost matrix), queries the N side.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "tuning.h"
#include "score_types.h"
#include "score_device.h"

namespace aspire {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int kBM = 128;   // candidate rows per tile
constexpr int kBK = 16;    // coordinates per LDS stage
constexpr float kDirectTau = 1e-4f;   // recompute (x-y)^2 directly when d^2 < tau * (|x|^2 + |y|^2)^2

struct GramArgs {
    RepSet q, c;
    int64_t cand0;
    uint32_t ncand, nq;
    int mr_q, mr_c;      // rows per document slot
    int dpt_q, dpt_c;    // documents per tile
    int n_qt, n_ct;      // tiles
    int E, ld;           // otAspire slot: entries per pair (64 T^2), row stride (8 T)
    int cdist_mode;
    float* cost;
    float* neg;
    float* scores;       // tsAspire: [nq][c.n]
    int center;          // ASPIRE_OT_FLAG_CENTER / ASPIRE_CDIST_CENTER: subtract the tile's first query row from every staged row
    const float* qbox;   // fused diameter (BOX): per-query coordinate boxes [nq][2][768]
    float* diam2;        //                       out [nq][ncand]
};

__device__ __forceinline__ uint32_t order_key(float f) {
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unorder_key(uint32_t u) {
    return __builtin_bit_cast(float, (u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
}

// Row addresses travel through LDS tables as integers and come back as GLOBAL-address-space pointers: a plain
// `const float*` read from LDS is a generic pointer, which the compiler serves with flat_load_dword -- four scalar
// loads per float4, each bumping lgkmcnt, so that every LDS wait of the MFMA loop also waited for HBM.  Rows that
// do not exist point at a row of zeros instead of being predicated.
typedef const float __attribute__((address_space(1)))* gfp;
__device__ float g_zero_row[kD];
__device__ __forceinline__ float4 ldg4(unsigned long long addr, int ofs) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float __attribute__((ext_vector_type(4))) v4;
    const v4 v = *reinterpret_cast<const v4 __attribute__((address_space(1)))*>(reinterpret_cast<gfp>(addr) + ofs);
    return make_float4(v.x, v.y, v.z, v.w);
#else
    (void)addr; (void)ofs;
    return make_float4(0.f, 0.f, 0.f, 0.f);   // host pass of the single-source compile; never called
#endif
}

__device__ __forceinline__ float ldg1(unsigned long long addr, int ofs) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *(reinterpret_cast<gfp>(addr) + ofs);
#else
    (void)addr; (void)ofs;
    return 0.f;
#endif
}

__device__ __forceinline__ float sq4f(const float4& a) { return fmaf(a.w, a.w, fmaf(a.z, a.z, fmaf(a.y, a.y, a.x * a.x))); }

// sum_d (x_d - y_d)^2 over the 768 coordinates (rare path: near-coincident sentences)
__device__ __noinline__ float direct_d2(unsigned long long x, unsigned long long y) {
    float s0 = 0.f, s1 = 0.f;
    for (int k = 0; k < kD; k += 8) {
        const float4 a = ldg4(x, k), b = ldg4(y, k), c = ldg4(x, k + 4), d = ldg4(y, k + 4);
        const float e0 = a.x - b.x, e1 = a.y - b.y, e2 = a.z - b.z, e3 = a.w - b.w;
        const float f0 = c.x - d.x, f1 = c.y - d.y, f2 = c.z - d.z, f3 = c.w - d.w;
        s0 = fmaf(e3, e3, fmaf(e2, e2, fmaf(e1, e1, fmaf(e0, e0, s0))));
        s1 = fmaf(f3, f3, fmaf(f2, f2, fmaf(f1, f1, fmaf(f0, f0, s1))));
    }
    return s0 + s1;
}

// Three workgroups per CU for the 32/64-column tiles costs 80-144 B of scratch per lane (epilogue values) against
// 178-194 registers at two per CU, and is still ahead: 1 x 20000 x 12 OT 347 vs 375 us, 1 x 4000 x 20 189 vs 235 us.
// Two workgroups per CU for every form: at three (168 registers) the 32- / 64-column forms spilled 80 / 144 bytes; without
// spills (178 / 194 registers) 1 x 20 000 x 12 is 8 % slower (371 vs 340 us) and 2 x 10 000 x 12 4 - 20 % faster (240 vs 248 us
// otAspire, 103 vs 126 us max-sim).
// X3 (the 128-column forms): the Gram tile on the bf16 matrix pipe at fp32 accuracy -- operands split into three bf16 planes
// when a tile is staged, six v_mfma_f32_32x32x16_bf16 products per term (see gemm_bf16x3_kernel in encoder.hip: same error
// against float64 as the fp32-input MFMA at 3/8 of its matrix-pipe time).
template <int BN, bool L2MAX, bool BOX, bool X3 = false>
__global__ void __launch_bounds__(256, 2) pair_gram_kernel(GramArgs g) {
    static_assert(!BOX || (!L2MAX && BN <= 64), "the fused diameter serves the few-query otAspire tiles");
    static_assert(!X3 || (BN == 128 && !BOX), "the bf16x3 form is built for the 128-column tiles");
    constexpr int WAVES_N = BN >= 64 ? 2 : 1, WAVES_M = 4 / WAVES_N;
    constexpr int WM = kBM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
    constexpr int LDA = kBM + 4, LDB = BN + 4;
    constexpr int A_F4 = kBM * kBK / 4 / 256;                        // 2
    constexpr int B_F4 = (BN * kBK / 4 + 255) / 256;                 // 2, 1, 1 (half the threads at BN = 32)
    constexpr bool B_ALL = BN * kBK / 4 >= 256;
    constexpr int kLdw = (kBK + 8) / 2;                               // X3: dwords per LDS row (16 bf16 + 8 of padding = 48 bytes)
    constexpr int A_WORDS = X3 ? 2 * 3 * kBM * kLdw : 2 * kBK * LDA, B_WORDS = X3 ? 2 * 3 * BN * kLdw : 2 * kBK * LDB;
    __shared__ __attribute__((aligned(16))) uint32_t As_raw[A_WORDS];
    __shared__ __attribute__((aligned(16))) uint32_t Bs_raw[B_WORDS];
    float (*As)[kBK][LDA] = reinterpret_cast<float (*)[kBK][LDA]>(As_raw);             // fp32 form: [buf][k][row]
    float (*Bs)[kBK][LDB] = reinterpret_cast<float (*)[kBK][LDB]>(Bs_raw);
    uint32_t (*As3)[3][kBM][kLdw] = reinterpret_cast<uint32_t (*)[3][kBM][kLdw]>(As_raw);   // X3: [buf][plane][row][k pairs]
    uint32_t (*Bs3)[3][BN][kLdw] = reinterpret_cast<uint32_t (*)[3][BN][kLdw]>(Bs_raw);
    __shared__ unsigned long long c_ptr[kBM];   // global addresses of the tile's rows (zero row where there is none)
    __shared__ unsigned long long q_ptr[BN];
    __shared__ long long c_off[kBM];
    __shared__ long long q_off[BN];
    __shared__ __attribute__((aligned(16))) float c_nrm[kBM];
    __shared__ float q_nrm[BN];
    __shared__ uint32_t pairmax[L2MAX ? 256 : 1];
    __shared__ int c_len_s[16];                  // BOX: lengths of the tile's candidate documents (0 = none)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    // workgroups that share a candidate tile sit next to each other in an XCD's launch order (its L2 then serves
    // the tile's second and later readers): linear id = position within the XCD's contiguous share
    uint32_t L;
    {
        const uint32_t nb = gridDim.x, b = blockIdx.x, x = b & 7, q8 = nb >> 3, r8 = nb & 7;
        L = x * q8 + (x < r8 ? x : r8) + (b >> 3);
    }
    const uint32_t ct = L / (uint32_t)g.n_qt, qt = L - ct * (uint32_t)g.n_qt;
    const unsigned long long zrow = (unsigned long long)(uintptr_t)&g_zero_row[0];

    // ---- tile tables -----------------------------------------------------------------------------------
    if (tid < kBM) {
        const int d = tid / g.mr_c, i = tid - d * g.mr_c;
        const uint32_t c_loc = ct * g.dpt_c + d;
        const bool doc_ok = d < g.dpt_c && c_loc < g.ncand;
        int len = 0, start = 0;
        if (doc_ok) {
            len = g.c.len[g.cand0 + c_loc];
            start = g.c.start[g.cand0 + c_loc];
        }
        c_ptr[tid] = (doc_ok && i < len) ? (unsigned long long)(uintptr_t)(g.c.rows + (size_t)(start + i) * kD) : zrow;
        c_off[tid] = !doc_ok ? -1 : L2MAX ? (long long)d : (long long)c_loc * g.E + i;
    } else if (tid < kBM + BN) {
        const int r = tid - kBM;
        const int d = r / g.mr_q, i = r - d * g.mr_q;
        const uint32_t q_loc = qt * g.dpt_q + d;
        const bool doc_ok = d < g.dpt_q && q_loc < g.nq;
        int len = 0, start = 0;
        if (doc_ok) {
            len = g.q.len[q_loc];
            start = g.q.start[q_loc];
        }
        q_ptr[r] = (doc_ok && i < len) ? (unsigned long long)(uintptr_t)(g.q.rows + (size_t)(start + i) * kD) : zrow;
        q_off[r] = !doc_ok ? -1 : L2MAX ? (long long)d : ((long long)q_loc * g.ncand) * g.E + (long long)i * g.ld;
    }
    if constexpr (BOX) {
        if (tid >= kBM + BN && tid < kBM + BN + 16) {   // every slot of the table is written (0 = no document)
            const int d = tid - (kBM + BN);
            const uint32_t c_loc = ct * g.dpt_c + d;
            c_len_s[d] = (d < g.dpt_c && c_loc < g.ncand) ? g.c.len[g.cand0 + c_loc] : 0;
        }
    }
    if constexpr (L2MAX) pairmax[tid] = 0u;
    __syncthreads();

    // ---- operand staging: thread -> (row, 16-byte k chunk) ---------------------------------------------
    const int lrow = tid >> 2, lk4 = tid & 3;
    unsigned long long pa[A_F4], pb[B_F4];
#pragma unroll
    for (int p = 0; p < A_F4; ++p) pa[p] = c_ptr[lrow + 64 * p];
#pragma unroll
    for (int p = 0; p < B_F4; ++p) pb[p] = q_ptr[(lrow + 64 * p) % BN];
    // Two register sets: while tile t is multiplied out of LDS, tile t+1 sits in one set (landed or landing) and
    // tile t+2's loads go out into the other -- candidate rows come from HBM, and one iteration of MFMAs
    // (~1 us) does not cover that latency, two do.
    float4 ra[2][A_F4], rb[2][B_F4];
    // Rows that share a large common component (anisotropic embeddings): x.y, |x|^2 and |y|^2 are then all dominated by it, the
    // expansion cancels for nearly EVERY entry and the direct-formula redo below takes the kernel over (128 x 16 384 x 12: 86 ms
    // instead of 9).  Distances do not move when every row of the tile is shifted by the same vector: with g.center the tile's
    // first query row comes off each staged float4 before it is squared or split (one more L1-resident load per k step).
    const unsigned long long pmu = q_ptr[0];
    float4 rm[2];
    rm[0] = rm[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto sub4 = [](const float4& v, const float4& m) { return make_float4(v.x - m.x, v.y - m.y, v.z - m.z, v.w - m.w); };
    float na[A_F4], nb[B_F4];
#pragma unroll
    for (int p = 0; p < A_F4; ++p) na[p] = 0.f;
#pragma unroll
    for (int p = 0; p < B_F4; ++p) nb[p] = 0.f;
    auto load_tiles = [&](auto setc, int k0) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int p = 0; p < A_F4; ++p) ra[S][p] = ldg4(pa[p], k0 + 4 * lk4);
#pragma unroll
        for (int p = 0; p < B_F4; ++p) rb[S][p] = ldg4(pb[p], k0 + 4 * lk4);
        if (g.center) rm[S] = ldg4(pmu, k0 + 4 * lk4);
    };
    // one float4 piece (A piece p < A_F4, B piece p - A_F4 otherwise) of a register set -> LDS, k-major
    auto split_store = [&](uint32_t (*dst)[kLdw], uint32_t (*dst1)[kLdw], uint32_t (*dst2)[kLdw], int row, const float4& v) {
        uint32_t a1, a2, a3, b1, b2, b3;
        split3_bf16(v.x, v.y, a1, a2, a3);
        split3_bf16(v.z, v.w, b1, b2, b3);
        *reinterpret_cast<uint2*>(&dst[row][2 * lk4]) = make_uint2(a1, b1);
        *reinterpret_cast<uint2*>(&dst1[row][2 * lk4]) = make_uint2(a2, b2);
        *reinterpret_cast<uint2*>(&dst2[row][2 * lk4]) = make_uint2(a3, b3);
    };
    // X3: one register set -> the three bf16 planes of LDS buffer `buf`; `count` = 1.f the first time a tile is stored
    // (its squared norms are accumulated), 0.f for the unread repeat after the last tile
    auto store_set3 = [&](auto setc, int buf, float count) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int p = 0; p < A_F4; ++p) {
            const float4 v = sub4(ra[S][p], rm[S]);
            split_store(As3[buf][0], As3[buf][1], As3[buf][2], lrow + 64 * p, v);
            na[p] = fmaf(count, sq4f(v), na[p]);
        }
#pragma unroll
        for (int p = 0; p < B_F4; ++p) {
            const float4 v = sub4(rb[S][p], rm[S]);
            split_store(Bs3[buf][0], Bs3[buf][1], Bs3[buf][2], lrow + 64 * p, v);
            nb[p] = fmaf(count, sq4f(v), nb[p]);
        }
    };
    auto store_piece = [&](auto setc, int piece, int buf) {
        constexpr int S = decltype(setc)::value;
        if (piece < A_F4) {
            const int p = piece, row = lrow + 64 * p;
            const float4 v = sub4(ra[S][p], rm[S]);
            As[buf][4 * lk4 + 0][row] = v.x;
            As[buf][4 * lk4 + 1][row] = v.y;
            As[buf][4 * lk4 + 2][row] = v.z;
            As[buf][4 * lk4 + 3][row] = v.w;
            na[p] += sq4f(v);
        } else if (piece < A_F4 + B_F4) {
            const int p = piece - A_F4;
            if (B_ALL || lrow < BN) {
                const int row = lrow + 64 * p;
                const float4 v = sub4(rb[S][p], rm[S]);
                Bs[buf][4 * lk4 + 0][row] = v.x;
                Bs[buf][4 * lk4 + 1][row] = v.y;
                Bs[buf][4 * lk4 + 2][row] = v.z;
                Bs[buf][4 * lk4 + 3][row] = v.w;
                nb[p] += sq4f(v);
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lr = lane & 31, lk = lane >> 5;
    // One tile's worth of MFMAs out of LDS buffer `buf`, with the iteration's other work threaded between the
    // eight k-steps (pinned by sched_barrier) instead of bunched before and after them: next-but-one tile's global
    // loads after step 0, the next tile's register -> LDS stores (into the other buffer, which nobody reads until
    // the barrier) after steps 2..5.  Bunched, the two workgroups that share a CU fall into step -- both in their
    // MFMA phase, then both out of it -- and the matrix pipe idles half the time (measured 42 % busy).
    // BOX: geomloss's per-pair diameter formed beside the MFMAs.  Thread (cd, bk) = (tid >> 4, tid & 15) scans
    // coordinate bk of the staged tile over candidate document cd's rows (min / max), widens it with each query
    // document's precomputed box for that coordinate, and accumulates the squared extent per query document; the 16
    // lanes of a document (one DPP row) add up at the end.  Replaces a second pass over every candidate row
    // (doc_box_range_kernel + pair_box_kernel: 276 us beside a 217 us Gram kernel at 1 x 20 000 x 12).
    constexpr int kBoxQ = 8;                       // query documents a BN <= 64 tile can hold
    const int cd = tid >> 4, bk = tid & 15;
    const int nq_tile = BOX ? (int)min((uint32_t)g.dpt_q, g.nq) : 0;
    float bacc[kBoxQ];
#pragma unroll
    for (int u = 0; u < kBoxQ; ++u) bacc[u] = 0.f;
    auto tile_step = [&](auto load_set, auto store_set, int buf, bool do_load, int k_load, bool do_store, int k_cur) {
        float qmn[kBoxQ], qmx[kBoxQ];
        float a[2][TM], b[2][TN];   // operands of k-step kk+1 are read while step kk's MFMAs run
        auto read_operands = [&](int kk, int slot) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[slot][i] = As[buf][kk + (kBK / 2) * lk][wr * WM + 32 * i + lr];   // k rows kk | kk+8: half-waves on opposite bank halves
#pragma unroll
            for (int j = 0; j < TN; ++j) b[slot][j] = Bs[buf][kk + (kBK / 2) * lk][wc * WN + 32 * j + lr];
        };
        read_operands(0, 0);
#pragma unroll
        for (int kk = 0; kk < kBK / 2; ++kk) {
            if (kk + 1 < kBK / 2) read_operands(kk + 1, (kk + 1) & 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk & 1][i], b[kk & 1][j], acc[i][j], 0, 0, 0);
            if constexpr (BOX) {
                // issued BEFORE the tile loads: vmcnt retires in order, so waiting for these (at step 6) must not
                // imply waiting for the next-but-one tile's rows
                if (kk == 0) {
#pragma unroll
                    for (int u = 0; u < kBoxQ; ++u)
                        if (u < nq_tile) {
                            qmn[u] = g.qbox[(size_t)u * 2 * kD + k_cur + bk];
                            qmx[u] = g.qbox[(size_t)u * 2 * kD + kD + k_cur + bk];
                        }
                    if (g.center) {                // the queries' boxes move with their rows
                        const float mub = ldg1(pmu, k_cur + bk);
#pragma unroll
                        for (int u = 0; u < kBoxQ; ++u) {
                            qmn[u] -= mub;
                            qmx[u] -= mub;
                        }
                    }
                }
            }
            if (kk == 0 && do_load) load_tiles(load_set, k_load);
            if constexpr (BOX) {
                if (kk == 6) {
                    const int len = min(c_len_s[cd], g.mr_c);
                    if (len > 0) {
                        float mn = INFINITY, mx = -INFINITY;
                        const float* col = &As[buf][bk][cd * g.mr_c];
                        int r4 = 0;
                        for (; r4 + 4 <= len; r4 += 4) {          // whole groups of four rows: no predicates, min3 / max3
                            const float4 v = *reinterpret_cast<const float4*>(col + r4);
                            mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w)));
                            mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
                        }
                        if (r4 < len) {                            // ragged tail (rows past len are zero rows: masked)
                            const float4 v = *reinterpret_cast<const float4*>(col + r4);
                            mn = fminf(mn, v.x); mx = fmaxf(mx, v.x);
                            if (r4 + 1 < len) { mn = fminf(mn, v.y); mx = fmaxf(mx, v.y); }
                            if (r4 + 2 < len) { mn = fminf(mn, v.z); mx = fmaxf(mx, v.z); }
                        }
#pragma unroll
                        for (int u = 0; u < kBoxQ; ++u)
                            if (u < nq_tile) {
                                const float ext = fmaxf(mx, qmx[u]) - fminf(mn, qmn[u]);
                                bacc[u] = fmaf(ext, ext, bacc[u]);
                            }
                    }
                }
            }
            if (kk >= 2 && kk < 6 && do_store) store_piece(store_set, kk - 2, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // X3: one tile = 12 fragment reads (lane = row lr, k half lk: eight consecutive k of a row per plane; A and B use the same
    // (lane half, element) -> k map) and 6 x TM x TN MFMAs, smallest terms first, the accumulators taking turns inside a term;
    // the next-but-one tile's loads go out first, the split + LDS stores of the next tile ride in the MFMAs' shadow.  The
    // body is branch free (past the end the last tile is loaded / stored again, unread) so that it schedules as one block.
    auto tile_step3 = [&](auto load_set, auto store_set, int buf, int k_load, float count) {
        load_tiles(load_set, k_load);
        bf16x8_t af[TM][3], bfr[TN][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i][pl] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(&As3[buf][pl][wr * WM + 32 * i + lr][4 * lk]));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j][pl] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(&Bs3[buf][pl][wc * WN + 32 * j + lr][4 * lk]));
        }
        store_set3(store_set, buf ^ 1, count);
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[term]], bfr[j][PB[term]], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, A_F4 + B_F4, 0);
#pragma unroll
        for (int m = 0; m < 6 * TM * TN; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    constexpr int nk = kD / kBK;
    static_assert(nk % 2 == 0, "the loop below is unrolled by two");
    static_assert(A_F4 + B_F4 <= 4, "four store slots per tile step");
    load_tiles(S0{}, 0);
    load_tiles(S1{}, kBK);
    if constexpr (X3) {
        store_set3(S0{}, 0, 1.f);
        __syncthreads();
#pragma unroll 1
        for (int t = 0; t < nk; t += 2) {
            // LDS buffer 0 = tile t, set 1 = tile t+1 (stored into buffer 1 here), set 0 free (tile t+2 loads into it)
            tile_step3(S0{}, S1{}, 0, min(t + 2, nk - 1) * kBK, 1.f);
            __syncthreads();
            // LDS buffer 1 = tile t+1, set 0 = tile t+2, set 1 free
            tile_step3(S1{}, S0{}, 1, min(t + 3, nk - 1) * kBK, t + 2 < nk ? 1.f : 0.f);
            __syncthreads();
        }
    } else {
#pragma unroll
    for (int piece = 0; piece < A_F4 + B_F4; ++piece) store_piece(S0{}, piece, 0);
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < nk; t += 2) {
        // LDS buffer 0 = tile t, set 1 = tile t+1 (stored into buffer 1 here), set 0 free (tile t+2 loads into it)
        tile_step(S0{}, S1{}, 0, t + 2 < nk, (t + 2) * kBK, true, t * kBK);
        __syncthreads();
        // LDS buffer 1 = tile t+1, set 0 = tile t+2, set 1 free
        tile_step(S1{}, S0{}, 1, t + 3 < nk, (t + 3) * kBK, t + 2 < nk, (t + 1) * kBK);
        __syncthreads();
    }
    }

    if constexpr (BOX) {
        const uint32_t c_loc = ct * g.dpt_c + cd;
#pragma unroll
        for (int u = 0; u < kBoxQ; ++u) {
            float sacc = bacc[u];
            sacc += lane_xor<1>(sacc);
            sacc += lane_xor<2>(sacc);
            sacc += lane_xor<4>(sacc);
            sacc += lane_xor<8>(sacc);
            if (bk == 0 && u < nq_tile && cd < g.dpt_c && c_loc < g.ncand) g.diam2[(size_t)u * g.ncand + c_loc] = sacc;
        }
    }
    // ---- squared norms: the 4 threads of a row hold its 4 interleaved k-chunk partial sums ----------------
#pragma unroll
    for (int p = 0; p < A_F4; ++p) {
        float s = na[p];
        s += lane_xor<1>(s);
        s += lane_xor<2>(s);
        if (lk4 == 0) c_nrm[lrow + 64 * p] = s;
    }
#pragma unroll
    for (int p = 0; p < B_F4; ++p) {
        float s = nb[p];
        s += lane_xor<1>(s);
        s += lane_xor<2>(s);
        if (lk4 == 0 && (B_ALL || lrow < BN)) q_nrm[lrow + 64 * p] = s;
    }
    __syncthreads();

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    // Entries whose expansion cancelled (d^2 tiny against the norms; only where torch.cdist would use the direct
    // formula) are not finished here: they go on a work list in LDS (the operand buffers are free now) and are
    // recomputed afterwards by WHOLE WAVES, one entry at a time, 12 coordinates per lane.  A lane-local recompute
    // would serialise 768 coordinates per flagged entry while its 63 wave mates wait: with real sentence vectors,
    // where a few percent of the pairs are that close, that was several times the cost of the tile's MFMAs.
    uint32_t* wlist = As_raw;
    constexpr int kCap = A_WORDS;                                 // (row << 16 | column) entries in As's footprint
    __shared__ uint32_t wl_count;
    if (tid == 0) wl_count = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = wc * WN + 32 * j + lr;
            const long long qo = q_off[n];
            const float xx = q_nrm[n];
            const unsigned long long qp = q_ptr[n];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int m0 = wr * WM + 32 * i + 8 * g4 + 4 * lk;
                const long long co = c_off[m0];
                if (qo < 0 || co < 0) continue;
                // (round 6: a cancelling entry is redone from the exact sum whatever formula torch.cdist would pick -- also beyond 25 rows: include/aspire_hip.h, SHARED SENTENCES)
                const float4 yy4 = *reinterpret_cast<const float4*>(&c_nrm[m0]);
                const float yy[4] = {yy4.x, yy4.y, yy4.z, yy4.w};
                float cost[4], neg[4];
                bool redo[4];
                bool any_redo = false;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float sq = fmaf(-2.f, acc[i][j][4 * g4 + k], xx) + yy[k];
                    cost[k] = sqrtf(fmaxf(sq, 1e-8f));
                    neg[k] = -sqrtf(fmaxf(sq, 0.f));
                    const float ns = xx + yy[k];
                    redo[k] = sq < kDirectTau * ns * ns && qp != zrow && c_ptr[m0 + k] != zrow;
                    if (redo[k]) {
                        const uint32_t slot = atomicAdd(&wl_count, 1u);
                        if (slot < (uint32_t)kCap) wlist[slot] = ((uint32_t)(m0 + k) << 16) | (uint32_t)n;
                        else {   // list full: lane-local
                            const float d2 = direct_d2(qp, c_ptr[m0 + k]);
                            neg[k] = -sqrtf(d2);
                            cost[k] = sqrtf(fmaxf(d2, 1e-8f));
                            redo[k] = false;
                        }
                    }
                    any_redo |= redo[k];
                }
                if constexpr (L2MAX) {
                    float best = -INFINITY;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (!redo[k] && qp != zrow && c_ptr[m0 + k] != zrow) best = fmaxf(best, neg[k]);
                    if (best > -INFINITY) atomicMax(&pairmax[(int)co * 16 + (int)qo], order_key(best));
                } else {
                    if (!any_redo) {
                        if (g.cost) *reinterpret_cast<float4*>(g.cost + qo + co) = make_float4(cost[0], cost[1], cost[2], cost[3]);
                        *reinterpret_cast<float4*>(g.neg + qo + co) = make_float4(neg[0], neg[1], neg[2], neg[3]);
                    } else {   // the work-list pass writes the flagged ones (-cdist AND geomloss's cost, from the same exact sum): no address is stored twice
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (!redo[k]) {
                                if (g.cost) g.cost[qo + co + k] = cost[k];
                                g.neg[qo + co + k] = neg[k];
                            }
                    }
                }
            }
        }
    __syncthreads();
    {
        // 16 lanes (one DPP row) per entry, 48 coordinates per lane: 16 entries per workgroup in flight, 24
        // independent 16-byte loads per lane, and a 4-step in-row reduction
        const uint32_t n_redo = min(wl_count, (uint32_t)kCap);
        const int grp = tid >> 4, l16 = tid & 15;
        for (uint32_t e0 = 0; e0 < n_redo; e0 += 16) {
            const uint32_t e = e0 + grp;
            const bool live = e < n_redo;
            const uint32_t mn = wlist[live ? e : 0];
            const int m = (int)(mn >> 16), n = (int)(mn & 0xFFFFu);
            const unsigned long long xp = q_ptr[n], yp = c_ptr[m];
            float p0 = 0.f, p1 = 0.f;
#pragma unroll
            for (int c = 0; c < 12; c += 2) {
                const float4 u0 = ldg4(xp, 4 * l16 + 64 * c), v0 = ldg4(yp, 4 * l16 + 64 * c);
                const float4 u1 = ldg4(xp, 4 * l16 + 64 * c + 64), v1 = ldg4(yp, 4 * l16 + 64 * c + 64);
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
            const float negd = -sqrtf(part);
            if (live && l16 == 0) {
                if constexpr (L2MAX) atomicMax(&pairmax[(int)c_off[m] * 16 + (int)q_off[n]], order_key(negd));
                else {
                    g.neg[q_off[n] + c_off[m]] = negd;
                    if (g.cost) g.cost[q_off[n] + c_off[m]] = sqrtf(fmaxf(part, 1e-8f));
                }
            }
        }
    }
    if constexpr (L2MAX) {
        __syncthreads();
        const int cd = tid >> 4, qd = tid & 15;
        const uint32_t c_loc = ct * g.dpt_c + cd, q_loc = qt * g.dpt_q + qd;
        if (cd < g.dpt_c && qd < g.dpt_q && c_loc < g.ncand && q_loc < g.nq) {
            const uint32_t key = pairmax[cd * 16 + qd];
            g.scores[(int64_t)q_loc * g.c.n + g.cand0 + c_loc] = key ? unorder_key(key) : -INFINITY;
        }
    }
}

// Per-coordinate bounding box of documents [first, first + gridDim.x): box[k][0][768] = min, box[k][1][768] = max
__global__ void __launch_bounds__(192) doc_box_range_kernel(RepSet d, int64_t first, float* __restrict__ box) {
    const int64_t k = blockIdx.x;
    const int n = d.len[first + k];
    const float* doc = d.rows + (size_t)d.start[first + k] * kD + threadIdx.x * 4;
    float4 mn, mx;
    doc_box_chunk(doc, n, mn, mx);
    *reinterpret_cast<float4*>(box + k * 2 * kD + threadIdx.x * 4) = mn;
    *reinterpret_cast<float4*>(box + k * 2 * kD + kD + threadIdx.x * 4) = mx;
}

// geomloss's diameter of one (query, candidate) call = norm of the joint bounding box's extent; squared here:
// diam2[q_loc * ncand + c_loc] = sum_d (max(qmax_d, cmax_d) - min(qmin_d, cmin_d))^2.
// Workgroup = 32 query docs x 32 candidate docs, thread = 2 x 2 of them, coordinates staged 64 at a time.
constexpr int kBoxLd = 68;
__global__ void __launch_bounds__(256) pair_box_kernel(const float* __restrict__ qbox, const float* __restrict__ cbox,
                                                       uint32_t nq, uint32_t ncand, float* __restrict__ diam2) {
    __shared__ __attribute__((aligned(16))) float s[4][32][kBoxLd];   // qmin, qmax, cmin, cmax
    const int tid = threadIdx.x;
    const uint32_t c0 = blockIdx.x * 32, q0 = blockIdx.y * 32;
    const int tq = tid & 15, tc = tid >> 4;
    // per (query, candidate) pair and coordinate: one v_max, one v_min (bare instructions: fmaxf / fminf would canonicalise all 32 values
    // read from LDS first), half a packed subtract and half a packed FMA -- 3 issue slots where the scalar form took 6
    typedef float f2b __attribute__((ext_vector_type(2)));
    f2b acc[2][2] = {{f2b{0.f, 0.f}, f2b{0.f, 0.f}}, {f2b{0.f, 0.f}, f2b{0.f, 0.f}}};
    auto vmx = [](float a, float b) {
        float r;
        asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    };
    auto vmn = [](float a, float b) {
        float r;
        asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    };
    for (int ch = 0; ch < kD / 64; ++ch) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int idx = tid + 256 * p;            // [arr 4][doc 32][f4 16]
            const int arr = idx >> 9, doc = (idx >> 4) & 31, f4 = idx & 15;
            const bool isq = arr < 2;
            const uint32_t gdoc = isq ? min(q0 + doc, nq - 1) : min(c0 + doc, ncand - 1);
            const float* src = (isq ? qbox : cbox) + (size_t)gdoc * 2 * kD + (arr & 1) * kD + ch * 64 + f4 * 4;
            *reinterpret_cast<float4*>(&s[arr][doc][f4 * 4]) = ld4(src);
        }
        __syncthreads();
#pragma unroll 4
        for (int f4 = 0; f4 < 16; ++f4) {
            float4 qn[2], qx[2], cn[2], cx[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                qn[u] = *reinterpret_cast<const float4*>(&s[0][2 * tq + u][f4 * 4]);
                qx[u] = *reinterpret_cast<const float4*>(&s[1][2 * tq + u][f4 * 4]);
                cn[u] = *reinterpret_cast<const float4*>(&s[2][2 * tc + u][f4 * 4]);
                cx[u] = *reinterpret_cast<const float4*>(&s[3][2 * tc + u][f4 * 4]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const f2b hi0 = {vmx(qx[u].x, cx[v].x), vmx(qx[u].y, cx[v].y)}, lo0 = {vmn(qn[u].x, cn[v].x), vmn(qn[u].y, cn[v].y)};
                    const f2b hi1 = {vmx(qx[u].z, cx[v].z), vmx(qx[u].w, cx[v].w)}, lo1 = {vmn(qn[u].z, cn[v].z), vmn(qn[u].w, cn[v].w)};
                    const f2b d0 = hi0 - lo0, d1 = hi1 - lo1;
                    acc[u][v] = __builtin_elementwise_fma(d1, d1, __builtin_elementwise_fma(d0, d0, acc[u][v]));
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const uint32_t q = q0 + 2 * tq + u, c = c0 + 2 * tc + v;
            if (q < nq && c < ncand) diam2[(size_t)q * ncand + c] = acc[u][v].x + acc[u][v].y;
        }
}

// The same for FEW queries (nq <= 8): the general kernel's 32 x 32 document blocks would run 1/32 .. 1/4 full and re-stage every
// candidate box through LDS (120 us at 1 .. 4 x 20 000).  Here the queries' boxes sit in LDS (6 KB each), a wave walks candidates --
// six coalesced 1 KB loads per candidate box, the lane's twelve coordinates against every query -- and reduces over its lanes.
constexpr int kBoxFewQ = 8;
__global__ void __launch_bounds__(256) pair_box_few_kernel(const float* __restrict__ qbox, const float* __restrict__ cbox, uint32_t nq,
                                                           uint32_t ncand, float* __restrict__ diam2) {
    extern __shared__ __attribute__((aligned(16))) float qs[];      // [nq][2][768]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t i = tid; i < nq * 2 * kD / 4; i += 256) reinterpret_cast<float4*>(qs)[i] = reinterpret_cast<const float4*>(qbox)[i];
    __syncthreads();
    for (uint32_t c = blockIdx.x * 4 + wave; c < ncand; c += gridDim.x * 4) {
        const float* cb = cbox + (size_t)c * 2 * kD + 4 * lane;
        float4 cn[3], cx[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            cn[t] = ld4_stream(cb + 256 * t);
            cx[t] = ld4_stream(cb + kD + 256 * t);
        }
        for (uint32_t u = 0; u < nq; ++u) {
            const float* qb = qs + (size_t)u * 2 * kD + 4 * lane;
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const float4 qn = *reinterpret_cast<const float4*>(qb + 256 * t), qx = *reinterpret_cast<const float4*>(qb + kD + 256 * t);
                const float dx = fmaxf(qx.x, cx[t].x) - fminf(qn.x, cn[t].x);
                const float dy = fmaxf(qx.y, cx[t].y) - fminf(qn.y, cn[t].y);
                const float dz = fmaxf(qx.z, cx[t].z) - fminf(qn.z, cn[t].z);
                const float dw = fmaxf(qx.w, cx[t].w) - fminf(qn.w, cn[t].w);
                acc = fmaf(dw, dw, fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, acc))));
            }
            acc = wave_sum(acc);
            if (lane == 0) diam2[(size_t)u * ncand + c] = acc;
        }
    }
}

int launch_pair_box(const float* qbox, const float* cbox, uint32_t nq, uint32_t ncand, float* diam2, hipStream_t stream) {
    if (nq <= (uint32_t)kBoxFewQ) {
        const uint32_t wgs = (ncand + 3) / 4 < 2048 ? (ncand + 3) / 4 : 2048;
        hipLaunchKernelGGL(pair_box_few_kernel, dim3(wgs), dim3(256), nq * 2 * kD * sizeof(float), stream, qbox, cbox, nq, ncand, diam2);
    } else {
        hipLaunchKernelGGL(pair_box_kernel, dim3((ncand + 31) / 32, (nq + 31) / 32), dim3(256), 0, stream, qbox, cbox, nq, ncand, diam2);
    }
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}

int slot_rows(int max_len) {
    const int r = (max_len + 3) / 4 * 4;
    return r < 8 ? 8 : r;
}

int fill_geometry(GramArgs& g, const ScoreArgs& a, int mr_q, int mr_c, int& bn) {
    g.q = a.q;
    g.c = a.c;
    g.cand0 = a.cand0;
    g.ncand = (uint32_t)(a.cand1 - a.cand0);
    g.nq = (uint32_t)a.q.n;
    g.mr_q = slot_rows(mr_q);
    g.mr_c = slot_rows(mr_c);
    const int64_t qrows = (int64_t)g.nq * g.mr_q;
    bn = qrows <= 32 ? 32 : qrows <= 64 ? 64 : 128;
    g.dpt_c = kBM / g.mr_c;
    g.dpt_q = bn / g.mr_q;
    g.n_ct = (int)((g.ncand + g.dpt_c - 1) / g.dpt_c);
    g.n_qt = (int)((g.nq + g.dpt_q - 1) / g.dpt_q);
    g.cdist_mode = a.cdist_mode;
    ASPIRE_REQUIRE((int64_t)g.n_ct * g.n_qt < ((int64_t)1 << 31), ASPIRE_ERR_UNSUPPORTED, "too many tiles in one launch");
    return ASPIRE_OK;
}

// the fp16-plane tiles (gramp.hip) of a call whose queries fill more than 64 rows.  128 x 128 by default: the 128 x 256 and
// 256 x 256 forms move fewer bytes per product and hold a higher clock, but end at the same matrix-pipe rate (the kernel is power
// bound: NOTES.md, round 4) and lose on calls with few query rows or few candidate tiles
GramGeometry planes_geometry(const GramArgs& g) {
    const int t = tuning().gram_tile;
    const int bn = t ? t % 1000 : 128, bm = t ? t / 1000 : 128;
    const int dpt_q = bn / g.mr_q, dpt_c = bm / g.mr_c;
    return GramGeometry{g.mr_q, g.mr_c, dpt_q, dpt_c, (int)((g.nq + dpt_q - 1) / dpt_q), (int)((g.ncand + dpt_c - 1) / dpt_c), g.E, g.ld, bm, bn};
}

template <bool L2MAX>
int launch_gram(const GramArgs& g, int bn, hipStream_t stream) {
    const dim3 grid((unsigned)(g.n_ct * g.n_qt));
    const bool box = g.diam2 != nullptr;
    if constexpr (L2MAX) {
        if (bn == 32) hipLaunchKernelGGL((pair_gram_kernel<32, true, false>), grid, dim3(256), 0, stream, g);
        else if (bn == 64) hipLaunchKernelGGL((pair_gram_kernel<64, true, false>), grid, dim3(256), 0, stream, g);
        else if (tuning().gemm_form == 1) hipLaunchKernelGGL((pair_gram_kernel<128, true, false>), grid, dim3(256), 0, stream, g);
        else hipLaunchKernelGGL((pair_gram_kernel<128, true, false, true>), grid, dim3(256), 0, stream, g);
    } else {
        if (bn == 32 && box) hipLaunchKernelGGL((pair_gram_kernel<32, false, true>), grid, dim3(256), 0, stream, g);
        else if (bn == 32) hipLaunchKernelGGL((pair_gram_kernel<32, false, false>), grid, dim3(256), 0, stream, g);
        else if (bn == 64 && box) hipLaunchKernelGGL((pair_gram_kernel<64, false, true>), grid, dim3(256), 0, stream, g);
        else if (bn == 64) hipLaunchKernelGGL((pair_gram_kernel<64, false, false>), grid, dim3(256), 0, stream, g);
        else if (tuning().gemm_form == 1) hipLaunchKernelGGL((pair_gram_kernel<128, false, false>), grid, dim3(256), 0, stream, g);
        else hipLaunchKernelGGL((pair_gram_kernel<128, false, false, true>), grid, dim3(256), 0, stream, g);
    }
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}

}  // namespace

// The matrix-core form serves CSR inputs (no padded extents) scored all-against-all, once the query side fills
// a useful part of an MFMA tile or documents are longer than the 8-row VALU tile.
// Max-sim on the fp16-plane tiles whatever the number of queries: both rep sets carry planes around one centre, the pool fills the
// chip (>= 128 candidate tiles), and it is not the one case the streaming kernel wins (ONE query of <= 8 rows against documents of
// <= 8: the fused kernel's max-sim form reads each row once at 6 TB/s and has no wasted columns -- 81 against 95 us at 1 x 20 000 x 8)
bool gram_planes_wanted_l2max(const aspire_repset* q, const aspire_repset* c, int pairing) {
    if (pairing != ASPIRE_PAIR_CROSS || q->ext != 0 || c->ext != 0 || !q->planes || !c->planes) return false;
    if (q->max_len <= 0 || c->max_len <= 0 || q->max_len > 32 || c->max_len > 32) return false;
    if (tuning().cost_path == 2 || tuning().gemm_form == 1 || tuning().gemm_form == 2) return false;
    if (q->planes->mu != c->planes->mu || !q->planes->planes || !c->planes->planes) return false;
    const int64_t tiles = (c->n + kBM / slot_rows(c->max_len) - 1) / (kBM / slot_rows(c->max_len));
    if (tiles < 128) return false;
    return !(q->n == 1 && q->max_len <= 8 && c->max_len <= 8);
}

// otAspire's cost stage on the plane tiles with FEW queries too: as above, and the pairs' diameters must not cost a pass over every
// candidate row -- the pool brings its documents' boxes along (aspire_repset.doc_box) or the caller its own diameters.  Two to eight
// queries against 20 000 documents of 8 rows then take cost tiles (~90 us) + pair_box + the block Sinkhorn kernel instead of the
// fused kernel re-staging every candidate group per query (Q = 4: 335 us, Q = 8: 626).  ONE query of <= 8 rows stays on the fused /
// chunk kernels.
bool gram_planes_wanted_ot(const aspire_repset* q, const aspire_repset* c, int pairing, bool caller_diameters) {
    if (!gram_planes_wanted_l2max(q, c, pairing)) return false;
    if (!c->doc_box && !caller_diameters) return false;
    return !(q->n == 1 && q->max_len <= 8);
}

bool gram_path_wanted(const aspire_repset* q, const aspire_repset* c, int pairing) {
    if (pairing != ASPIRE_PAIR_CROSS || q->ext != 0 || c->ext != 0) return false;
    if (q->max_len <= 0 || c->max_len <= 0 || q->max_len > 32 || c->max_len > 32) return false;
    // ASPIRE_HIP_COST_PATH=mfma|valu pins the choice (parity tests compare the two forms); default: by shape
    if (tuning().cost_path == 1) return true;
    if (tuning().cost_path == 2) return false;
    const int max_rows = q->max_len > c->max_len ? q->max_len : c->max_len;
    const int64_t qrows = q->n * (int64_t)slot_rows(q->max_len);
    const int64_t tiles = (c->n + kBM / slot_rows(c->max_len) - 1) / (kBM / slot_rows(c->max_len));
    if (max_rows > 8) {
        // long documents: the VALU tile-loop kernel costs ~5 ns per (pair x 8x8 tile), the Gram kernel ~75 us of
        // latency (48 dependent K stages) before its throughput counts: 1 x 3000 x 12 is 82 vs 88 us, 1 x 4000 x 20
        // 194 vs 120 us
        const int64_t T = (max_rows + 7) / 8;
        return tiles >= 32 && q->n * c->n * T * T >= 16000;
    }
    if (tiles < 128) return false;          // small pools are latency bound: the per-pair kernels start faster
    return qrows >= 24;
}

size_t gram_extra_bytes_per_cand(void) { return (size_t)2 * kD * sizeof(float); }   // the candidate's box

int launch_pair_gram_ot(const ScoreArgs& a, int T, int mr_q, int mr_c, float* cost, float* neg, float* diam2, float* qbox,
                        float* cbox, hipStream_t stream) {
    GramArgs g{};
    int bn = 128;
    if (int rc = fill_geometry(g, a, mr_q, mr_c, bn)) return rc;
    g.E = 64 * T * T;
    g.center = a.center;
    g.ld = 8 * T;
    g.cost = a.cost_from_neg ? nullptr : cost;      // (ScoreArgs::cost_from_neg: the solve stage derives geomloss's cost from the -cdist tile)
    g.neg = neg;
    if (gram_planes_ok(a) && bn < 128 && (a.c_box || !diam2)) {
        // few queries on a plane pool (gram_planes_wanted_ot): the 128-column plane tiles, diameters from the cached boxes
        if (diam2 && a.cand0 == 0) {
            hipLaunchKernelGGL(doc_box_range_kernel, dim3((unsigned)g.nq), dim3(192), 0, stream, a.q, (int64_t)0, qbox);
            ASPIRE_LAUNCH_OK();
        }
        if (int rc = launch_pair_gram_planes(a, planes_geometry(g), false, cost, neg, stream)) return rc;
        if (diam2) return launch_pair_box(qbox, a.c_box + (size_t)a.cand0 * 2 * kD, g.nq, g.ncand, diam2, stream);
        return ASPIRE_OK;
    }
    if (diam2 && a.cand0 == 0) {   // per-coordinate boxes of the queries, once per call
        hipLaunchKernelGGL(doc_box_range_kernel, dim3((unsigned)g.nq), dim3(192), 0, stream, a.q, (int64_t)0, qbox);
        ASPIRE_LAUNCH_OK();
    }
    if (diam2 && bn <= 64) {       // few queries: the diameter is formed inside the Gram kernel
        g.qbox = qbox;
        g.diam2 = diam2;
        return launch_gram<false>(g, bn, stream);
    }
    if (bn == 128 && gram_planes_ok(a)) {
        if (int rc = launch_pair_gram_planes(a, planes_geometry(g), false, cost, neg, stream)) return rc;
    } else if (int rc = launch_gram<false>(g, bn, stream)) return rc;
    if (diam2) {
        if (a.c_box) {        // a resident pool brought its documents' boxes along (aspire_repset.doc_box)
            cbox = const_cast<float*>(a.c_box) + (size_t)a.cand0 * 2 * kD;
        } else {
            hipLaunchKernelGGL(doc_box_range_kernel, dim3(g.ncand), dim3(192), 0, stream, a.c, a.cand0, cbox);
            ASPIRE_LAUNCH_OK();
        }
        if (int rc = launch_pair_box(qbox, cbox, g.nq, g.ncand, diam2, stream)) return rc;
    }
    return ASPIRE_OK;
}

int launch_pair_gram_l2max(const ScoreArgs& a, int mr_q, int mr_c, hipStream_t stream) {
    GramArgs g{};
    int bn = 128;
    ScoreArgs b = a;
    b.cand0 = 0;
    b.cand1 = a.c.n;
    if (int rc = fill_geometry(g, b, mr_q, mr_c, bn)) return rc;
    g.scores = a.scores;
    g.center = a.center;
    // (with few query rows too: most of the 128 query columns of a tile are then zero rows, and the tiles still run at the rate the
    // candidate planes stream in -- 1 x 20 000 x 12: 137 us = 5.4 TB/s against 192 on the 16-row streaming kernel)
    if (gram_planes_ok(b)) return launch_pair_gram_planes(b, planes_geometry(g), true, nullptr, nullptr, stream);
    return launch_gram<true>(g, bn, stream);
}

}  // namespace aspire

extern "C" int aspire_repset_boxes_f32(const aspire_repset* set, int64_t D, float* boxes, void* stream) {
    using namespace aspire;
    ASPIRE_REQUIRE(set && (boxes || set->n == 0), ASPIRE_ERR_INVALID_ARG, "null argument");
    ASPIRE_REQUIRE(D == kD, ASPIRE_ERR_UNSUPPORTED, "encoding dim must be %d", kD);
    if (set->n == 0) return ASPIRE_OK;
    ASPIRE_REQUIRE(set->n < ((int64_t)1 << 31), ASPIRE_ERR_UNSUPPORTED, "too many documents for one launch");
    const RepSet d{set->rows, set->start, set->len, set->n, set->ext};
    hipLaunchKernelGGL(doc_box_range_kernel, dim3((unsigned)set->n), dim3(192), 0, (hipStream_t)stream, d, (int64_t)0, boxes);
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}
