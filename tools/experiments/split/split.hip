// EXPERIMENT, NOT part of libaspire_hip.so since round 6 (tools/experiments/split/build.sh builds a variant library with it; there
// ASPIRE_HIP_FUSED_SPLIT=1 turns it on and tools/experiments/split/test_split.py pins its bits to the fused kernel's): round 5's attempt
// to take the Sinkhorn solves off the streaming waves.  On par with the fused kernel (106.5 - 109 us per 20 x 1000 call against 105 - 107); with the
// solves skipped (ASPIRE_HIP_SPLIT_PRIO=3) 91.5 -- the structure's floor.  NOTES.md, round 5 log; profiles/r05_split_*.
// otAspire throughput kernel, role-split form: the fused kernel's two halves on DIFFERENT waves of one workgroup (A5-A8; reference
// arithmetic: src/learning/facetid_models/pair_distances.py:21-92 + geomloss 0.2.4's sinkhorn_tensorized, restated -- fused_solve.h).
//
// Why.  pair_fused_kernel (fused.hip) lets the wave that streamed an item solve it, in slices inside the next item's stages: one
// wave carries the staging registers AND the solve state (239 VGPRs: two waves per SIMD), and its issue stream alternates between
// a stage's side products and a slice of ~7 epsilon steps.  Measured (NOTES.md round 4): the stream is then bound by the wave's own
// issue slots -- ~30 us per item whether the chip is full or not -- and a 20 x 1000 call takes 102 us where its cost phase alone
// takes 77.  Here the streaming waves keep only staging state (<= 168 VGPRs: THREE waves per SIMD), hand an item's finished 8 x 8
// blocks through an LDS ring to solver waves of the same workgroup, and go on streaming; the solver waves never wait for memory and
// fill the issue slots the streamers leave.
//
// Work decomposition (gfx950), documents of <= 8 sentence rows, CSR inputs, batches of <= 64 jobs (or ONE query against a pool):
//   * one workgroup of 12 waves per CU: waves 0 .. 7 stream (two per SIMD), waves 8 .. 11 solve (one per SIMD); a streamer that
//     runs out of items turns solver, so the launch's tail -- the last items' solves -- is spread over all twelve;
//   * a workgroup owns a contiguous run of items (item = four consecutive candidates of one job, as in fused.hip); its streamers
//     claim them one at a time from an LDS counter;
//   * the query's per-coordinate box (geomloss's diameter) is formed ONCE per workgroup and job by the solver waves while the
//     streamers' first loads are in flight (fused.hip's SELF form: every wave, from its staged query rows, during its first item);
//   * hand-over: the four pairs' 8 x 8 costs and -cdist entries row-major + 16 words (lengths, candidates, diam^2) into one of 12
//     ring slots; sequence numbers per slot (a bounded multi-producer / multi-consumer queue), polled with s_sleep.
#include <mutex>

#include "fused_solve.h"
#include "tuning.h"

namespace aspire {
namespace {

constexpr int kSpStreamers = 8;
constexpr int kSpSolvers = 4;
constexpr int kSpWaves = kSpStreamers + kSpSolvers;
constexpr int kSpJobs = 3;             // query boxes a workgroup keeps (its run of items rarely touches more jobs; beyond: formed in-wave)
constexpr int kSpSlots = 12;           // hand-over ring
constexpr int kSpItemWords = 2 * 256 + 16;      // a slot: cost [4][64] | neg [4][64] | meta: q_len, then per pair c_len | real << 8, c_idx, diam^2
// LDS, in floats: control words | boxes [kSpJobs][2][768] | ring [kSpSlots][kSpItemWords] | stage buffers [kSpStreamers][kWaveLds]
constexpr int kSpCtl = 64 + 128;        // 64 control words, then the jobs' candidate ranges [2][64]
constexpr int kSpBoxOfs = kSpCtl;
constexpr int kSpRingOfs = kSpBoxOfs + kSpJobs * 2 * kD;
constexpr int kSpStageOfs = kSpRingOfs + kSpSlots * kSpItemWords;
constexpr int kSpLdsFloats = kSpStageOfs + kSpStreamers * kWaveLds;
// control words
constexpr int kCtlClaim = 0, kCtlTail = 1, kCtlHead = 2, kCtlBox = 3, kCtlDone = 4, kCtlSeq = 8;
static_assert(kCtlSeq + kSpSlots <= 64, "control block");

#ifdef ASPIRE_PHASE_CLOCK
// debug build only (tools/splitphases.py): per-wave time stamps (100 MHz wall clock): [workgroup * 12 + wave][16] = start, then
// per streamed item its hand-over; [8] = the wave turns solver, [9 ..] = end of each solve (up to 6), [15] = end
static __device__ long long* g_sdbg = nullptr;
#define S_STAMP(k)                                                                                                   \
    do {                                                                                                             \
        if (g_sdbg && lane == 0 && (k) < 16) g_sdbg[(size_t)(blockIdx.x * kSpWaves + wave) * 16 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define S_STAMP(k) \
    do {           \
    } while (0)
#endif

__device__ __forceinline__ uint32_t lds_fetch_add(uint32_t* p) {
    return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t lds_load(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// SINGLE: ONE query (q.n == 1, CROSS pairing) against candidates [cand0, cand1) -- one job; else MAPPED jobs [job0, job1) of a batch.
template <bool SINGLE>
__global__ void __launch_bounds__(kSpWaves * 64) pair_split_kernel(ScoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    uint32_t* ctl = reinterpret_cast<uint32_t*>(lds_all);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // lane l holds job l's candidate range and the items (groups of four) up to and including it
    int jo0 = 0, jo1 = 0;
    if constexpr (SINGLE) {
        if (lane == 0) {
            jo0 = (int)a.cand0;
            jo1 = (int)a.cand1;
        }
    } else if (lane < a.job1 - a.job0) {
        jo0 = a.job_off[a.job0 + lane];
        jo1 = a.job_off[a.job0 + lane + 1];
    }
    int gend = (jo1 - jo0 + 3) >> 2;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int t = __shfl_up(gend, m);
        if (lane >= m) gend += t;
    }
    const uint32_t n_items = (uint32_t)__builtin_amdgcn_readlane(gend, 63);
    // this workgroup's run of items
    const uint32_t nb = gridDim.x, per = n_items / nb, rem = n_items % nb;
    const uint32_t lo = blockIdx.x * per + min((uint32_t)blockIdx.x, rem);
    const uint32_t cnt = per + (blockIdx.x < rem ? 1u : 0u);
    const int job_lo = __popcll(__ballot(gend <= (int)lo));              // job of the run's first item
    if (threadIdx.x < 64) {
        ctl[threadIdx.x] = (threadIdx.x >= kCtlSeq && threadIdx.x < kCtlSeq + kSpSlots) ? threadIdx.x - kCtlSeq : 0u;
        ctl[64 + threadIdx.x] = (uint32_t)jo0;           // (the ranges live in LDS: two registers less in every wave)
        ctl[128 + threadIdx.x] = (uint32_t)jo1;
    }
    __syncthreads();
    if (cnt == 0) return;
    S_STAMP(0);

    if (wave < kSpStreamers) {
        float* lds = lds_all + kSpStageOfs + wave * kWaveLds;
        float* nscr = lds + kNormOfs;
        // lane roles as in fused.hip: p = candidate of this lane (compute AND staging); (li, lj) = its 2 x 2 block of the 8 x 8 entries;
        // the 16 lanes of group p stage the 8 rows of candidate p and query rows 2 p, 2 p + 1, chunk sc each
        const int p = lane >> 4, lp = lane & 15, li = lp >> 2, lj = lp & 3;
        const int sg = p, sc = lp;
        // an item's documents, fetched one item AHEAD (the table round trips run under the current item's stages); the query's
        // part is wave-uniform and lives in scalar registers
        struct Ctx {
            int c_idx, c_start, c_lr;      // per lane group: candidate, its first row, its length | real << 8
            int q_len, q_start, job;       // wave-uniform
        };
        auto load_ctx = [&](uint32_t item) {
            Ctx x;
            const int job = SINGLE ? 0 : __popcll(__ballot(gend <= (int)item));
            const int g0 = job > 0 ? __builtin_amdgcn_readlane(gend, job - 1) : 0;
            const int cj0 = (int)ctl[64 + job], cj1 = (int)ctl[128 + job];
            const int first = cj0 + 4 * ((int)item - g0);
            const int cand = min(first + p, cj1 - 1);
            const int q_idx = SINGLE ? 0 : a.job0 + job;
            x.job = job;
            x.q_len = __builtin_amdgcn_readfirstlane(a.q.len[q_idx]);
            x.q_start = __builtin_amdgcn_readfirstlane(a.q.start[q_idx]);
            x.c_idx = cand;
            x.c_lr = a.c.len[cand] | (first + p < cj1 ? 256 : 0);
            x.c_start = a.c.start[cand];
            return x;
        };
        auto claim = [&]() {
            uint32_t t = 0;
            if (lane == 0) t = lds_fetch_add(&ctl[kCtlClaim]);
            return (uint32_t)__builtin_amdgcn_readfirstlane(t);
        };
        uint32_t cur_it = claim();
        Ctx next = load_ctx(lo + min(cur_it, cnt - 1));
        bool box_ready = false;
        int n_done = 0;
        (void)n_done;

        while (cur_it < cnt) {
            const Ctx cur = next;
            const uint32_t nxt_it = claim();
            next = load_ctx(lo + (nxt_it < cnt ? nxt_it : cur_it));          // (the last item fetches itself again)
            const int q_len = cur.q_len;
            const int bslot = cur.job - job_lo;
            const int c_idx = cur.c_idx, c_start = cur.c_start, c_len = cur.c_lr & 255;
            const bool real = (cur.c_lr & 256) != 0;
            const float* qdoc = a.q.rows + (size_t)cur.q_start * kD;
            const float* sy_doc = a.c.rows + (size_t)c_start * kD;
            const char* qbase = reinterpret_cast<const char*>(qdoc);
            {
            mfma4_t macc[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) macc[m] = mfma4_t{0.f, 0.f, 0.f, 0.f};
            f2_t ny[8], nx[2], dsq = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 8; ++k) ny[k] = f2_t{0.f, 0.f};
            nx[0] = nx[1] = f2_t{0.f, 0.f};
            float4 vy[8], vx[2];
            auto issue_loads = [&](int st) {
                const int dofs = (st * kCh + sc) * 4;
#pragma unroll
                for (int j = 0; j < 8; ++j) vy[j] = ld4_stream(sy_doc + (size_t)min(j, c_len - 1) * kD + dofs);   // pad rows: copies of the last row
#pragma unroll
                for (int k = 0; k < 2; ++k) vx[k] = ld4(qdoc + (size_t)min(2 * sg + k, q_len - 1) * kD + dofs);
            };
            auto sq_acc = [](f2_t acc, const float4& v) {
                acc = __builtin_elementwise_fma(f2_t{v.x, v.y}, f2_t{v.x, v.y}, acc);
                return __builtin_elementwise_fma(f2_t{v.z, v.w}, f2_t{v.z, v.w}, acc);
            };
            issue_loads(0);
            if (!box_ready) {
                box_ready = lds_load(&ctl[kCtlBox]) == (uint32_t)kSpSolvers;
                if (box_ready) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            const bool have_box = box_ready && bslot < kSpJobs;                 // wave-uniform
            const float* qcb = lds_all + kSpBoxOfs + (have_box ? bslot : 0) * 2 * kD;
#pragma unroll 1
            for (int st = 0; st < kStages; ++st) {
                float4 mu = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.center) {
                    mu = make_float4(vx[0].x + vx[1].x, vx[0].y + vx[1].y, vx[0].z + vx[1].z, vx[0].w + vx[1].w);
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
                }
                float4 mn, mx;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    ny[j] = sq_acc(ny[j], vy[j]);
                    *reinterpret_cast<float4*>(lds + (8 + sg * 8 + j) * kRowStride + sc * 4) = vy[j];
                }
                mn = vmin3_4(vy[0], vy[1], vy[2]); mx = vmax3_4(vy[0], vy[1], vy[2]);
                mn = vmin3_4(mn, vy[3], vy[4]); mx = vmax3_4(mx, vy[3], vy[4]);
                mn = vmin3_4(mn, vy[5], vy[6]); mx = vmax3_4(mx, vy[5], vy[6]);
                mn = vmin3_4(mn, vy[7], vy[7]); mx = vmax3_4(mx, vy[7], vy[7]);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    nx[k] = sq_acc(nx[k], vx[k]);
                    *reinterpret_cast<float4*>(lds + (2 * sg + k) * kRowStride + sc * 4) = vx[k];
                }
                asm volatile("" : "+v"(ny[0]), "+v"(ny[1]), "+v"(ny[2]), "+v"(ny[3]), "+v"(ny[4]), "+v"(ny[5]), "+v"(ny[6]), "+v"(ny[7]),
                                  "+v"(nx[0]), "+v"(nx[1]), "+v"(mn.x), "+v"(mn.y), "+v"(mn.z), "+v"(mn.w), "+v"(mx.x), "+v"(mx.y),
                                  "+v"(mx.z), "+v"(mx.w) : : "memory");
                if (st + 1 < kStages) issue_loads(st + 1);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                {
                    float4 qn, qx;
                    if (have_box) {
                        qn = *reinterpret_cast<const float4*>(qcb + (st * kCh + sc) * 4);
                        qx = *reinterpret_cast<const float4*>(qcb + kD + (st * kCh + sc) * 4);
                        qn.x -= mu.x; qn.y -= mu.y; qn.z -= mu.z; qn.w -= mu.w;
                        qx.x -= mu.x; qx.y -= mu.y; qx.z -= mu.z; qx.w -= mu.w;
                    } else {
                        qn = qx = *reinterpret_cast<const float4*>(lds + sc * 4);
#pragma unroll
                        for (int r = 1; r < 8; ++r) {
                            const float4 qv = *reinterpret_cast<const float4*>(lds + r * kRowStride + sc * 4);
                            qn.x = vmin1(qn.x, qv.x); qn.y = vmin1(qn.y, qv.y); qn.z = vmin1(qn.z, qv.z); qn.w = vmin1(qn.w, qv.w);
                            qx.x = vmax1(qx.x, qv.x); qx.y = vmax1(qx.y, qv.y); qx.z = vmax1(qx.z, qv.z); qx.w = vmax1(qx.w, qv.w);
                        }
                    }
                    const f2_t dlo = {vmax1(mx.x, qx.x) - vmin1(mn.x, qn.x), vmax1(mx.y, qx.y) - vmin1(mn.y, qn.y)};
                    const f2_t dhi = {vmax1(mx.z, qx.z) - vmin1(mn.z, qn.z), vmax1(mx.w, qx.w) - vmin1(mn.w, qn.w)};
                    dsq = __builtin_elementwise_fma(dhi, dhi, __builtin_elementwise_fma(dlo, dlo, dsq));
                }
                {
                    const int mb = lane >> 2, mt = lane & 3, miq = (mb >> 1) & 1, mjq = mb & 1;
                    const float* xm = lds + (4 * miq + mt) * kRowStride;
                    const float* ym = lds + (8 + p * 8 + 4 * mjq + mt) * kRowStride;
#pragma unroll 4
                    for (int c = 0; c < kCh; ++c) {
                        const float4 xa = *reinterpret_cast<const float4*>(xm + c * 4);
                        const float4 yb = *reinterpret_cast<const float4*>(ym + c * 4);
                        macc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa.x, yb.x, macc[0], 0, 0, 0);
                        macc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa.y, yb.y, macc[1], 0, 0, 0);
                        macc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa.z, yb.z, macc[2], 0, 0, 0);
                        macc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa.w, yb.w, macc[3], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the stage buffer is rewritten next
                __builtin_amdgcn_wave_barrier();
            }

            // block columns -> the solve's 2 x 2 layout, through the (idle) stage buffer: pair p's 8 x 8 entries row-major
            float accg[2][2];
            {
                const int mb = lane >> 2, mt = lane & 3, miq = (mb >> 1) & 1, mjq = mb & 1;
                float* tr = lds + p * 64;
#pragma unroll
                for (int r = 0; r < 4; ++r) tr[(4 * miq + r) * 8 + 4 * mjq + mt] = (macc[0][r] + macc[1][r]) + (macc[2][r] + macc[3][r]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    const float2 t2 = *reinterpret_cast<const float2*>(tr + (2 * li + x) * 8 + 2 * lj);
                    accg[x][0] = t2.x;
                    accg[x][1] = t2.y;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            // norms: sum the staging lanes' partials through the scratch table nscr[value][lane] (fused.hip)
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
            float diam2 = dsq.x + dsq.y;
            diam2 += lane_xor<1>(diam2); diam2 += lane_xor<2>(diam2); diam2 += lane_xor<4>(diam2); diam2 += lane_xor<8>(diam2);

            // finish the entries.  Only x.y was accumulated: -cdist comes from the same expansion as geomloss's cost, and the
            // entries where it cancels (torch.cdist's direct formula differs there) are redone below
            float cost[2][2], neg[2][2];
            bool redo[2][2];
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) {
                    const int i = 2 * li + x, j = 2 * lj + y;
                    const float sq = fmaf(-2.f, accg[x][y], xx[x]) + yy[y];
                    const float ns = xx[x] + yy[y];
                    redo[x][y] = i < q_len && j < c_len && sq < 1e-4f * ns * ns;
                    cost[x][y] = sqrtf(fmaxf(sq, 1e-8f));
                    neg[x][y] = -sqrtf(fmaxf(sq, 0.f));
                }
            {
                // direct-formula redo, the whole wave on one entry (12 coordinates per lane), one entry per memory round trip (the fused
                // kernel takes four: their 96 transient registers do not fit this kernel's 168)
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) {
                        unsigned long long wm = __ballot(redo[x][y]);
                        while (wm != 0) {
                            const int o = (int)__builtin_ctzll(wm);
                            wm &= wm - 1;
                            const int ol = o & 15, i = 2 * (ol >> 2) + x, j = 2 * (ol & 3) + y;
                            const int cs_e = __builtin_amdgcn_readlane(c_start, o);
                            const float* qrow = reinterpret_cast<const float*>(qbase) + (size_t)i * kD + 4 * lane;
                            const float* crow = a.c.rows + ((size_t)cs_e + j) * kD + 4 * lane;
                            float part = 0.f;
#pragma unroll
                            for (int t = 0; t < 3; ++t) {
                                const float4 u = ld4(qrow + 256 * t), v = ld4(crow + 256 * t);
                                const float d0 = u.x - v.x, d1 = u.y - v.y, d2 = u.z - v.z, d3 = u.w - v.w;
                                part = fmaf(d3, d3, fmaf(d2, d2, fmaf(d1, d1, fmaf(d0, d0, part))));
                            }
                            const float tot = wave_sum(part);
                            if (lane == o) {
                                neg[x][y] = -sqrtf(tot);
                                cost[x][y] = sqrtf(fmaxf(tot, 1e-8f));      // geomloss's cost from the same exact sum (fused.hip)
                            }
                        }
                    }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the scratch table is rewritten by the next item
            __builtin_amdgcn_wave_barrier();

            // ---- hand the item over: ticket, wait for the slot (twelve items behind), payload, publish --------------------------
            uint32_t t = 0;
            if (lane == 0) t = lds_fetch_add(&ctl[kCtlTail]);
            t = (uint32_t)__builtin_amdgcn_readfirstlane(t);
            const uint32_t slot = t % kSpSlots;
            while (lds_load(&ctl[kCtlSeq + slot]) != t) __builtin_amdgcn_s_sleep(2);
            float* pl = lds_all + kSpRingOfs + slot * kSpItemWords;
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                *reinterpret_cast<float2*>(pl + p * 64 + (2 * li + x) * 8 + 2 * lj) = make_float2(cost[x][0], cost[x][1]);
                *reinterpret_cast<float2*>(pl + 256 + p * 64 + (2 * li + x) * 8 + 2 * lj) = make_float2(neg[x][0], neg[x][1]);
            }
            if (lane == 0) pl[512] = __builtin_bit_cast(float, q_len);
            if (lp == 0) {
                pl[512 + 4 + p] = __builtin_bit_cast(float, c_len | (real ? 256 : 0));
                pl[512 + 8 + p] = __builtin_bit_cast(float, c_idx);
                pl[512 + 12 + p] = diam2;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) lds_store(&ctl[kCtlSeq + slot], t + 1);
            }
            ++n_done;
            S_STAMP(n_done);
            cur_it = nxt_it;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) lds_fetch_add(&ctl[kCtlDone]);           // (after the wave's last publication)
    } else {
        // ---- the solver waves first form the query boxes of the run's jobs (256 threads, three coordinates each) --------------------
        const int tid = threadIdx.x - kSpStreamers * 64;
        const int job_hi = SINGLE ? 0 : __popcll(__ballot(gend <= (int)(lo + cnt - 1)));
#pragma unroll 1
        for (int s = 0; s < kSpJobs; ++s) {
            const int job = job_lo + s;
            if (job > job_hi) break;
            const int q_idx = SINGLE ? 0 : a.job0 + job;
            const int q_len = a.q.len[q_idx];
            const float* qdoc = a.q.rows + (size_t)a.q.start[q_idx] * kD;
            float* qb = lds_all + kSpBoxOfs + s * 2 * kD;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int d = tid + 256 * t;
                float v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = qdoc[(size_t)min(r, q_len - 1) * kD + d];
                float mn = v[0], mx = v[0];
#pragma unroll
                for (int r = 1; r < 8; ++r) {
                    mn = fminf(mn, v[r]);
                    mx = fmaxf(mx, v[r]);
                }
                qb[d] = mn;
                qb[kD + d] = mx;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) lds_fetch_add(&ctl[kCtlBox]);
    }
    S_STAMP(8);

    // ---- solve: pop an item, solve its four pairs, store the scores -------------------------------------------------------------------
    {
        const int p = lane >> 4, lp = lane & 15, li = lp >> 2, lj = lp & 3;
        int n_solved = 0;
        (void)n_solved;
        for (;;) {
            uint32_t h = 0;
            if (lane == 0) h = lds_fetch_add(&ctl[kCtlHead]);
            h = (uint32_t)__builtin_amdgcn_readfirstlane(h);
            const uint32_t slot = h % kSpSlots;
            bool none = false;
            while (lds_load(&ctl[kCtlSeq + slot]) != h + 1) {
                // no more items: every streamer is through and this ticket is past the last publication
                if (lds_load(&ctl[kCtlDone]) == (uint32_t)kSpStreamers && h >= lds_load(&ctl[kCtlTail])) {
                    none = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            if (none) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const float* pl = lds_all + kSpRingOfs + slot * kSpItemWords;
            float cost[2][2], neg[2][2];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const float2 c2 = *reinterpret_cast<const float2*>(pl + p * 64 + (2 * li + x) * 8 + 2 * lj);
                const float2 n2 = *reinterpret_cast<const float2*>(pl + 256 + p * 64 + (2 * li + x) * 8 + 2 * lj);
                cost[x][0] = c2.x; cost[x][1] = c2.y;
                neg[x][0] = n2.x; neg[x][1] = n2.y;
            }
            const int q_len = __builtin_bit_cast(int, pl[512]);
            const int lr = __builtin_bit_cast(int, pl[512 + 4 + p]);
            const int c_idx = __builtin_bit_cast(int, pl[512 + 8 + p]);
            const float diam2 = pl[512 + 12 + p];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the payload is in registers before the slot is released
            if (lane == 0) lds_store(&ctl[kCtlSeq + slot], h + kSpSlots);
            const int c_len = lr & 255;
            const bool real = (lr & 256) != 0;
            bool rv[2], cv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                rv[t] = 2 * li + t < q_len;
                cv[t] = 2 * lj + t < c_len;
            }
            if (a.skip_tail) {          // timing probe (SPLIT_PRIO=3): no solve -- what the streamers and the ring cost on their own
                if (real && lp == 0) a.scores[c_idx] = diam2 + cost[0][0] + neg[0][0];
                continue;
            }
            Solve s;
            solve_begin<false>(s, a, cost, neg, rv, cv, fmaxf(sqrtf(diam2), kMinDiameter));
            s.out = real ? (int64_t)c_idx : (int64_t)-1;
            if (q_len > 8 || c_len > 8) s.valid |= 16u;
            solve_finish<false>(s, a);
            ++n_solved;
            S_STAMP(8 + n_solved);
        }
    }
    S_STAMP(15);
}

int cu_count() {
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

template <bool SINGLE>
int launch_split_as(const ScoreArgs& a, hipStream_t stream) {
    static std::once_flag raised;
    static hipError_t raise_rc = hipSuccess;
    std::call_once(raised, [] {
        raise_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(pair_split_kernel<SINGLE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       kSpLdsFloats * (int)sizeof(float));
    });
    ASPIRE_HIP_OK(raise_rc);
    hipLaunchKernelGGL((pair_split_kernel<SINGLE>), dim3((unsigned)cu_count()), dim3(kSpWaves * 64), kSpLdsFloats * sizeof(float), stream, a);
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}

}  // namespace

// the role-split form pays once every CU's eight streamers have a couple of items each
bool split_path_ok(int64_t groups_bound, const aspire_ot_params* prm) {
    const int f = tuning().fused_split;
    if (f != 1) return false;
    return groups_bound >= (int64_t)cu_count() * 2 && prm->scaling >= 0.25 && tuning().fused_nosolve == 0 && !tuning().fused_valu &&
           !tuning().fused_noself;
}

// a: MAPPED jobs [job0, job1) (<= 64 of them), or CROSS with ONE query against candidates [cand0, cand1)
int launch_pair_split(const ScoreArgs& a_in, hipStream_t stream) {
    ScoreArgs a = a_in;
    const bool single = a.pairing == ASPIRE_PAIR_CROSS;
    const int prio = tuning().split_prio;
    a.skip_tail = prio == 3;
    return single ? launch_split_as<true>(a, stream) : launch_split_as<false>(a, stream);
}

}  // namespace aspire
#ifdef ASPIRE_PHASE_CLOCK
extern "C" void aspire_debug_split_buffer(void* p) {
    long long* q = (long long*)p;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(aspire::g_sdbg), &q, sizeof(q));
}
#endif
