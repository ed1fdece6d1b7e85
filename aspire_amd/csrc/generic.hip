// Documents of more than 32 sentence rows (the tile kernels' limit): one workgroup per pair, everything in LDS.
//
// The reference has no sentence-count limit, only the 500-word-piece cap (examples/ex_aspire_consent.py:120, 142-155), and
// AspireNER appends entity "sentences" to an abstract (src/evaluation/utils/models.py:224-233), so an evaluation pool can
// hold the odd document of 40 or 60 rows.  Such pairs are rare: this kernel favours being obviously right over being
// fast -- geomloss 0.2.4's own formulation (sinkhorn_tensorized restated, see score.hip): log-weights in the exponent,
// max-shifted log-sum-exps, the float64 epsilon schedule exactly as numpy builds it, expf / logf from libm.
//   * the regular kernels run first and leave NaN for a pair that holds a document longer than their tile; this kernel is
//     then launched over ALL pairs and a workgroup returns at once unless its pair is one of those (`skip_up_to`);
//   * padded reference tensors whose extent exceeds 32 rows go through it for every pair (skip_up_to = 0), with the
//     return_pair_sims extras (pair_distances.py:86).
// Reference arithmetic: src/learning/facetid_models/pair_distances.py:21-92 (otAspire), :138-186 (tsAspire max-sim).
#include <math.h>

#include "common.h"
#include "score_device.h"
#include "score_types.h"

namespace aspire {
namespace {

constexpr int kGenThreads = 256;

struct GenLds {           // offsets in floats into the dynamic LDS block
    int cost, neg, f, g, ft, gt, la, lb, wa, wb, xx, yy, red, total;
};
__host__ __device__ inline GenLds gen_layout(int rows_q, int rows_c) {
    GenLds L;
    const int ld = rows_c + 1, m = rows_q > rows_c ? rows_q : rows_c;
    int o = 0;
    L.cost = o; o += rows_q * ld;
    L.neg = o; o += rows_q * ld;
    L.f = o; o += m;
    L.g = o; o += m;
    L.ft = o; o += m;
    L.gt = o; o += m;
    L.la = o; o += m;
    L.lb = o; o += m;
    L.wa = o; o += m;
    L.wb = o; o += m;
    L.xx = o; o += m;
    L.yy = o; o += m;
    L.red = o; o += kGenThreads;
    L.total = o;
    return L;
}

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = kGenThreads / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] = is_max ? fmaxf(red[tid], red[tid + s]) : red[tid] + red[tid + s];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

// mode 0: otAspire (a.want, extras);  mode 1: the aggregations of -cdist (a.agg: tsAspire max-sim, top-2, attention; pair outputs);
__device__ __forceinline__ void pair_generic_body(const ScoreArgs& a, int mode, int skip_up_to, int rows_q, int rows_c, int64_t p,
                                                  float* lds) {
    const int tid = threadIdx.x;
    const bool paired = a.pairing != ASPIRE_PAIR_CROSS;                // one query per candidate (PAIRED, MAPPED)
    const int64_t c_idx = paired ? p : p % a.c.n;
    int64_t q_idx = a.pairing == ASPIRE_PAIR_PAIRED ? p : p / a.c.n;
    if (a.pairing == kPairMapped) {
        if (a.qmap != nullptr) {
            q_idx = (int64_t)a.qmap[p];
        } else {                                                       // no candidate -> job table (the fused kernel's SELF form): search job_off
            int j = a.job0;
            while (j + 1 < a.job1 && (int64_t)a.job_off[j + 1] <= p) ++j;
            q_idx = j;
        }
    }
    const int q_len = a.q.len[q_idx], c_len = a.c.len[c_idx];
    if (q_len <= skip_up_to && c_len <= skip_up_to) return;          // the tile kernels scored this pair
    if (q_len > rows_q || c_len > rows_c) {                            // longer than the host-known bound: poison
        if (tid == 0) a.scores[p] = __builtin_nanf("");
        return;
    }
    const int q_av = a.q.ext > 0 ? a.q.ext : q_len, c_av = a.c.ext > 0 ? a.c.ext : c_len;      // readable rows (pad rows of padded tensors)
    const GenLds L = gen_layout(rows_q, rows_c);
    const int ld = rows_c + 1;
    float* cost = lds + L.cost;
    float* neg = lds + L.neg;
    const float* qdoc = a.q.rows + (size_t)a.q.start[q_idx] * kD;
    const float* cdoc = a.c.rows + (size_t)a.c.start[c_idx] * kD;

    // ---- squared norms of every readable row ----------------------------------------------------------------------------
    for (int r = tid; r < q_av + c_av; r += kGenThreads) {
        const float* row = r < q_av ? qdoc + (size_t)r * kD : cdoc + (size_t)(r - q_av) * kD;
        float s0 = 0.f, s1 = 0.f;
        for (int d = 0; d < kD; d += 8) {
            s0 += sq4(ld4(row + d));
            s1 += sq4(ld4(row + d + 4));
        }
        if (r < q_av) lds[L.xx + r] = s0 + s1;
        else lds[L.yy + r - q_av] = s0 + s1;
    }
    __syncthreads();
    // ---- the two L2 forms of every entry: geomloss's cost (matmul expansion, clamp 1e-8) and torch.cdist (direct
    // differences up to 25 rows on both sides, the expansion beyond) ----------------------------------------------------
    const bool mm = use_mm_formula(a.cdist_mode, q_av, c_av);
    for (int e = tid; e < q_av * c_av; e += kGenThreads) {
        const int i = e / c_av, j = e - i * c_av;
        const float* x = qdoc + (size_t)i * kD;
        const float* y = cdoc + (size_t)j * kD;
        float g0 = 0.f, g1 = 0.f, d0 = 0.f, d1 = 0.f;
        for (int d = 0; d < kD; d += 8) {
            const float4 u0 = ld4(x + d), v0 = ld4(y + d), u1 = ld4(x + d + 4), v1 = ld4(y + d + 4);
            g0 += dot4(u0, v0);
            g1 += dot4(u1, v1);
            const float a0 = u0.x - v0.x, a1 = u0.y - v0.y, a2 = u0.z - v0.z, a3 = u0.w - v0.w;
            const float b0 = u1.x - v1.x, b1 = u1.y - v1.y, b2 = u1.z - v1.z, b3 = u1.w - v1.w;
            d0 = fmaf(a3, a3, fmaf(a2, a2, fmaf(a1, a1, fmaf(a0, a0, d0))));
            d1 = fmaf(b3, b3, fmaf(b2, b2, fmaf(b1, b1, fmaf(b0, b0, d1))));
        }
        const float sq = fmaf(-2.f, g0 + g1, lds[L.xx + i]) + lds[L.yy + j];
        const float ns = lds[L.xx + i] + lds[L.yy + j];
        // (where the expansion cancels -- the streaming kernels' test -- geomloss's cost comes from the exact sum too: what its own
        // formula gives in float64; in fp32 the reference returns the square root of rounding noise there)
        // (round 6: ... and so does -cdist, also under torch.cdist's matmul formula: one rule for a cancelling entry in every kernel family)
        const bool cancels = sq < 1e-4f * ns * ns;
        cost[i * ld + j] = sqrtf(fmaxf(cancels ? d0 + d1 : sq, 1e-8f));
        neg[i * ld + j] = (mm && !cancels) ? -sqrtf(fmaxf(sq, 0.f)) : -sqrtf(d0 + d1);
    }
    __syncthreads();

    if (mode == 1) {
        // ---- the aggregations of the masked -cdist block -------------------------------------------------------------------
        float score;
        if (a.agg == ASPIRE_AGG_MAX) {
            // tsAspire: max over the valid block (pair_distances.py:167-176)
            float best = -INFINITY;
            for (int e = tid; e < q_len * c_len; e += kGenThreads) best = fmaxf(best, neg[(e / c_len) * ld + e % c_len]);
            score = block_reduce(best, lds + L.red, true);
        } else if (a.agg == ASPIRE_AGG_TOP2) {
            // torch.topk(k = 2) over the padded [q.ext, c.ext] block, masked entries taking part with -cdist - 10e8
            // (pair_distances.py:295-345); without padded extents over the valid block, a missing second entry counting -10e8.
            // The largest value, then: the same again if it occurs twice, else the largest value below it.
            const int eq = a.q.ext > 0 ? a.q.ext : q_len, ec = a.c.ext > 0 ? a.c.ext : c_len;
            auto entry = [&](int e) {
                const int i = e / ec, j = e - i * ec;
                return neg[i * ld + j] + ((i < q_len && j < c_len) ? 0.f : -10e8f);
            };
            float m1 = -INFINITY;
            for (int e = tid; e < eq * ec; e += kGenThreads) m1 = fmaxf(m1, entry(e));
            m1 = block_reduce(m1, lds + L.red, true);
            float cnt = 0.f, below = -INFINITY;
            for (int e = tid; e < eq * ec; e += kGenThreads) {
                const float v = entry(e);
                if (v == m1) cnt += 1.f;
                else below = fmaxf(below, v);
            }
            cnt = block_reduce(cnt, lds + L.red, false);
            below = block_reduce(below, lds + L.red, true);
            float m2 = cnt >= 2.f ? m1 : below;
            if (m2 == -INFINITY) m2 = -10e8f;
            score = m1 + m2;
        } else {
            // AllPairMaskedAttention (pair_distances.py:95-135): soft-max of -d / temp over the valid block, sum p * (-d)
            const float temp = (float)a.temp;
            float mx = -INFINITY;
            for (int e = tid; e < q_len * c_len; e += kGenThreads) mx = fmaxf(mx, neg[(e / c_len) * ld + e % c_len] / temp);
            mx = block_reduce(mx, lds + L.red, true);
            float se = 0.f, sn = 0.f;
            for (int e = tid; e < q_len * c_len; e += kGenThreads) {
                const float v = neg[(e / c_len) * ld + e % c_len];
                const float w = expf(v / temp - mx);
                se += w;
                sn = fmaf(w, v, sn);
            }
            se = block_reduce(se, lds + L.red, false);
            sn = block_reduce(sn, lds + L.red, false);
            score = sn / se;
            if (a.out_plan)
                for (int e = tid; e < a.q.ext * a.c.ext; e += kGenThreads) {
                    const int i = e / a.c.ext, j = e - i * a.c.ext;
                    a.out_plan[(p * a.q.ext + i) * a.c.ext + j] = (i < q_len && j < c_len) ? expf(neg[i * ld + j] / temp - mx) / se : 0.f;
                }
        }
        if (tid == 0) a.scores[p] = score;
        if (a.out_pairsims)
            for (int e = tid; e < a.q.ext * a.c.ext; e += kGenThreads) {
                const int i = e / a.c.ext, j = e - i * a.c.ext;
                a.out_pairsims[(p * a.q.ext + i) * a.c.ext + j] =
                    neg[i * ld + j] + (((i < q_len && j < c_len) || a.agg == ASPIRE_AGG_ATTENTION) ? 0.f : -10e8f);
            }
        return;
    }

    // ---- diameter: the caller's (one per group) or the bounding box of the pair's own valid rows ------------------------
    float diam;
    if (a.diameter != nullptr) {
        diam = a.pairing == ASPIRE_PAIR_CROSS ? a.diameter[q_idx * a.n_groups + c_idx / a.diam_group] : a.diameter[c_idx / a.diam_group];
    } else {
        float acc = 0.f;
        for (int d = tid; d < kD; d += kGenThreads) {
            float mn = INFINITY, mx = -INFINITY;
            for (int r = 0; r < q_len; ++r) { const float v = qdoc[(size_t)r * kD + d]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
            for (int r = 0; r < c_len; ++r) { const float v = cdoc[(size_t)r * kD + d]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
            acc += (mx - mn) * (mx - mn);
        }
        diam = sqrtf(block_reduce(acc, lds + L.red, false));
    }
    // two documents that are one and the same point (a one-sentence candidate equal to a one-sentence query): diameter 0,
    // where geomloss's schedule (log diam) is undefined -- a tiny diameter gives the obvious answer, the cost of that one entry
    diam = fmaxf(diam, kMinDiameter);
    // ---- marginals (pair_distances.py:57-60): softmax over sentences of the best match / temp ----------------------------
    const float temp = (float)a.temp;
    if (tid < q_len) {
        float m = -INFINITY;
        for (int j = 0; j < c_len; ++j) m = fmaxf(m, neg[tid * ld + j]);
        lds[L.ft + tid] = m / temp;
    }
    if (tid < c_len) {
        float m = -INFINITY;
        for (int i = 0; i < q_len; ++i) m = fmaxf(m, neg[i * ld + tid]);
        lds[L.gt + tid] = m / temp;
    }
    __syncthreads();
    {
        float mq = -INFINITY, mc = -INFINITY, sq = 0.f, sc = 0.f;
        for (int i = 0; i < q_len; ++i) mq = fmaxf(mq, lds[L.ft + i]);
        for (int j = 0; j < c_len; ++j) mc = fmaxf(mc, lds[L.gt + j]);
        for (int i = 0; i < q_len; ++i) sq += expf(lds[L.ft + i] - mq);
        for (int j = 0; j < c_len; ++j) sc += expf(lds[L.gt + j] - mc);
        const float lsq = logf(sq), lsc = logf(sc);
        __syncthreads();
        if (tid < rows_q) {
            const float w = tid < q_len ? expf(lds[L.ft + tid] - mq - lsq) : 0.f;      // log_softmax(...).exp()
            lds[L.wa + tid] = w;
            lds[L.la + tid] = w > 0.f ? logf(w) : -100000.f;                             // geomloss log_weights
        }
        if (tid < rows_c) {
            const float w = tid < c_len ? expf(lds[L.gt + tid] - mc - lsc) : 0.f;
            lds[L.wb + tid] = w;
            lds[L.lb + tid] = w > 0.f ? logf(w) : -100000.f;
        }
    }
    __syncthreads();
    // ---- Sinkhorn loop (geomloss sinkhorn_loop): softmin(eps, C, h)_i = -eps * LSE_j(h_j - C_ij / eps) ------------------
    float* f = lds + L.f;
    float* g = lds + L.g;
    float* ft = lds + L.ft;
    float* gt = lds + L.gt;
    const float* la = lds + L.la;
    const float* lb = lds + L.lb;
    // first: true = the initialisation (bare log-weights); else h = log-weight + potential / eps
    auto softmins = [&](float eps, bool first) {
        if (tid < c_len) {          // gt_j = softmin over i
            float m = -INFINITY;
            for (int i = 0; i < q_len; ++i) m = fmaxf(m, la[i] + (first ? 0.f : f[i] / eps) - cost[i * ld + tid] / eps);
            float s = 0.f;
            for (int i = 0; i < q_len; ++i) s += expf(la[i] + (first ? 0.f : f[i] / eps) - cost[i * ld + tid] / eps - m);
            gt[tid] = -eps * (m + logf(s));
        }
        if (tid < q_len) {          // ft_i = softmin over j
            float m = -INFINITY;
            for (int j = 0; j < c_len; ++j) m = fmaxf(m, lb[j] + (first ? 0.f : g[j] / eps) - cost[tid * ld + j] / eps);
            float s = 0.f;
            for (int j = 0; j < c_len; ++j) s += expf(lb[j] + (first ? 0.f : g[j] / eps) - cost[tid * ld + j] / eps - m);
            ft[tid] = -eps * (m + logf(s));
        }
        __syncthreads();
    };
    auto update = [&](bool averaged) {
        if (tid < q_len) f[tid] = averaged ? 0.5f * (f[tid] + ft[tid]) : ft[tid];
        if (tid < c_len) g[tid] = averaged ? 0.5f * (g[tid] + gt[tid]) : gt[tid];
        __syncthreads();
    };
    const float eb = (float)a.blur;
    softmins(diam, true);
    update(false);
    softmins(diam, false);          // eps_s[0] = diam
    update(true);
    const double ldm = log((double)diam);
    int n_mid = (int)ceil((a.log_blur - ldm) / a.log_scaling);      // len(arange(log diam, log blur, log scaling))
    n_mid = n_mid < 0 ? 0 : n_mid;
    for (int k = 0; k < n_mid; ++k) {
        softmins((float)exp(ldm + (double)k * a.log_scaling), false);
        update(true);
    }
    softmins(eb, false);
    update(true);
    softmins(eb, false);            // last extrapolation: simultaneous, not averaged
    update(false);

    // ---- outputs ---------------------------------------------------------------------------------------------------------
    float acc = 0.f;
    if (a.want != ASPIRE_OT_PLAN_SIM) {
        if (tid < q_len) acc += lds[L.wa + tid] * f[tid];
        if (tid < c_len) acc += lds[L.wb + tid] * g[tid];
    }
    const bool dump = a.out_plan != nullptr || a.out_pairsims != nullptr;
    if (a.want == ASPIRE_OT_PLAN_SIM || dump) {
        const int eq = dump ? a.q.ext : q_len, ec = dump ? a.c.ext : c_len;
        for (int e = tid; e < eq * ec; e += kGenThreads) {
            const int i = e / ec, j = e - i * ec;
            const bool valid = i < q_len && j < c_len;
            const float negm = valid ? neg[i * ld + j] : 0.f;
            const float plan = valid ? expf((f[i] + g[j] + negm) / eb) * (lds[L.wa + i] * lds[L.wb + j]) : 0.f;
            if (a.want == ASPIRE_OT_PLAN_SIM) acc += plan * negm;
            if (dump) {
                const int64_t o = (p * a.q.ext + i) * a.c.ext + j;
                if (a.out_plan) a.out_plan[o] = plan;
                if (a.out_pairsims) a.out_pairsims[o] = negm;
            }
        }
    }
    float score = block_reduce(acc, lds + L.red, false);
    if (a.want == ASPIRE_OT_SIMILARITY) score = -score;
    if (tid == 0) a.scores[p] = score;
    if (a.out_qdistr)
        for (int i = tid; i < a.q.ext; i += kGenThreads) a.out_qdistr[p * a.q.ext + i] = i < q_len ? lds[L.wa + i] : 0.f;
    if (a.out_cdistr)
        for (int j = tid; j < a.c.ext; j += kGenThreads) a.out_cdistr[p * a.c.ext + j] = j < c_len ? lds[L.wb + j] : 0.f;
}

__global__ void __launch_bounds__(kGenThreads) pair_generic_kernel(ScoreArgs a, int mode, int skip_up_to, int rows_q, int rows_c) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int64_t p = (int64_t)blockIdx.x + (int64_t)blockIdx.y * gridDim.x;
    const int64_t P = a.pairing != ASPIRE_PAIR_CROSS ? a.c.n : a.q.n * a.c.n;
    if (p >= P) return;
    pair_generic_body(a, mode, skip_up_to, rows_q, rows_c, p, lds);
}

}  // namespace

int generic_max_rows(void) { return 128; }

// mode 0 otAspire / 1 max-sim over every pair of `a` (CROSS or PAIRED); pairs whose two documents both have at most
// `skip_up_to` rows are left alone.  rows_q / rows_c: host-known bounds of the documents' rows (<= generic_max_rows()).
int launch_pair_generic(const ScoreArgs& a, int mode, int skip_up_to, int rows_q, int rows_c, hipStream_t stream) {
    ASPIRE_REQUIRE(rows_q <= generic_max_rows() && rows_c <= generic_max_rows(), ASPIRE_ERR_UNSUPPORTED,
                   "documents with more than %d sentence rows are not supported (got %d x %d)", generic_max_rows(), rows_q, rows_c);
    const int64_t P = a.pairing == ASPIRE_PAIR_CROSS ? a.q.n * a.c.n : a.c.n;
    if (P == 0) return ASPIRE_OK;
    const size_t lds_bytes = (size_t)gen_layout(rows_q, rows_c).total * sizeof(float);
    if (lds_bytes > 64 * 1024) {      // more than the default dynamic LDS limit: raise it (per function, sticky, harmless to repeat)
        ASPIRE_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(pair_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          160 * 1024));
    }
    // grid.x * grid.y >= P with grid.y <= 65535: pair = x + y * grid.x
    const int64_t gxx = P < 1048576 ? P : 1048576;
    const int64_t gy = (P + gxx - 1) / gxx;
    ASPIRE_REQUIRE(gy <= 65535, ASPIRE_ERR_UNSUPPORTED, "too many pairs (%lld) for the long-document kernel", (long long)P);
    hipLaunchKernelGGL(pair_generic_kernel, dim3((unsigned)gxx, (unsigned)gy), dim3(kGenThreads), lds_bytes, stream, a, mode, skip_up_to,
                       rows_q, rows_c);
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}

}  // namespace aspire
