// NOT BUILT.  Kernel forms that were measured and superseded in round 1, kept for reference (they compiled against
// aspire_amd/csrc/score.hip as of commit fb9bf10; see DESIGN.md "Tried and dropped"):
//   pair_cost_mfma1_kernel  matrix-core (v_mfma_f32_16x16x4_f32) form of the small-pool cost kernel: 829 instead of 1825
//                           vector instructions per pair, but 9.8 us per launch against 8.7 and 100 vs 124 M overlapped
//   sinkhorn4_kernel        16 lanes per pair, 2 x 2 entries per lane, two exponentials per entry, schedule table in LDS:
//                           superseded by sinkhorn_block_kernel (one exponential per entry)
// ---------------------------------------------------------------------------------------------
// Kernel 1, matrix-core form for small single-tile pools (T == 1, CSR inputs, all-pairs).
//
// The VALU forms above spend ~1800 vector instructions per pair (768 multiply-adds per wave-slice plus the 64-lane
// reduction of their 64 partial sums, norms, bounding box), and with many queries' launches overlapped the chip is
// VALU-issue bound.  Here x.y runs on v_mfma_f32_16x16x4_f32 (exact fp32 multiply-adds; the matrix pipe is idle
// otherwise): a workgroup takes TWO candidates (16 rows = the M side) against the query's 8 rows (N side, columns
// 8-15 repeat them), its four waves a quarter of the 768 coordinates each.  Lane (r, g) = (l & 15, l >> 4) loads row
// r's coordinates 16 t + 4 g .. + 3 of its quarter as float4s; component c of chunk t is one K = 4 step for BOTH
// operands (the K index only has to agree between A and B, and both use the same lane -> coordinate map), so the
// accumulators are finished sums over the quarter and nothing is reduced across lanes.  Norms: 48 multiply-adds per
// lane on the same registers + two swaps over g.  The pair's bounding box (own epsilon schedule) wants all rows of a
// coordinate in one lane: a second, row-per-register view of the same bytes (L1 / L2 hits) feeds v_min3 / v_max3.
// Measured (bench.py, 1 x 1000 x 8): 829 vector instructions per pair instead of 1825, but 9.8 us per launch against
// 9.3 and 100 M alignments/s overlapped against 124 -- the second view's load round trip and the 16-row gathers
// (every load instruction touches 16 half-used cache lines) cost more than the issue slots saved.  Kept behind
// ASPIRE_HIP_COST1=mfma (parity-tested) as the starting point for few-query pools whose rows fill the N side.
// ---------------------------------------------------------------------------------------------
typedef float mfma4_t __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256, 2) pair_cost_mfma1_kernel(ScoreArgs a, PairWs<1> ws) {
    __shared__ float g_part[4][16][17];      // [wave][candidate row 0..15][query row 0..15 (+1 pad)]
    __shared__ float yn_s[4][16], xn_s[4][8], box_s[4][2];
    __shared__ unsigned long long redo_s[2];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4, which = r >> 3, rr = r & 7;
    const uint32_t nq = (uint32_t)a.q.n, ncand = (uint32_t)(a.cand1 - a.cand0);
    const uint32_t item = blockIdx.x;                    // grid = nq * ceil(ncand / 2)
    const uint32_t cp = nq == 1 ? item : item / nq, q_loc = nq == 1 ? 0u : item - cp * nq;
    const uint32_t cl0 = 2 * cp, cl1 = min(2 * cp + 1, ncand - 1);       // odd tail: the last candidate twice
    const int64_t c_idx0 = a.cand0 + cl0, c_idx1 = a.cand0 + cl1, q_idx = (int64_t)q_loc;
    const int c_len0 = a.c.len[c_idx0], c_len1 = a.c.len[c_idx1], q_len = a.q.len[q_idx];
    const float* cdoc0 = a.c.rows + (size_t)a.c.start[c_idx0] * kD;
    const float* cdoc1 = a.c.rows + (size_t)a.c.start[c_idx1] * kD;
    const float* qdoc = a.q.rows + (size_t)a.q.start[q_idx] * kD;
    const bool own_diam = a.diameter == nullptr;
    const int dbase = wave * 192;

    // ---- operands in the matrix layout (rows beyond a document's length repeat its last row: masked downstream)
    const float* yptr = (which ? cdoc1 : cdoc0) + (size_t)min(rr, (which ? c_len1 : c_len0) - 1) * kD + dbase + 4 * g;
    const float* xptr = qdoc + (size_t)min(rr, q_len - 1) * kD + dbase + 4 * g;
    float4 xb[12], ya[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) xb[t] = ld4(xptr + 16 * t);     // the query first: L2 resident, lands early
#pragma unroll
    for (int t = 0; t < 12; ++t) ya[t] = ld4(yptr + 16 * t);
    __builtin_amdgcn_sched_barrier(0);       // all 24 loads in flight before the first multiply (left alone the scheduler
                                             // issues them five at a time between the MFMAs: several HBM round trips)
    mfma4_t acc = {0.f, 0.f, 0.f, 0.f};
    float yn0 = 0.f, yn1 = 0.f, xn0 = 0.f, xn1 = 0.f;
#pragma unroll
    for (int t = 0; t < 12; ++t) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[t].x, xb[t].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[t].y, xb[t].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[t].z, xb[t].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[t].w, xb[t].w, acc, 0, 0, 0);
        yn0 = fmaf(ya[t].y, ya[t].y, fmaf(ya[t].x, ya[t].x, yn0));
        yn1 = fmaf(ya[t].w, ya[t].w, fmaf(ya[t].z, ya[t].z, yn1));
        xn0 = fmaf(xb[t].y, xb[t].y, fmaf(xb[t].x, xb[t].x, xn0));
        xn1 = fmaf(xb[t].w, xb[t].w, fmaf(xb[t].z, xb[t].z, xn1));
    }
    float yn = yn0 + yn1, xn = xn0 + xn1;
    yn = swap_add<16>(yn, yn);      // over g (lane bits 4, 5)
    xn = swap_add<16>(xn, xn);
    yn = swap_add<32>(yn, yn);
    xn = swap_add<32>(xn, xn);
    if (g == 0) {
        yn_s[wave][r] = yn;
        if (r < 8) xn_s[wave][r] = xn;
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) g_part[wave][4 * g + v][r] = acc[v];    // D[i][j]: lane holds rows 4 g + v of column r

    // ---- bounding boxes of (query + candidate) per coordinate: row-per-register view, 48 lanes x 4 coordinates
    if (own_diam) {
        float s0 = 0.f, s1 = 0.f;
        if (lane < 48) {
            const int d = dbase + 4 * lane;
            auto box8 = [&](const float* doc, int len, float4& mn, float4& mx, bool init) {
                float4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = ld4(doc + (size_t)min(k, len - 1) * kD + d);
                if (init) mn = mx = v[0];
#pragma unroll
                for (int k = init ? 1 : 0; k < 8; ++k) {
                    mn.x = fminf(mn.x, v[k].x); mn.y = fminf(mn.y, v[k].y); mn.z = fminf(mn.z, v[k].z); mn.w = fminf(mn.w, v[k].w);
                    mx.x = fmaxf(mx.x, v[k].x); mx.y = fmaxf(mx.y, v[k].y); mx.z = fmaxf(mx.z, v[k].z); mx.w = fmaxf(mx.w, v[k].w);
                }
            };
            float4 qmn, qmx;
            box8(qdoc, q_len, qmn, qmx, true);
            auto span2 = [&](const float* doc, int len) {
                float4 mn = qmn, mx = qmx;
                box8(doc, len, mn, mx, false);
                const float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z, dw = mx.w - mn.w;
                return fmaf(dw, dw, fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
            };
            s0 = span2(cdoc0, c_len0);
            s1 = span2(cdoc1, c_len1);
        }
        s0 = wave_sum(s0);
        s1 = wave_sum(s1);
        if (lane == 0) {
            box_s[wave][0] = s0;
            box_s[wave][1] = s1;
        }
    }
    __syncthreads();

    // ---- finish: waves 0 and 1 take a candidate each, lane e = 8 i + j as in pair_cost1_kernel ----
    if (wave < 2) {
        const int cand = wave, li = lane >> 3, lj = lane & 7;
        const int c_len = cand ? c_len1 : c_len0;
        float gsum = 0.f, xx = 0.f, yy = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            gsum += g_part[w][8 * cand + lj][li];
            xx += xn_s[w][li];
            yy += yn_s[w][8 * cand + lj];
        }
        const float sq = fmaf(-2.f, gsum, xx) + yy;
        const float ns = xx + yy;
        const bool mm = use_mm_formula(a.cdist_mode, q_len, c_len);
        const bool redo = !mm && li < q_len && lj < c_len && sq < 1e-4f * ns * ns;
        const int64_t slot = (int64_t)q_loc * ncand + (cand ? cl1 : cl0);
        const int64_t o = slot * 64 + lane;
        ws.cost[o] = sqrtf(fmaxf(sq, 1e-8f));
        if (!redo) ws.neg[o] = -sqrtf(fmaxf(sq, 0.f));
        const unsigned long long m = __ballot(redo);
        if (lane == 0) {
            redo_s[cand] = m;
            if (own_diam) ws.diam2[slot] = (box_s[0][cand] + box_s[1][cand]) + (box_s[2][cand] + box_s[3][cand]);
        }
    }
    __syncthreads();
    // ---- entries whose expansion cancelled: torch.cdist's direct formula, 16 lanes per entry (see pair_cost1_kernel)
#pragma unroll 1
    for (int cand = 0; cand < 2; ++cand) {
        const unsigned long long todo = redo_s[cand];        // workgroup-uniform
        if (__builtin_expect(todo == 0, 1)) continue;
        const float* cdoc = cand ? cdoc1 : cdoc0;
        const int64_t slot = (int64_t)q_loc * ncand + (cand ? cl1 : cl0);
        const int n_flag = __builtin_popcountll(todo), l16 = lane & 15;
        for (int base = wave * 4; base < n_flag; base += 16) {
            const int my = base + (lane >> 4);
            const bool live = my < n_flag;
            unsigned long long m = todo;
            for (int t = 0; t < (live ? my : 0); ++t) m &= m - 1;      // drop the first `my` set bits
            const int e = __builtin_ctzll(m);
            const float* xr = qdoc + (size_t)(e >> 3) * kD + 4 * l16;
            const float* yr = cdoc + (size_t)(e & 7) * kD + 4 * l16;
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
            if (live && l16 == 0) ws.neg[slot * 64 + e] = -sqrtf(part);
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Kernel 2, packed form for <= 8 x 8 problems: FOUR Sinkhorn solves per wave.  A pair lives in one DPP row of 16
// lanes, lane (li, lj) = ((l >> 2) & 3, l & 3) owning the 2 x 2 entries (2 li + a, 2 lj + b).  Both reductions of
// the update then stay inside a DPP row -- over j: in-register + quad_perm xor 1, xor 2; over i: in-register +
// row_ror:4, row_ror:8 -- so the 22-cycle v_permlane swaps of the one-pair-per-wave layout disappear and a step
// costs ~70 issue cycles per pair instead of ~145 (tools: build/dbg/thr.hip for the per-op prices).
// Every pair follows its own epsilon schedule (own diameter): lane k of a pair evaluates steps k, k+16, ... in
// float64 and parks the per-step constants {log2(e)/eps, eps*ln2} in an LDS table that its 16 lanes read back
// (one broadcast ds_read_b64 per step, fetched a step ahead).
// ---------------------------------------------------------------------------------------------
constexpr int kMaxSteps4 = 160;   // eps steps per pair the table holds (diam/blur up to ~1e7 at scaling 0.9)

__device__ __forceinline__ float row16_sum_i(float v) {   // all-reduce over lane bits 2,3 (the 4 values of li)
    v += dpp_mov<0x124>(v, v);                             // row_ror:4
    return v + dpp_mov<0x128>(v, v);                       // row_ror:8
}
__device__ __forceinline__ float row16_max_i(float v) {
    v = fmaxf(v, dpp_mov<0x124>(v, v));
    return fmaxf(v, dpp_mov<0x128>(v, v));
}
__device__ __forceinline__ float quad_sum_j(float v) {     // all-reduce over lane bits 0,1 (the 4 values of lj)
    v += lane_xor<1>(v);
    return v + lane_xor<2>(v);
}
__device__ __forceinline__ float quad_max_j(float v) {
    v = fmaxf(v, lane_xor<1>(v));
    return fmaxf(v, lane_xor<2>(v));
}

__global__ void __launch_bounds__(256) sinkhorn4_kernel(ScoreArgs a, PairWs<1> ws, int64_t n_slots) {
    __shared__ float2 sched[4][4][kMaxSteps4];               // [wave][pair][step] = {r2, eln2}
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pp = lane >> 4, l16 = lane & 15, li = (lane >> 2) & 3, lj = lane & 3;
    const int64_t slot0 = ((int64_t)blockIdx.x * 4 + wave) * 4;
    if (slot0 >= n_slots) return;
    const bool real = slot0 + pp < n_slots;                  // tail wave: surplus groups redo the last pair, store nothing
    const int64_t slot = real ? slot0 + pp : n_slots - 1;
    const bool paired = a.pairing == ASPIRE_PAIR_PAIRED;
    const uint32_t ncand = (uint32_t)(a.cand1 - a.cand0);
    const uint32_t q_loc = paired ? 0u : (uint32_t)slot / ncand;
    const int64_t q_idx = paired ? a.cand0 + slot : (int64_t)q_loc;
    const int64_t c_idx = paired ? a.cand0 + slot : a.cand0 + ((uint32_t)slot - q_loc * ncand);
    const int64_t p = paired ? c_idx : q_idx * a.c.n + c_idx;
    const int q_len = a.q.len[q_idx], c_len = a.c.len[c_idx];

    float cost[2][2], neg[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const float2 cc = *reinterpret_cast<const float2*>(ws.cost + slot * 64 + (2 * li + x) * 8 + 2 * lj);
        const float2 nn = *reinterpret_cast<const float2*>(ws.neg + slot * 64 + (2 * li + x) * 8 + 2 * lj);
        cost[x][0] = cc.x; cost[x][1] = cc.y;
        neg[x][0] = nn.x; neg[x][1] = nn.y;
    }
    float diam;
    if (a.diameter == nullptr) {
        diam = sqrtf(ws.diam2[slot]);
    } else {
        diam = paired ? a.diameter[c_idx / a.diam_group] : a.diameter[q_idx * a.n_groups + c_idx / a.diam_group];
    }
    bool rv[2], cv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        rv[t] = 2 * li + t < q_len;
        cv[t] = 2 * lj + t < c_len;
    }
    // ---- marginals (pair_distances.py:57-60) -------------------------------------------------------------
    const float temp = (float)a.temp;
    float la2[2], lb2[2], wa[2], wb[2];
    {
        float qm[2], cm[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            float m = fmaxf((rv[x] && cv[0]) ? neg[x][0] : kNegBig, (rv[x] && cv[1]) ? neg[x][1] : kNegBig);
            qm[x] = quad_max_j(m) / temp;
        }
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            float m = fmaxf((rv[0] && cv[y]) ? neg[0][y] : kNegBig, (rv[1] && cv[y]) ? neg[1][y] : kNegBig);
            cm[y] = row16_max_i(m) / temp;
        }
        const float mq = row16_max_i(fmaxf(rv[0] ? qm[0] : kNegBig, rv[1] ? qm[1] : kNegBig));
        const float mc = quad_max_j(fmaxf(cv[0] ? cm[0] : kNegBig, cv[1] ? cm[1] : kNegBig));
        const float sq = (rv[0] ? fast_exp(qm[0] - mq) : 0.f) + (rv[1] ? fast_exp(qm[1] - mq) : 0.f);
        const float sc = (cv[0] ? fast_exp(cm[0] - mc) : 0.f) + (cv[1] ? fast_exp(cm[1] - mc) : 0.f);
        const float lsq = fast_log(row16_sum_i(sq)), lsc = fast_log(quad_sum_j(sc));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            wa[t] = rv[t] ? fast_exp(qm[t] - mq - lsq) : 0.f;
            wb[t] = cv[t] ? fast_exp(cm[t] - mc - lsc) : 0.f;
            la2[t] = (wa[t] > 0.f ? fast_log(wa[t]) : -100000.f) * kLog2e;   // geomloss log_weights, in base-2 units
            lb2[t] = (wb[t] > 0.f ? fast_log(wb[t]) : -100000.f) * kLog2e;
        }
    }
    // ---- this pair's epsilon schedule -> LDS ---------------------------------------------------------------
    float ldf;
    int n_mid = schedule_mid_steps(a, diam, ldf);
    const float lscf = a.log2_scaling;
    const bool overflow = n_mid + 3 > kMaxSteps4;             // schedule longer than the table: poison the score
    if (overflow) n_mid = kMaxSteps4 - 3;
    // table rows: 0 = diam (the first loop step), 1 .. n_mid = the annealed values, n_mid+1, n_mid+2 = blur
    float2* tab = sched[wave][pp];
    // The annealed values exp(ld + k*lsc) are formed in fp32 here (5 per lane; in float64 they cost more than the
    // whole annealing loop): a relative 1e-6 on an intermediate temperature moves the final potentials by < 1e-7.
    for (int k = l16; k < n_mid + 3; k += 16) {
        float e;
        if (k == 0) e = diam;
        else if (k <= n_mid) e = __builtin_amdgcn_exp2f(fmaf((float)(k - 1), lscf, ldf));
        else e = (float)a.blur;
        tab[k] = make_float2(kLog2e * rcp_refined(e), e * kLn2);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int n_steps = n_mid + 3;                            // the last one is the un-averaged extrapolation
    int max_steps = n_steps;
    max_steps = max(max_steps, __shfl_xor(max_steps, 16));
    max_steps = max(max_steps, __shfl_xor(max_steps, 32));

    // ---- initialisation at eps = diam: softmin of the bare log-weights.  No max shift is needed: the largest
    // weight of a probability vector over <= 8 atoms is >= 1/8 and C/diam <= ~1, so the sum stays in range. ----
    float f[2], g[2];
    {
        const float2 e0 = tab[0];
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            float sum = 0.f;
#pragma unroll
            for (int x = 0; x < 2; ++x) sum += __builtin_amdgcn_exp2f(rv[x] ? fmaf(-cost[x][y], e0.x, la2[x]) : kNegBig);
            g[y] = -e0.y * __builtin_amdgcn_logf(row16_sum_i(sum));
        }
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            float sum = 0.f;
#pragma unroll
            for (int y = 0; y < 2; ++y) sum += __builtin_amdgcn_exp2f(cv[y] ? fmaf(-cost[x][y], e0.x, lb2[y]) : kNegBig);
            f[x] = -e0.y * __builtin_amdgcn_logf(quad_sum_j(sum));
        }
    }
    // ---- the annealing loop (see step2 of sinkhorn_pair for the derivation of the shifted base-2 update) ----
    float2 ek = tab[0];
    for (int k = 0; k < max_steps; ++k) {
        const float2 enext = tab[min(k + 1, n_steps - 1)];   // fetched a step ahead
        const bool active = k < n_steps;
        const bool averaged = k < n_steps - 1;
        const float r2 = ek.x, eln2 = ek.y;
        float f2[2], g2[2], av[2], bv[2], ft[2], gt[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f2[t] = f[t] * r2;
            g2[t] = g[t] * r2;
            av[t] = la2[t] + f2[t];
            bv[t] = lb2[t] + g2[t];
        }
        float sc_[2] = {0.f, 0.f}, sr_[2] = {0.f, 0.f};
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const float uc = fmaf(-cost[x][y], r2, av[x]) + g2[y];
                const float ur = fmaf(-cost[x][y], r2, bv[y]) + f2[x];
                sc_[y] += __builtin_amdgcn_exp2f(rv[x] ? uc : kNegBig);
                sr_[x] += __builtin_amdgcn_exp2f(cv[y] ? ur : kNegBig);
            }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            gt[t] = eln2 * (g2[t] - __builtin_amdgcn_logf(row16_sum_i(sc_[t])));
            ft[t] = eln2 * (f2[t] - __builtin_amdgcn_logf(quad_sum_j(sr_[t])));
            const float gn = averaged ? 0.5f * (g[t] + gt[t]) : gt[t];
            const float fn = averaged ? 0.5f * (f[t] + ft[t]) : ft[t];
            g[t] = active ? gn : g[t];
            f[t] = active ? fn : f[t];
        }
        ek = enext;
    }
    // ---- outputs ---------------------------------------------------------------------------------------------
    float score;
    if (a.want != ASPIRE_OT_PLAN_SIM) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            acc += (lj == 0 && rv[t]) ? wa[t] * f[t] : 0.f;
            acc += (li == 0 && cv[t]) ? wb[t] * g[t] : 0.f;
        }
        score = row16_sum_i(quad_sum_j(acc));
        if (a.want == ASPIRE_OT_SIMILARITY) score = -score;
    } else {
        const float eb = (float)a.blur, rb = rcp_refined(eb);
        float acc = 0.f;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const bool valid = rv[x] && cv[y];
                const float negm = valid ? neg[x][y] : 0.f;
                const float outer = valid ? f[x] + g[y] : 0.f;
                acc += fast_exp(div_r(outer + negm, eb, rb)) * (wa[x] * wb[y]) * negm;
            }
        score = row16_sum_i(quad_sum_j(acc));
    }
    // the shifted log-sum-exp cannot leave fp32 range on sane inputs; if it did, or the schedule outgrew the
    // table, or a document is longer than the tile, the pair is poisoned rather than silently wrong.
    if (!(fabsf(score) < 1e30f) || overflow || q_len > 8 || c_len > 8) score = __builtin_nanf("");
    if (real && l16 == 0) a.scores[p] = score;
}

