// Device-side argument structs shared by the scoring kernels (score.hip, gram.hip).
#pragma once
#include "common.h"

namespace aspire {

struct RepSet {
    const float* rows;
    const int32_t* start;
    const int32_t* len;
    int64_t n;
    int32_t ext;
};

struct ScoreArgs {
    RepSet q, c;
    int pairing;     // ASPIRE_PAIR_*
    int cdist_mode;  // ASPIRE_CDIST_*
    int q_per_block; // CROSS: queries handled by one block (grid.y chunks)
    int64_t cand0, cand1;  // otAspire: the chunk of candidates this launch covers
    // OT
    double blur, scaling, temp;
    double log_blur, log_scaling;        // natural logs, float64, formed on the host (schedule lengths)
    float log2_blur, log2_scaling;       // the same in log2 units, fp32
    // fp16 planes of the two row matrices (include/aspire_hip.h: aspire_rep_planes; HOST structs, null = none): the launchers of
    // gramp.hip read them, no kernel does
    const aspire_rep_planes* q_planes;
    const aspire_rep_planes* c_planes;
    const float* c_box;      // the candidates' cached per-document boxes (aspire_repset.doc_box: device [c.n][2][768]) or null
    const float* diameter;
    int64_t diam_group;
    int64_t n_groups;
    int want;
    int agg;         // ASPIRE_AGG_* (l2 aggregation kernels)
    float* scores;
    float* out_qdistr;
    float* out_cdistr;
    float* out_pairsims;
    float* out_plan;
    // MAPPED pairing (aspire_ot_rank_batch_f32: J independent (query, pool) jobs in one launch): candidate p is
    // scored against query qmap[p], P = C.  job j owns candidates [job_off[j], job_off[j+1]) and the groups of four
    // candidates [grp_off[j], grp_off[j+1]) (a group never straddles two jobs); grp_job[g] = job of group g.
    const int32_t* qmap;
    const int32_t* job_off;
    const int32_t* grp_off;
    const int32_t* grp_job;
    // 16 int32 per group, everything a wave needs to start the group, in one 64-byte line (one memory round trip instead of
    // four dependent ones): [0] query, [1] its len, [2] its first row, [3] real candidates in the group (1..4),
    // [4..7] candidate index (clamped to the job's last), [8..11] their lens, [12..15] their first rows
    const int32_t* grp_rec;
    int32_t job0, job1;       // the jobs this launch covers (chunks of a batch run side by side on two streams)
    int32_t max_job_groups;   // host-side launch geometry: upper bound of a job's groups of four
    int32_t tile_form;        // host-side: the batch runs on the throughput kernels
    int32_t cost_from_neg;    // the cost stage wrote -cdist tiles only (the matrix-pipe tiles: both numbers come from one d there): geomloss's cost = max(-(-cdist), 1e-4)
    int32_t skip_tail;        // diagnostics (FUSED_NOSOLVE=2): the fused kernel drops every wave's LAST solve (timing of the exposed tail)
    // CHUNK items without a counter: with <= 64 classification blocks ("slices") every block writes its item count to grp_off[slice]
    // and its records into its own region of chunk_region_cap records; a wave of the scoring kernel prefix-scans the 64 counts
    // itself (as the SELF form does with job_off) -- no atomics, no memset in front of the call.  0: grp_off[0] = the item counter.
    int32_t chunk_regions, chunk_region_cap;
    // Rows that share a large common component (sentence-embedding spaces are anisotropic: mean cosine 0.5 .. 0.9 is common): the
    // streaming kernels form -cdist and geomloss's cost from |x|^2 - 2 x.y + |y|^2 and redo an entry with the direct formula where
    // that cancels -- on such rows nearly EVERY entry (20 x 1000 x 8 at mean cosine 0.8: 526 us instead of 100).  L2 distances do
    // not change under a common shift: with `center` set the kernels subtract the mean of the staged query rows from every row on
    // its way into LDS (ASPIRE_OT_FLAG_CENTER / ASPIRE_CDIST_CENTER); norms, boxes and dot products are then those of the spread.
    int32_t center;
    // Hybrid for pools of mostly short documents with a few of 9 .. 16 rows: a census of the long pairs (gate[0], written by
    // long_pair_census_kernel earlier on the stream) decides ON THE DEVICE how the queued kernels work.  Few long pairs
    // (gate[0] <= gate_limit): the fused kernel scores the short pairs and poisons the long ones, the 16-row streaming kernel
    // and the block Sinkhorn kernel then touch ONLY the long pairs.  Many: the fused kernel returns at once, the other two
    // take every pair.
    const int32_t* gate;
    int32_t gate_limit;
    long long* dbg;  // phase cycle stamps (only with -DASPIRE_PHASE_CLOCK)
};

// Internal third pairing next to ASPIRE_PAIR_CROSS / ASPIRE_PAIR_PAIRED (see ScoreArgs::qmap).
constexpr int kPairMapped = 2;
// Floor of a pair's bounding-box diameter in the Sinkhorn solvers: two documents that are one and the same point (a
// one-sentence candidate equal to a one-sentence query) have diameter 0, where geomloss's schedule (log diam) is undefined --
// with a tiny diameter every solver gives the obvious answer, the cost of that one entry.
constexpr float kMinDiameter = 1e-6f;

__device__ __forceinline__ bool gate_few_long(const ScoreArgs& a) { return a.gate != nullptr && *a.gate <= a.gate_limit; }

// (query, candidate, output index) of workspace slot `slot` of the current candidate chunk.
struct PairIdx {
    int64_t q_idx, c_idx, p;
};
__device__ __forceinline__ PairIdx pair_of_slot(const ScoreArgs& a, int64_t slot) {
    PairIdx r;
    if (a.pairing == ASPIRE_PAIR_CROSS) {
        const uint32_t ncand = (uint32_t)(a.cand1 - a.cand0);
        const uint32_t q_loc = (uint32_t)slot / ncand;     // 32-bit: 64-bit division costs hundreds of cycles here
        r.q_idx = (int64_t)q_loc;
        r.c_idx = a.cand0 + ((uint32_t)slot - q_loc * ncand);
        r.p = r.q_idx * a.c.n + r.c_idx;
    } else {
        r.c_idx = a.cand0 + slot;
        r.q_idx = a.pairing == kPairMapped ? (int64_t)a.qmap[r.c_idx] : r.c_idx;
        r.p = r.c_idx;
    }
    return r;
}
__device__ __forceinline__ float group_diameter_of(const ScoreArgs& a, const PairIdx& i) {
    return a.pairing == ASPIRE_PAIR_CROSS ? a.diameter[i.q_idx * a.n_groups + i.c_idx / a.diam_group]
                                          : a.diameter[i.c_idx / a.diam_group];
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// Per-pair intermediates between the two kernels of the otAspire path, one slot per pair of the current
// chunk: cost [8T x 8T] row-major, neg [8T x 8T], diam2 (sum over coordinates of (max-min)^2).
template <int T>
struct PairWs {
    static constexpr int kEntries = 64 * T * T;
    float* cost;
    float* neg;
    float* diam2;
};

__device__ __forceinline__ bool use_mm_formula(int mode, int nq, int nc) {
    // torch.cdist default: matmul expansion iff either side has more than 25 rows.
    return mode == ASPIRE_CDIST_MM || (mode == ASPIRE_CDIST_AUTO && (nq > 25 || nc > 25));
}


// gram.hip: matrix-core form of the pairwise-cost stage (host launchers; see the file header)
bool gram_path_wanted(const aspire_repset* q, const aspire_repset* c, int pairing);
bool gram_planes_wanted_l2max(const aspire_repset* q, const aspire_repset* c, int pairing);
bool gram_planes_wanted_ot(const aspire_repset* q, const aspire_repset* c, int pairing, bool caller_diameters);
size_t gram_extra_bytes_per_cand(void);
int launch_pair_gram_ot(const ScoreArgs& a, int T, int mr_q, int mr_c, float* cost, float* neg, float* diam2, float* qbox,
                        float* cbox, hipStream_t stream);
int launch_pair_gram_l2max(const ScoreArgs& a, int mr_q, int mr_c, hipStream_t stream);

// gramp.hip: the 128-column Gram tiles on pre-split fp16 planes (three matrix-pipe products per term, operands by LDS-DMA)
bool gram_planes_ok(const ScoreArgs& a);
struct GramGeometry {
    int mr_q, mr_c, dpt_q, dpt_c, n_qt, n_ct, E, ld;
    int bm, bn;  // candidate / query rows per tile: 128 x 128, 128 x 256, 256 x 256
};
int launch_pair_gram_planes(const ScoreArgs& a, const GramGeometry& geo, bool l2max, float* cost, float* neg, hipStream_t stream);

// generic.hip: one workgroup per pair for documents beyond the tile kernels' 32 rows (mode 0 otAspire, 1 max-sim)
int generic_max_rows(void);
int launch_pair_generic(const ScoreArgs& a, int mode, int skip_up_to, int rows_q, int rows_c, hipStream_t stream);

// fused.hip: cost + Sinkhorn solve in one launch for documents of <= 8 rows (CSR inputs, CROSS or MAPPED pairing)
bool fused_self_ok(int64_t jobs, const aspire_ot_params* prm);
bool fused_inbox_ok(const aspire_repset* q, const float* diameter);
bool fused_path_ok(const aspire_repset* q, const aspire_repset* c);
// (a pair whose shifted sums leave fp32 range is solved again in the max-shifted form by the wave that finds it: fused.hip, solve_safe)
int launch_pair_fused(const ScoreArgs& a, int64_t groups_bound, const float* qbox, hipStream_t stream);
// split.hip: the same work with streaming waves and solver waves as two roles of one workgroup (where the fused kernel's SELF / QBOX
// forms apply and the launch fills the chip)
#ifdef ASPIRE_EXPERIMENT_SPLIT      // tools/experiments/split/split.hip (round 5's role-split kernel; not part of the product library)
bool split_path_ok(int64_t groups_bound, const aspire_ot_params* prm);
int launch_pair_split(const ScoreArgs& a, hipStream_t stream);
#endif
// CHUNK form: items = four 8-row chunks (chunk_prep_kernel's records in a.grp_rec, their count in a.grp_off[0])
int launch_pair_fused_chunk(const ScoreArgs& a, int64_t items_bound, const float* qbox, hipStream_t stream);
int launch_pair_fused_chunk_l2max(const ScoreArgs& a, int64_t items_bound, hipStream_t stream);
int launch_pair_fused_l2max(const ScoreArgs& a, int64_t groups_bound, hipStream_t stream, bool self = false);
bool tile16_path_ok(const aspire_repset* q, const aspire_repset* c, int pairing);
int launch_pair_tile16_l2max(const ScoreArgs& a, int64_t items_bound, hipStream_t stream);
int launch_pair_tile16(const ScoreArgs& a, float* cost, float* neg, float* diam2, int64_t items_bound, const float* qbox,
                       hipStream_t stream);
int launch_pair_tile16_rec(const ScoreArgs& a, int T, float* cost, float* neg, float* diam2, int64_t items_bound, const float* qbox,
                           hipStream_t stream);
int launch_pair_tile16_rec_l2max(const ScoreArgs& a, int64_t items_bound, hipStream_t stream);

}  // namespace aspire
