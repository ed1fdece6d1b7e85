/*
 * aspire_hip.h -- C ABI of libaspire_hip.so: the MI355X (gfx950) implementation of Aspire's
 * query-vs-candidate scoring path.
 *
 * The reference (allenai/aspire) is pure Python and has no FFI; the drop-in boundary is the Python
 * call surface of examples/ex_aspire_consent{,_multimatch}.py.  Every entry point below replaces the
 * PyTorch-eager arithmetic of one reference function (cited per function, paths relative to the
 * reference root) and is what a ctypes binding of that function calls.
 *
 * Conventions
 *   - plain C: device pointers + sizes, no torch types.  All pointers are DEVICE pointers unless
 *     the parameter name ends in `_host`.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are asynchronous
 *     on that stream; inputs are borrowed, outputs must be pre-allocated by the caller.
 *   - every function returns an aspire_status; aspire_last_error() gives the message (thread local).
 *   - fp32 arithmetic throughout ("within 1e-4 of the reference CPU path").  D (encoding dim) must
 *     be 768 (BERT-base, `bert_encoding_dim` at ex_aspire_consent.py:31).
 *
 * Rep store layout ("rows + CSR"): sentence reps of many documents are one row-major fp32 matrix
 * rows[total_sents, D]; document k owns rows [start[k], start[k] + len[k]).  A padded reference
 * tensor [B, S, D] is the special case start[k] = k*S.
 */
#ifndef ASPIRE_HIP_H
#define ASPIRE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASPIRE_ABI_VERSION 6

typedef enum {
    ASPIRE_OK = 0,
    ASPIRE_ERR_INVALID_ARG = 1,  /* the reference would raise AssertionError / IndexError          */
    ASPIRE_ERR_UNSUPPORTED = 2,  /* shape outside what the kernels are built for (message says)    */
    ASPIRE_ERR_HIP = 3           /* a HIP runtime call failed                                      */
} aspire_status;

int aspire_abi_version(void);
const char* aspire_last_error(void);
/* number of sentence rows per document the scoring kernels accept (larger -> ASPIRE_ERR_UNSUPPORTED) */
int aspire_max_sents(void);

/* ---------------------------------------------------------------------------------------------
 * A2 + A3  CLS read-out and span mean pooling.
 * Replaces AspireConSent.consent_reps_bert's pooling loop, examples/ex_aspire_consent.py:75-100
 * (original src/learning/facetid_models/disent_models.py:487-535).
 *   hidden    [B, L, D]  final hidden states
 *   tok_idx   flat int32 token positions; slot (b, s) owns tok_idx[span_off[b*S+s] .. span_off[b*S+s+1])
 *   span_off  [B*S + 1]; an empty slot (doc has fewer than S sentences) yields exact zeros
 *   sent_reps [B, S, D]  out: sum of the slot's token rows / max(count, 1)
 *   cls_reps  [B, D]     out (may be NULL): hidden[b, 0, :]
 * ------------------------------------------------------------------------------------------- */
int aspire_span_mean_pool_f32(const float* hidden, int64_t B, int64_t L, int64_t D,
                              const int32_t* tok_idx, const int32_t* span_off, int64_t S,
                              float* sent_reps, float* cls_reps, void* stream);
/* The same pooling written straight into a resident rep store (rows + CSR, no padding rows): slot (b, s) goes to row
 * out_row[b*S+s] of `rows` (device int32 [B*S]; < 0 = document b has no sentence s, nothing is written).  This is the
 * un-padding of caching_encode (src/learning/facetid_models/disent_models.py:363-370: sent_reps[i, :, :num_sents]) and of
 * AspireModel.encode (src/evaluation/utils/models.py:208) done by the kernel's store addresses instead of a host loop,
 * so encoded documents never leave HBM on their way into the candidate pool. */
int aspire_span_mean_pool_rows_f32(const float* hidden, int64_t B, int64_t L, int64_t D,
                                   const int32_t* tok_idx, const int32_t* span_off, int64_t S,
                                   const int32_t* out_row, float* rows, float* cls_reps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A1  BERT-base encoder forward.  Replaces `self.bert_encoder(tokid_tt, token_type_ids=seg_tt,
 * attention_mask=attnmask_tt).last_hidden_state` at examples/ex_aspire_consent.py:72-73 (HuggingFace
 * BertModel: embeddings + LayerNorm, 12 x [QKV, masked softmax attention, output proj + residual +
 * LayerNorm, 768->3072 GELU(erf) 3072->768 + residual + LayerNorm]; the pooler is not computed, the
 * reference never reads it).  fp32 ACCURACY throughout (1e-4 of HuggingFace's fp32 CPU forward, also on weights with a trained
 * checkpoint's outliers: tests/test_gpu_encoder_heavy.py): with `planes` prepared and >= 1024 token rows every GEMM and the
 * attention run on the fp16 matrix pipe over operands held as two fp16 planes (three exact products per term, fp32 sums); otherwise
 * on operands split on the fly / the fp32-input matrix cores.  An activation beyond fp16's range (|x| > 65504) makes the plane path
 * return non-finite rows: check the output and run again after aspire_debug_set("GEMM", "bf16x3") + ("ATTN", "f32") (aspire_amd does).
 *   weights are borrowed device pointers in nn.Linear layout ([out, in] row-major);
 *   w_qkv is query/key/value weights concatenated along `out` ([2304, 768]), b_qkv likewise.
 *   tok_ids / type_ids / attn_mask  int64 [B, L] (type_ids may be NULL = all zero); attn_mask != 0 = real token
 *   hidden_out [B, L, 768]
 *   workspace  device scratch of aspire_bert_workspace_bytes(w, B, L) bytes
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    const float *w_qkv, *b_qkv;      /* [2304, 768], [2304] */
    const float *w_o, *b_o;          /* [768, 768],  [768]  */
    const float *ln1_g, *ln1_b;      /* attention.output.LayerNorm */
    const float *w_ffn1, *b_ffn1;    /* intermediate.dense [ffn, 768], [ffn] */
    const float *w_ffn2, *b_ffn2;    /* output.dense       [768, ffn], [768] */
    const float *ln2_g, *ln2_b;      /* output.LayerNorm */
} aspire_bert_layer;

typedef struct {
    const float *word_emb, *pos_emb, *type_emb; /* [vocab,768] [max_pos,768] [n_types,768] */
    const float *emb_ln_g, *emb_ln_b;
    const aspire_bert_layer* layers;            /* HOST array of n_layers entries (device pointers inside) */
    int32_t n_layers, n_heads, hidden, ffn_dim, vocab, max_pos, n_types;
    float ln_eps;                               /* layer_norm_eps (1e-12) */
    /* NULL, or the nn.Linear weights' pre-split planes: a DEVICE buffer of aspire_bert_planes_bytes(w) bytes filled once by
     * aspire_bert_prepare_planes when the model is loaded (the weights never change; the call synchronises the stream and
     * returns ASPIRE_ERR_UNSUPPORTED for a weight beyond +-1023 or not finite).  With it the forward's GEMMs (from 1024 token
     * rows on) stream pre-split operands: every fp32 value as two fp16 planes h + l (24 significant bits, the weights scaled by
     * 2^6 first), three exact matrix-pipe products per term (h.h' + h.l' + l.h') summed in fp32 -- fp32's accuracy, measured
     * against float64 in tests/test_gpu_encoder.py -- and no conversion work in the main loop; without it every workgroup splits
     * its fp32 tiles into three bf16 planes on the fly (six products per term). */
    const void* planes;
} aspire_bert_weights;

size_t aspire_bert_planes_bytes(const aspire_bert_weights* w);
int aspire_bert_prepare_planes(const aspire_bert_weights* w, void* planes, size_t planes_bytes, void* stream);
size_t aspire_bert_workspace_bytes(const aspire_bert_weights* w, int64_t B, int64_t L);
int aspire_bert_forward_f32(const aspire_bert_weights* w, const int64_t* tok_ids, const int64_t* type_ids,
                            const int64_t* attn_mask, int64_t B, int64_t L, float* hidden_out,
                            void* workspace, size_t workspace_bytes, void* stream);
/* The encoder kernels' sticky per-device status word.  From 6144 token rows on (gfx950 in SPX mode) the two N = 768 GEMMs of a layer
 * carry the LayerNorm in their epilogue: the six column tiles of a 128-row block exchange their rows' moments through device memory
 * and a tile WAITS inside the kernel for its partners.  The wait is bounded (20 ms): a tile that gives up sets
 * ASPIRE_BERT_STATUS_LN_TIMEOUT and the outputs of the forwards in flight are invalid.  aspire_bert_status copies the word to
 * *status_host, SYNCHRONISES `stream`, and clears it; on a non-zero word run the forwards since the last check again with
 * aspire_debug_set("GEMM_LN", "off") (the separate LayerNorm pass; aspire_amd/encoder.py and consent.py do exactly that).  Never seen
 * set outside the fault-injection test (tests/test_gpu_encoder.py); the bound turns a hang under a broken dispatch-order assumption
 * (CU masking, a serialising debugger) into an error. */
#define ASPIRE_BERT_STATUS_LN_TIMEOUT 1
int aspire_bert_status(int32_t* status_host, void* stream);

/* ---------------------------------------------------------------------------------------------
 * caching_score's document-level term (src/learning/facetid_models/disent_models.py:305-307, taken when
 * abs_loss_prop > 0): functional.pairwise_distance(query_cls_reps, cand_cls_reps, p=2.0) = ||q - c + eps||_2 with torch's
 * eps = 1e-6 added to every coordinate of the difference.  (The caller negates and scales it.)
 *   q_cls [Q, 768], c_cls [C, 768]; dist [P] out: P = Q * C (CROSS, pair = q * C + c) or Q == C (PAIRED)
 * ------------------------------------------------------------------------------------------- */
int aspire_cls_l2_f32(const float* q_cls, int64_t Q, const float* c_cls, int64_t C, int64_t D, int pairing, double eps,
                      float* dist, void* stream);

/* cdist formula selection, mirroring torch.cdist's default compute mode (used at
 * pair_distances.py:49 and :167): rows <= 25 on both sides -> direct sqrt(sum (x-y)^2),
 * otherwise the matmul expansion. */
#define ASPIRE_CDIST_AUTO 0
#define ASPIRE_CDIST_DIRECT 1
#define ASPIRE_CDIST_MM 2

/* How documents are paired.  CROSS: every query with every candidate, P = Q*C, pair p = q*C + c
 * (the ranking loops evaluate.py:72-74, pp_gen_nearest.py:182-202).  PAIRED: query p with candidate
 * p, P = Q = C (the reference's batched `compute_distance(query, cand)` signature). */
#define ASPIRE_PAIR_CROSS 0
#define ASPIRE_PAIR_PAIRED 1

/* ---------------------------------------------------------------------------------------------
 * fp16 planes of a rep store's row matrix: the many-query cost tiles (pair_distances.py:48-55 with >= ~8 query
 * documents: torch.cdist of every query sentence against every candidate sentence is a [sum S_c, sum S_q] x 768 GEMM)
 * run on v_mfma_f32_32x32x16_f16 at fp32 accuracy when both rep sets carry their rows a second time as two fp16
 * planes.  Row r is stored as h + l of s_r * (rows[r] - mu): mu = one vector per store (L2 distances do not change
 * under a common shift: the anisotropic common component of sentence embeddings comes off before anything is
 * rounded), s_r = the power of two that puts the row's largest entry into [2^14, 2^15) (any finite fp32 row fits:
 * nothing is rejected), h = fp16(.), l = fp16(. - h): 22+ significant bits per element, three exact matrix-pipe
 * products per term (h.h' + h.l' + l.h') accumulated in fp32 -- the encoder's scheme (aspire_bert_prepare_planes).
 * Layout: 48 k blocks of 16 coordinates; k block kb, row r: 64 bytes = pieces (plane pl, k half kh) at
 * ((kb * plane_rows + r) * 4 + 2 pl + kh) * 16, each the 8 fp16 of plane pl at coordinates 16 kb + 8 kh .. + 7;
 * rows [total_rows, plane_rows) are zero (what the tiles read for the rows a short document does not have).
 * The planes are prepared ONCE per resident store (4 B per element, like the fp32 rows: one more copy in HBM) and
 * per call for queries that are not part of the store, with the STORE's mu: two rep sets can meet in one call iff
 * their `mu` pointers are equal.  A rep set without planes (planes == NULL), or one whose mu differs, takes the
 * kernels that read the fp32 rows -- same results to rounding.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    const void* planes;      /* [48][plane_rows][64 B] */
    const float* row_nrm;    /* [plane_rows]  |rows[r] - mu|^2 (0 for the zero rows) */
    const float* row_iscale; /* [plane_rows]  1 / s_r */
    const float* mu;         /* [D] */
    int64_t total_rows;      /* rows of the fp32 matrix the planes were made of */
    int64_t plane_rows;      /* > total_rows, a multiple of 16 */
} aspire_rep_planes;

/* bytes of the device blob aspire_rep_planes_prepare fills for a matrix of total_rows rows */
size_t aspire_rep_planes_bytes(int64_t total_rows);
/* rows [total_rows, D] -> blob; *out_host (a HOST struct) receives the pointers into it.
 *   mu   device [D] or NULL.  NULL: the mean of a sample of up to 4096 of the rows is formed (deterministic: rows
 *        k * stride) and kept in the blob.  Given: used as it is and out_host->mu == mu -- pass the store's mu when
 *        preparing query rows, and rank 0's mu on every shard of a sharded store: every shard then rounds its rows
 *        around the same point, and wherever the plane tiles run (a shard of >= 128 candidate tiles; a smaller or
 *        uneven last shard takes the kernels that read the fp32 rows, within 5e-5 of the tiles) sharded and
 *        un-sharded scores are the same bits. */
int aspire_rep_planes_prepare(const float* rows, int64_t total_rows, int64_t D, const float* mu, void* blob,
                              size_t blob_bytes, aspire_rep_planes* out_host, void* stream);

typedef struct {
    const float* rows;     /* [total_rows, D] */
    const int32_t* start;  /* [n] first row of each document */
    const int32_t* len;    /* [n] valid sentence rows (abs_lens) */
    int64_t n;             /* number of documents */
    /* Padded extent: if > 0 every document has `ext` readable rows (len[k] <= ext) and pair outputs
     * are laid out [.., ext, ..] with the reference's padding semantics; 0 = no padding (ext = len). */
    int32_t ext;
    /* Host-known upper bound of len[] (lens live on the device; the launcher needs the bound to size
     * the tile).  Ignored when ext > 0.  A document longer than the bound yields a NaN score.
     * Every document needs len >= 1: the reference raises on a document without sentences (torch.max over an
     * empty dimension, pair_distances.py:57) -- lens live on the device, so the HOST layer rejects it
     * (aspire_amd.ops.DeviceRepSet raises ValueError); the library does not look. */
    int32_t max_len;
    /* optional (NULL): fp16 planes of `rows` (aspire_rep_planes above; a HOST pointer, read during the call) */
    const aspire_rep_planes* planes;
    /* optional (NULL): the documents' per-coordinate bounding boxes, device [n][2][D] (min row, max row), as
     * aspire_repset_boxes_f32 forms them.  geomloss's epsilon schedule starts at the diameter of the two documents' joint box;
     * the many-query otAspire calls form every candidate's box per call (a pass over all its rows) unless a resident pool
     * brings them along. */
    const float* doc_box;
} aspire_repset;

/* boxes [n][2][D] of the documents of `set` (len[] >= 1): boxes[k][0] = per-coordinate minimum over document k's rows, [k][1] = maximum */
int aspire_repset_boxes_f32(const aspire_repset* set, int64_t D, float* boxes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A9  tsAspire max-sim.  Replaces allpair_masked_dist_l2max,
 * src/learning/facetid_models/pair_distances.py:138-186.
 *   scores    [P]  out: max over valid (i < q_len, j < c_len) of -||q_i - c_j||   (the "sims";
 *                  the reference's distance output is its negation)
 *   pair_sims [P, q.ext, c.ext] out, optional (NULL): -cdist + pad_mask, pad_mask = -10e8 outside
 *                  the valid block (pair_distances.py:156-170); requires q.ext > 0 and c.ext > 0.
 * ------------------------------------------------------------------------------------------- */
int aspire_l2max_scores_f32(const aspire_repset* q, const aspire_repset* c, int64_t D, int pairing,
                            int cdist_mode, float* scores, float* pair_sims, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sibling aggregations of the same masked -cdist block (score_agg_type 'l2top2' / 'l2attention',
 * src/learning/facetid_models/disent_models.py:238-245):
 *   ASPIRE_AGG_MAX        aspire_l2max_scores_f32 above.
 *   ASPIRE_AGG_TOP2       allpair_masked_dist_l2topk, pair_distances.py:295-345: sum of the two largest entries of
 *                         -cdist + pad_mask over the padded [q.ext, c.ext] block (torch.topk k = 2; with fewer than
 *                         two valid entries a masked one, ~ -1e9, is picked exactly as the reference does; without
 *                         padded extents the missing entry counts as -10e8).
 *   ASPIRE_AGG_ATTENTION  AllPairMaskedAttention.compute_distance, pair_distances.py:95-135 with
 *                         models_common/activations.py:35-61: sum_ij p_ij * (-d_ij), p = soft-max over the valid
 *                         block of -d_ij / temp (temp = cdatt_sm_temp).
 *   scores    [P]  out: the similarity (return_pair_sims=True value); the reference's distance is its negation
 *   pair_sims [P, q.ext, c.ext] out, optional: TOP2: -cdist + pad_mask; ATTENTION: -cdist, unmasked (:125)
 *   pair_softmax [P, q.ext, c.ext] out, optional, ATTENTION only: p_ij (zero outside the valid block)
 * ------------------------------------------------------------------------------------------- */
#define ASPIRE_AGG_MAX 0
#define ASPIRE_AGG_TOP2 1
#define ASPIRE_AGG_ATTENTION 2
int aspire_l2agg_scores_f32(const aspire_repset* q, const aspire_repset* c, int64_t D, int pairing,
                            int cdist_mode, int agg, double temp, float* scores, float* pair_sims,
                            float* pair_softmax, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A5-A8  otAspire.  Replaces AllPairMaskedWasserstein.compute_distance,
 * src/learning/facetid_models/pair_distances.py:21-92 (copy at
 * examples/ex_aspire_consent_multimatch.py:118-189), including the geomloss==0.2.4
 * SamplesLoss("sinkhorn", p=1, blur, reach=None, scaling, debias=False) solver it calls.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    /* doubles: the reference holds these as Python floats and geomloss builds the epsilon schedule
     * from them in float64 (np.log / np.arange / np.exp) before any fp32 arithmetic. */
    double blur;         /* geoml_blur    (default 0.05) */
    double scaling;      /* geoml_scaling (default 0.9)  */
    double sent_sm_temp; /* sent_sm_temp  (default 1.0)  */
    int32_t cdist_mode;  /* ASPIRE_CDIST_*               */
    int32_t flags;       /* ASPIRE_OT_FLAG_* (0 = default) */
} aspire_ot_params;

/* ONE_FORM: score every pair with ONE kernel (the one-workgroup-per-pair long form, documents of 1 .. 128 rows) whatever the
 * size of the call.  By default the grid size selects among kernel families whose summation orders differ, so the same
 * (query, candidate) pair can come back a few 1e-5 apart from a per-query call and from a batched one and near-ties of a
 * ranking may swap; with this flag a pair's score depends on its two documents only -- aspire_ot_rank_f32 per query and
 * aspire_ot_rank_batch_f32 over all queries return the same bits and the same order.  Several times slower.  The max-sim
 * entry points take the same request as ASPIRE_CDIST_ONE_FORM or'ed into cdist_mode. */
#define ASPIRE_OT_FLAG_ONE_FORM 1
#define ASPIRE_CDIST_ONE_FORM 0x100
/* CENTER: the rows share a large common component (anisotropic embedding spaces: mean cosine of 0.5 .. 0.9 between unrelated
 * sentences is common).  The streaming kernels then subtract the mean of the query's rows from every row before forming
 * |x|^2 - 2 x.y + |y|^2 -- L2 distances do not change under a common shift, the expansion stops cancelling, and the
 * direct-formula fix-up that otherwise redoes nearly every entry (5 x slower at mean cosine 0.8) is back to the exception.
 * Same results to rounding (closer to the float64 value than the un-centred expansion).  Honoured by the streaming kernels
 * (fused / chunk forms, the 16-row tiles: centre = mean of the staged query rows) and by the matrix-pipe cost tiles of the
 * many-query calls and the small-pool cost kernels (centre = the tile's / the query's first row: 128 x 16 384 x 12 at mean cosine
 * 0.97 86 -> 8 ms; 1 x 1000 x 12 at 0.8 123 -> 49 us); the 33 .. 128-row kernel and padded reference tensors ignore it.
 * aspire_amd.ops sets it from a sample of the pool. */
#define ASPIRE_OT_FLAG_CENTER 2
#define ASPIRE_CDIST_CENTER 0x200

/* SHARED SENTENCES -- the one place where the product deliberately does NOT reproduce the reference's fp32 bits.  geomloss forms its
 * cost as sqrt(max(|x|^2 - 2 x.y + |y|^2, 1e-8)) (and torch.cdist beyond 25 rows forms -cdist the same way,
 * pair_distances.py:48-56).  Where a candidate sentence (nearly) EQUALS a query sentence the expansion cancels: in fp32 its value is
 * rounding noise of size ~1e-6 (|x|^2 + |y|^2), and the reference returns the square root of that noise -- 1.5e-2 .. 5.4e-2 from its
 * own float64 value on 768-d reps, different for every summation order (another BLAS, another batch size: other bits).  Every kernel
 * family here tests d^2 < 1e-4 (|x|^2 + |y|^2)^2 (the d below which the expansion's error exceeds ~5e-5 ABSOLUTE in the distance --
 * the bar is absolute, hence the squared norm on the right) and takes BOTH -cdist and geomloss's cost of such an entry from the exact
 * sum of squared differences: within 1.3e-5 of the float64 oracle (tests/test_gpu_coincident.py), and therefore up to 5e-2 away from
 * what the fp32 reference happens to return for that pair.  Rankings: a shared sentence gives the pair a cost entry of ~1e-4 instead
 * of ~2e-2, i.e. it scores (correctly) slightly better than the reference scores it.  The rule holds at ANY document length: also
 * where torch.cdist itself would pick its matmul formula (a side beyond 25 rows, ASPIRE_CDIST_MM) a cancelling entry's -cdist and cost
 * come from the exact sum (round 6; before, the kernels left those entries on the expansion, and geomloss's cost with them).  There is
 * no flag that reproduces the reference's noise. */
#define ASPIRE_OT_DISTANCE 0 /* return_pair_sims=False: OT_eps = <a,f> + <b,g>  (positive)          */
#define ASPIRE_OT_PLAN_SIM 1 /* return_pair_sims=True : sum_ij P_ij * neg_ij     (negative)         */
#define ASPIRE_OT_SIMILARITY 2 /* -OT_eps: what AspireModel.get_similarity returns (models.py:197), the ranking key */

/*   diameter   NULL: each pair uses the bounding-box diameter of its own valid rows (what the
 *              reference computes when called with B = 1, src/evaluation/utils/models.py:190-197).
 *              else: pair (q, c) uses diameter[q * ngroups + c / diam_group]
 *              with ngroups = ceil(C / diam_group) in CROSS mode, diameter[p / diam_group] in PAIRED
 *              mode -- geomloss derives ONE epsilon schedule per call from the whole batch
 *              (consecutive groups of 64 candidates in pp_gen_nearest.py:182-196); compute it with
 *              aspire_group_diameter_f32.
 *   scores     [P] out
 *   out_qdistr [P, q.ext], out_cdistr [P, c.ext], out_pairsims / out_plan [P, q.ext, c.ext]:
 *              optional (NULL) extra outputs of return_pair_sims=True (pair_distances.py:86):
 *              query_distr, cand_distr, pair_sims (masked neg L2, pads = 0), transport_plan.
 *              masked_sims = plan * pair_sims is left to the caller.  Require ext > 0.
 *   workspace  device scratch for the per-pair cost matrices handed from the cost kernel to the Sinkhorn
 *              kernel.  aspire_ot_workspace_bytes() gives the size that lets the whole job run in one
 *              pass (capped at 1 GiB); any size that holds one candidate's pairs is accepted and larger jobs
 *              are processed in candidate chunks.
 */
size_t aspire_ot_workspace_bytes(const aspire_repset* q, const aspire_repset* c, int pairing);
int aspire_ot_sinkhorn_f32(const aspire_repset* q, const aspire_repset* c, int64_t D, int pairing,
                           const aspire_ot_params* prm, const float* diameter, int64_t diam_group,
                           int want, float* scores, float* out_qdistr, float* out_cdistr,
                           float* out_pairsims, float* out_plan, void* workspace, size_t workspace_bytes,
                           void* stream);

/* geomloss max_diameter (sinkhorn_divergence.py of geomloss 0.2.4) for batched calls: the L2 norm of
 * the per-coordinate bounding box over ALL rows of the call's x and y tensors, zero pad rows included.
 *   CROSS : diameter[q * ngroups + g] covers query q's rows and the rows of candidates
 *           [g*group, min(C, (g+1)*group)); a zero row joins the box iff the group's candidates have
 *           unequal lengths (caching_score pads them to the group max, disent_models.py:269-281).
 *   PAIRED: diameter[g] covers queries and candidates [g*group, (g+1)*group); with ext > 0 all ext
 *           rows are read (the padded tensors as they are), else len rows.
 */
int aspire_group_diameter_f32(const aspire_repset* q, const aspire_repset* c, int64_t D, int pairing,
                              int64_t group, float* diameter, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A12  rank.  Per-query top-k of a [Q, C] score matrix, descending, ties by ascending candidate
 * index -- the order Python's stable sorted(..., reverse=True) gives (src/evaluation/evaluate.py:76,
 * src/pre_process/pp_gen_nearest.py:266,339).
 *   idx_base  added to every output index (the shard's first global candidate id)
 *   top_scores [Q, k], top_idx [Q, k] out; if C < k the tail is (-inf, -1).
 *   workspace  device scratch of at least aspire_topk_workspace_bytes(Q, C, k) bytes (0 when C <= 4096).
 *   Any C < 2^32 - 1 and any k: pools beyond one 4096-key chunk are ranked by chunk winners (k < 1024) or fully sorted
 *   (k >= 1024: sorted chunks + merge passes), e.g. k = C for the whole-pool sort of evaluate.py:76.
 */
size_t aspire_topk_workspace_bytes(int64_t Q, int64_t C, int64_t k);
int aspire_topk_desc_f32(const float* scores, int64_t Q, int64_t C, int64_t k, int64_t idx_base,
                         float* top_scores, int64_t* top_idx, void* workspace, size_t workspace_bytes,
                         void* stream);

/* ---------------------------------------------------------------------------------------------
 * A5-A8 + A12 in one call: the ranking step of evaluate.py:58-76 / pp_gen_nearest.py:131-204 for a pool --
 * scores [Q, C] exactly as aspire_ot_sinkhorn_f32 (pairing = ASPIRE_PAIR_CROSS, no pair outputs) followed by the
 * per-query rank exactly as aspire_topk_desc_f32 (top_scores, top_idx) or aspire_topk_keys_f32 (keys != NULL).
 * One host call, the kernels queued back to back on `stream`.  (A variant with the rank fused into the Sinkhorn
 * kernel's last workgroup was measured slower than the separate rank kernel -- NOTES.md -- and is not kept.)
 * Workspace: aspire_ot_rank_workspace_bytes(q, c, k).
 * ------------------------------------------------------------------------------------------- */
size_t aspire_ot_rank_workspace_bytes(const aspire_repset* q, const aspire_repset* c, int64_t k);
int aspire_ot_rank_f32(const aspire_repset* q, const aspire_repset* c, int64_t D, const aspire_ot_params* prm,
                       const float* diameter, int64_t diam_group, int want, float* scores, int64_t k,
                       int64_t idx_base, float* top_scores, int64_t* top_idx, uint64_t* keys, void* workspace,
                       size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The per-query loop of evaluate.py:58-76 batched over queries: J independent (query, pool) re-ranks in ONE call.
 * Job j scores the candidates [job_off[j], job_off[j+1]) of `c` -- query j's own pool (evaluate.py:60-62), all pools
 * laid back to back in one CSR rep set -- against query j of `q`, one epsilon schedule per pair
 * (AspireModel.get_similarity, src/evaluation/utils/models.py:190-197), and ranks each pool on its own (stable
 * descending, evaluate.py:76).  Launches on `stream`: ONE scoring launch over all pairs (costs and Sinkhorn solves fused
 * once the batch fills the chip; a pair whose sums leave fp32 range is re-solved in the max-shifted form by the wave that finds
 * it) and one rank launch with a workgroup per job; batches of more than 64 jobs, batches whose documents exceed 8 rows (a launch
 * that sorts the candidates into chunk items) and the small-batch kernel forms put a launch in front that builds the job tables
 * and the query boxes.  Documents of up to 128 rows are accepted (33 .. 128: one workgroup per pair).  Everything runs on the
 * caller's stream: independent calls on different streams (each with its own outputs and workspace) overlap.
 *   q          J query documents (q->n == J), CSR (ext == 0)
 *   c          every job's candidates (c->n == C == job_off[J]), CSR (ext == 0)
 *   job_off    DEVICE int32 [J + 1], non-decreasing, job_off[0] == 0, job_off[J] == C
 *   max_job    host-known upper bound of a job's candidate count (launch geometry only)
 *   scores     [C] out: candidate p against its job's query, as aspire_ot_sinkhorn_f32 would give it (`want`)
 *   k          0: scores only; else top_scores [J, k], top_idx [J, k] out: per job, index = job_base[j] + position in
 *              the job's pool, (-inf, -1) beyond the pool's size; or keys [J, k] (non-NULL: instead of top_scores /
 *              top_idx) in the sortable key form of aspire_topk_keys_f32, for the shard exchange of section 8(e)
 *   job_base   DEVICE int32 [J] or NULL (= 0): global index of the job's first candidate when this rank holds one
 *              contiguous block of every job's pool
 *   workspace  16-byte aligned, aspire_ot_rank_batch_workspace_bytes(q, c, max_job, k) bytes
 * Results equal J separate aspire_ot_rank_f32 calls up to the kernel form the grid size selects (bit for bit when the
 * forms are pinned to the same ones, see aspire_debug_set).
 * ------------------------------------------------------------------------------------------- */
size_t aspire_ot_rank_batch_workspace_bytes(const aspire_repset* q, const aspire_repset* c, int64_t max_job, int64_t k);
int aspire_ot_rank_batch_f32(const aspire_repset* q, const aspire_repset* c, int64_t D, const int32_t* job_off,
                             int64_t max_job, const aspire_ot_params* prm, int want, float* scores, int64_t k,
                             const int32_t* job_base, float* top_scores, int64_t* top_idx, uint64_t* keys, void* workspace,
                             size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The same per-query loop for tsAspire (allpair_masked_dist_l2max, pair_distances.py:138-186; caching_score's 'l2lse'
 * branch, disent_models.py:294-295): J independent (query, pool) re-ranks by max-sim in ONE call.  Arguments as
 * aspire_ot_rank_batch_f32 (no OT parameters; cdist_mode as aspire_l2max_scores_f32); scores [C] = -min L2 over the valid
 * sentence pairs of candidate p and its job's query.  Documents of up to 128 rows.  Launches: tables, then the streaming
 * max-sim kernels (documents of <= 16 rows, from 384 groups of four candidates) or one workgroup per candidate, then the
 * segmented rank.
 * ------------------------------------------------------------------------------------------- */
size_t aspire_l2max_rank_batch_workspace_bytes(const aspire_repset* q, const aspire_repset* c, int64_t max_job, int64_t k);
int aspire_l2max_rank_batch_f32(const aspire_repset* q, const aspire_repset* c, int64_t D, const int32_t* job_off,
                                int64_t max_job, int cdist_mode, float* scores, int64_t k, const int32_t* job_base,
                                float* top_scores, int64_t* top_idx, uint64_t* keys, void* workspace,
                                size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * SURVEY.md 8(e)  shard merge.  The same rank in KEY form for the candidate-pool shards of a multi-GPU job:
 *   aspire_topk_keys_f32    per-query local top-k as sortable 64-bit keys [Q, k]:
 *                           (order-preserving score bits << 32) | (0xFFFFFFFF - global index), 0 = padding.
 *                           One unsigned descending sort of keys from ANY set of shards is the order of Python's
 *                           stable sorted(..., reverse=True) over the un-sharded pool (evaluate.py:76).
 *                           Needs idx_base + C < 2^32 - 1.  Workspace as aspire_topk_desc_f32.
 *   aspire_topk_merge_keys  keys [R, Q, k_in] as an all-gather of R ranks' [Q, k_in] blocks leaves them ->
 *                           top_scores [Q, k], top_idx [Q, k] (global indices; (-inf, -1) padding).
 *                           Limit: R * k_in <= 4096.
 * ------------------------------------------------------------------------------------------- */
int aspire_topk_keys_f32(const float* scores, int64_t Q, int64_t C, int64_t k, int64_t idx_base,
                         uint64_t* keys, void* workspace, size_t workspace_bytes, void* stream);
int aspire_topk_merge_keys(const uint64_t* keys, int64_t R, int64_t Q, int64_t k_in, int64_t k,
                           float* top_scores, int64_t* top_idx, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Diagnostics (tests, bench.py, tuning) -- not part of the surface that replaces reference code.
 *   aspire_debug_set   pin a kernel form / grid: key = "SINKHORN" (wave | block | block-norepair | block16 | block-dense | block-wide), "COST_PATH"
 *                      (mfma | valu), "OT_FORM" (small | tile | fused), "COST1_BLOCKS" (n), "ATTN" (gemm), "GEMM" (f32 | bf16x3 | planes), "GEMM_TILE" (96 | 128 | 64), "GEMM_RING" (2 | 3), "GEMM_LN" (on | off: LayerNorm in / behind the N = 768 GEMMs),
 *                      "FUSED_VALU" (1),
 *                      "FUSED_NOSELF" (1), "FUSED_WAVES" (n), "FUSED_NOSOLVE" (1 | 2: timing only, invalid scores);
 *                      value NULL or "" restores the default.  The same switches are read
 *                      ONCE from the environment (ASPIRE_HIP_<key>) when the library is first used; nothing on the
 *                      launch path reads the environment.
 *   aspire_debug_ot_cost_stage_f32         the cost stage of aspire_ot_sinkhorn_f32 alone (no solve, scores untouched)
 *   aspire_debug_ot_rank_batch_stages_f32  chosen stages of aspire_ot_rank_batch_f32 alone: 1 tables + query boxes (nothing
 *                      for a batch whose scoring kernel derives them itself),
 *                      2 costs, 4 Sinkhorn solves (2 and 4 are ONE kernel in the fused form: either bit launches it),
 *                      8 rank (bench.py times the stages of a pass one by one this way, after a full call has filled
 *                      the workspace)
 * ------------------------------------------------------------------------------------------- */
int aspire_debug_set(const char* key, const char* value);
/* the switch's current value as aspire_debug_set takes it ("" = default) into buf[0 .. len); for scoped pins that restore
 * what was set before them (an ASPIRE_HIP_* environment setting, an enclosing pin) */
int aspire_debug_get(const char* key, char* buf, size_t len);
int aspire_debug_ot_cost_stage_f32(const aspire_repset* q, const aspire_repset* c, int64_t D, int pairing,
                                   const aspire_ot_params* prm, float* scores, void* workspace, size_t workspace_bytes,
                                   void* stream);
int aspire_debug_ot_rank_batch_stages_f32(const aspire_repset* q, const aspire_repset* c, int64_t D, const int32_t* job_off,
                                          int64_t max_job, const aspire_ot_params* prm, int want, float* scores, int64_t k,
                                          float* top_scores, int64_t* top_idx, void* workspace, size_t workspace_bytes,
                                          void* stream, int stages);

/* Measurement aid: one wave spins for wall_us microseconds on `stream` and writes out[0] = shader-clock ticks, out[1] = 100 MHz
 * wall ticks: launched beside a kernel under study (another stream), out[0] / out[1] x 100 MHz is the clock the chip held under
 * it (the matrix-pipe kernels run power-limited well below the nominal 2.4 GHz; rooflines quote both). */
int aspire_debug_clock_probe(long long* out, long long wall_us, void* stream);

/* Cross-lane primitive self test (DPP / permlane forms vs ds_bpermute); out_mismatch_host[16] receives
 * the number of mismatching lanes per check (all 0 = ok; order: xor 1,2,4,8,16,32, row sum, col sum,
 * row max, col max, butterfly 64, butterfly 16).  Synchronous. */
int aspire_selftest_xlane(int* out_mismatch_host);

#ifdef __cplusplus
}
#endif
#endif /* ASPIRE_HIP_H */
