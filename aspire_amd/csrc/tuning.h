// Diagnostic switches of libaspire_hip.so.  Defaults are the product behaviour; the parity tests pin a kernel form to run
// two forms on the same inputs, and tuning experiments pin grids.  The environment (ASPIRE_HIP_*) is read ONCE, on first
// use -- never on the launch path -- and aspire_debug_set() (include/aspire_hip.h) changes a switch at run time.
#pragma once

namespace aspire {

struct Tuning {
    int sinkhorn_form = 0;   // ASPIRE_HIP_SINKHORN: 0 by grid size, 1 wave, 3 block, 4 block-norepair, 5 block16, 6 block-dense, 7 block-wide (lanes per pair of the block form)
    int cost_path = 0;       // ASPIRE_HIP_COST_PATH: 0 by shape, 1 mfma (Gram kernel), 2 valu
    int cost1_blocks = 0;    // ASPIRE_HIP_COST1_BLOCKS: cap on the small-pool cost kernel's workgroups (0 = default)
    int attn_f32 = 0;        // ASPIRE_HIP_ATTN=f32: the fused attention kernel on fp32-input MFMAs (round 2) instead of the fp16-plane form
    int attn_gemm = 0;       // ASPIRE_HIP_ATTN=gemm: three-kernel attention instead of the fused kernel
    int attn_form = 0;       // ASPIRE_HIP_ATTN=p64 (2): round 6's kernel on 64-key tiles, three workgroups per CU (A/B form); ASPIRE_HIP_ATTN=f16x2 (1): round 5's fused kernel (fp32 Q / K / V split into planes inside the attention kernel) instead of round 6's (planes from the QKV GEMM, LDS-DMA staging)
    int gemm_form = 0;       // ASPIRE_HIP_GEMM: 0 default (pre-split operands from 1024 token rows on, else bf16x3), 1 f32 = fp32-input MFMA everywhere, 2 bf16x3 = operands split on the fly, 3 planes = pre-split operands at any size
    int gemm_tile96 = 0;     // ASPIRE_HIP_GEMM_TILE=96: force 128 x 96 GEMM tiles where N allows
    int gemm_probe = 0;      // ASPIRE_HIP_GEMM_PROBE=1: the P-layout GEMM without its MFMAs, 2: without its LDS-DMA (timing probes, wrong results)
    int gemm_ring = 0;       // ASPIRE_HIP_GEMM_RING=22 | 23 | 13 | 14: 10 x (k blocks per stage) + (stages in the LDS ring) of the P-layout GEMM; 113: ring 13 as a persistent tile loop
    int gemm_ln = 0;         // ASPIRE_HIP_GEMM_LN: 0 by size (LayerNorm in the N = 768 GEMMs' epilogue from 48 row tiles on), 1 off = always its own pass, 2 on = always fused
    int gemm_tile = 0;       // ASPIRE_HIP_GEMM_TILE=128 | 64: force 128 x 128 / 128 x 64 tiles in the bf16x3 form (tuning)
    int ot_form = 0;         // ASPIRE_HIP_OT_FORM: otAspire on documents of <= 8 rows: 0 by size, 1 small = small-pool kernels,
                             // 2 tile = throughput cost kernel + block Sinkhorn kernel, 3 fused = both in one launch, 4 chunk, 5 one = one wave per pair,
                             // costs + solve in one launch (the default for a few dozen .. a few thousand pairs of short documents)
    int fused_valu = 0;      // ASPIRE_HIP_FUSED_VALU=1: the fused kernel's dot products as VALU FMAs instead of MFMA (A/B, parity tests)
    int fused_nosolve = 0;   // ASPIRE_HIP_FUSED_NOSOLVE=1: the fused kernel's cost phase alone (timing experiments)
    int fused_noself = 0;    // ASPIRE_HIP_FUSED_NOSELF=1: batches of <= 64 jobs also take the tables launch + the table-driven kernel (A/B)
    int gram_tile = 0;       // ASPIRE_HIP_GRAM_TILE=128128 | 128256 | 256256: candidate x query rows per tile of the fp16-plane cost tiles (0: by shape)
    int gram_pp = 0;         // ASPIRE_HIP_GRAM_PP=1: the 256 x 256 tiles in the ping-pong form (two wave groups a segment apart)
    int gram_ring = 0;       // ASPIRE_HIP_GRAM_RING=4: four-stage ring for the 256 x 256 tiles (default 3)
    int fused_waves = 0;     // ASPIRE_HIP_FUSED_WAVES: cap on the fused kernel's resident waves (0 = default; grid experiments)
    int fused_split = 0;     // ASPIRE_HIP_FUSED_SPLIT: 0 / 2 the fused kernel, 1 the role-split kernel where it applies -- ONLY in the experiment build (tools/experiments/split/build.sh: -DASPIRE_EXPERIMENT_SPLIT); ignored by the product library
    int split_prio = 0;      // ASPIRE_HIP_SPLIT_PRIO: the role-split kernel's solver mode: 0 solve, 3 skip the solves (timing probe: wrong scores)
};

const Tuning& tuning();
// key = the environment variable's name without the ASPIRE_HIP_ prefix (e.g. "SINKHORN"), value as in the environment;
// value NULL or "" restores the default.  Returns false on an unknown key / value.
bool tuning_set(const char* key, const char* value);
// the switch's current value in the form tuning_set takes ("" = default); false on an unknown key / short buffer
bool tuning_get(const char* key, char* buf, unsigned long len);

}  // namespace aspire
