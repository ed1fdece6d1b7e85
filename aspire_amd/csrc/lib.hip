// Error reporting, ABI version, and the cross-lane primitive self test.
#include <stdarg.h>
#include <stdio.h>

#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "common.h"
#include "tuning.h"

namespace aspire {
namespace {
thread_local char g_err[512] = "";
Tuning g_tuning;
std::once_flag g_tuning_once;

bool apply_tuning(Tuning& t, const char* key, const char* v) {
    const bool unset = !v || !*v;
    if (!strcmp(key, "SINKHORN")) {
        const int f = unset ? 0 : !strcmp(v, "wave") ? 1 : !strcmp(v, "block") ? 3 : !strcmp(v, "block-norepair") ? 4 : !strcmp(v, "block16") ? 5 : !strcmp(v, "block-dense") ? 6 : !strcmp(v, "block-wide") ? 7 : -1;
        if (f < 0) return false;
        t.sinkhorn_form = f;
    } else if (!strcmp(key, "COST_PATH")) {
        const int f = unset ? 0 : !strcmp(v, "mfma") ? 1 : !strcmp(v, "valu") ? 2 : -1;
        if (f < 0) return false;
        t.cost_path = f;
    } else if (!strcmp(key, "COST1_BLOCKS")) {
        t.cost1_blocks = unset ? 0 : atoi(v);
    } else if (!strcmp(key, "ATTN")) {
        if (!unset && strcmp(v, "gemm") && strcmp(v, "f32") && strcmp(v, "f16x2") && strcmp(v, "p64")) return false;
        t.attn_gemm = !unset && !strcmp(v, "gemm");
        t.attn_f32 = !unset && !strcmp(v, "f32");
        t.attn_form = unset ? 0 : !strcmp(v, "f16x2") ? 1 : !strcmp(v, "p64") ? 2 : 0;
    } else if (!strcmp(key, "GEMM")) {
        const int f = unset ? 0 : !strcmp(v, "f32") ? 1 : !strcmp(v, "bf16x3") ? 2 : !strcmp(v, "planes") ? 3 : -1;
        if (f < 0) return false;
        t.gemm_form = f;
    } else if (!strcmp(key, "GEMM_TILE")) {
        const int f = unset ? 0 : atoi(v);
        if (f != 0 && f != 96 && f != 128 && f != 64 && f != 256) return false;
        t.gemm_tile96 = f == 96;
        t.gemm_tile = f;
    } else if (!strcmp(key, "GEMM_RING")) {
        int f = unset ? 0 : atoi(v);
        if (f == 2 || f == 3) f += 20;        // the round-3 spellings: ring depth at two k blocks per stage
        if (f != 0 && f != 22 && f != 23 && f != 12 && f != 13 && f != 14 && f != 113) return false;
        t.gemm_ring = f;
    } else if (!strcmp(key, "GEMM_LN")) {
        if (!unset && strcmp(v, "off") && strcmp(v, "on")) return false;
        t.gemm_ln = unset ? 0 : !strcmp(v, "off") ? 1 : 2;
    } else if (!strcmp(key, "GEMM_PROBE")) {
        t.gemm_probe = unset ? 0 : atoi(v);
    } else if (!strcmp(key, "FUSED_VALU")) {
        t.fused_valu = unset ? 0 : atoi(v);
    } else if (!strcmp(key, "FUSED_NOSOLVE")) {
        t.fused_nosolve = unset ? 0 : atoi(v);
    } else if (!strcmp(key, "FUSED_NOSELF")) {
        t.fused_noself = unset ? 0 : atoi(v);
    } else if (!strcmp(key, "FUSED_WAVES")) {
        t.fused_waves = unset ? 0 : atoi(v);
    } else if (!strcmp(key, "FUSED_SPLIT")) {
        t.fused_split = unset ? 0 : atoi(v);
    } else if (!strcmp(key, "SPLIT_PRIO")) {
        t.split_prio = unset ? 0 : atoi(v);
    } else if (!strcmp(key, "GRAM_TILE")) {
        const int f = unset ? 0 : atoi(v);
        if (f != 0 && f != 128128 && f != 128256 && f != 256128 && f != 256256) return false;
        t.gram_tile = f;
    } else if (!strcmp(key, "GRAM_PP")) {
        t.gram_pp = unset ? 0 : atoi(v);
    } else if (!strcmp(key, "GRAM_RING")) {
        const int f = unset ? 0 : atoi(v);
        if (f != 0 && f != 3 && f != 4) return false;
        t.gram_ring = f;
    } else if (!strcmp(key, "OT_FORM")) {
        const int f = unset ? 0 : !strcmp(v, "small") ? 1 : !strcmp(v, "tile") ? 2 : !strcmp(v, "fused") ? 3 : !strcmp(v, "chunk") ? 4 : !strcmp(v, "one") ? 5 : -1;
        if (f < 0) return false;
        t.ot_form = f;
    } else {
        return false;
    }
    return true;
}

// the value apply_tuning would take to set the switch to what it is now ("" = default)
bool render_tuning(const Tuning& t, const char* key, char* buf, size_t len) {
    const char* v = nullptr;
    char num[32] = "";
    auto number = [&](int n) {
        if (n != 0) snprintf(num, sizeof(num), "%d", n);
        return (const char*)num;
    };
    if (!strcmp(key, "SINKHORN")) {
        static const char* names[] = {"", "wave", "", "block", "block-norepair", "block16", "block-dense", "block-wide"};
        v = names[t.sinkhorn_form];
    } else if (!strcmp(key, "COST_PATH")) v = t.cost_path == 1 ? "mfma" : t.cost_path == 2 ? "valu" : "";
    else if (!strcmp(key, "COST1_BLOCKS")) v = number(t.cost1_blocks);
    else if (!strcmp(key, "ATTN")) v = t.attn_gemm ? "gemm" : t.attn_f32 ? "f32" : t.attn_form == 1 ? "f16x2" : t.attn_form == 2 ? "p64" : "";
    else if (!strcmp(key, "GEMM")) v = t.gemm_form == 1 ? "f32" : t.gemm_form == 2 ? "bf16x3" : t.gemm_form == 3 ? "planes" : "";
    else if (!strcmp(key, "GEMM_TILE")) v = number(t.gemm_tile);
    else if (!strcmp(key, "GEMM_RING")) v = number(t.gemm_ring);
    else if (!strcmp(key, "GEMM_LN")) v = t.gemm_ln == 1 ? "off" : t.gemm_ln == 2 ? "on" : "";
    else if (!strcmp(key, "GEMM_PROBE")) v = number(t.gemm_probe);
    else if (!strcmp(key, "FUSED_VALU")) v = number(t.fused_valu);
    else if (!strcmp(key, "FUSED_NOSOLVE")) v = number(t.fused_nosolve);
    else if (!strcmp(key, "FUSED_NOSELF")) v = number(t.fused_noself);
    else if (!strcmp(key, "FUSED_WAVES")) v = number(t.fused_waves);
    else if (!strcmp(key, "FUSED_SPLIT")) v = number(t.fused_split);
    else if (!strcmp(key, "SPLIT_PRIO")) v = number(t.split_prio);
    else if (!strcmp(key, "GRAM_TILE")) v = number(t.gram_tile);
    else if (!strcmp(key, "GRAM_RING")) v = number(t.gram_ring);
    else if (!strcmp(key, "GRAM_PP")) v = number(t.gram_pp);
    else if (!strcmp(key, "OT_FORM")) v = t.ot_form == 1 ? "small" : t.ot_form == 2 ? "tile" : t.ot_form == 3 ? "fused" : t.ot_form == 4 ? "chunk" : t.ot_form == 5 ? "one" : "";
    if (!v || strlen(v) + 1 > len) return false;
    strcpy(buf, v);
    return true;
}

void tuning_from_env() {
    static const char* keys[] = {"SINKHORN", "COST_PATH", "COST1_BLOCKS", "ATTN", "GEMM_TILE", "GEMM_RING", "GEMM_LN", "GEMM_PROBE", "GEMM", "OT_FORM", "FUSED_VALU", "FUSED_NOSOLVE", "FUSED_WAVES", "FUSED_SPLIT", "SPLIT_PRIO", "FUSED_NOSELF", "GRAM_TILE", "GRAM_RING", "GRAM_PP"};
    for (const char* k : keys) {
        char name[64];
        snprintf(name, sizeof(name), "ASPIRE_HIP_%s", k);
        if (const char* v = getenv(name)) apply_tuning(g_tuning, k, v);
    }
}
}  // namespace

const Tuning& tuning() {
    std::call_once(g_tuning_once, tuning_from_env);
    return g_tuning;
}
bool tuning_set(const char* key, const char* value) {
    std::call_once(g_tuning_once, tuning_from_env);
    return apply_tuning(g_tuning, key, value);
}
bool tuning_get(const char* key, char* buf, size_t len) {
    std::call_once(g_tuning_once, tuning_from_env);
    return render_tuning(g_tuning, key, buf, len);
}

namespace {
}
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {
// Every fast cross-lane form against ds_bpermute (__shfl_xor) on lane-distinct data.
__global__ void xlane_selftest_kernel(int* mismatch) {
    const int lane = threadIdx.x;
    const float v = (float)(lane * 7 + 3);
    int n = 0;
#define CHECK(expr) atomicAdd(mismatch + (n++), (expr) ? 1 : 0)
    CHECK(lane_xor<1>(v) != __shfl_xor(v, 1));
    CHECK(lane_xor<2>(v) != __shfl_xor(v, 2));
    CHECK(lane_xor<4>(v) != __shfl_xor(v, 4));
    CHECK(lane_xor<8>(v) != __shfl_xor(v, 8));
    CHECK(lane_xor<16>(v) != __shfl_xor(v, 16));
    CHECK(lane_xor<32>(v) != __shfl_xor(v, 32));
    // reductions
    float rs = v;
    for (int m = 1; m < 8; m <<= 1) rs += __shfl_xor(rs, m);
    CHECK(row8_sum(v) != rs);
    float cs = v;
    for (int m = 8; m < 64; m <<= 1) cs += __shfl_xor(cs, m);
    CHECK(col8_sum(v) != cs);
    float rm = v, cm = v;
    for (int m = 1; m < 8; m <<= 1) rm = fmaxf(rm, __shfl_xor(rm, m));
    for (int m = 8; m < 64; m <<= 1) cm = fmaxf(cm, __shfl_xor(cm, m));
    CHECK(row8_max(v) != rm);
    CHECK(col8_max(v) != cm);
    // butterflies: element k of lane l is (l + 1) * (k + 1); sum over lanes = (k + 1) * 2080
    {
        float a[64];
#pragma unroll
        for (int k = 0; k < 64; ++k) a[k] = (float)((lane + 1) * (k + 1));
        CHECK(butterfly_sum<64>(a, lane) != (float)((lane + 1) * 2080));
        float c[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) c[k] = (float)((lane + 1) * (k + 1));
        CHECK(butterfly_sum<32>(c, lane) != (float)(((lane >> 1) + 1) * 2080));
        float b[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) b[k] = (float)((lane + 1) * (k + 1));
        CHECK(butterfly_sum<16>(b, lane) != (float)(((lane >> 2) + 1) * 2080));
    }
#undef CHECK
}
}  // namespace
}  // namespace aspire

using namespace aspire;

extern "C" int aspire_abi_version(void) { return ASPIRE_ABI_VERSION; }

extern "C" int aspire_debug_set(const char* key, const char* value) {
    ASPIRE_REQUIRE(key, ASPIRE_ERR_INVALID_ARG, "null key");
    ASPIRE_REQUIRE(tuning_set(key, value), ASPIRE_ERR_INVALID_ARG, "unknown diagnostic switch %s=%s", key, value ? value : "");
    return ASPIRE_OK;
}
extern "C" int aspire_debug_get(const char* key, char* buf, size_t len) {
    ASPIRE_REQUIRE(key && buf && len > 0, ASPIRE_ERR_INVALID_ARG, "null key / buffer");
    ASPIRE_REQUIRE(tuning_get(key, buf, len), ASPIRE_ERR_INVALID_ARG, "unknown diagnostic switch %s (or buffer too small)", key);
    return ASPIRE_OK;
}
extern "C" const char* aspire_last_error(void) { return g_err; }

namespace aspire {
namespace {
// out[0] = shader-clock ticks (s_memtime), out[1] = 100 MHz wall ticks (s_memrealtime) over ~wall_ticks of spinning by one wave
__global__ void clock_probe_kernel(long long* out, long long wall_ticks) {
    const long long t0 = (long long)__builtin_amdgcn_s_memtime(), r0 = (long long)__builtin_amdgcn_s_memrealtime();
    long long r1 = r0;
    while (r1 - r0 < wall_ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = (long long)__builtin_amdgcn_s_memrealtime();
    }
    if (threadIdx.x == 0) {
        out[0] = (long long)__builtin_amdgcn_s_memtime() - t0;
        out[1] = r1 - r0;
    }
}
}  // namespace
}  // namespace aspire

/* debug: one wave spins for wall_us microseconds on `stream` and reports the clock the shader engines ran at meanwhile
 * (launch it beside the kernel under study, on another stream): out[0] / out[1] x 100 MHz */
extern "C" int aspire_debug_clock_probe(long long* out, long long wall_us, void* stream) {
    ASPIRE_REQUIRE(out, ASPIRE_ERR_INVALID_ARG, "null output");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, wall_us * 100);
    ASPIRE_LAUNCH_OK();
    return ASPIRE_OK;
}

extern "C" int aspire_selftest_xlane(int* out_mismatch_host) {
    ASPIRE_REQUIRE(out_mismatch_host, ASPIRE_ERR_INVALID_ARG, "null output");
    int* d = nullptr;
    ASPIRE_HIP_OK(hipMalloc(&d, 16 * sizeof(int)));
    ASPIRE_HIP_OK(hipMemset(d, 0, 16 * sizeof(int)));
    hipLaunchKernelGGL(xlane_selftest_kernel, dim3(1), dim3(64), 0, 0, d);
    ASPIRE_LAUNCH_OK();
    ASPIRE_HIP_OK(hipMemcpy(out_mismatch_host, d, 16 * sizeof(int), hipMemcpyDeviceToHost));
    ASPIRE_HIP_OK(hipFree(d));
    return ASPIRE_OK;
}
