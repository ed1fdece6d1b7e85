// Shared host/device helpers for libaspire_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/aspire_hip.h"

namespace aspire {

void set_error(const char* fmt, ...);

#define ASPIRE_REQUIRE(cond, code, ...)    \
    do {                                   \
        if (!(cond)) {                     \
            ::aspire::set_error(__VA_ARGS__); \
            return (code);                 \
        }                                  \
    } while (0)

#define ASPIRE_HIP_OK(expr)                                                                 \
    do {                                                                                    \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess) {                                                            \
            ::aspire::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return ASPIRE_ERR_HIP;                                                          \
        }                                                                                   \
    } while (0)

#define ASPIRE_LAUNCH_OK()                                                            \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            ::aspire::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
            return ASPIRE_ERR_HIP;                                                    \
        }                                                                             \
    } while (0)

constexpr int kD = 768;  // bert_encoding_dim, examples/ex_aspire_consent.py:31

// ---------------------------------------------------------------------------------------------
// Cross-lane primitives.  A wave is 64 lanes = 4 DPP rows of 16.  DPP control words (GFX9 encoding):
//   quad_perm [1,0,3,2] = 0xB1   quad_perm [2,3,0,1] = 0x4E   row_shl:n = 0x100+n   row_shr:n = 0x110+n
//   row_ror:n = 0x120+n          row_mirror = 0x140           row_half_mirror = 0x141
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF, bool BOUND = true>
__device__ __forceinline__ float dpp_mov(float old, float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK,
                                           BANK_MASK, BOUND));
}

// v[lane ^ M] for every lane.
template <int M>
__device__ __forceinline__ float lane_xor(float v) {
    static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "xor mask");
    if constexpr (M == 1) {
        return dpp_mov<0xB1>(v, v);
    } else if constexpr (M == 2) {
        return dpp_mov<0x4E>(v, v);
    } else if constexpr (M == 4) {
        // lanes in banks 0,2 (lane%8 < 4) read lane+4; lanes in banks 1,3 read lane-4.
        float t = dpp_mov<0x104, 0xF, 0x5, false>(v, v);
        return dpp_mov<0x114, 0xF, 0xA, false>(t, v);
    } else if constexpr (M == 8) {
        return dpp_mov<0x128>(v, v);
    } else if constexpr (M == 16) {
        const int iv = __builtin_bit_cast(int, v);
        auto r = __builtin_amdgcn_permlane16_swap(iv, iv, false, false);  // r[0]=[r0,r0,r2,r2] r[1]=[r1,r1,r3,r3]
        return __builtin_bit_cast(float, (__lane_id() & 16) ? r[0] : r[1]);
    } else {
        const int iv = __builtin_bit_cast(int, v);
        auto r = __builtin_amdgcn_permlane32_swap(iv, iv, false, false);  // r[0]=[lo,lo] r[1]=[hi,hi]
        return __builtin_bit_cast(float, (__lane_id() & 32) ? r[0] : r[1]);
    }
}

// Exchange for the halving butterfly: returns A' + B' where (A', B') = swap(A, B):
//   M == 32: lanes <32 get A[l] + A[l+32]; lanes >=32 get B[l-32] + B[l]
//   M == 16: even rows get A[l] + A[l^16]; odd rows get B[l^16] + B[l]
template <int M>
__device__ __forceinline__ float swap_add(float a, float b) {
    static_assert(M == 16 || M == 32, "swap width");
    const int ia = __builtin_bit_cast(int, a), ib = __builtin_bit_cast(int, b);
    // NB: copy the two results to ints first.  __builtin_bit_cast applied directly to the vector element
    // r[1] reads element 0 with this clang (ROCm 7.2): r[0] + r[1] silently became r[0] + r[0].
    int x0, x1;
    if constexpr (M == 32) {
        auto r = __builtin_amdgcn_permlane32_swap(ia, ib, false, false);
        x0 = r[0];
        x1 = r[1];
    } else {
        auto r = __builtin_amdgcn_permlane16_swap(ia, ib, false, false);
        x0 = r[0];
        x1 = r[1];
    }
    return __builtin_bit_cast(float, x0) + __builtin_bit_cast(float, x1);
}

// Halving butterfly: every lane holds N partial sums v[0..N) (N = 64, 32 or 16); afterwards lane l
// holds (in the return value) the sum over all 64 lanes of element l >> log2(64/N), replicated over the
// low log2(64/N) lane bits.  Each level halves the live values: the lane-bit-b partner pairs exchange the
// half the other one keeps.  Levels for lane bits 5 and 4 are one v_permlane{32,16}_swap + add per value;
// lower bits are a DPP add per value plus a select.
template <int M>
__device__ __forceinline__ float pair_fold(float keep0, float keep1, int lane) {
    // lanes with bit M clear keep element keep0, the others keep1
    const float t = keep0 + lane_xor<M>(keep0);
    const float u = keep1 + lane_xor<M>(keep1);
    return (lane & M) ? u : t;
}

template <int N>
__device__ __forceinline__ float butterfly_sum(float (&v)[N], int lane) {
    static_assert(N == 64 || N == 32 || N == 16, "sizes used here");
#pragma unroll
    for (int k = 0; k < N / 2; ++k) v[k] = swap_add<32>(v[k], v[k + N / 2]);
#pragma unroll
    for (int k = 0; k < N / 4; ++k) v[k] = swap_add<16>(v[k], v[k + N / 4]);
#pragma unroll
    for (int k = 0; k < N / 8; ++k) v[k] = pair_fold<8>(v[k], v[k + N / 8], lane);
#pragma unroll
    for (int k = 0; k < N / 16; ++k) v[k] = pair_fold<4>(v[k], v[k + N / 16], lane);
    if constexpr (N == 16) {
        float r = v[0];
        r += lane_xor<2>(r);
        return r + lane_xor<1>(r);
    } else {
#pragma unroll
        for (int k = 0; k < N / 32; ++k) v[k] = pair_fold<2>(v[k], v[k + N / 32], lane);
        if constexpr (N == 32) {
            return v[0] + lane_xor<1>(v[0]);
        } else {
            return pair_fold<1>(v[0], v[1], lane);
        }
    }
}

// All-reduce over the 8 lanes that share lane>>3 (the "row" of an 8x8 tile: varies lane bits 0..2).
__device__ __forceinline__ float row8_max(float v) {
    v = fmaxf(v, lane_xor<1>(v));
    v = fmaxf(v, lane_xor<2>(v));
    return fmaxf(v, dpp_mov<0x141>(v, v));  // row_half_mirror: quads already uniform -> acts as xor 4
}
__device__ __forceinline__ float row8_sum(float v) {
    v += lane_xor<1>(v);
    v += lane_xor<2>(v);
    return v + dpp_mov<0x141>(v, v);
}
// All-reduce over the 8 lanes that share lane&7 (the "column": varies lane bits 3..5).
__device__ __forceinline__ float col8_max(float v) {
    v = fmaxf(v, lane_xor<8>(v));
    const int iv = __builtin_bit_cast(int, v);
    auto r = __builtin_amdgcn_permlane16_swap(iv, iv, false, false);
    const int r0 = r[0], r1 = r[1];  // see swap_add: no bit_cast on vector elements
    v = fmaxf(__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1));
    const int iw = __builtin_bit_cast(int, v);
    auto s = __builtin_amdgcn_permlane32_swap(iw, iw, false, false);
    const int s0 = s[0], s1 = s[1];
    return fmaxf(__builtin_bit_cast(float, s0), __builtin_bit_cast(float, s1));
}
__device__ __forceinline__ float col8_sum(float v) {
    v += lane_xor<8>(v);
    v = swap_add<16>(v, v);
    return swap_add<32>(v, v);
}
__device__ __forceinline__ float wave_sum(float v) { return col8_sum(row8_sum(v)); }
__device__ __forceinline__ float wave_max(float v) { return col8_max(row8_max(v)); }

// fp32 -> three bf16 planes, x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (24 mantissa bits;
// both subtractions are exact): two values at a time, each plane as one packed dword (low half = first value).  The operand
// split of the bf16x3 matrix products (encoder.hip: gemm_bf16x3_kernel, gram.hip: pair_gram_kernel<..., X3>).
__device__ __forceinline__ void split3_bf16(float x, float y, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {x, y};
    p1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2));
    const f2 r1 = v - f2{__builtin_bit_cast(float, p1 << 16), __builtin_bit_cast(float, p1 & 0xffff0000u)};
    p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1, bf2));
    const f2 r2 = r1 - f2{__builtin_bit_cast(float, p2 << 16), __builtin_bit_cast(float, p2 & 0xffff0000u)};
    p3 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r2, bf2));
}
// 16-byte load of a read-once stream (candidate rows, hidden states): the non-temporal hint keeps them from displacing what is
// re-read (query rows, tables) -- the fused scoring kernel 108.5 -> 102 us on the same box
__device__ __forceinline__ float4 ld4_stream(const float* p) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}
#endif  // __HIPCC__

}  // namespace aspire
