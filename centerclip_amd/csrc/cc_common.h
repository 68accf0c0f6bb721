// Shared device/host helpers for libcenterclip_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/centerclip_hip.h"

#define CC_WAVE 64

#define CC_LAUNCH_CHECK()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return CC_ERR_HIP;           \
    } while (0)

__host__ __device__ static inline size_t cc_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- order-preserving float <-> int key (for atomicMax on floats of either sign) -------
__device__ __forceinline__ int cc_float_to_ordered_int(float f) {
    int b = __float_as_int(f);
    return b >= 0 ? b : (b ^ 0x7FFFFFFF);
}
__device__ __forceinline__ float cc_ordered_int_to_float(int k) {
    return __int_as_float(k >= 0 ? k : (k ^ 0x7FFFFFFF));
}
// unsigned key whose unsigned order equals the float order (used inside 64-bit arg-reduce keys)
__device__ __forceinline__ unsigned cc_float_to_ordered_uint(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// x combined with the value 16 (32) lanes away, without the LDS crossbar: v_permlane16_swap / v_permlane32_swap (gfx950)
// exchange DPP rows between two copies of the register - afterwards the two registers hold, in every lane, this lane's value
// and its partner's.  The same two operands as `x op __shfl_xor(x, 16)`, so max and (commutative) add give the same bits as
// the ds_bpermute form; a dependent LDS round trip (~100 cycles) becomes two VALU instructions.
__device__ __forceinline__ void cc_lane_xor16_pair(float x, float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void cc_lane_xor32_pair(float x, float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ float cc_rows_max(float x) {          // max over the 4 lanes l15 + 16 * {0, 1, 2, 3}
    float a, b;
    cc_lane_xor16_pair(x, a, b); x = fmaxf(a, b);
    cc_lane_xor32_pair(x, a, b); return fmaxf(a, b);
}
__device__ __forceinline__ float cc_rows_sum(float x) {          // (x[l] + x[l ^ 16]) + the same of the other half
    float a, b;
    cc_lane_xor16_pair(x, a, b); x = a + b;
    cc_lane_xor32_pair(x, a, b); return a + b;
}

__device__ __forceinline__ float cc_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, CC_WAVE);
    return v;
}
// The same sum over the 64 lanes on the DPP path: four in-row steps (quad swaps, half mirror, mirror - all 16 lanes of a
// DPP row then hold the row's sum) + four v_readlane, instead of six dependent ds_bpermute round trips through the LDS
// crossbar.  Another association than cc_wave_sum ((l ^ 1, l ^ 2, ... inside a row, then (r0 + r1) + (r2 + r3)): for the
// tolerance-bound reductions (LayerNorm statistics, norms of feature rows), not where a summation order is pinned.
template <int CTRL>
__device__ __forceinline__ float cc_dpp_f32(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float cc_wave_sum_fast(float v) {
    v += cc_dpp_f32<0xB1>(v);
    v += cc_dpp_f32<0x4E>(v);
    v += cc_dpp_f32<0x141>(v);
    v += cc_dpp_f32<0x140>(v);
    const int iv = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(iv, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(iv, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(iv, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float cc_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, CC_WAVE));
    return v;
}
__device__ __forceinline__ int cc_wave_imax(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, CC_WAVE));
    return v;
}
__device__ __forceinline__ unsigned long long cc_wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long t = __shfl_xor(v, o, CC_WAVE);
        v = t > v ? t : v;
    }
    return v;
}

// token (p, j) -> first float of its W-vector (see cc_token_layout in the public header)
__device__ __forceinline__ const float* cc_token_ptr(const float* x, const cc_token_layout& l, int p, int j) {
    const int b = p % l.B, s = p / l.B;
    const int f = j / l.n, i = j - f * l.n;
    return x + (int64_t)b * l.stride_b + (int64_t)s * l.stride_s + (int64_t)f * l.stride_f + (int64_t)i * l.stride_i;
}
