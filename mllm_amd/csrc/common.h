// mllm_amd/csrc/common.h -- shared host/device helpers of libmllm_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/mllm_hip.h"

namespace mllm_hip {

void set_error(const char *what, hipError_t e, const char *file, int line);
int check_launch(const char *what, const char *file, int line);

#define MH_CHECK(expr)                                                        \
    do {                                                                      \
        hipError_t _e = (expr);                                               \
        if (_e != hipSuccess) {                                               \
            ::mllm_hip::set_error(#expr, _e, __FILE__, __LINE__);             \
            return MLLM_HIP_ERR_HIP;                                          \
        }                                                                     \
    } while (0)
#define MH_LAUNCH_OK(name) ::mllm_hip::check_launch(name, __FILE__, __LINE__)

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// ---- device helpers -------------------------------------------------------------------------------------------
__device__ __forceinline__ float h2f(uint16_t h) { return __half2float(__ushort_as_half(h)); }
__device__ __forceinline__ uint16_t f2h(float f) { return __half_as_ushort(__float2half_rn(f)); }

// nearest_int of ggml (Quantize.hpp:174-180): magic-add, round-to-nearest-even. Explicit _rn ops: no contraction.
__device__ __forceinline__ int nearest_int(float v) {
    float val = __fadd_rn(v, 12582912.0f);
    return (__float_as_int(val) & 0x007fffff) - 0x00400000;
}

template <typename T>
__device__ __forceinline__ T wave_xor(T v, int mask) { return __shfl_xor(v, mask, 64); }

// ---- wavefront reductions on DPP (no LDS crossbar round trips): quad_perm x2, row_half_mirror, row_mirror leave every lane
// with its 16-lane row total; row_bcast15 / row_bcast31 carry it across rows so lane 63 holds the wave total, which
// v_readlane broadcasts.  (`__shfl_xor` lowers to ds_bpermute + s_waitcnt lgkmcnt(0) per step on gfx950.)
#define MH_DPP(old, src, ctrl, rmask) __builtin_amdgcn_update_dpp((old), (src), (ctrl), (rmask), 0xF, false)
constexpr int DPP_QUAD_X1 = 0xB1, DPP_QUAD_X2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_BCAST15 = 0x142, DPP_BCAST31 = 0x143;

#define MH_DPPF(old, v, ctrl, rmask) __int_as_float(MH_DPP(__float_as_int(old), __float_as_int(v), ctrl, rmask))

// sum over aligned groups of 8 lanes, result in every lane of the group (exact for ints)
__device__ __forceinline__ int group8_sum(int v) {
    v += MH_DPP(0, v, DPP_QUAD_X1, 0xF);
    v += MH_DPP(0, v, DPP_QUAD_X2, 0xF);
    v += MH_DPP(0, v, DPP_HALF_MIRROR, 0xF);
    return v;
}
__device__ __forceinline__ int group4_sum(int v) {
    v += MH_DPP(0, v, DPP_QUAD_X1, 0xF);
    v += MH_DPP(0, v, DPP_QUAD_X2, 0xF);
    return v;
}
__device__ __forceinline__ float group4_sum(float v) {
    v += MH_DPPF(0.0f, v, DPP_QUAD_X1, 0xF);
    v += MH_DPPF(0.0f, v, DPP_QUAD_X2, 0xF);
    return v;
}
__device__ __forceinline__ float group8_sum(float v) {
    v += MH_DPPF(0.0f, v, DPP_QUAD_X1, 0xF);
    v += MH_DPPF(0.0f, v, DPP_QUAD_X2, 0xF);
    v += MH_DPPF(0.0f, v, DPP_HALF_MIRROR, 0xF);
    return v;
}
// sum of the 8 per-group values of a wave whose 8-lane groups are each uniform; total valid in lane 63 only
__device__ __forceinline__ float groups_total_lane63(float v) {
    v += MH_DPPF(0.0f, v, DPP_MIRROR, 0xF);
    v += MH_DPPF(0.0f, v, DPP_BCAST15, 0xA);
    v += MH_DPPF(0.0f, v, DPP_BCAST31, 0xC);
    return v;
}
__device__ __forceinline__ float group8_max(float v) {
    v = fmaxf(v, MH_DPPF(v, v, DPP_QUAD_X1, 0xF));
    v = fmaxf(v, MH_DPPF(v, v, DPP_QUAD_X2, 0xF));
    v = fmaxf(v, MH_DPPF(v, v, DPP_HALF_MIRROR, 0xF));
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {   // every lane ends with its 16-lane row's sum
    v += MH_DPPF(0.0f, v, DPP_QUAD_X1, 0xF);
    v += MH_DPPF(0.0f, v, DPP_QUAD_X2, 0xF);
    v += MH_DPPF(0.0f, v, DPP_HALF_MIRROR, 0xF);
    v += MH_DPPF(0.0f, v, DPP_MIRROR, 0xF);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = group16_sum(v);
    v += MH_DPPF(0.0f, v, DPP_BCAST15, 0xA);
    v += MH_DPPF(0.0f, v, DPP_BCAST31, 0xC);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_sum_i(int v) {
    v = group8_sum(v);
    v += MH_DPP(0, v, DPP_MIRROR, 0xF);
    v += MH_DPP(0, v, DPP_BCAST15, 0xA);
    v += MH_DPP(0, v, DPP_BCAST31, 0xC);
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ float wave_max(float v) {
    v = group8_max(v);
    v = fmaxf(v, MH_DPPF(v, v, DPP_MIRROR, 0xF));
    v = fmaxf(v, MH_DPPF(v, v, DPP_BCAST15, 0xA));
    v = fmaxf(v, MH_DPPF(v, v, DPP_BCAST31, 0xC));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#define MH_DPPD_STEP(v, ctrl, rmask)                                                                     \
    {                                                                                                    \
        const int _lo = MH_DPP(0, __double2loint(v), ctrl, rmask), _hi = MH_DPP(0, __double2hiint(v), ctrl, rmask); \
        v += __hiloint2double(_hi, _lo);                                                                 \
    }
__device__ __forceinline__ double wave_sum_d(double v) {
    MH_DPPD_STEP(v, DPP_QUAD_X1, 0xF)
    MH_DPPD_STEP(v, DPP_QUAD_X2, 0xF)
    MH_DPPD_STEP(v, DPP_HALF_MIRROR, 0xF)
    MH_DPPD_STEP(v, DPP_MIRROR, 0xF)
    MH_DPPD_STEP(v, DPP_BCAST15, 0xA)
    MH_DPPD_STEP(v, DPP_BCAST31, 0xC)
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
// value of `mine` in the first lane whose flag is set (wave-uniform result); flag must be set in at least one lane
__device__ __forceinline__ float first_flagged(bool flag, float mine) {
    const unsigned long long mask = __ballot(flag);
    const int src = __ffsll((long long)mask) - 1;
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), src));
}

// 6-bit scale/min unpack of block_q4_K (get_scale_min_k4, ggml QuantizeQ4.cpp:177-184) from the 3 scale dwords.
// Returns sc in the low byte lanes of `sc8[2]` (8 bytes) and mins in `mn8[2]`, the kmask form of VecDotQ4.cpp:231-236.
__device__ __forceinline__ void unpack_q4k_scales(uint32_t u0, uint32_t u1, uint32_t u2, uint32_t sc8[2], uint32_t mn8[2]) {
    const uint32_t kmask1 = 0x3f3f3f3f, kmask2 = 0x0f0f0f0f, kmask3 = 0x03030303;
    mn8[1] = ((u2 >> 4) & kmask2) | (((u1 >> 6) & kmask3) << 4);
    mn8[0] = u1 & kmask1;
    sc8[1] = (u2 & kmask2) | (((u0 >> 6) & kmask3) << 4);
    sc8[0] = u0 & kmask1;
}
__device__ __forceinline__ int byte_of(const uint32_t v[2], int j) { return (v[j >> 2] >> (8 * (j & 3))) & 0xff; }

// 4x int8 dot with int32 accumulate (v_dot4_i32_i8): a, b hold four signed bytes each.
__device__ __forceinline__ int dot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }

}  // namespace mllm_hip
