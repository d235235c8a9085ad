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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
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
