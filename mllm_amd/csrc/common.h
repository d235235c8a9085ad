// mllm_amd/csrc/common.h -- shared host/device helpers of libmllm_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/mllm_hip.h"

namespace mllm_hip {

void set_error(const char *what, hipError_t e, const char *file, int line);
void set_error_msg(const char *fmt, ...);      // a formatted message for mllm_hip_last_error() that is not a HIP error (shape / argument refusals)
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

// Measurement / bring-up switches of the launch paths (mllm_hip_set_option, include/mllm_hip.h): a table in the library, never the caller's environment.
// -1 = "not set": the launch path uses its built-in choice.
// first-maximum argmax over the chip (kernels_elem.hip): nparts workgroups leave (value, index) partials, a fold picks the largest value and, among equals, the smallest index
int argmax_parts_launch(const float *x, int n, float *part_val, int *part_idx, int nparts, hipStream_t st);
int argmax_final_launch(const float *part_val, const int *part_idx, int nparts, int *out, hipStream_t st);
// kernels_decode.hip: a whole M = 1 Linear on raw Q4_K rows in one launch; returns 1 when the shape is not covered
int dec_linear_row_q4k(const void *Wraw, const float *x, const float *addend, float *y, int N, int K, hipStream_t st);
int row_fused_launch(const mllm_hip_row_fused &in, hipStream_t st);      // kernels_decode.hip: mllm_hip_row_fused_launch
// kernels_elem.hip: the rotary of mllm_hip_rope_apply, fp32 in place, over rows that repeat one table every `period` rows
int rope_apply_periodic(float *x, int64_t ldx, const float *sin_t, const float *cos_t, int ld_tab, int rows, int period, int H, int D, hipStream_t st);
enum Option { OPT_VISION_BATCH, OPT_TIME_LAYERS, OPT_NO_GUB, OPT_NO_PJB, OPT_PJB_MIN_NS, OPT_ATTN_FLAGS, OPT_ATTN_DS, OPT_HEAD_WPC, OPT_GEMM_ORDER, OPT_NO_LNF, OPT_MERGE_O, OPT_CHAIN_CONT, OPT_GU_PERSIST, OPT_QKV_PERSIST, OPT_COUNT };
int option(Option o);

// ---- device helpers -------------------------------------------------------------------------------------------
__device__ __forceinline__ float h2f(uint16_t h) { return __half2float(__ushort_as_half(h)); }
// fp32 -> fp16 of an ALREADY ROUNDED fp32 value (the reference stores fp32 results, then converts: two roundings).  The empty asm
// pins the fp32 value in a register; without it the compiler folds a preceding fma/add into v_fma_mixlo_f16, which rounds the
// exact result once and differs from the reference in ~2^-13 of the values.
__device__ __forceinline__ uint16_t f2h(float f) {
    asm volatile("" : "+v"(f));
    return __half_as_ushort(__float2half_rn(f));
}

// NOTE: sqrtf() is the correctly rounded square root here (v_sqrt_f32 + refinement); __fsqrt_rn() lowers to the bare 1-ulp
// v_sqrt_f32 on gfx950 and must not be used where the reference calls sqrtf/std::sqrt.
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

// ---- Q8_K of one 256-block held as 4 consecutive values per lane, counted per instruction (the decode form over several blocks at once is
// wave_quant_blocks of kernels_decode.hip, where the reasoning is written out) ----
__device__ __forceinline__ unsigned wave_umax(unsigned v) {      // integer max: DPP folds into v_max_u32 (fmaxf on a DPP move costs a move, a canonicalise and the max)
    v = max(v, (unsigned)MH_DPP(0, (int)v, DPP_QUAD_X1, 0xF));
    v = max(v, (unsigned)MH_DPP(0, (int)v, DPP_QUAD_X2, 0xF));
    v = max(v, (unsigned)MH_DPP(0, (int)v, DPP_HALF_MIRROR, 0xF));
    v = max(v, (unsigned)MH_DPP(0, (int)v, DPP_MIRROR, 0xF));
    v = max(v, (unsigned)MH_DPP(0, (int)v, DPP_BCAST15, 0xA));
    v = max(v, (unsigned)MH_DPP(0, (int)v, DPP_BCAST31, 0xC));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// bits of amax = max |x| over the block (wave-uniform, >= +0) and the value ggml's strict `>` scan keeps: x[first j with |x[j]| == amax].  The first lane holding +amax
// or -amax decides the sign; only a lane that holds both (amax != 0) needs the order of its four elements, behind a wave-uniform branch.
__device__ __forceinline__ float q8k_first_max(const float4 &v, unsigned &abits) {
    float t, hi, lo, am;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(v.x), "v"(v.y), "v"(v.z));
    asm("v_max_f32 %0, %1, %2" : "=v"(hi) : "v"(t), "v"(v.w));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(v.x), "v"(v.y), "v"(v.z));
    asm("v_min_f32 %0, %1, %2" : "=v"(lo) : "v"(t), "v"(v.w));
    asm("v_max_f32 %0, |%1|, |%2|" : "=v"(am) : "v"(hi), "v"(lo));
    abits = wave_umax(__float_as_uint(am));
    const float amax = __uint_as_float(abits);
    const unsigned long long pos = __ballot(hi == amax), neg = __ballot(lo == -amax);
    const unsigned long long both = pos | neg, first = both & (0ull - both);
    unsigned mbits = abits ^ ((neg & first) ? 0x80000000u : 0u);
    if ((pos & neg & first) != 0 && abits != 0) {
        const float a0 = fabsf(v.x), a1 = fabsf(v.y), a2 = fabsf(v.z), a3 = fabsf(v.w);
        const float mine = a0 == amax ? v.x : (a1 == amax ? v.y : (a2 == amax ? v.z : v.w));
        mbits = __float_as_uint(first_flagged(a0 == amax || a1 == amax || a2 == amax || a3 == amax, mine));
    }
    return __uint_as_float(mbits);
}
// min(127, nearest_int(iscale * x)) of the lane's four values, left as the BITS of 12582912 + q (the low byte is q's byte; q as a float is bits - 12582912.0f exactly).
// Identical to the reference for every finite product.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void q8k_round4(const float4 &v, float iscale, uint32_t (&b)[4]) {
    const f32x2_t s = {iscale, iscale}, magic = {12582912.0f, 12582912.0f};
    const f32x2_t m01 = f32x2_t{v.x, v.y} * s + magic, m23 = f32x2_t{v.z, v.w} * s + magic;      // -ffp-contract=off: a product, then a sum (v_pk_mul_f32, v_pk_add_f32)
    const uint32_t top = 0x4B40007Fu;      // bits of 12582912 + 127
    b[0] = min(__float_as_uint(m01.x), top); b[1] = min(__float_as_uint(m01.y), top);
    b[2] = min(__float_as_uint(m23.x), top); b[3] = min(__float_as_uint(m23.y), top);
}
__device__ __forceinline__ uint32_t q8k_bytes(const uint32_t (&b)[4]) {
    return __builtin_amdgcn_perm(b[1], b[0], 0x0c0c0400u) | __builtin_amdgcn_perm(b[3], b[2], 0x04000c0cu);
}
__device__ __forceinline__ int q8k_sum4(uint32_t bytes) { return __builtin_amdgcn_sdot4((int)bytes, 0x01010101, 0, false); }

// expf exactly as the reference's attention gets it from glibc 2.35 libm on an AVX2+FMA x86-64 host (__expf_fma: N = 32 table,
// cubic in double, kd = fma(InvLn2N, x, Shift), r = fma(InvLn2N, x, -(kd - Shift))); oracle/restate.c:orc_expf is the same
// restatement and is checked against libm on ~10^9 arguments.  Finite or -inf arguments only.
// The 2^(i/32) table is read through `T`: kernels stage EXPF_TAB into LDS once (expf_tab_stage) -- indexed straight from constant
// memory every call is a dependent global load (cold after each layer's weight stream) on the softmax's critical path.
static __device__ const uint64_t EXPF_TAB[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
__device__ __forceinline__ uint64_t expf_tab_fetch() { return EXPF_TAB[threadIdx.x & 31]; }   // issue early, store with expf_tab_store
__device__ __forceinline__ void expf_tab_store(uint64_t *lds_tab, uint64_t v) { if (threadIdx.x < 32) lds_tab[threadIdx.x] = v; }
__device__ __forceinline__ float glibc_expf(float x, const uint64_t *T) {
    // the three range cases are selects behind the main path, not early returns: a return per lane is a divergent branch around everything that follows, and the five
    // calls per (row, key tile) of the prefill softmax spent 1,500 cycles per chunk in it (scratch/fa_stamps.py).  Out of range the main path computes a value nobody
    // takes (no operation in it traps).
    const double xd = (double)x, InvLn2N = 0x1.71547652b82fep+5, Shift = 0x1.8p+52;
    double kd = __fma_rn(InvLn2N, xd, Shift);
    const uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd = kd - Shift;
    const double r = __fma_rn(InvLn2N, xd, -kd);
    const double sc = __longlong_as_double((long long)(T[ki & 31] + (ki << 47)));
    const double z = __fma_rn(0x1.c6af84b912394p-20, r, 0x1.ebfce50fac4f3p-13);
    const double r2 = r * r;
    double y = __fma_rn(r, 0x1.62e42ff0c52d6p-6, 1.0);
    y = __fma_rn(z, r2, y);
    float res = (float)(y * sc);
    res = x > 0x1.62e42ep6f ? __int_as_float(0x7f800000) : res;
    res = x < -0x1.9d1d9ep6f ? 0x1p-149f : res;
    res = x < -0x1.9fe368p6f ? 0.0f : res;
    return res;
}
// inclusive max scan over the 64 lanes of a wave (lane i ends with max of lanes 0..i)
__device__ __forceinline__ float wave_scan_max(float v) {
    v = fmaxf(v, MH_DPPF(v, v, 0x111 /* row_shr:1 */, 0xF));
    v = fmaxf(v, MH_DPPF(v, v, 0x112 /* row_shr:2 */, 0xF));
    v = fmaxf(v, MH_DPPF(v, v, 0x114 /* row_shr:4 */, 0xF));
    v = fmaxf(v, MH_DPPF(v, v, 0x118 /* row_shr:8 */, 0xF));
    v = fmaxf(v, MH_DPPF(v, v, DPP_BCAST15, 0xA));
    v = fmaxf(v, MH_DPPF(v, v, DPP_BCAST31, 0xC));
    return v;
}
// value of lane i-1 (lane 0 receives `first`)
__device__ __forceinline__ float wave_shift_up(float v, float first) { return MH_DPPF(first, v, 0x138 /* wave_shr:1 */, 0xF); }

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

// ---- packed operand layout of the Q4_K GEMM (kernels_linear.hip): bytes per (32-row tile, super-block) of the three planes of the ACTIVATION side
constexpr size_t Q4KP_W_PER_BLK = 32 * 512, Q4KP_M_PER_BLK = 32 * 4 * 8, Q4KP_D_PER_BLK = 32 * 8;
__host__ __device__ static inline size_t q4kp_tile_blocks(int rows, int nb) { return (size_t)((rows + 31) / 32) * nb; }
static inline size_t q4kp_bytes(int rows, int K) {
    const size_t tb = q4kp_tile_blocks(rows, K / 256);
    return tb * (Q4KP_W_PER_BLK + Q4KP_M_PER_BLK + Q4KP_D_PER_BLK);
}
// weight side: nibbles (4 KiB), fp16 sub-block scale pairs (512 B), mins operands (1 KiB), (d, dmin) (256 B) per (32-row tile, super-block) = 0.72 B per weight
constexpr size_t Q4KW_Q_PER_BLK = 4096, Q4KW_S_PER_BLK = 512, Q4KW_M_PER_BLK = 1024, Q4KW_D_PER_BLK = 256;
static inline size_t q4kw_bytes(int rows, int K) { return q4kp_tile_blocks(rows, K / 256) * (Q4KW_Q_PER_BLK + Q4KW_S_PER_BLK + Q4KW_M_PER_BLK + Q4KW_D_PER_BLK); }

}  // namespace mllm_hip
