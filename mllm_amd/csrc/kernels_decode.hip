// mllm_amd/csrc/kernels_decode.hip -- the fused single-token (decode) layer of the Qwen2 / LLaMA style decoder for gfx950.
//
// One decode step of one layer of QWen2Decoder (models/qwen2_vl/modeling_qwen2_vl.hpp:316-325) is 5 launches instead
// of the 14 ops the reference dispatches; every launch is a weight-streaming GEMV whose prologue recomputes, per
// workgroup and redundantly, the small activation-side work the reference does as separate ops:
//
//   dec_qkv     [RMSNorm(x) -> Q8_K]            . Wqkv^T + b               -> qkv fp32            (A9+A4+A1, layer 0: +A8)
//   dec_attn    M-RoPE(q), M-RoPE(k)->fp16 slab, v->fp16 slab, __fa2_decode over the slab in key order, one workgroup per head (A11+A12+A13)
//   dec_oproj   [attn -> Q8_K]                  . Wo^T + x                -> tmp                  (A4+A1+A20)
//   dec_gateup  [RMSNorm(tmp) -> Q8_K]          . (Wgate|Wup)^T, silu(g)*u -> act                 (A9+A4+A1+A14+A20)
//   dec_down    [act -> Q8_K]                   . Wdown^T + tmp           -> x                    (A4+A1+A20)
//   dec_head    [RMSNorm(x) -> Q8_0]            . Wemb^T (tied lm_head, Q4_0 planes) -> logits + per-workgroup argmax
//   dec_next    final argmax (first maximum), token history, advance the device-side step state
//
// The arithmetic of every fused piece is the same sequence of fp32/int operations as the stand-alone launchers in
// kernels_elem/linear/attn.hip (which the prefill path uses) and as the reference (q4k_dot.h, kernels_attn.hip: its
// accumulation order), so decode, prefill and the oracle agree bit for bit.  Per-step scalars (KV length, rotary row, token id) live in device memory, so one captured
// hipGraph replays every step.  Weight loads are issued BEFORE the prologue so the HBM latency of the first rows overlaps
// the prologue's arithmetic (vmcnt counts in issue order: the prologue's own small loads are issued first).
#include <cmath>
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "decode_launch.h"

namespace mllm_hip {

// diagnostic build only (-DMLLM_HIP_STAMPS): wave 0 of every workgroup records s_memrealtime (100 MHz) at named points
#ifndef MLLM_HIP_NT
#define MLLM_HIP_NT 0
#endif
#ifndef MLLM_HIP_GUB
#define MLLM_HIP_GUB 1
#endif
#ifndef MLLM_HIP_EARLY_ROWS
#define MLLM_HIP_EARLY_ROWS 1
#endif
#ifndef MLLM_HIP_GUB_DMA_FIRST
#define MLLM_HIP_GUB_DMA_FIRST 1
#endif
#ifndef MLLM_HIP_PRE_ROWS
#define MLLM_HIP_PRE_ROWS 0
#endif
constexpr bool g_nt = MLLM_HIP_NT != 0;

#if defined(MLLM_HIP_STAMPS) || defined(MLLM_HIP_STAMPS_GUB) || defined(MLLM_HIP_STAMPS_PJB) || defined(MLLM_HIP_STAMPS_CHAIN)
__device__ unsigned long long g_stamps[8192 * 16];
#endif
#ifdef MLLM_HIP_STAMPS
#define STAMP(i)                                                                                         \
    do {                                                                                                 \
        if (threadIdx.x == 0 && (blockIdx.x + gridDim.x * blockIdx.y) < 8192) {                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            g_stamps[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + (i)] = __builtin_amdgcn_s_memrealtime();                           \
            __builtin_amdgcn_sched_barrier(0);                                                           \
        }                                                                                                \
    } while (0)
#define STAMPCLK(i)                                                                                      \
    do {                                                                                                 \
        if (threadIdx.x == 0 && (blockIdx.x + gridDim.x * blockIdx.y) < 8192) {                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            g_stamps[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + (i)] = __builtin_amdgcn_s_memtime();                               \
            __builtin_amdgcn_sched_barrier(0);                                                           \
        }                                                                                                \
    } while (0)
#define STAMPB(i)                                                                                        \
    do {                                                                                                 \
        if ((threadIdx.x & 63) == 0 && (i) < 16) {                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            g_stamps[(4096 + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memrealtime();               \
            __builtin_amdgcn_sched_barrier(0);                                                           \
        }                                                                                                \
    } while (0)
#define STAMPV(i, v) do { if (threadIdx.x == 0 && (blockIdx.x + gridDim.x * blockIdx.y) < 8192) g_stamps[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + (i)] = (v); } while (0)
#define STAMPT(i, t)                                                                                     \
    do {                                                                                                 \
        if (threadIdx.x == (t) && (blockIdx.x + gridDim.x * blockIdx.y) < 8192) {                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            g_stamps[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + (i)] = __builtin_amdgcn_s_memrealtime();                           \
            __builtin_amdgcn_sched_barrier(0);                                                           \
        }                                                                                                \
    } while (0)
#else
#define STAMP(i)
#define STAMPV(i, v)
#define STAMPT(i, t)
#define STAMPB(i)
#endif
// the chain launch's roles (diagnostic build -DMLLM_HIP_STAMPS_CHAIN, scratch/stamps_chain.py): rows 2048 + workgroup, slot 0 entry, 1 exit, 2.. the role's milestones
#if defined(MLLM_HIP_STAMPS_CHAIN)
#define CSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 2048) { __builtin_amdgcn_sched_barrier(0); mllm_hip::g_stamps[(2048 + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define CSTAMP(i)
#endif
// stamps of the gate|up GEMV only (diagnostic build with -DMLLM_HIP_STAMPS_GUB: scratch/stamps_gub.py); the attention's stamps stay silent in that build
// (the same for the down projection with -DMLLM_HIP_STAMPS_PJB: scratch/stamps_pjb.py)
#if defined(MLLM_HIP_STAMPS_PJB)
#define GSTAMP2(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192 && K > 4096) mllm_hip::g_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GSTAMP2(i)
#endif
#if defined(MLLM_HIP_STAMPS_GUB)
#define GSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) mllm_hip::g_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GSTAMP(i)
#endif
}  // namespace mllm_hip
#include "q4k_dot.h"
#include "q40_dot.h"
#include "kernels_attn_core.h"
namespace mllm_hip {

__device__ __forceinline__ float v_expf_dec(float x) {  // same polynomial as kernels_elem.hip v_expf (mllm_v_expf)
    const float r = 0x1.8p23f;
    const float z = __fmaf_rn(x, 0x1.715476p+0f, r);
    const float n = z - r;
    const float b = __fmaf_rn(-n, 0x1.7f7d1cp-20f, __fmaf_rn(-n, 0x1.62e4p-1f, x));
    const uint32_t e = __float_as_uint(z) << 23;
    const float k = __uint_as_float(e + __float_as_uint(1.0f));
    const bool c = fabsf(n) > 126.0f;
    const float u = b * b;
    const float j = __fmaf_rn(__fmaf_rn(__fmaf_rn(0x1.0e4020p-7f, b, 0x1.573e2ep-5f), u, __fmaf_rn(0x1.555e66p-3f, b, 0x1.fffdb6p-2f)), u,
                              0x1.ffffecp-1f * b);
    if (!c) return __fmaf_rn(j, k, k);
    const uint32_t g = (n <= 0.0f) ? 0x82000000u : 0u;
    const float s1 = __uint_as_float(g + 0x7f000000u);
    const float s2 = __uint_as_float(e - g);
    if (fabsf(n) > 192.0f) return s1 * s1;
    return __fmaf_rn(s2, j, s2) * s1;
}

// ---- shared-memory image of one Q8_K quantised activation row ---------------------------------------------------------
// layout in dynamic LDS (16-B aligned pieces): qs[K] | d[K/256] | q8s[K/32] (int32 sums of 32) ; xf[K] scratch (fp32)
struct ActLds {
    int8_t *qs;
    float *d;
    int *q8s;
    float *xf;
    int qstride;    // bytes between the q8 bytes of consecutive 256-blocks (256, or 272 for the bank-staggered image of dec_gateup_blk)
};
__device__ __forceinline__ ActLds carve_act(char *smem, int K) {
    ActLds a;
    a.qs = reinterpret_cast<int8_t *>(smem);
    a.d = reinterpret_cast<float *>(smem + K);
    a.q8s = reinterpret_cast<int *>(smem + K + ((K / 256 * 4 + 15) & ~15));
    a.xf = reinterpret_cast<float *>(smem + K + ((K / 256 * 4 + 15) & ~15) + ((K / 32 * 4 + 15) & ~15));
    a.qstride = 256;
    return a;
}
__host__ __device__ static inline size_t act_lds_bytes(int K, bool with_xf) {
    return (size_t)K + ((K / 256 * 4 + 15) & ~15) + ((K / 32 * 4 + 15) & ~15) + (with_xf ? (size_t)K * 4 : 0) + 64;
}

// one wave quantises NB 256-blocks (block wid + WPB*i, 4 consecutive values per lane) into LDS: quantize_row_q8_K_reference (Quantize.hpp / ggml).
// Written block-parallel (arrays over i) so the NB reduction chains interleave, and counted per instruction: the down projection's prologue quantises 35
// blocks on 8 waves = 2 waves per SIMD, and is bound by VALU issue, not latency (scratch/stamps_pjb.py: 1.68 us of the kernel's 5 at ~106 VALU per block).
//   * |max| is reduced on the value BITS with integer max: DPP folds into v_max_u32 (fmaxf on a DPP move costs a move, a canonicalise and the max);
//   * the SIGNED first maximum (ggml keeps x[j] of the first |x[j]| == amax, in index order) comes from the lanes' max / min: the first lane holding +amax or -amax
//     decides, and only a lane holding both (with amax != 0) needs the element order -- that case takes the per-element path below behind a wave-uniform branch;
//   * the two IEEE divisions (-128 / max, 1 / iscale) of all NB blocks are done once, block i in lane i, instead of NB times on uniform values;
//   * iscale * x and the magic add (nearest_int) are the two-wide v_pk_mul_f32 / v_pk_add_f32; the byte of q IS the low byte of the sum's bits, so the clamp to 127
//     is an unsigned min on the bits, the pack two v_perm + or, and q0+q1+q2+q3 one v_dot4 (identical for every finite input).
// The form for one or two blocks per wave (q|k|v, o, gate|up: K / 256 <= 16 over 8 waves), where the quantiser is a single dependent chain and not an instruction count:
// the per-instruction form below, with its ballots, 64-bit scalar mask arithmetic and uniform branch, is SLOWER there (same-box A/B: Qwen1.5-0.5B 1,688 -> 1,646 tok/s,
// TinyLlama-1.1B 1,562 -> 1,478 with it in every kernel) -- it pays only where a wave quantises three or more blocks (the 8,960-wide down projection).
template <int NB, int WPB>
__device__ __forceinline__ void wave_quant_blocks_chain(const float4 (&v)[NB], int lane, int wid, int nblk, const ActLds &a) {
    float amax[NB], mx[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) amax[i] = fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
#pragma unroll
    for (int i = 0; i < NB; ++i) amax[i] = wave_max(amax[i]);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const float a0 = fabsf(v[i].x), a1 = fabsf(v[i].y), a2 = fabsf(v[i].z), a3 = fabsf(v[i].w);
        const float mine = a0 == amax[i] ? v[i].x : (a1 == amax[i] ? v[i].y : (a2 == amax[i] ? v[i].z : v[i].w));
        mx[i] = first_flagged(a0 == amax[i] || a1 == amax[i] || a2 == amax[i] || a3 == amax[i], mine);
    }
    int qsum[NB];
    uint32_t packed[NB];
    float dd[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const bool nz = amax[i] != 0.0f;
        const float iscale = nz ? -128.0f / mx[i] : 0.0f;
        const int q0 = min(127, nearest_int(iscale * v[i].x)), q1 = min(127, nearest_int(iscale * v[i].y));
        const int q2 = min(127, nearest_int(iscale * v[i].z)), q3 = min(127, nearest_int(iscale * v[i].w));
        dd[i] = nz ? 1.0f / iscale : 0.0f;
        packed[i] = (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
        qsum[i] = q0 + q1 + q2 + q3;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) qsum[i] = group8_sum(qsum[i]);   // 8 lanes = 32 values (bsums[2k] + bsums[2k+1])
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int blk = wid + WPB * i;
        if (blk < nblk) {
            reinterpret_cast<uint32_t *>(a.qs + blk * a.qstride)[lane] = packed[i];
            if ((lane & 7) == 0) a.q8s[blk * 8 + (lane >> 3)] = qsum[i];
            if (lane == 0) a.d[blk] = dd[i];
        }
    }
}

template <int NB, int WPB>
__device__ __forceinline__ void wave_quant_blocks(const float4 (&v)[NB], int lane, int wid, int nblk, const ActLds &a) {
    if constexpr (NB < 3) { wave_quant_blocks_chain<NB, WPB>(v, lane, wid, nblk, a); return; }
    float hi[NB], lo[NB];
    unsigned abits[NB], mbits[NB];      // wave-uniform: bits of amax (>= +0) and of the signed first maximum
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        // v_max3 / v_min3 spelled out: fmaxf() on loaded values is preceded by a canonicalising v_max x, x per operand (IEEE mode)
        float t;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(v[i].x), "v"(v[i].y), "v"(v[i].z));
        asm("v_max_f32 %0, %1, %2" : "=v"(hi[i]) : "v"(t), "v"(v[i].w));
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(v[i].x), "v"(v[i].y), "v"(v[i].z));
        asm("v_min_f32 %0, %1, %2" : "=v"(lo[i]) : "v"(t), "v"(v[i].w));
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        float am;
        asm("v_max_f32 %0, |%1|, |%2|" : "=v"(am) : "v"(hi[i]), "v"(lo[i]));
        abits[i] = wave_umax(__float_as_uint(am));
    }
    bool tangled = false;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const float amax = __uint_as_float(abits[i]);
        const unsigned long long pos = __ballot(hi[i] == amax), neg = __ballot(lo[i] == -amax);
        const unsigned long long both = pos | neg, first = both & (0ull - both);
        mbits[i] = abits[i] ^ ((neg & first) ? 0x80000000u : 0u);
        tangled |= (pos & neg & first) != 0 && abits[i] != 0;
    }
    if (tangled) {      // some lane holds +amax and -amax and is the first to hold either: the element order inside it decides
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const float amax = __uint_as_float(abits[i]);
            const float a0 = fabsf(v[i].x), a1 = fabsf(v[i].y), a2 = fabsf(v[i].z), a3 = fabsf(v[i].w);
            const float mine = a0 == amax ? v[i].x : (a1 == amax ? v[i].y : (a2 == amax ? v[i].z : v[i].w));
            mbits[i] = __float_as_uint(first_flagged(a0 == amax || a1 == amax || a2 == amax || a3 == amax, mine));
        }
    }
    float isc[NB], dd[NB];
    if constexpr (NB >= 3) {
        int mxv = __float_as_int(1.0f);
#pragma unroll
        for (int i = 0; i < NB; ++i) asm("v_writelane_b32 %0, %1, %2" : "+v"(mxv) : "s"((int)mbits[i]), "n"(i));
        const float isv = -128.0f / __int_as_float(mxv), ddv = 1.0f / isv;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            isc[i] = abits[i] ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(isv), i)) : 0.0f;
            dd[i] = abits[i] ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ddv), i)) : 0.0f;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            isc[i] = abits[i] ? -128.0f / __uint_as_float(mbits[i]) : 0.0f;
            dd[i] = abits[i] ? 1.0f / isc[i] : 0.0f;
        }
    }
    int qsum[NB];
    uint32_t packed[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const f32x2_t s = {isc[i], isc[i]}, magic = {12582912.0f, 12582912.0f};
        const f32x2_t m01 = f32x2_t{v[i].x, v[i].y} * s + magic, m23 = f32x2_t{v[i].z, v[i].w} * s + magic;      // -ffp-contract=off: a product, then a sum
        const uint32_t top = 0x4B40007Fu;      // bits of 12582912 + 127
        const uint32_t b0 = min(__float_as_uint(m01.x), top), b1 = min(__float_as_uint(m01.y), top);
        const uint32_t b2 = min(__float_as_uint(m23.x), top), b3 = min(__float_as_uint(m23.y), top);
        packed[i] = __builtin_amdgcn_perm(b1, b0, 0x0c0c0400u) | __builtin_amdgcn_perm(b3, b2, 0x04000c0cu);
        qsum[i] = __builtin_amdgcn_sdot4((int)packed[i], 0x01010101, 0, false);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) qsum[i] = group8_sum(qsum[i]);   // 8 lanes = 32 values (bsums[2k] + bsums[2k+1])
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int blk = wid + WPB * i;
        if (blk < nblk) {
            reinterpret_cast<uint32_t *>(a.qs + blk * a.qstride)[lane] = packed[i];
            a.q8s[blk * 8 + (lane >> 3)] = qsum[i];      // the 8 lanes of a group hold the same sum, every lane the same d: same-address stores, no exec masks
            a.d[blk] = dd[i];
        }
    }
}
__device__ __forceinline__ void wave_quant_to_lds(float4 v, int lane, int blk, const ActLds &a) {
    const float4 vv[1] = {v};
    wave_quant_blocks<1, 4>(vv, lane, blk, blk + 1, a);
}

// RMSNorm of one row (CPURMSNorm.cpp:31-136) by a 256-thread workgroup, then Q8_K into LDS. dim % 256 == 0, dim <= 1024*NV.
// The row is loaded into registers by load_row() BEFORE the caller issues its weight loads (vmcnt retires in issue order).
template <int NV, int WPB>
__device__ __forceinline__ void load_row(float4 (&xv)[NV], const float *__restrict__ x, int dim) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int d = threadIdx.x * 4 + i * WPB * 256;
        // unconditional load of a clamped address + select: a predicated load makes the compiler wait for it inside the branch, which
        // serialises the row's and the norm weights' round trips
        const float4 v = *reinterpret_cast<const float4 *>(x + (d < dim ? d : 0));
        xv[i] = d < dim ? v : make_float4(0, 0, 0, 0);
    }
}
// the activation row and the norm weights together: both loads are issued before either is waited for.  (Written as a select of an
// unconditional load the compiler sinks each load back under its predicate and waits for it there, one round trip after the other;
// the empty asm makes both values live at one point outside any predicate.)
template <int NV, int WPB>
__device__ __forceinline__ void load_rows2(float4 (&xv)[NV], float4 (&wv)[NV], const float *__restrict__ x, const float *__restrict__ w, int dim) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int d = threadIdx.x * 4 + i * WPB * 256, dc = d < dim ? d : 0;
        float4 a = *reinterpret_cast<const float4 *>(x + dc), b = *reinterpret_cast<const float4 *>(w + dc);
        asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
        xv[i] = d < dim ? a : make_float4(0, 0, 0, 0);
        wv[i] = d < dim ? b : make_float4(0, 0, 0, 0);
    }
}
// thread (wid, lane) holds values [(wid + WPB*i)*256 + 4*lane, +4): exactly the 4 values of lane `lane` of quant block wid + WPB*i
template <int NV, int WPB>
__device__ __forceinline__ void wg_rmsnorm_quant(const float4 (&xv)[NV], const float4 (&wv)[NV], int dim, float eps, const ActLds &a, double *red, float *norm_out = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float4 v = xv[i];   // zero beyond dim
        ss += (double)v.x * (double)v.x + (double)v.y * (double)v.y + (double)v.z * (double)v.z + (double)v.w * (double)v.w;
    }
    ss = wave_sum_d(ss);
    if (lane == 0) red[wid] = ss;
    __syncthreads();
    ss = red[0];
#pragma unroll
    for (int w = 1; w < WPB; ++w) ss += red[w];
    const float m = (float)(ss / (double)dim);
    const float inv = 1.0f / sqrtf(m + eps);
    const int nblk = dim >> 8;
    float4 o[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        o[i].x = (xv[i].x * inv) * wv[i].x; o[i].y = (xv[i].y * inv) * wv[i].y; o[i].z = (xv[i].z * inv) * wv[i].z; o[i].w = (xv[i].w * inv) * wv[i].w;
        if (norm_out && wid + WPB * i < nblk) *reinterpret_cast<float4 *>(norm_out + (wid + WPB * i) * 256 + lane * 4) = o[i];      // the RMSNORM Op's own output (adapter runs)
    }
    wave_quant_blocks<NV, WPB>(o, lane, wid, nblk, a);
    __syncthreads();
}

__device__ __forceinline__ uint4 ld_nt(const uint4 *p) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 v = g_nt ? __builtin_nontemporal_load(reinterpret_cast<const u4 *>(p)) : *reinterpret_cast<const u4 *>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// ---- GEMV core: this wave computes ROWS consecutive rows of a Q4_K matrix against the LDS activation row ----------------
template <int NSTEPS, int ROWS>
struct RowLoads { uint4 hdr[ROWS][NSTEPS], q[ROWS][NSTEPS]; };

template <int NSTEPS, int ROWS, int R0 = 0, int R1 = ROWS>
__device__ __forceinline__ void issue_rows(RowLoads<NSTEPS, ROWS> &L, const uint8_t *__restrict__ W, int nb, const int *rows, int lane) {
    const int g = lane >> 3, qoff = 16 * (lane & 7);          // decode-order rows: the lane's 16 bytes = column class lane & 7
#pragma unroll
    for (int rr = R0; rr < R1; ++rr)
#pragma unroll
        for (int st = 0; st < NSTEPS; ++st) {
            const int blk = st * 8 + g < nb ? st * 8 + g : 0;
            const uint8_t *wb = W + ((int64_t)rows[rr] * nb + blk) * 144;
            // streamed once per token: non-temporal (keeps L2/MALL for the activations and partials that are re-read)
            L.hdr[rr][st] = ld_nt(reinterpret_cast<const uint4 *>(wb));
            L.q[rr][st] = ld_nt(reinterpret_cast<const uint4 *>(wb + 16 + qoff));
        }
}

// out[rr] is wave-uniform.  tab: this wave's chain table (q4k_dot.h) inside the workgroup's dynamic LDS, after the activation image
template <int NSTEPS, int ROWS>
__device__ __forceinline__ void dot_rows(const RowLoads<NSTEPS, ROWS> &L, const ActLds &a, int nb, int lane, float2 *tab, float (&out)[ROWS]) {
    Q4KActC<NSTEPS> A;
    q4kc_load_act<NSTEPS>(A, a.qs, a.d, a.q8s, nb, lane);
    q4kc_dot_rows<NSTEPS, ROWS>(L.hdr, L.q, A, nb, lane, tab, out);
}
template <int NSTEPS, int ROWS>
__device__ __forceinline__ float2 *wave_tab(char *smem, int K, bool with_xf, int wid) {
    return reinterpret_cast<float2 *>(smem + ((act_lds_bytes(K, with_xf) + 15) & ~(size_t)15) + (size_t)wid * q4k_tab_bytes(NSTEPS, ROWS));
}
template <int NSTEPS, int ROWS>
static inline size_t fused_lds_bytes(int K, bool with_xf, int wpb) {
    return ((act_lds_bytes(K, with_xf) + 15) & ~(size_t)15) + (size_t)wpb * q4k_tab_bytes(NSTEPS, ROWS);
}

// what dec_qkv needs to touch the layer's cache rows ahead of the attention launch (kslab == nullptr: off)
struct KvWarm {
    static constexpr int PER_THREAD = 4;      // 16-byte requests per thread: 16 workgroups x 512 threads x 4 cover K and V of one head up to ~1000 keys
    const uint16_t *kslab, *vslab;
    uint32_t *sink;
    int Hkv, D, ldk, vt_ld, cache_limit;
};
// ------------------------------------------------------------------------------------------------------------------------
// dec_qkv: x (or the embedding row of state->token for layer 0) -> RMSNorm -> Q8_K -> Wqkv rows (+bias) -> qkv fp32
// ------------------------------------------------------------------------------------------------------------------------
// PAIRS (the merged q|k|v + attention + o-projection launch): the rows go out as {value, DecodeState::serial} pairs for the attention's workgroups of the same launch;
// wg / grid = this role's workgroup index and count (the stand-alone kernel passes its own block index and grid size)
// XPOLL (four-role launch): the input row itself arrives as pairs from the previous layer's down-projection role (xpairs); the weight rows go out first
// LOOP (stand-alone launches of long rows with a grid smaller than the row count): the workgroup walks further row groups, `wave_stride` waves apart, behind its one prologue
template <int NSTEPS, int ROWS, bool EMBED, int NV, int WPB, bool PAIRS, bool XPOLL = false, bool LOOP = false>
__device__ __forceinline__ void dec_qkv_body(const DecodeState *__restrict__ state, const float *__restrict__ x, float *__restrict__ x_out,
                                             const uint8_t *__restrict__ emb_qs, const uint16_t *__restrict__ emb_d, int vocab, const float *__restrict__ norm_w, float eps,
                                             const uint8_t *__restrict__ W, const float *__restrict__ bias, float *__restrict__ y, unsigned long long *__restrict__ ypairs, int N, int K,
                                             KvWarm kw, int wg, int grid, const unsigned long long *__restrict__ xpairs = nullptr, int *__restrict__ poll_err = nullptr,
                                             int wave_stride = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[WPB];
    const ActLds a = carve_act(smem, K);
    const int lane = threadIdx.x & 63, nb = K >> 8;
    typename std::conditional<LOOP, int, const int>::type wave = wg * WPB + (threadIdx.x >> 6);
    const unsigned serial = PAIRS ? (unsigned)state->serial : 0u;
    const int T_warm = kw.kslab ? state->T : 0;      // a scalar load (uniform address): in flight under the prologue, waited for on lgkmcnt only
    int rows[ROWS];
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) rows[rr] = min(wave * ROWS + rr, N - 1);
    RowLoads<NSTEPS, ROWS> L;
    float4 xv[NV];
    if (EMBED) {
        // CPUEmbedding (Q4_0 row dequant) of the current token: thread t owns values 4t..4t+3 (+1024 i) like load_row();
        // workgroup 0 also stores the row to x_out (it is the residual the o-projection adds back)
        int id = state->token;
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const uint8_t *q = emb_qs + (int64_t)id * (K / 2);
        const uint16_t *dd = emb_d + (int64_t)id * (K / 32);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int d0 = threadIdx.x * 4 + i * WPB * 256;
            float4 v = make_float4(0, 0, 0, 0);
            if (d0 < K) {
                const int blk = d0 >> 5, jj = d0 & 31;           // 4 consecutive values of one 32-block: all low or all high nibbles
                const float d = h2f(dd[blk]);
                const uint32_t b4 = *reinterpret_cast<const uint32_t *>(q + blk * 16 + (jj & 15));
                const int sh = jj >= 16 ? 4 : 0;
                v.x = (float)((int)((b4 >> sh) & 0xF) - 8) * d;
                v.y = (float)((int)((b4 >> (8 + sh)) & 0xF) - 8) * d;
                v.z = (float)((int)((b4 >> (16 + sh)) & 0xF) - 8) * d;
                v.w = (float)((int)((b4 >> (24 + sh)) & 0xF) - 8) * d;
                if (wg == 0) *reinterpret_cast<float4 *>(x_out + d0) = v;
            }
            xv[i] = v;
        }
    }
    float4 wv[NV];
    if constexpr (XPOLL) {
        static_assert(NSTEPS == 1 && NV == 1 && !EMBED, "the polled form: short rows, one quant block per wave");
        issue_rows<NSTEPS, ROWS>(L, W, nb, rows, lane);
        load_row<NV, WPB>(wv, norm_w, K);
        __builtin_amdgcn_sched_barrier(0);
        const int d = threadIdx.x * 4;
        xv[0] = make_float4(0, 0, 0, 0);
        if (d < K) {
            const unsigned long long *p = xpairs + d;
            unsigned long long e0, e1, e2, e3;
            int polls = 0;
            for (;;) {
                e0 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); e1 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                e2 = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); e3 = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(e0 >> 32) == serial && (unsigned)(e1 >> 32) == serial && (unsigned)(e2 >> 32) == serial && (unsigned)(e3 >> 32) == serial) break;
                if (++polls > (1 << 18)) { __hip_atomic_store(poll_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            xv[0] = make_float4(__uint_as_float((unsigned)e0), __uint_as_float((unsigned)e1), __uint_as_float((unsigned)e2), __uint_as_float((unsigned)e3));
        } else wv[0] = make_float4(0, 0, 0, 0);
        CSTAMP(2);
        wg_rmsnorm_quant<NV, WPB>(xv, wv, K, eps, a, red);
        CSTAMP(3);
    } else {
    if (EMBED) load_row<NV, WPB>(wv, norm_w, K);
    else load_rows2<NV, WPB>(xv, wv, x, norm_w, K);
    __builtin_amdgcn_sched_barrier(0);
#if MLLM_HIP_EARLY_ROWS
    if (NSTEPS == 1) { issue_rows<NSTEPS, ROWS>(L, W, nb, rows, lane); __builtin_amdgcn_sched_barrier(0); }
#endif
    wg_rmsnorm_quant<NV, WPB>(xv, wv, K, eps, a, red);
    if (!(MLLM_HIP_EARLY_ROWS && NSTEPS == 1)) issue_rows<NSTEPS, ROWS>(L, W, nb, rows, lane);   // long rows after the prologue: see dec_gateup_kernel
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMPV(15, (unsigned long long)(__builtin_amdgcn_s_getreg(0x1814) & 15));
    // Warm THIS XCD's L2 with the cache rows the attention kernel -- the next launch -- reads through one CU per head: its workgroups for K/V head k carry
    // blockIdx.x % 8 == k % 8 (dec_attn_grid), and so do the workgroups that touch head k's rows here; under the dispatcher's round-robin both land on one XCD.
    // A speed matter only: nothing depends on where a workgroup runs.  The requests go out behind the weight rows and are consumed (xor) at the very end.
    uint32_t warm = 0;
    uint4 wq[KvWarm::PER_THREAD];
    const int kvh_w = wg & 7;
    const bool warms = kw.kslab && kvh_w < kw.Hkv && T_warm > 0;
    if (warms) {
        const int rank = wg >> 3, nrank = (grid - kvh_w + 7) >> 3, T = min(T_warm, kw.cache_limit - 1);
        const int rowk = kw.D * 2 / 16, nk = T * rowk, rowv = (T + 7) >> 3, nv = kw.D * rowv;
#pragma unroll
        for (int i = 0; i < KvWarm::PER_THREAD; ++i) {
            const int v = (rank + i * nrank) * (64 * WPB) + (int)threadIdx.x;
            const uint16_t *src = kw.kslab;      // an address that always exists
            if (v < nk) src = kw.kslab + (int64_t)(v / rowk) * kw.ldk + kvh_w * kw.D + (v % rowk) * 8;
            else if (v - nk < nv) src = kw.vslab + (int64_t)(kvh_w * kw.D + (v - nk) / rowv) * kw.vt_ld + ((v - nk) % rowv) * 8;
            wq[i] = *reinterpret_cast<const uint4 *>(src);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LOOP) {
        static_assert(!PAIRS && !XPOLL, "the walking form is the stand-alone kernel's");
        for (;;) {
            float out[ROWS];
            dot_rows<NSTEPS, ROWS>(L, a, nb, lane, wave_tab<NSTEPS, ROWS>(smem, K, false, threadIdx.x >> 6), out);
            const int cur = wave;
            wave += wave_stride;
            const bool more = wave_stride > 0 && wave * ROWS < N;
            if (more) {      // the next group's rows go out before this group's results are written
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) rows[rr] = min(wave * ROWS + rr, N - 1);
                issue_rows<NSTEPS, ROWS>(L, W, nb, rows, lane);
            }
            if (lane == 63) {
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) {
                    const int rw = cur * ROWS + rr;
                    if (rw < N) y[rw] = bias ? out[rr] + bias[rw] : out[rr];
                }
            }
            if (!more) break;
        }
    } else {
    float out[ROWS];
    dot_rows<NSTEPS, ROWS>(L, a, nb, lane, wave_tab<NSTEPS, ROWS>(smem, K, false, threadIdx.x >> 6), out);
    if (lane == 63) {
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            const int rw = wave * ROWS + rr;
            if (rw < N) {
                const float r = bias ? out[rr] + bias[rw] : out[rr];
                if constexpr (PAIRS) __hip_atomic_store(ypairs + rw, ((unsigned long long)serial << 32) | (unsigned long long)__float_as_uint(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else y[rw] = r;
            }
        }
    }
    }
    if (warms) {
#pragma unroll
        for (int i = 0; i < KvWarm::PER_THREAD; ++i) warm ^= wq[i].x ^ wq[i].y ^ wq[i].z ^ wq[i].w;
        if (kw.sink) kw.sink[wg * (64 * WPB) + threadIdx.x] = warm;      // never set: the loads need a consumer the compiler can see
    }
}

template <int NSTEPS, int ROWS, bool EMBED, int NV, int WPB>
__global__ __launch_bounds__(64 * WPB) void dec_qkv_kernel(const DecodeState *__restrict__ state, const float *__restrict__ x, float *__restrict__ x_out,
                                                      const uint8_t *__restrict__ emb_qs, const uint16_t *__restrict__ emb_d, int vocab,
                                                      const float *__restrict__ norm_w, float eps, const uint8_t *__restrict__ W,
                                                      const float *__restrict__ bias, float *__restrict__ y, int N, int K, KvWarm kw) {
    dec_qkv_body<NSTEPS, ROWS, EMBED, NV, WPB, false>(state, x, x_out, emb_qs, emb_d, vocab, norm_w, eps, W, bias, y, nullptr, N, K, kw, (int)blockIdx.x, (int)gridDim.x);
}
// the walking form of the same kernel (long rows: launch_norm_gemv with NS >= 2)
template <int NSTEPS, int ROWS, int NV, int WPB>
__global__ __launch_bounds__(64 * WPB) void dec_qkv_walk_kernel(const DecodeState *__restrict__ state, const float *__restrict__ x, float *__restrict__ x_out,
                                                           const uint8_t *__restrict__ emb_qs, const uint16_t *__restrict__ emb_d, int vocab,
                                                           const float *__restrict__ norm_w, float eps, const uint8_t *__restrict__ W,
                                                           const float *__restrict__ bias, float *__restrict__ y, int N, int K, KvWarm kw) {
    dec_qkv_body<NSTEPS, ROWS, false, NV, WPB, false, false, true>(state, x, x_out, emb_qs, emb_d, vocab, norm_w, eps, W, bias, y, nullptr, N, K, kw, (int)blockIdx.x, (int)gridDim.x,
                                                                   nullptr, nullptr, (int)gridDim.x * WPB);
}

// ------------------------------------------------------------------------------------------------------------------------
// dec_gateup: tmp -> RMSNorm -> Q8_K -> rows (gate n, up n) -> act[n] = silu(gate) * up   (gate rows [0,I), up rows [I,2I))
// ------------------------------------------------------------------------------------------------------------------------
template <int NSTEPS, int PAIRS, int NV, int WPB>
__global__ __launch_bounds__(64 * WPB) void dec_gateup_kernel(const float *__restrict__ x, const float *__restrict__ norm_w, float eps,
                                                         const uint8_t *__restrict__ W, float *__restrict__ act, int I, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[WPB];
    const ActLds a = carve_act(smem, K);
    const int lane = threadIdx.x & 63, nb = K >> 8;
    int wave = blockIdx.x * WPB + (threadIdx.x >> 6);
    int rows[2 * PAIRS];
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
        const int n = min(wave * PAIRS + p, I - 1);
        rows[2 * p] = n;
        rows[2 * p + 1] = I + n;
    }
    RowLoads<NSTEPS, 2 * PAIRS> L;
    float4 xv[NV], wv[NV];
    STAMP(0);
    load_rows2<NV, WPB>(xv, wv, x, norm_w, K);
    // The activation row is tiny and L2-resident; the weight rows saturate the CU's memory queue for ~2 us.  Issuing them
    // first makes every wave of the workgroup sit in load issue while the prologue's barriers wait for it, so only PRE rows
    // go out before the prologue and the rest right after it (measured with the stamp build: 5.3 -> see DESIGN.md).
    constexpr int PRE = MLLM_HIP_PRE_ROWS < 2 * PAIRS ? MLLM_HIP_PRE_ROWS : 2 * PAIRS;
    issue_rows<NSTEPS, 2 * PAIRS, 0, PRE>(L, W, nb, rows, lane);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(1);
    STAMP(2);
    wg_rmsnorm_quant<NV, WPB>(xv, wv, K, eps, a, red);
    STAMP(3);
    issue_rows<NSTEPS, 2 * PAIRS, PRE, 2 * PAIRS>(L, W, nb, rows, lane);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(4);
    // A grid smaller than the row count (option gu_persist: workgroups per CU) makes the workgroup walk further row groups behind the one prologue: the Q8_K image of the
    // row is made once per workgroup instead of once per eight row pairs.  With the full grid the loop runs once.
    const int stride = (int)gridDim.x * WPB;
    for (;;) {
        float out[2 * PAIRS];
        dot_rows<NSTEPS, 2 * PAIRS>(L, a, nb, lane, wave_tab<NSTEPS, 2 * PAIRS>(smem, K, false, threadIdx.x >> 6), out);
        STAMP(5);
        const int cur = wave;
        wave += stride;
        const bool more = wave * PAIRS < I;
        if (more) {      // the next group's rows go out before this group's results are written
#pragma unroll
            for (int p = 0; p < PAIRS; ++p) {
                const int n = min(wave * PAIRS + p, I - 1);
                rows[2 * p] = n;
                rows[2 * p + 1] = I + n;
            }
            issue_rows<NSTEPS, 2 * PAIRS>(L, W, nb, rows, lane);
        }
        if (lane == 63) {
#pragma unroll
            for (int p = 0; p < PAIRS; ++p) {
                const int n = cur * PAIRS + p;
                if (n < I) {
                    const float g = out[2 * p], u = out[2 * p + 1];
                    act[n] = (g / (1.0f + v_expf_dec(0.0f - g))) * u;   // mllm_v_silu then F_TTMUL
                }
            }
        }
        if (!more) break;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// dec_gateup_blk: the same step with ONE LANE PER SUPER-BLOCK.  The kernels above are VALU-bound (weights resident in L2 run no
// faster than cold ones): with 8 lanes per super-block every lane repeats the header unpack and the table bookkeeping for 32 weights.
// Here a wave takes PAIRS gate rows and PAIRS up rows (2 * PAIRS * K/256 <= 64 super-blocks), streams them -- contiguous in memory --
// into LDS by LDS-DMA, and lane l keeps super-block l whole: one header unpack and one (scale, value) table row per 256 weights,
// 7 VALU operations per nibble dword (and, shift, and, 2 dot4, 2 24-bit mads).  The chain walk and the table layout are q4k_dot.h's.
// The activation image is staggered (272-byte block stride) so that the 6 distinct blocks read by one ds_read_b128 hit distinct banks.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int GUB_QSTRIDE = 272;
__host__ __device__ static inline size_t gub_act_bytes(int K) {
    return (size_t)(K / 256) * GUB_QSTRIDE + ((K / 256 * 4 + 15) & ~15) + ((K / 32 * 4 + 15) & ~15) + 64;
}
__host__ __device__ static inline size_t gub_half_bytes(int pairs, int nb) { return ((size_t)pairs * nb * 144 + 1023) & ~(size_t)1023; }
__host__ __device__ static inline size_t gub_wave_bytes(int pairs, int nb) { return 2 * gub_half_bytes(pairs, nb) + (size_t)2 * pairs * nb * Q4K_SLOTS * 8 + 64; }
static inline size_t gub_lds_bytes(int K, int pairs, int wpb) { return ((gub_act_bytes(K) + 15) & ~(size_t)15) + (size_t)wpb * gub_wave_bytes(pairs, K / 256); }

__device__ __forceinline__ void glds16_dec(const void *gsrc, unsigned lds_dst_in) {
    unsigned keep;
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane(lds_dst_in);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// One super-block (144 bytes in LDS, disk layout) against block b of the staggered activation image: the 12 (scale, value) table entries of
// q4k_dot.h -- slots 0..7 = classes [0,4,2,6,1,5,3,7], 8..11 = mins [0,2,1,3] -- written as float2, the type q4k_chain reads (a float4 store
// would be a different TBAA base and may be moved across the fence).
__device__ __forceinline__ void blk_emit(const char *blk, const ActLds &a, int b, float2 *e) {
    const uint4 hdr = *reinterpret_cast<const uint4 *>(blk);
    const float d = h2f((uint16_t)(hdr.x & 0xffff)), dmin = h2f((uint16_t)(hdr.x >> 16));
    uint32_t sc8[2], mn8[2];
    unpack_q4k_scales(hdr.y, hdr.z, hdr.w, sc8, mn8);
    const int8_t *xb = a.qs + b * a.qstride;
    int cls[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int sl = byte_of(sc8, 2 * j), sh = byte_of(sc8, 2 * j + 1);
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 q0 = *reinterpret_cast<const u32x4 *>(blk + 16 + 32 * j), q1 = *reinterpret_cast<const u32x4 *>(blk + 32 + 32 * j);
        const i32x4 xl0 = *reinterpret_cast<const i32x4 *>(xb + 64 * j), xl1 = *reinterpret_cast<const i32x4 *>(xb + 64 * j + 16);
        const i32x4 xh0 = *reinterpret_cast<const i32x4 *>(xb + 64 * j + 32), xh1 = *reinterpret_cast<const i32x4 *>(xb + 64 * j + 48);
        int lo[8], hi[8], xl[8], xh[8], dl[8], dh[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const unsigned q = t < 4 ? q0[t & 3] : q1[t & 3];
            lo[t] = (int)(q & 0x0f0f0f0fu); hi[t] = (int)((q >> 4) & 0x0f0f0f0fu);
            xl[t] = t < 4 ? xl0[t & 3] : xl1[t & 3]; xh[t] = t < 4 ? xh0[t & 3] : xh1[t & 3];
        }
        dot4z_x8(dl, lo, xl);
        dot4z_x8(dh, hi, xh);
#pragma unroll
        for (int t = 0; t < 8; ++t) cls[t] = __mul24(sh, dh[t]) + (__mul24(sl, dl[t]) + cls[t]);
    }
    const int4 s0 = *reinterpret_cast<const int4 *>(a.q8s + b * 8), s1 = *reinterpret_cast<const int4 *>(a.q8s + b * 8 + 4);
    const int prod0 = __mul24(byte_of(mn8, 0), s0.x) + __mul24(byte_of(mn8, 1), s0.y), prod1 = __mul24(byte_of(mn8, 2), s0.z) + __mul24(byte_of(mn8, 3), s0.w);
    const int prod2 = __mul24(byte_of(mn8, 4), s1.x) + __mul24(byte_of(mn8, 5), s1.y), prod3 = __mul24(byte_of(mn8, 6), s1.z) + __mul24(byte_of(mn8, 7), s1.w);
    const float xd = a.d[b];
    const float dy = xd * d, dm = (-xd) * dmin;      // VecDotQ4.cpp:228-229
    e[0] = make_float2(dy, (float)cls[0]); e[1] = make_float2(dy, (float)cls[4]);
    e[2] = make_float2(dy, (float)cls[2]); e[3] = make_float2(dy, (float)cls[6]);
    e[4] = make_float2(dy, (float)cls[1]); e[5] = make_float2(dy, (float)cls[5]);
    e[6] = make_float2(dy, (float)cls[3]); e[7] = make_float2(dy, (float)cls[7]);
    e[8] = make_float2(dm, (float)prod0); e[9] = make_float2(dm, (float)prod2);
    e[10] = make_float2(dm, (float)prod1); e[11] = make_float2(dm, (float)prod3);
}

// The same table entries by a PAIR of adjacent lanes: lane h = 0 / 1 takes nibble words j = 2h, 2h+1 (sub-blocks 4h .. 4h+3) and the mins of its four sub-blocks; the
// integer class sums of the two halves are added across the pair by DPP (integer sums: any order), then lane 0 stores slots 0-3, 8, 10 and lane 1 slots 4-7, 9, 11.
// For the down projection (6 rows x 35 super-blocks per workgroup of 512 lanes) this puts 420 lanes to work instead of 210 on an emit that is bound by one wave's
// instruction stream (323 VALU per super-block; scratch/stamps_pjb.py).  Every lane of a quad must be active (the caller rounds the active range up).
__device__ __forceinline__ void blk_emit_pair(const char *blk, const ActLds &a, int b, int h, float2 *e) {
    const uint4 hdr = *reinterpret_cast<const uint4 *>(blk);
    const float d = h2f((uint16_t)(hdr.x & 0xffff)), dmin = h2f((uint16_t)(hdr.x >> 16));
    uint32_t sc8[2], mn8[2];
    unpack_q4k_scales(hdr.y, hdr.z, hdr.w, sc8, mn8);
    const uint32_t sc = h ? sc8[1] : sc8[0], mn = h ? mn8[1] : mn8[0];
    const int8_t *xb = a.qs + b * a.qstride + h * 128;
    const char *qb = blk + 16 + h * 64;
    int cls[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int sl = (int)((sc >> (16 * j)) & 0xff), sh = (int)((sc >> (16 * j + 8)) & 0xff);
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 q0 = *reinterpret_cast<const u32x4 *>(qb + 32 * j), q1 = *reinterpret_cast<const u32x4 *>(qb + 16 + 32 * j);
        const i32x4 xl0 = *reinterpret_cast<const i32x4 *>(xb + 64 * j), xl1 = *reinterpret_cast<const i32x4 *>(xb + 64 * j + 16);
        const i32x4 xh0 = *reinterpret_cast<const i32x4 *>(xb + 64 * j + 32), xh1 = *reinterpret_cast<const i32x4 *>(xb + 64 * j + 48);
        int lo[8], hi[8], xl[8], xh[8], dl[8], dh[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const unsigned q = t < 4 ? q0[t & 3] : q1[t & 3];
            lo[t] = (int)(q & 0x0f0f0f0fu); hi[t] = (int)((q >> 4) & 0x0f0f0f0fu);
            xl[t] = t < 4 ? xl0[t & 3] : xl1[t & 3]; xh[t] = t < 4 ? xh0[t & 3] : xh1[t & 3];
        }
        dot4z_x8(dl, lo, xl);
        dot4z_x8(dh, hi, xh);
#pragma unroll
        for (int t = 0; t < 8; ++t) cls[t] = __mul24(sh, dh[t]) + (__mul24(sl, dl[t]) + cls[t]);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) cls[t] += MH_DPP(0, cls[t], DPP_QUAD_X1, 0xF);
    const int4 s4 = *reinterpret_cast<const int4 *>(a.q8s + b * 8 + 4 * h);
    const int pa = __mul24((int)(mn & 0xff), s4.x) + __mul24((int)((mn >> 8) & 0xff), s4.y), pb = __mul24((int)((mn >> 16) & 0xff), s4.z) + __mul24((int)(mn >> 24), s4.w);
    const float xd = a.d[b];
    const float dy = xd * d, dm = (-xd) * dmin;      // VecDotQ4.cpp:228-229
    float2 *eh = e + 4 * h;
    eh[0] = make_float2(dy, (float)(h ? cls[1] : cls[0])); eh[1] = make_float2(dy, (float)(h ? cls[5] : cls[4]));
    eh[2] = make_float2(dy, (float)(h ? cls[3] : cls[2])); eh[3] = make_float2(dy, (float)(h ? cls[7] : cls[6]));
    e[8 + h] = make_float2(dm, (float)pa); e[10 + h] = make_float2(dm, (float)pb);
}

// ADAPT (integration/hip's lazy window, mllm_hip_row_fused_launch mode 1): the gate and the up projection are two Linears' own row sets (W = gate rows, E.Wup = up rows), the row may
// come as xa + xb (the F_TTADD in front), and every Op of the run keeps its output: the sum, the normalised row (workgroup 0 stores both), gate, silu(gate), up, and the product
struct GubExtra { const uint8_t *Wup; const float *xb; float *sum_out, *norm_out, *g_out, *silu_out, *u_out; };
template <int PAIRS, int NV, int WPB, bool ADAPT = false>
__global__ __launch_bounds__(64 * WPB) void dec_gateup_blk_kernel(const float *__restrict__ x, const float *__restrict__ norm_w, float eps,
                                                                  const uint8_t *__restrict__ W, float *__restrict__ act, int I, int K, const GubExtra E = GubExtra{}) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[(WPB + 1) & ~1];      // a multiple of 16 bytes: statics precede the dynamic region unpadded
    const int nb = K >> 8;
    ActLds a;
    a.qs = reinterpret_cast<int8_t *>(smem);
    a.d = reinterpret_cast<float *>(smem + (size_t)nb * GUB_QSTRIDE);
    a.q8s = reinterpret_cast<int *>(smem + (size_t)nb * GUB_QSTRIDE + ((nb * 4 + 15) & ~15));
    a.xf = nullptr;
    a.qstride = GUB_QSTRIDE;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wave = blockIdx.x * WPB + wid;
    const int p0 = wave * PAIRS;                                  // first pair of this wave (the launcher guarantees I % PAIRS == 0)
    const bool live = p0 < I;
    const size_t half = gub_half_bytes(PAIRS, nb);
    char *stage = smem + ((gub_act_bytes(K) + 15) & ~(size_t)15) + (size_t)wid * gub_wave_bytes(PAIRS, nb);
    float2 *tab = reinterpret_cast<float2 *>(stage + 2 * half);
    float *outv = reinterpret_cast<float *>(stage + 2 * half + (size_t)2 * PAIRS * nb * Q4K_SLOTS * 8);
    float4 xv[NV], wv[NV];
    GSTAMP(0);
    load_rows2<NV, WPB>(xv, wv, x, norm_w, K);
    float4 bv[ADAPT ? NV : 1];
    if constexpr (ADAPT) {
        const float *pb = E.xb ? E.xb : x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int d = threadIdx.x * 4 + i * WPB * 256;
            bv[i] = *reinterpret_cast<const float4 *>(pb + (d < K ? d : 0));
            asm volatile("" : "+v"(bv[i].x), "+v"(bv[i].y), "+v"(bv[i].z), "+v"(bv[i].w));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#if MLLM_HIP_GUB_DMA_FIRST
    // ---- the wave's rows: two contiguous runs of PAIRS * nb super-blocks -> LDS (global_load_lds_dwordx4, 1 KiB per instruction) ----------------
    const int run = PAIRS * nb * 144;
    if (live) {
        const unsigned st0 = (unsigned)(size_t)stage;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint8_t *src = (ADAPT && r ? E.Wup : W) + ((int64_t)(r && !ADAPT ? I + p0 : p0) * nb) * 144;
            for (int o = 0; o < run; o += 1024) {
                const int off = o + lane * 16;
                glds16_dec(src + (off < run ? off : 0), st0 + (unsigned)(r * half) + (unsigned)o);   // past the run: a dummy 16 bytes into the pad
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#endif
#if defined(MLLM_HIP_STAMPS_GUB)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (diagnostic build: the activation row and, in the DMA-first order, the weight rows have landed)
    GSTAMP(1);
#endif
    if constexpr (ADAPT) {
        if (E.xb) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int d = threadIdx.x * 4 + i * WPB * 256;
                xv[i].x = __fadd_rn(xv[i].x, bv[i].x); xv[i].y = __fadd_rn(xv[i].y, bv[i].y); xv[i].z = __fadd_rn(xv[i].z, bv[i].z); xv[i].w = __fadd_rn(xv[i].w, bv[i].w);
                if (d >= K) xv[i] = make_float4(0, 0, 0, 0);
                else if (E.sum_out && blockIdx.x == 0) *reinterpret_cast<float4 *>(E.sum_out + d) = xv[i];
            }
        }
        wg_rmsnorm_quant<NV, WPB>(xv, wv, K, eps, a, red, blockIdx.x == 0 ? E.norm_out : nullptr);
    } else
    wg_rmsnorm_quant<NV, WPB>(xv, wv, K, eps, a, red);
    GSTAMP(2);
#if !MLLM_HIP_GUB_DMA_FIRST
    // ---- the wave's rows: two contiguous runs of PAIRS * nb super-blocks -> LDS (global_load_lds_dwordx4, 1 KiB per instruction) ----------------
    const int run = PAIRS * nb * 144;
    if (live) {
        const unsigned st0 = (unsigned)(size_t)stage;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint8_t *src = (ADAPT && r ? E.Wup : W) + ((int64_t)(r && !ADAPT ? I + p0 : p0) * nb) * 144;
            for (int o = 0; o < run; o += 1024) {
                const int off = o + lane * 16;
                glds16_dec(src + (off < run ? off : 0), st0 + (unsigned)(r * half) + (unsigned)o);   // past the run: a dummy 16 bytes into the pad
            }
        }
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    GSTAMP(3);
    // ---- lane l: super-block (region = gate / up, pair, b) ------------------------------------------------------------------------------------
    const int region = lane >= PAIRS * nb, within = lane - (region ? PAIRS * nb : 0), pair = within / nb, b = within - pair * nb;
    if (live && lane < 2 * PAIRS * nb) {
        blk_emit(stage + (size_t)region * half + (size_t)within * 144, a, b, tab + ((size_t)(2 * pair + region) * nb + b) * Q4K_SLOTS);
    }
    wave_lds_fence();
    GSTAMP(4);
#pragma unroll
    for (int p = 0; p < 2 * PAIRS; p += 4) {
        const int nr = 2 * PAIRS - p < 4 ? 2 * PAIRS - p : 4;
        const float res = q4k_chain(tab + (size_t)p * nb * Q4K_SLOTS, nb, nb, nr, lane);
        if ((lane & 15) == 8 && (lane >> 4) < nr) outv[p + (lane >> 4)] = res;
    }
    wave_lds_fence();
    GSTAMP(5);
    if (live && lane < PAIRS) {
        const float g = outv[2 * lane], u = outv[2 * lane + 1];
        if constexpr (ADAPT) {
            const float sg = g / (1.0f + v_expf_dec(0.0f - g));
            E.g_out[p0 + lane] = g;      // in the Ops' order: the frontend may have handed the gate's block to the up projection's output
            if (E.silu_out) E.silu_out[p0 + lane] = sg;
            E.u_out[p0 + lane] = u;
            act[p0 + lane] = sg * u;
        } else
        act[p0 + lane] = (g / (1.0f + v_expf_dec(0.0f - g))) * u;   // mllm_v_silu then F_TTMUL
    }
    GSTAMP(6);
}

// ------------------------------------------------------------------------------------------------------------------------
// dec_down / dec_oproj: fp32 activation row (act, or the attention output row) -> Q8_K -> W rows + residual -> y
// ------------------------------------------------------------------------------------------------------------------------
// POLL (the merged attention + o-projection launch): the activation row arrives as {value, epoch} pairs written by workgroups of the SAME launch (agent-scope relaxed 64-bit
// atomics, the data is the flag: profiles/r04_seam_overlap_microbench.md); this workgroup has its weight rows in flight before it starts to poll.  Spins are bounded: a time-out
// sets *poll_err and the workgroup goes on with what it has (the host reports the step as failed).
template <int NSTEPS, int ROWS, int WPB, bool POLL>
__device__ __forceinline__ void dec_proj_body(const float *__restrict__ xin, const unsigned long long *__restrict__ xpairs, unsigned epoch, int *__restrict__ poll_err, int wg,
                                              const uint8_t *__restrict__ W, const float *__restrict__ residual, float *__restrict__ y, int N, int K,
                                              const unsigned long long *__restrict__ res_pairs = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ActLds a = carve_act(smem, K);
    constexpr int NQ = (NSTEPS * 8 + WPB - 1) / WPB;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wave = wg * WPB + wid, nb = K >> 8;
    int rows[ROWS];
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) rows[rr] = min(wave * ROWS + rr, N - 1);
    RowLoads<NSTEPS, ROWS> L;
    STAMP(0);
    float4 v[NQ];
    if constexpr (POLL) {
        static_assert(NSTEPS == 1, "the polled form issues its rows first: short rows only");
        issue_rows<NSTEPS, ROWS>(L, W, nb, rows, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int blk = wid + WPB * i;
            v[i] = make_float4(0, 0, 0, 0);
            if (blk < nb) {
                const unsigned long long *p = xpairs + blk * 256 + lane * 4;
                unsigned long long e0, e1, e2, e3;
                int polls = 0;
                for (;;) {
                    e0 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); e1 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    e2 = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); e3 = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const bool ok = (unsigned)(e0 >> 32) == epoch && (unsigned)(e1 >> 32) == epoch && (unsigned)(e2 >> 32) == epoch && (unsigned)(e3 >> 32) == epoch;
                    if (ok) break;
                    if (++polls > (1 << 18)) { __hip_atomic_store(poll_err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                v[i] = make_float4(__uint_as_float((unsigned)e0), __uint_as_float((unsigned)e1), __uint_as_float((unsigned)e2), __uint_as_float((unsigned)e3));
            }
        }
        CSTAMP(2);
        wave_quant_blocks<NQ, WPB>(v, lane, wid, nb, a);
    } else {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int blk = wid + WPB * i;
            v[i] = blk < nb ? *reinterpret_cast<const float4 *>(xin + blk * 256 + lane * 4) : make_float4(0, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#if MLLM_HIP_EARLY_ROWS
        // short rows (o-proj: 1.3 MB of weights in all): the rows go out right behind the activation loads and fly underneath the
        // quantisation -- the burst is too small to hold the activation row up, unlike the 15 MB of gate|up
        if (NSTEPS == 1) { issue_rows<NSTEPS, ROWS>(L, W, nb, rows, lane); __builtin_amdgcn_sched_barrier(0); }
#endif
        wave_quant_blocks<NQ, WPB>(v, lane, wid, nb, a);
        if (!(MLLM_HIP_EARLY_ROWS && NSTEPS == 1)) issue_rows<NSTEPS, ROWS>(L, W, nb, rows, lane);
    }
    __syncthreads();
    STAMP(5);
    if constexpr (POLL) CSTAMP(3);
    float out[ROWS];
    dot_rows<NSTEPS, ROWS>(L, a, nb, lane, wave_tab<NSTEPS, ROWS>(smem, K, false, wid), out);
    STAMP(7);
    if constexpr (POLL) CSTAMP(4);
    if (lane == 63) {
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            const int rw = wave * ROWS + rr;
            if (rw < N) {
                // res_pairs (four-role launch): the residual row was written by the down-projection role of THIS launch -- read it from its pairs (all of them have been seen
                // with this epoch by the q|k|v role before the attention could run), not through a plain load
                const float rsd = POLL && res_pairs ? __uint_as_float((unsigned)__hip_atomic_load(res_pairs + rw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : (residual ? residual[rw] : 0.0f);
                y[rw] = (residual || (POLL && res_pairs)) ? out[rr] + rsd : out[rr];
            }
        }
    }
}
template <int NSTEPS, int ROWS, int WPB>
__global__ __launch_bounds__(64 * WPB) void dec_proj_kernel(const float *__restrict__ xin, const uint8_t *__restrict__ W, const float *__restrict__ residual,
                                                            float *__restrict__ y, int N, int K) {
    dec_proj_body<NSTEPS, ROWS, WPB, false>(xin, nullptr, 0u, nullptr, blockIdx.x, W, residual, y, N, K);
}

// dec_proj_blk: the down projection with one lane per super-block (blk_emit), a workgroup per RPW consecutive rows.  The rows (one
// contiguous run of RPW * K/256 super-blocks, disk layout) go to LDS by LDS-DMA, issued once the activation row has arrived and
// running underneath its quantisation; lane g of the workgroup owns super-block g of the run; RPW / 4 waves walk the chains.
__host__ __device__ static inline size_t pjb_stage_bytes(int rpw, int nb) { return ((size_t)rpw * nb * 144 + 1023) & ~(size_t)1023; }
// rows per workgroup of the lane-per-super-block projection (dec_proj_blk): ONE definition for the launchers and for the warming table of the attention launch, whose
// regions must be the consumer workgroups' row groups (eight waves walk four rows' chains each: at most 32 rows per workgroup)
static inline int pjb_rows_per_wg(int N, int K) { return std::max(1, std::min(std::min(512 / (K / 256), 32), (N + 255) / 256)); }
static inline size_t pjb_lds_bytes(int K, int rpw) {
    return ((gub_act_bytes(K) + 15) & ~(size_t)15) + pjb_stage_bytes(rpw, K / 256) + (size_t)rpw * (K / 256) * Q4K_SLOTS * 8;
}
// OUTP (the down-projection role of the four-role launch): besides y the rows go out as {value, epoch} pairs for the next layer's q|k|v role of the same launch
template <int WPB, int NQ, bool OUTP>
__device__ __forceinline__ void dec_proj_blk_body(const float *__restrict__ xin, const uint8_t *__restrict__ W, const float *__restrict__ residual, float *__restrict__ y, int N,
                                                  int K, int RPW, int wg, unsigned long long *__restrict__ ypairs, unsigned epoch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nb = K >> 8;
    ActLds a;
    a.qs = reinterpret_cast<int8_t *>(smem);
    a.d = reinterpret_cast<float *>(smem + (size_t)nb * GUB_QSTRIDE);
    a.q8s = reinterpret_cast<int *>(smem + (size_t)nb * GUB_QSTRIDE + ((nb * 4 + 15) & ~15));
    a.xf = nullptr;
    a.qstride = GUB_QSTRIDE;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = min(wg * RPW, N - RPW);     // the last workgroup re-does rows of its neighbour (same values)
    char *stage = smem + ((gub_act_bytes(K) + 15) & ~(size_t)15);
    float2 *tab = reinterpret_cast<float2 *>(stage + pjb_stage_bytes(RPW, nb));
    float4 v[NQ];
    GSTAMP2(0);
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int blk = wid + WPB * i;
        v[i] = *reinterpret_cast<const float4 *>(xin + (blk < nb ? blk : 0) * 256 + lane * 4);
    }
    // all NQ loads in flight together, then (the activation row being the dependent fetch) the weight stream behind it
#pragma unroll
    for (int i = 0; i < NQ; ++i) asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i].z), "+v"(v[i].w));
    {
        const int run = RPW * nb * 144;
        const uint8_t *src = W + (int64_t)row0 * nb * 144;
        const unsigned st0 = (unsigned)(size_t)stage;
        for (int o = wid * 1024; o < run; o += WPB * 1024) {
            const int off = o + lane * 16;
            glds16_dec(src + (off < run ? off : 0), st0 + (unsigned)o);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NQ; ++i) if (wid + WPB * i >= nb) v[i] = make_float4(0, 0, 0, 0);
    GSTAMP2(1);
    wave_quant_blocks<NQ, WPB>(v, lane, wid, nb, a);
    GSTAMP2(2);
    if constexpr (OUTP) CSTAMP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GSTAMP2(3);
    __syncthreads();
    GSTAMP2(4);
    if constexpr (OUTP) CSTAMP(3);
    if (RPW * nb > 64 && 2 * RPW * nb <= 64 * WPB) {      // a lane pair per super-block (see blk_emit_pair) once one wave no longer holds them all
        const int sb = tid >> 1;
        if (sb < RPW * nb) blk_emit_pair(stage + (size_t)sb * 144, a, sb % nb, tid & 1, tab + (size_t)sb * Q4K_SLOTS);
    } else if (tid < RPW * nb) {
        blk_emit(stage + (size_t)tid * 144, a, tid % nb, tab + (size_t)tid * Q4K_SLOTS);
    }
    GSTAMP2(5);
    __syncthreads();
    GSTAMP2(6);
    if constexpr (OUTP) CSTAMP(4);
    if (4 * wid < RPW) {
        const int nr = RPW - 4 * wid < 4 ? RPW - 4 * wid : 4;
        const float res = q4k_chain(tab + (size_t)4 * wid * nb * Q4K_SLOTS, nb, nb, nr, lane);
        const int rw = row0 + 4 * wid + (lane >> 4);
        if ((lane & 15) == 8 && (lane >> 4) < nr) {
            const float r = residual ? res + residual[rw] : res;
            y[rw] = r;
            if constexpr (OUTP) __hip_atomic_store(ypairs + rw, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    GSTAMP2(7);
}
template <int WPB, int NQ>
__global__ __launch_bounds__(64 * WPB) void dec_proj_blk_kernel(const float *__restrict__ xin, const uint8_t *__restrict__ W, const float *__restrict__ residual,
                                                                float *__restrict__ y, int N, int K, int RPW) {
    dec_proj_blk_body<WPB, NQ, false>(xin, W, residual, y, N, K, RPW, (int)blockIdx.x, nullptr, 0u);
}

// ------------------------------------------------------------------------------------------------------------------------
// dec_attn: one workgroup (1024 threads) per query head.  Rotates q and the new k with the step's sin/cos row (rope_hf:
// fma(a,c,-(b*s)), fma(a,s,b*c)), rounds the new k, v to fp16 (what the reference's cache holds; the first head of a GQA group
// appends them to the slab), then __fa2_decode over keys 0..T in key order (kernels_attn_core.h).
// qkv: [Hq*D | Hkv*D | Hkv*D] fp32 of this token.  K slab [cache_limit][Hkv*D] fp16, V slab transposed [Hkv*D][vt_ld] fp16.  out: [Hq*D] fp32.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int DEC_ATTN_NT = 1024;
// Workgroup placement: the heads of one K/V group read the same slab rows, so they are given equal blockIdx.x % 8 -- workgroups b and b + 8 share an XCD
// (and its L2) under the dispatcher's round-robin, a speed matter only.  K/V head k sits in column k % 8; inside a column the order is (k / 8, head of the
// group).  Grid = 8 * ceil(Hkv / 8) * (Hq / Hkv) workgroups; those whose K/V head does not exist leave at once.
// DS: workgroups per head (dim split): each computes the head's scores and softmax statistics and walks D / DS of its value dims.
__host__ __device__ static inline int dec_attn_grid(int Hq, int Hkv, int ds) { return 8 * ((Hkv + 7) / 8) * (Hq / Hkv) * ds; }
template <int D, int DS>
__global__ __launch_bounds__(DEC_ATTN_NT) void dec_attn_kernel(const DecodeState *__restrict__ state, const float *__restrict__ qkv, const float *__restrict__ sin_t,
                                                               const float *__restrict__ cos_t, uint16_t *__restrict__ kslab, uint16_t *__restrict__ vslab,
                                                               float *__restrict__ out, int Hq, int Hkv, int cache_limit, int vt_ld, int nslots, int flags) {
    constexpr int HALF = D / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ __attribute__((aligned(16))) uint16_t knew[D];
    __shared__ __attribute__((aligned(16))) uint16_t vnew[D];
    constexpr int DV = D / DS;
    const int gsize = Hq / Hkv, per_kv = gsize * DS;
    int kvh, sub;
    if (flags & 1) { const int col = blockIdx.x & 7, idx = blockIdx.x >> 3; kvh = (idx / per_kv) * 8 + col; sub = idx % per_kv; }
    else { kvh = blockIdx.x / per_kv; sub = blockIdx.x % per_kv; }
    if (kvh >= Hkv) return;
    const int gh = sub / DS, vdim0 = (sub % DS) * DV;
    const int head = kvh * gsize + gh;
    const DecodeLds L = carve_decode(smem, cache_limit, D, DEC_ATTN_NT, nslots);
    const int tid = threadIdx.x;
    const int HD = Hq * D, KVD = Hkv * D;
    // the step's own small operands first (vmcnt retires in issue order), then -- speculatively, T only masks them afterwards -- the
    // slab rows of the first keys; the remaining rows of the first pass once T has arrived
    STAMP(1);
    const int T_raw = state->T;
    float qa = 0.0f, qb = 0.0f, sn = 0.0f, cs = 0.0f;
    if (tid < HALF) { const float *qp = qkv + head * D; qa = qp[tid]; qb = qp[tid + HALF]; sn = sin_t[tid]; cs = cos_t[tid]; }
    else if (tid < D) { const float *kp = qkv + HD + kvh * D; qa = kp[tid - HALF]; qb = kp[tid]; sn = sin_t[tid - HALF]; cs = cos_t[tid - HALF]; }
    else if (tid < 2 * D) qa = qkv[HD + KVD + kvh * D + (tid - D)];
    __builtin_amdgcn_sched_barrier(0);
    DecodePrefetch<D, true, DEC_ATTN_NT, true, DV> P;
    const bool two_stage = flags & 2;
    fa2_decode_prefetch<D, true, DEC_ATTN_NT, true, DV>(P, kslab, KVD, vslab, vt_ld, kvh * D, cache_limit, nslots, two_stage ? 1 : 0);
    __builtin_amdgcn_sched_barrier(0);
    const int T = min(T_raw, cache_limit - 1), Sk = T + 1;
    if (two_stage) fa2_decode_prefetch<D, true, DEC_ATTN_NT, true, DV>(P, kslab, KVD, vslab, vt_ld, kvh * D, cache_limit, nslots, 2, T);
    __builtin_amdgcn_sched_barrier(0);
    if (tid < HALF) {
        L.qs[tid] = __fmaf_rn(qa, cs, -(qb * sn));
        L.qs[tid + HALF] = __fmaf_rn(qa, sn, qb * cs);
    } else if (tid < D) {
        knew[tid - HALF] = f2h(__fmaf_rn(qa, cs, -(qb * sn)));
        knew[tid] = f2h(__fmaf_rn(qa, sn, qb * cs));
    } else if (tid < 2 * D) {
        vnew[tid - D] = f2h(qa);
    }
    __syncthreads();
    if (sub == 0 && tid < D) {
        kslab[(int64_t)T * KVD + kvh * D + tid] = knew[tid];
        vslab[(int64_t)(kvh * D + tid) * vt_ld + T] = vnew[tid];
    }
    fa2_decode_head<D, true, DEC_ATTN_NT, true, DV>(L, P, kslab, KVD, vslab, vt_ld, kvh * D, kvh * D + vdim0, Sk, cache_limit, knew, vnew + vdim0, T);
    if (tid < DV) out[head * D + vdim0 + tid] = L.ob[tid];
}

// The same step on fa2_decode_head_pipe (kernels_attn_core.h): the scores / softmax statistics of later key blocks are computed while the walk over the first ones runs.
// Same grid, same placement, same arithmetic (bit-identical results); dynamic LDS = pipe_lds_bytes<D, DV>(cache_limit, NP).
// 12 waves: the walker's two register stages (a block each: 50 registers) need more than the 128 registers a 16-wave workgroup leaves a lane -- with 1024 threads the
// allocator folded the two stages into one and the read-ahead was gone
constexpr int DEC_PIPE_NT = 768;
// warmer `slot` of the `nw` on XCD column `col` takes the targets g = col + 8 j, j = slot, slot + nw, ...; per target every region's bytes, 1 KiB per wave and request, as
// LDS-DMA into a landing pad nobody reads (`pad`: 1 KiB per wave of the workgroup's dynamic LDS): no registers, no waits between the requests, nothing for the compiler to
// keep alive -- with ordinary loads the eight 16-byte results per thread and their addresses pushed the kernel into spills and scratch
__device__ __forceinline__ void warm_weights(const WeightWarm *__restrict__ wt, char *pad, int col, int slot, int nw) {
    constexpr int NWV = DEC_PIPE_NT / 64;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const unsigned dst = (unsigned)(size_t)pad + (unsigned)wid * 1024u;
    const int n = wt->n;
    for (int j = slot; j < WeightWarm::GROUPS; j += nw) {
        const int g = col + 8 * j;
        for (int r = 0; r < n; ++r) {
            const int bytes = wt->bytes[r];
            if (g >= wt->count[r]) continue;
            const uint8_t *src = wt->base[r] + (int64_t)g * bytes;
            for (int o = wid * 1024; o < bytes; o += NWV * 1024) {
                const int off = o + lane * 16;
                glds16_dec(src + (off < bytes ? off : 0), dst);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// PAIRS (the merged attention + o-projection launch): the output row is stored as {value, DecodeState::serial} pairs for the o-projection workgroups of the same launch
// to poll (dec_proj_body<.., POLL>); grid_attn = the workgroups of the launch that belong to the attention (the o-projection's come behind them)
// QPOLL: q | k | v arrive as pairs from the q|k|v role of the same launch (qkv_pairs); the speculative cache fetches go out first, the poll sits where the values are needed
template <int D, int DS, bool PAIRS, bool QPOLL = false>
__device__ __forceinline__ void dec_attn_pipe_body(const DecodeState *__restrict__ state, const float *__restrict__ qkv, const float *__restrict__ sin_t,
                                                   const float *__restrict__ cos_t, uint16_t *__restrict__ kslab, uint16_t *__restrict__ vslab,
                                                   float *__restrict__ out, unsigned long long *__restrict__ out_pairs, int Hq, int Hkv, int cache_limit, int vt_ld, int flags,
                                                   int attn_groups, const WeightWarm *__restrict__ ww, int grid_attn, int wg = -1,
                                                   const unsigned long long *__restrict__ qkv_pairs = nullptr, int *__restrict__ poll_err = nullptr) {
    if (wg < 0) wg = (int)blockIdx.x;
    constexpr int HALF = D / 2, DV = D / DS, NWK = (DV + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ __attribute__((aligned(16))) uint16_t knew[D];
    __shared__ __attribute__((aligned(16))) uint16_t vnew[D];
    const int gsize = Hq / Hkv, per_kv = gsize * DS;
    int kvh, sub;
    if (flags & 1) {
        const int col = wg & 7, idx = wg >> 3;
        kvh = (idx / per_kv) * 8 + col; sub = idx % per_kv;
        // warming workgroups (only when the layer has a table): the extra ones behind the attention's grid, and -- on an XCD column no K/V head lives on -- the attention slots too
        const bool col_dead = col >= Hkv;
        if (ww && (idx >= attn_groups || col_dead)) {
            const int ext = (grid_attn >> 3) - attn_groups;
            const int nw = ext + (col_dead ? attn_groups : 0), slot = idx >= attn_groups ? idx - attn_groups + (col_dead ? attn_groups : 0) : idx;
            warm_weights(ww, smem, col, slot, nw);
            return;
        }
    } else { kvh = wg / per_kv; sub = wg % per_kv; }
    if (kvh >= Hkv) return;
    const int gh = sub / DS, vdim0 = (sub % DS) * DV;
    const int head = kvh * gsize + gh;
    const PipeLds L = carve_pipe(smem, cache_limit, D, DV);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int HD = Hq * D, KVD = Hkv * D;
    STAMP(1);
    STAMPV(14, (unsigned long long)(__builtin_amdgcn_s_getreg(0x1814) & 15));      // HW_REG_XCC_ID of this workgroup (diagnostic build only)
    // the step's own small operands first (vmcnt retires in issue order), then -- speculatively, T only masks them afterwards -- the first-round block of every producer wave
    const int T_raw = state->T;
    const unsigned serial = PAIRS ? (unsigned)state->serial : 0u;
    float qa = 0.0f, qb = 0.0f, sn = 0.0f, cs = 0.0f;
    if constexpr (!QPOLL) {
        if (tid < HALF) { const float *qp = qkv + head * D; qa = qp[tid]; qb = qp[tid + HALF]; sn = sin_t[tid]; cs = cos_t[tid]; }
        else if (tid < D) { const float *kp = qkv + HD + kvh * D; qa = kp[tid - HALF]; qb = kp[tid]; sn = sin_t[tid - HALF]; cs = cos_t[tid - HALF]; }
        else if (tid < 2 * D) qa = qkv[HD + KVD + kvh * D + (tid - D)];
    } else {
        if (tid < HALF) { sn = sin_t[tid]; cs = cos_t[tid]; }
        else if (tid < D) { sn = sin_t[tid - HALF]; cs = cos_t[tid - HALF]; }
    }
    const uint64_t etab_v = expf_tab_fetch();
    __builtin_amdgcn_sched_barrier(0);
    PipeRegs<D, DV> R;
    const int pw = wid - NWK - 1;      // producer index (< 0: walker / logsum wave)
    constexpr int NSPEC = 4;           // producers whose first block is requested before T is known (keys below 128 always exist in the slab; T only masks them)
    if (pw >= 0 && pw < NSPEC) {
        pipe_fetch_k<D, DV>(R, kslab, KVD, kvh * D, pw, cache_limit, lane);
        pipe_fetch_v<D, DV>(R, vslab, vt_ld, kvh * D + vdim0, pw, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    // the walk is the critical path and the first blocks feed it first: static priorities (walkers > early producers > the rest)
    if (pw < 0) __builtin_amdgcn_s_setprio(3);
    else if (pw < NSPEC) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(1);
    for (int i = tid; i < 3 * ((pipe_nblk(cache_limit) + 3) & ~3) + 4; i += DEC_PIPE_NT) L.flagT[i] = 0u;      // flagT | rm (two words per block) | walked are contiguous
    expf_tab_store(L.etab, etab_v);
    const int T = min(T_raw, cache_limit - 1), Sk = T + 1;
    if (pw >= NSPEC && pw * FP_B < Sk) {      // the later blocks of the first round, now that the key count is known
        pipe_fetch_k<D, DV>(R, kslab, KVD, kvh * D, pw, cache_limit, lane);
        pipe_fetch_v<D, DV>(R, vslab, vt_ld, kvh * D + vdim0, pw, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (QPOLL) {      // this thread's one or two values of the q | k | v row, from the q|k|v role's pairs
        const unsigned epoch = (unsigned)state->serial;
        const int ia = tid < HALF ? head * D + tid : (tid < D ? HD + kvh * D + tid - HALF : HD + KVD + kvh * D + (tid - D));
        const int ib = tid < D ? ia + HALF : ia;
        if (tid < 2 * D) {
            int polls = 0;
            for (;;) {
                const unsigned long long ea = __hip_atomic_load(qkv_pairs + ia, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long eb = __hip_atomic_load(qkv_pairs + ib, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(ea >> 32) == epoch && (unsigned)(eb >> 32) == epoch) { qa = __uint_as_float((unsigned)ea); qb = __uint_as_float((unsigned)eb); break; }
                if (++polls > (1 << 18)) { __hip_atomic_store(poll_err, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        CSTAMP(2);
    }
    if (tid < HALF) {
        L.qs[tid] = __fmaf_rn(qa, cs, -(qb * sn));
        L.qs[tid + HALF] = __fmaf_rn(qa, sn, qb * cs);
    } else if (tid < D) {
        knew[tid - HALF] = f2h(__fmaf_rn(qa, cs, -(qb * sn)));
        knew[tid] = f2h(__fmaf_rn(qa, sn, qb * cs));
    } else if (tid < 2 * D) {
        vnew[tid - D] = f2h(qa);
    }
    __syncthreads();
    STAMP(0);
    if (sub == 0 && tid < D) {
        kslab[(int64_t)T * KVD + kvh * D + tid] = knew[tid];
        vslab[(int64_t)(kvh * D + tid) * vt_ld + T] = vnew[tid];
    }
    if constexpr (QPOLL) CSTAMP(3);
    fa2_decode_head_pipe<D, DV, DEC_PIPE_NT>(L, R, kslab, KVD, vslab, vt_ld, kvh * D, kvh * D + vdim0, Sk, cache_limit, knew, vnew + vdim0, T);
    if constexpr (QPOLL) CSTAMP(4);
    if (tid < DV) {
        if constexpr (PAIRS) __hip_atomic_store(out_pairs + head * D + vdim0 + tid, ((unsigned long long)serial << 32) | (unsigned long long)__float_as_uint(L.ob[tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else out[head * D + vdim0 + tid] = L.ob[tid];
    }
    STAMP(5);
}
template <int D, int DS>
__global__ __launch_bounds__(DEC_PIPE_NT) void dec_attn_pipe_kernel(const DecodeState *__restrict__ state, const float *__restrict__ qkv, const float *__restrict__ sin_t,
                                                                    const float *__restrict__ cos_t, uint16_t *__restrict__ kslab, uint16_t *__restrict__ vslab,
                                                                    float *__restrict__ out, int Hq, int Hkv, int cache_limit, int vt_ld, int flags, int attn_groups,
                                                                    const WeightWarm *__restrict__ ww) {
    dec_attn_pipe_body<D, DS, false>(state, qkv, sin_t, cos_t, kslab, vslab, out, nullptr, Hq, Hkv, cache_limit, vt_ld, flags, attn_groups, ww, (int)gridDim.x);
}
// The attention of a decode step and the o-projection behind it in ONE launch: workgroups [0, grid_attn) are the attention's (heads, dim halves, warmers), the rest the
// o-projection's (dec_proj_kernel<1, 2, 8>'s body on the first 512 threads).  The latter are dispatched behind the former, issue their weight rows at once, and poll the
// attention's output row -- {value, epoch} pairs -- instead of waiting for a kernel boundary: the launch, the row fetch and the weight stream of the projection run under the
// attention (profiles/r04_seam_overlap_microbench.md: a polled hand-off costs 1.6 - 1.9 us at this workgroup count against the 2.2 us of a dependent launch's start).
struct OProjRole { const uint8_t *W; const float *residual; float *y; unsigned long long *pairs; int *poll_err; int N, K, grid_attn, rows; };
template <int D, int DS>
__global__ __launch_bounds__(DEC_PIPE_NT) void dec_attn_oproj_kernel(const DecodeState *__restrict__ state, const float *__restrict__ qkv, const float *__restrict__ sin_t,
                                                                     const float *__restrict__ cos_t, uint16_t *__restrict__ kslab, uint16_t *__restrict__ vslab, int Hq, int Hkv,
                                                                     int cache_limit, int vt_ld, int flags, int attn_groups, const WeightWarm *__restrict__ ww, const OProjRole P) {
    if ((int)blockIdx.x < P.grid_attn) {
        dec_attn_pipe_body<D, DS, true>(state, qkv, sin_t, cos_t, kslab, vslab, nullptr, P.pairs, Hq, Hkv, cache_limit, vt_ld, flags, attn_groups, ww, P.grid_attn);
        return;
    }
    if (threadIdx.x >= 512) return;      // the projection's body is an eight-wave workgroup
    if (P.rows == 1) dec_proj_body<1, 1, 8, true>(nullptr, P.pairs, (unsigned)state->serial, P.poll_err, (int)blockIdx.x - P.grid_attn, P.W, P.residual, P.y, P.N, P.K);
    else dec_proj_body<1, 2, 8, true>(nullptr, P.pairs, (unsigned)state->serial, P.poll_err, (int)blockIdx.x - P.grid_attn, P.W, P.residual, P.y, P.N, P.K);
}


// q|k|v projection + attention + o-projection of a layer in ONE launch: workgroups [0, grid_q) run dec_qkv_kernel<1, 2, EMBED, 1, 8>'s body and publish q | k | v as pairs, the
// attention's workgroups behind them send their speculative cache fetches and poll those pairs, the o-projection's behind those poll the attention's output.  Producers always
// have the lower workgroup indices.
struct QkvFront { const float *x; float *x_out; const uint8_t *emb_qs; const uint16_t *emb_d; const float *norm_w; const uint8_t *W; const float *bias; unsigned long long *pairs;
                  float eps; int vocab, N, K, grid_q; KvWarm kw; };
template <int D, int DS, bool EMBED>
__global__ __launch_bounds__(DEC_PIPE_NT) void dec_qkv_attn_oproj_kernel(const DecodeState *__restrict__ state, const float *__restrict__ sin_t, const float *__restrict__ cos_t,
                                                                         uint16_t *__restrict__ kslab, uint16_t *__restrict__ vslab, int Hq, int Hkv, int cache_limit, int vt_ld, int flags,
                                                                         int attn_groups, const WeightWarm *__restrict__ ww, const QkvFront F, const OProjRole P) {
    const int b = (int)blockIdx.x;
    if (b < F.grid_q) {
        if (threadIdx.x >= 512) return;
        dec_qkv_body<1, 2, EMBED, 1, 8, true>(state, F.x, F.x_out, F.emb_qs, F.emb_d, F.vocab, F.norm_w, F.eps, F.W, F.bias, nullptr, F.pairs, F.N, F.K, F.kw, b, F.grid_q);
        return;
    }
    if (b < F.grid_q + P.grid_attn) {
        dec_attn_pipe_body<D, DS, true, true>(state, nullptr, sin_t, cos_t, kslab, vslab, nullptr, P.pairs, Hq, Hkv, cache_limit, vt_ld, flags, attn_groups, ww, P.grid_attn,
                                              b - F.grid_q, F.pairs, P.poll_err);
        return;
    }
    if (threadIdx.x >= 512) return;
    const int wo = b - F.grid_q - P.grid_attn;
    if (P.rows == 1) dec_proj_body<1, 1, 8, true>(nullptr, P.pairs, (unsigned)state->serial, P.poll_err, wo, P.W, P.residual, P.y, P.N, P.K);
    else dec_proj_body<1, 2, 8, true>(nullptr, P.pairs, (unsigned)state->serial, P.poll_err, wo, P.W, P.residual, P.y, P.N, P.K);
}

// Layer l's down projection and layer l + 1's q|k|v projection + attention + o-projection in ONE launch.  The down projection's 256 workgroups fill the chip (the launch's LDS
// size leaves one workgroup per CU); as they leave, the q|k|v role's workgroups are dispatched, send their weight rows and poll the layer's output row (pairs), and so on down
// the chain -- what disappears is the drain / launch gap between the two kernels.  The o-projection's residual is that same output row: read from the pairs, not through a plain
// load of what another workgroup of this launch stored.
struct DownRole { const float *xin; const uint8_t *W; const float *residual; float *y; unsigned long long *ypairs; int N, K, rpw, grid_d, cont; };
template <int D, int DS, int NQ>
__global__ __launch_bounds__(DEC_PIPE_NT) void dec_down_front_kernel(const DecodeState *__restrict__ state, const float *__restrict__ sin_t, const float *__restrict__ cos_t,
                                                                     uint16_t *__restrict__ kslab, uint16_t *__restrict__ vslab, int Hq, int Hkv, int cache_limit, int vt_ld, int flags,
                                                                     int attn_groups, const WeightWarm *__restrict__ ww, const DownRole Dn, const QkvFront F, const OProjRole P) {
    int b = (int)blockIdx.x;
    CSTAMP(0);
    if (b < Dn.grid_d) {
        if (threadIdx.x >= 512) return;
        dec_proj_blk_body<8, NQ, true>(Dn.xin, Dn.W, Dn.residual, Dn.y, Dn.N, Dn.K, Dn.rpw, b, Dn.ypairs, (unsigned)state->serial);
        CSTAMP(1);
        if (Dn.cont && b < F.grid_q) {      // the first grid_q down-projection workgroups carry on as the q|k|v role: no wait for a CU, no dispatch
            __syncthreads();      // (the role reuses the LDS the down projection's last phase read)
            dec_qkv_body<1, 2, false, 1, 8, true, true>(state, nullptr, nullptr, F.emb_qs, F.emb_d, F.vocab, F.norm_w, F.eps, F.W, F.bias, nullptr, F.pairs, F.N, F.K, F.kw, b, F.grid_q,
                                                        Dn.ypairs, P.poll_err);
            CSTAMP(5);
        }
        return;
    }
    b -= Dn.grid_d;
    if (!Dn.cont) {
        if (b < F.grid_q) {
            if (threadIdx.x >= 512) return;
            dec_qkv_body<1, 2, false, 1, 8, true, true>(state, nullptr, nullptr, F.emb_qs, F.emb_d, F.vocab, F.norm_w, F.eps, F.W, F.bias, nullptr, F.pairs, F.N, F.K, F.kw, b, F.grid_q,
                                                        Dn.ypairs, P.poll_err);
            CSTAMP(1);
            return;
        }
        b -= F.grid_q;
    }
    if (b < P.grid_attn) {
        dec_attn_pipe_body<D, DS, true, true>(state, nullptr, sin_t, cos_t, kslab, vslab, nullptr, P.pairs, Hq, Hkv, cache_limit, vt_ld, flags, attn_groups, ww, P.grid_attn,
                                              b, F.pairs, P.poll_err);
        CSTAMP(1);
        return;
    }
    if (threadIdx.x >= 512) return;
    const int wo = b - P.grid_attn;
    if (P.rows == 1) dec_proj_body<1, 1, 8, true>(nullptr, P.pairs, (unsigned)state->serial, P.poll_err, wo, P.W, nullptr, P.y, P.N, P.K, Dn.ypairs);
    else dec_proj_body<1, 2, 8, true>(nullptr, P.pairs, (unsigned)state->serial, P.poll_err, wo, P.W, nullptr, P.y, P.N, P.K, Dn.ypairs);
    CSTAMP(1);
}

// ------------------------------------------------------------------------------------------------------------------------
// dec_head: x -> RMSNorm -> Q8_0 -> tied lm_head rows (Q4_0 planes) -> logits, plus this workgroup's (max, first index)
// ------------------------------------------------------------------------------------------------------------------------
template <int BPL>
__global__ __launch_bounds__(256) void dec_head_kernel(const float *__restrict__ x, const float *__restrict__ norm_w, float eps,
                                                       const uint8_t *__restrict__ Wqs, const uint16_t *__restrict__ Wd, float *__restrict__ logits,
                                                       float *__restrict__ part_val, int *__restrict__ part_idx, int N, int K, int rows_per_wave) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[4];
    __shared__ float bv[4];
    __shared__ int bi[4];
    float *xf = reinterpret_cast<float *>(smem);                 // [K] normalised row
    int8_t *xq = reinterpret_cast<int8_t *>(smem + (size_t)K * 4);  // [K] q8_0
    float *xdd = reinterpret_cast<float *>(smem + (size_t)K * 5);   // [K/32] fp16-rounded scales as fp32
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, sub = lane & 15, rsel = lane >> 4;
    const int wave = blockIdx.x * 4 + wid, nblk = BPL * 16;
    const int row0 = wave * rows_per_wave, row1 = min(N, row0 + rows_per_wave);
    // first batch of weight rows in flight before the prologue
    uint4 q0[2][BPL];
    uint16_t d0[2][BPL];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int rw = min(row0 + 4 * u + rsel, max(row1 - 1, 0));
#pragma unroll
        for (int b = 0; b < BPL; ++b) {
            const int64_t bidx = (int64_t)rw * nblk + sub + 16 * b;
            q0[u][b] = *reinterpret_cast<const uint4 *>(Wqs + bidx * 16);
            d0[u][b] = Wd[bidx];
        }
    }
    // RMSNorm (final norm, eps 1e-6) of the single row
    double ss = 0.0;
    for (int d = tid * 4; d < K; d += 1024) {
        const float4 v = *reinterpret_cast<const float4 *>(x + d);
        *reinterpret_cast<float4 *>(xf + d) = v;
        ss += (double)v.x * (double)v.x + (double)v.y * (double)v.y + (double)v.z * (double)v.z + (double)v.w * (double)v.w;
    }
    ss = wave_sum_d(ss);
    if (lane == 0) red[wid] = ss;
    __syncthreads();
    ss = red[0] + red[1] + red[2] + red[3];
    const float inv = 1.0f / sqrtf((float)(ss / (double)K) + eps);
    // Q8_0 (quantize_row_q8_0_reference): 8 lanes per 32-block
    for (int blk = tid >> 3; blk < K / 32; blk += 32) {
        const int d4 = blk * 32 + (tid & 7) * 4;
        float4 v = *reinterpret_cast<const float4 *>(xf + d4);
        const float4 ww = *reinterpret_cast<const float4 *>(norm_w + d4);
        v.x = (v.x * inv) * ww.x; v.y = (v.y * inv) * ww.y; v.z = (v.z * inv) * ww.z; v.w = (v.w * inv) * ww.w;
        float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        amax = group8_max(amax);
        const float dd = amax / 127.0f;
        const float id = dd != 0.0f ? 1.0f / dd : 0.0f;
        const int a0 = (int)roundf(v.x * id), a1 = (int)roundf(v.y * id), a2 = (int)roundf(v.z * id), a3 = (int)roundf(v.w * id);
        *reinterpret_cast<uint32_t *>(xq + d4) = (uint32_t)(a0 & 0xff) | ((uint32_t)(a1 & 0xff) << 8) | ((uint32_t)(a2 & 0xff) << 16) | ((uint32_t)(a3 & 0xff) << 24);
        if ((tid & 7) == 0) xdd[blk] = h2f(f2h(dd));
    }
    __syncthreads();
    Q40Act<BPL> A;
    q40_load_act<BPL, 16>(A, xq, xdd, nullptr, sub);
    float *ts = reinterpret_cast<float *>(smem + (((size_t)K * 5 + (size_t)K / 32 * 4 + 15) & ~(size_t)15)) + wid * q40_tab_floats(nblk), *td = ts + 8 * q40_ts_stride(nblk);
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int base = row0; base < row1; base += 8) {
        // this pass's rows are in q0 / d0 (loaded one pass ago); the next pass's rows are requested before the dot so that the HBM
        // round trip runs underneath the emit and the chain walk instead of in front of them
        uint4 q[2][BPL];
        uint16_t dw[2][BPL];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int b = 0; b < BPL; ++b) { q[u][b] = q0[u][b]; dw[u][b] = d0[u][b]; }
        if (base + 8 < row1) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int rw = min(base + 8 + 4 * u + rsel, row1 - 1);
#pragma unroll
                for (int b = 0; b < BPL; ++b) {
                    const int64_t bidx = (int64_t)rw * nblk + sub + 16 * b;
                    q0[u][b] = *reinterpret_cast<const uint4 *>(Wqs + bidx * 16);
                    d0[u][b] = Wd[bidx];
                }
            }
        }
        wave_lds_fence();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int rl = 4 * u + rsel;
            q40_emit<BPL, 16>(q[u], dw[u], A, sub, ts + (size_t)rl * q40_ts_stride(nblk), td + (size_t)rl * q40_td_stride(nblk));
        }
        wave_lds_fence();
        const float acc = q40_chain(ts, td, nblk, min(8, row1 - base), lane);
        const int rw = base + (lane >> 3);
        if (rw < row1) {
            if ((lane & 7) == 0) logits[rw] = acc;
            if (acc > best) { best = acc; besti = rw; }   // rows visited in increasing order per lane group
        }
    }
    // workgroup argmax with first-index tie break
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = __shfl_xor(best, m, 64);
        const int oi = __shfl_xor(besti, m, 64);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if (lane == 0) { bv[wid] = best; bi[wid] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) if (bv[w] > best || (bv[w] == best && bi[w] < besti)) { best = bv[w]; besti = bi[w]; }
        part_val[blockIdx.x] = best;
        part_idx[blockIdx.x] = besti;
    }
}

// final argmax over the workgroup partials (std::max_element: first maximum), record the token, advance the step state
__global__ __launch_bounds__(256) void dec_next_kernel(DecodeState *__restrict__ state, const float *__restrict__ part_val, const int *__restrict__ part_idx,
                                                       int nparts, int *__restrict__ tok_out, int *__restrict__ history, const float *__restrict__ tab_sin,
                                                       const float *__restrict__ tab_cos, float *__restrict__ cur_sin, float *__restrict__ cur_cos, int half,
                                                       int max_rows) {
    // the next step's rotary row moves to a fixed address, so the attention kernel's loads do not depend on the step index
    {
        const int nxt = min(state->step + 1, max_rows - 1);
        if ((int)threadIdx.x < half) { cur_sin[threadIdx.x] = tab_sin[(int64_t)nxt * half + threadIdx.x]; cur_cos[threadIdx.x] = tab_cos[(int64_t)nxt * half + threadIdx.x]; }
    }
    __shared__ float bv[4];
    __shared__ int bi[4];
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int i = threadIdx.x; i < nparts; i += 256) {
        const float v = part_val[i];
        const int ix = part_idx[i];
        if (v > best || (v == best && ix < besti)) { best = v; besti = ix; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = __shfl_xor(best, m, 64);
        const int oi = __shfl_xor(besti, m, 64);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) if (bv[w] > best || (bv[w] == best && bi[w] < besti)) { best = bv[w]; besti = bi[w]; }
        *tok_out = besti;
        if (history) history[state->step] = besti;
        state->token = besti;
        state->T += 1;
        state->step += 1;
        state->serial += 1;
    }
}
#if defined(MLLM_HIP_STAMPS) || defined(MLLM_HIP_STAMPS_GUB) || defined(MLLM_HIP_STAMPS_PJB) || defined(MLLM_HIP_STAMPS_CHAIN)
}  // namespace mllm_hip
extern "C" int mllm_hip_debug_read_stamps(unsigned long long *host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(mllm_hip::g_stamps), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
namespace mllm_hip {
#endif
}  // namespace mllm_hip

using namespace mllm_hip;

// ---- launch helpers used by engine.hip (decode_launch.h) ---------------------------------------------------------------

namespace mllm_hip {

template <typename KernelT>
static int allow_lds(KernelT kern, size_t lds) {
    if (lds > 160 * 1024) return MLLM_HIP_ERR_SHAPE;
    if (lds > 48 * 1024) MH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return MLLM_HIP_OK;
}
// RMSNorm(x) -> Q8_K -> W rows (+bias) -> y: the q|k|v projection of a layer, and the Linear lm_head of the models whose head is not tied
template <int NS>
static int launch_norm_gemv(const DecodeCtx &c, const float *norm_w, float eps, const uint8_t *W, const float *bias, int N, bool embed, const float *x,
                            float *x_out, float *y, hipStream_t st, const KvWarm &kw = KvWarm{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0}) {
#ifndef QKV_ROWS
#define QKV_ROWS 2
#endif
#ifndef QKV_WPB
#define QKV_WPB 8
#endif
    constexpr int ROWS = NS == 1 ? QKV_ROWS : 1, WPB = QKV_WPB;
    const int waves = (N + ROWS - 1) / ROWS;
    const size_t lds = fused_lds_bytes<NS, ROWS>(c.H, false, WPB);
    constexpr int NV = (NS * 8 + WPB - 1) / WPB;   // quant blocks per wave
    // long rows (LLaVA-1.5-7B: 12,288 one-row waves = 1,536 workgroups, each with its own RMSNorm + Q8_K of the 4,096-value row): a grid of qkv_persist workgroups per CU walks them
    // (short rows too when there are at least four times as many workgroups as the walking grid: the untied lm_heads -- TinyLlama 2,000, Qwen1.5 9,496 workgroups)
    const int persist = option(OPT_QKV_PERSIST) >= 0 ? option(OPT_QKV_PERSIST) : (NS >= 2 || (waves + WPB - 1) / WPB >= 1024 ? 2 : 0);
    if (!embed && persist > 0 && (waves + WPB - 1) / WPB > 256 * persist) {
        int wrc = allow_lds(dec_qkv_walk_kernel<NS, ROWS, NV, WPB>, lds);
        if (wrc) return wrc;
        hipLaunchKernelGGL((dec_qkv_walk_kernel<NS, ROWS, NV, WPB>), dim3(256 * persist), dim3(64 * WPB), lds, st, c.state, x, x_out, c.emb_qs, c.emb_d, c.vocab, norm_w, eps, W, bias, y,
                           N, c.H, kw);
        return MH_LAUNCH_OK("dec_qkv_walk");
    }
    int rc = embed ? allow_lds(dec_qkv_kernel<NS, ROWS, true, NV, WPB>, lds) : allow_lds(dec_qkv_kernel<NS, ROWS, false, NV, WPB>, lds);
    if (rc) return rc;
    if (embed)
        hipLaunchKernelGGL((dec_qkv_kernel<NS, ROWS, true, NV, WPB>), dim3((waves + WPB - 1) / WPB), dim3(64 * WPB), lds, st, c.state, x, x_out, c.emb_qs, c.emb_d, c.vocab, norm_w,
                           eps, W, bias, y, N, c.H, kw);
    else
        hipLaunchKernelGGL((dec_qkv_kernel<NS, ROWS, false, NV, WPB>), dim3((waves + WPB - 1) / WPB), dim3(64 * WPB), lds, st, c.state, x, x_out, c.emb_qs, c.emb_d, c.vocab, norm_w,
                           eps, W, bias, y, N, c.H, kw);
    return MH_LAUNCH_OK("dec_qkv");
}
template <int NS>
static int launch_gateup(const DecodeLayer &L, const DecodeCtx &c, const float *x, hipStream_t st) {
#ifndef GU_PAIRS
#define GU_PAIRS 2
#endif
#ifndef GU_WPB
#define GU_WPB 8
#endif
#ifndef GUB_PAIRS
#define GUB_PAIRS 5
#endif
#ifndef GUB_WPB
#define GUB_WPB 7
#endif
    const bool gub_off = option(OPT_NO_GUB) > 0;      // bring-up switch: the 8-lanes-per-block kernel instead
    if (MLLM_HIP_GUB && !gub_off && NS == 1 && 2 * GUB_PAIRS * (c.H >> 8) <= 64 && c.I % GUB_PAIRS == 0 && (c.H >> 8) <= GUB_WPB) {
        constexpr int BP = GUB_PAIRS, BW = GUB_WPB, BNV = 1;      // one quant block per wave in the prologue: K/256 <= waves
        const int bw = c.I / BP;
        const size_t blds = gub_lds_bytes(c.H, BP, BW);
        auto bk = dec_gateup_blk_kernel<BP, BNV, BW>;
        int brc = allow_lds(bk, blds);
        if (brc) return brc;
        hipLaunchKernelGGL(bk, dim3((bw + BW - 1) / BW), dim3(64 * BW), blds, st, x, L.post_norm, c.eps, L.Wgu_raw, c.act, c.I, c.H, GubExtra{});
        return MH_LAUNCH_OK("dec_gateup_blk");
    }
    constexpr int PAIRS = NS == 1 ? GU_PAIRS : 1, WPB = GU_WPB;
    const int waves = (c.I + PAIRS - 1) / PAIRS;
    const size_t lds = fused_lds_bytes<NS, 2 * PAIRS>(c.H, false, WPB);
    auto kern = dec_gateup_kernel<NS, PAIRS, (NS * 8 + WPB - 1) / WPB, WPB>;
    int rc = allow_lds(kern, lds);
    if (rc) return rc;
    int grid = (waves + WPB - 1) / WPB;
    // workgroups per CU of the persistent form (LLaVA-1.5-7B: 1,376 -> 512 workgroups, each walking 2.7 row groups behind one prologue: 19.5 -> 15.2 us; 1 / 3 / 4 per CU: 17.3 - 17.6)
    const int persist = option(OPT_GU_PERSIST) >= 0 ? option(OPT_GU_PERSIST) : (NS >= 2 ? 2 : 0);
    if (persist > 0) grid = std::min(grid, 256 * persist);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WPB), lds, st, x, L.post_norm, c.eps, L.Wgu, c.act, c.I, c.H);
    return MH_LAUNCH_OK("dec_gateup");
}
template <int NS>
static int launch_proj(const uint8_t *W, const uint8_t *Wraw, const float *xin, const float *residual, float *y, int N, int K, hipStream_t st) {
#ifndef PJ_ROWS1
#define PJ_ROWS1 2
#endif
#ifndef PJ_ROWS5
#define PJ_ROWS5 1
#endif
#ifndef PJ_WPB5
#define PJ_WPB5 8
#endif
#ifndef PJ_WPB1
#define PJ_WPB1 8
#endif
    // one lane per super-block: rows per workgroup chosen for one workgroup per CU (256) within the 512 lanes of a workgroup
    const bool pjb_off = option(OPT_NO_PJB) > 0;
    const int pjb_min_ns = option(OPT_PJB_MIN_NS) >= 0 ? option(OPT_PJB_MIN_NS) : 3;   // short rows: the 8-lane kernel is faster
    {
        const int nb = K >> 8;
        const int rpw = pjb_rows_per_wg(N, K);
        if (NS >= pjb_min_ns && NS <= 5 && Wraw && !pjb_off && N >= rpw) {
            constexpr int BW = 8, NQ = NS;
            const size_t blds = pjb_lds_bytes(K, rpw);
            auto bk = dec_proj_blk_kernel<BW, NQ>;
            int brc = allow_lds(bk, blds);
            if (brc) return brc;
            hipLaunchKernelGGL(bk, dim3((N + rpw - 1) / rpw), dim3(64 * BW), blds, st, xin, Wraw, residual, y, N, K, rpw);
            return MH_LAUNCH_OK("dec_proj_blk");
        }
    }
    constexpr int ROWS = NS == 1 ? PJ_ROWS1 : PJ_ROWS5, WPB = NS >= 3 ? PJ_WPB5 : PJ_WPB1;   // long rows: 1024-thread workgroups share the row quantisation
    const int waves = (N + ROWS - 1) / ROWS;
    const size_t lds = fused_lds_bytes<NS, ROWS>(K, false, WPB);
    auto kern = dec_proj_kernel<NS, ROWS, WPB>;
    int rc = allow_lds(kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3((waves + WPB - 1) / WPB), dim3(64 * WPB), lds, st, xin, W, residual, y, N, K);
    return MH_LAUNCH_OK("dec_proj");
}

// one activation row through a Linear with raw Q4_K rows in ONE launch (Q8_K of the row in the prologue, one lane per super-block, + addend[n] -- a bias or a residual):
// what mllm_hip_linear uses for M == 1 instead of a quantiser launch and a GEMV launch.  Returns 1 when the shape is not covered (the caller keeps its two launches).
int dec_linear_row_q4k(const void *Wraw, const float *x, const float *addend, float *y, int N, int K, hipStream_t st) {
    if (K <= 0 || K % 256 || N <= 0) return 1;
    const int nb = K >> 8, nsr = (nb + 7) / 8;
    const int rpw = pjb_rows_per_wg(N, K);
    if (nsr > 5 || N < rpw) return 1;
    const size_t lds = pjb_lds_bytes(K, rpw);
#define ROW_CASE(NSV)                                                                                                                                     \
    case NSV: {                                                                                                                                           \
        auto bk = dec_proj_blk_kernel<8, NSV>;                                                                                                            \
        const int rc = allow_lds(bk, lds);                                                                                                                \
        if (rc) return rc;                                                                                                                                \
        hipLaunchKernelGGL(bk, dim3((N + rpw - 1) / rpw), dim3(512), lds, st, x, (const uint8_t *)Wraw, addend, y, N, K, rpw);                            \
    } break;
    switch (nsr) { ROW_CASE(1) ROW_CASE(2) ROW_CASE(3) ROW_CASE(4) ROW_CASE(5) }
#undef ROW_CASE
    return MH_LAUNCH_OK("dec_proj_blk");
}

// ---- one activation row through UP TO SIX Ops of the reference's decoder layer in one launch (mllm_hip_row_fused; integration/hip's lazy window) ------------------------------
// The reference's frontend issues a decode layer one Op at a time: F_TTADD -> RMSNorm -> Linear q, k, v ... Linear o -> F_TTADD ... RMSNorm -> Linear gate -> SiLU -> Linear up ->
// F_TTMUL -> Linear down -> F_TTADD.  This kernel is dec_proj_blk_kernel (one lane per super-block, rows by LDS-DMA) with the neighbouring Ops folded in, every Op's own output still
// written (the frontend owns those tensors), every value computed by the arithmetic of the Op's own kernel:
//   prologue   s = xa + xb (F_TTADD, CPUBinaryFunc.hpp)  ->  n = RMSNorm(s) (norm_kernel's double sum, (x * inv) * w)  ->  Q8_K of n   (each optional; workgroup 0 stores s and n)
//   body       up to three Linears on that row (q | k | v), each its own raw Q4_K rows + bias: y = dot + bias
//   epilogue   mode 0: post_out = y + post_add (the F_TTADD behind an o / down projection);   mode 1: segment 0 = gate, 1 = up: silu_out = silu(y0), mul_out = silu_out * y1
// A workgroup covers `rpw` consecutive rows of one segment (mode 1: rpw / 2 rows of the gate and the same rows of the up projection).
typedef mllm_hip_row_seg RowFusedSeg;
typedef mllm_hip_row_fused RowFusedArgs;
static inline size_t rowf_lds_bytes(int K, int rpw, int mode) {
    return 384 + ((gub_act_bytes(K) + 15) & ~(size_t)15) + (mode == 1 ? 2 * pjb_stage_bytes(rpw / 2, K / 256) : pjb_stage_bytes(rpw, K / 256)) + (size_t)rpw * (K / 256) * Q4K_SLOTS * 8;
}
template <int WPB, int NQ>
__global__ __launch_bounds__(64 * WPB) void row_fused_kernel(const RowFusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = A.K, nb = K >> 8, RPW = A.rpw_;
    double *red = reinterpret_cast<double *>(smem);                 // [WPB <= 16]
    float *resbuf = reinterpret_cast<float *>(smem + 128);          // [rpw <= 64]
    char *act0 = smem + 384;
    ActLds a;
    a.qs = reinterpret_cast<int8_t *>(act0);
    a.d = reinterpret_cast<float *>(act0 + (size_t)nb * GUB_QSTRIDE);
    a.q8s = reinterpret_cast<int *>(act0 + (size_t)nb * GUB_QSTRIDE + ((nb * 4 + 15) & ~15));
    a.xf = nullptr;
    a.qstride = GUB_QSTRIDE;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool gateup = A.mode == 1;
    int s = 0;
    if (!gateup) {
        if (A.nseg > 1 && (int)blockIdx.x >= A.seg[1].wg0_) s = 1;
        if (A.nseg > 2 && (int)blockIdx.x >= A.seg[2].wg0_) s = 2;
    }
    const uint8_t *W0 = (const uint8_t *)(s == 0 ? A.seg[0].W : (s == 1 ? A.seg[1].W : A.seg[2].W));
    const int Ns = s == 0 ? A.seg[0].N : (s == 1 ? A.seg[1].N : A.seg[2].N);
    const int wg0 = s == 0 ? 0 : (s == 1 ? A.seg[1].wg0_ : A.seg[2].wg0_);
    const int half = gateup ? RPW >> 1 : RPW;      // rows of one segment in this workgroup
    const int row0 = min(((int)blockIdx.x - wg0) * half, Ns - half);     // the last workgroup re-does rows of its neighbour (same values)
    char *stage = act0 + ((gub_act_bytes(K) + 15) & ~(size_t)15);
    const int run = half * nb * 144;
    char *stage2 = stage + ((run + 1023) & ~1023);                        // mode 1: the up rows' run (1 KB steps of the LDS-DMA loop must not run into it)
    float2 *tab = reinterpret_cast<float2 *>(stage + (gateup ? 2 * (size_t)((run + 1023) & ~1023) : pjb_stage_bytes(RPW, nb)));
    const bool has_b = A.xb != nullptr, has_n = A.norm_w != nullptr;
    const float *pb = has_b ? A.xb : A.xa, *pw = has_n ? A.norm_w : A.xa;      // unconditional loads (a predicated load is waited for inside its branch)
    float4 v[NQ], vb[NQ], wv[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int blk = wid + WPB * i, o = (blk < nb ? blk : 0) * 256 + lane * 4;
        v[i] = *reinterpret_cast<const float4 *>(A.xa + o);
        vb[i] = *reinterpret_cast<const float4 *>(pb + o);
        wv[i] = *reinterpret_cast<const float4 *>(pw + o);
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i)
        asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i].z), "+v"(v[i].w), "+v"(vb[i].x), "+v"(vb[i].y), "+v"(vb[i].z), "+v"(vb[i].w), "+v"(wv[i].x), "+v"(wv[i].y), "+v"(wv[i].z), "+v"(wv[i].w));
    {
        const uint8_t *src = W0 + (int64_t)row0 * nb * 144;
        const unsigned st0 = (unsigned)(size_t)stage;
        for (int o = wid * 1024; o < run; o += WPB * 1024) {
            const int off = o + lane * 16;
            glds16_dec(src + (off < run ? off : 0), st0 + (unsigned)o);
        }
        if (gateup) {
            const uint8_t *src2 = (const uint8_t *)A.seg[1].W + (int64_t)row0 * nb * 144;
            const unsigned st2 = (unsigned)(size_t)stage2;
            for (int o = wid * 1024; o < run; o += WPB * 1024) {
                const int off = o + lane * 16;
                glds16_dec(src2 + (off < run ? off : 0), st2 + (unsigned)o);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        if (has_b) { v[i].x = __fadd_rn(v[i].x, vb[i].x); v[i].y = __fadd_rn(v[i].y, vb[i].y); v[i].z = __fadd_rn(v[i].z, vb[i].z); v[i].w = __fadd_rn(v[i].w, vb[i].w); }
        if (wid + WPB * i >= nb) v[i] = make_float4(0, 0, 0, 0);
        else if (A.sum_out && blockIdx.x == 0) *reinterpret_cast<float4 *>(A.sum_out + (wid + WPB * i) * 256 + lane * 4) = v[i];
    }
    if (has_n) {      // norm_kernel<false>: double sum of squares, inv = 1 / sqrt(mean + eps), (x * inv) * w
        double ss = 0.0;
#pragma unroll
        for (int i = 0; i < NQ; ++i) ss += (double)v[i].x * (double)v[i].x + (double)v[i].y * (double)v[i].y + (double)v[i].z * (double)v[i].z + (double)v[i].w * (double)v[i].w;
        ss = wave_sum_d(ss);
        if (lane == 0) red[wid] = ss;
        __syncthreads();
        ss = red[0];
#pragma unroll
        for (int w = 1; w < WPB; ++w) ss += red[w];
        const float m = (float)(ss / (double)K);
        const float inv = __fdiv_rn(1.0f, sqrtf(__fadd_rn(m, A.eps)));
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            v[i].x = __fmul_rn(__fmul_rn(v[i].x, inv), wv[i].x); v[i].y = __fmul_rn(__fmul_rn(v[i].y, inv), wv[i].y);
            v[i].z = __fmul_rn(__fmul_rn(v[i].z, inv), wv[i].z); v[i].w = __fmul_rn(__fmul_rn(v[i].w, inv), wv[i].w);
            if (wid + WPB * i >= nb) v[i] = make_float4(0, 0, 0, 0);
            else if (A.norm_out && blockIdx.x == 0) *reinterpret_cast<float4 *>(A.norm_out + (wid + WPB * i) * 256 + lane * 4) = v[i];
        }
    }
    wave_quant_blocks<NQ, WPB>(v, lane, wid, nb, a);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // stage row r < half sits in the first run, the rest (mode 1: the up rows) in the second
    auto blk_of = [&](int sb) -> const char * { return sb < half * nb ? stage + (size_t)sb * 144 : stage2 + (size_t)(sb - half * nb) * 144; };
    if (RPW * nb > 64 && 2 * RPW * nb <= 64 * WPB) {
        const int sb = tid >> 1;
        if (sb < RPW * nb) blk_emit_pair(blk_of(sb), a, sb % nb, tid & 1, tab + (size_t)sb * Q4K_SLOTS);
    } else if (tid < RPW * nb) {
        blk_emit(blk_of(tid), a, tid % nb, tab + (size_t)tid * Q4K_SLOTS);
    }
    __syncthreads();
    float res = 0.0f;
    int nr = 0;
    if (4 * wid < RPW) {
        nr = RPW - 4 * wid < 4 ? RPW - 4 * wid : 4;
        res = q4k_chain(tab + (size_t)4 * wid * nb * Q4K_SLOTS, nb, nb, nr, lane);
    }
    if (!gateup) {
        const float *bias = s == 0 ? A.seg[0].bias : (s == 1 ? A.seg[1].bias : A.seg[2].bias);
        float *y = s == 0 ? A.seg[0].y : (s == 1 ? A.seg[1].y : A.seg[2].y);
        const float *padd = s == 0 ? A.seg[0].post_add : (s == 1 ? A.seg[1].post_add : A.seg[2].post_add);
        float *pout = s == 0 ? A.seg[0].post_out : (s == 1 ? A.seg[1].post_out : A.seg[2].post_out);
        const int rw = row0 + 4 * wid + (lane >> 4);
        if ((lane & 15) == 8 && (lane >> 4) < nr) {
            const float yv = bias ? __fadd_rn(res, bias[rw]) : res;
            y[rw] = yv;
            if (pout) pout[rw] = __fadd_rn(yv, padd[rw]);
        }
        return;
    }
    if ((lane & 15) == 8 && (lane >> 4) < nr) resbuf[4 * wid + (lane >> 4)] = res;
    __syncthreads();
    if (tid < half) {
        const int rw = row0 + tid;
        float g = resbuf[tid], u = resbuf[half + tid];
        if (A.seg[0].bias) g = __fadd_rn(g, A.seg[0].bias[rw]);
        if (A.seg[1].bias) u = __fadd_rn(u, A.seg[1].bias[rw]);
        const float sg = __fdiv_rn(g, __fadd_rn(1.0f, v_expf_dec(__fsub_rn(0.0f, g))));      // silu_kernel's silu_ref
        A.seg[0].y[rw] = g;                 // in the Ops' order: the frontend may have handed the gate's block to the up projection's output
        if (A.silu_out) A.silu_out[rw] = sg;
        A.seg[1].y[rw] = u;
        A.mul_out[rw] = __fmul_rn(sg, u);
    }
}
// host side of the launch: the rows a workgroup takes and where each segment's workgroups start; > 0 = the grid, < 0 = an error code (shape not covered)
static int row_fused_plan(RowFusedArgs &A) {
    const int K = A.K;
    if (K <= 0 || K % 256 || A.nseg < 1 || A.nseg > 3 || !A.xa) return MLLM_HIP_ERR_SHAPE;
    const int nb = K >> 8, nsr = (nb + 7) / 8;
    if (nsr > 5) return MLLM_HIP_ERR_SHAPE;
    int total = 0;
    if (A.mode == 1) {
        if (A.nseg != 2 || A.seg[0].N != A.seg[1].N || !A.mul_out || !A.seg[0].W || !A.seg[1].W || !A.seg[0].y || !A.seg[1].y) return MLLM_HIP_ERR_SHAPE;
        const int N = A.seg[0].N;
        // 16 rows of each projection per 512-thread workgroup (32 rows on 1024 threads halve the repeated prologues but measured slower: 11.7 against 9.8 us on 1536 x 8960)
        const int half = std::max(1, std::min(std::min(256 / nb, 16), (N + 255) / 256));
        if (N < half) return MLLM_HIP_ERR_SHAPE;
        A.rpw_ = 2 * half;
        total = (N + half - 1) / half;
    } else if (A.mode == 0) {
        int sumN = 0;
        for (int i = 0; i < A.nseg; ++i) sumN += A.seg[i].N;
        A.rpw_ = pjb_rows_per_wg(sumN, K);
        for (int i = 0; i < A.nseg; ++i) {
            if (A.seg[i].N < A.rpw_ || !A.seg[i].W || !A.seg[i].y || (A.seg[i].post_out && !A.seg[i].post_add)) return MLLM_HIP_ERR_SHAPE;
            A.seg[i].wg0_ = total;
            total += (A.seg[i].N + A.rpw_ - 1) / A.rpw_;
        }
    } else return MLLM_HIP_ERR_ARG;
    if (rowf_lds_bytes(K, A.rpw_, A.mode) > 160 * 1024) return MLLM_HIP_ERR_SHAPE;
    return total;
}
int row_fused_supported(const RowFusedArgs &in) {
    RowFusedArgs A = in;
    return row_fused_plan(A) > 0 ? 1 : 0;
}
int row_fused_launch(const RowFusedArgs &in, hipStream_t st) {
    RowFusedArgs A = in;
    const int total = row_fused_plan(A);
    if (total <= 0) return total ? total : MLLM_HIP_ERR_SHAPE;
    const int K = A.K, nsr = (K / 256 + 7) / 8;
    // the MLP's run on the 2 B model's shape: the engine's own gate|up kernel (a wave per five row pairs, no workgroup barrier behind the prologue) with the Ops' outputs added
    if (A.mode == 1 && option(OPT_NO_GUB) <= 0 && 2 * 5 * (K >> 8) <= 64 && (K >> 8) <= 7 && A.seg[0].N % 5 == 0 && !A.seg[0].bias && !A.seg[1].bias && A.norm_w) {
        constexpr int BP = 5, BW = 7;
        const int I = A.seg[0].N, bw = I / BP;
        const size_t blds = gub_lds_bytes(K, BP, BW);
        auto bk = dec_gateup_blk_kernel<BP, 1, BW, true>;
        const int brc = allow_lds(bk, blds);
        if (brc) return brc;
        const GubExtra E{(const uint8_t *)A.seg[1].W, A.xb, A.sum_out, A.norm_out, A.seg[0].y, A.silu_out, A.seg[1].y};
        hipLaunchKernelGGL(bk, dim3((bw + BW - 1) / BW), dim3(64 * BW), blds, st, A.xa, A.norm_w, A.eps, (const uint8_t *)A.seg[0].W, A.mul_out, I, K, E);
        return MH_LAUNCH_OK("dec_gateup_blk(adapt)");
    }
    const size_t lds = rowf_lds_bytes(K, A.rpw_, A.mode);
#define ROWF_CASE(NSV)                                                                  \
    case NSV: {                                                                         \
        auto bk = row_fused_kernel<8, NSV>;                                             \
        const int rc = allow_lds(bk, lds);                                              \
        if (rc) return rc;                                                              \
        hipLaunchKernelGGL(bk, dim3(total), dim3(512), lds, st, A);                     \
    } break;
    switch (nsr) { ROWF_CASE(1) ROWF_CASE(2) ROWF_CASE(3) ROWF_CASE(4) ROWF_CASE(5) }
#undef ROWF_CASE
    return MH_LAUNCH_OK("row_fused");
}

#define NS_DISPATCH(K, CALL)                           \
    switch (((K) / 256 + 7) / 8) {                     \
    case 1: { constexpr int NS = 1; CALL; } break;     \
    case 2: { constexpr int NS = 2; CALL; } break;     \
    case 3: { constexpr int NS = 3; CALL; } break;     \
    case 4: { constexpr int NS = 4; CALL; } break;     \
    case 5: { constexpr int NS = 5; CALL; } break;     \
    case 6: { constexpr int NS = 6; CALL; } break;     \
    default: return MLLM_HIP_ERR_SHAPE;                \
    }

// one fused kernel of layer `li`: 0 qkv, 1 attn, 2 o-proj, 3 gate|up, 4 down. x = layer input / output, t = post-attention residual
// raw Q4_K rows -> decode order (one thread per dword of the nibble area; headers copied)
__global__ void q4k_decode_order_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int64_t n_blocks) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // dword index over n_blocks * 36
    if (i >= n_blocks * 36) return;
    const int64_t blk = i / 36;
    const int w = (int)(i - blk * 36);
    const uint32_t *s = reinterpret_cast<const uint32_t *>(src + blk * 144);
    uint32_t *o = reinterpret_cast<uint32_t *>(dst + blk * 144);
    if (w < 4) o[w] = s[w];
    else { const int n = w - 4, t = n >> 2, j = n & 3; o[w] = s[4 + 8 * j + t]; }
}
int decode_order_q4k(const void *src, void *dst, int64_t n_blocks, hipStream_t st) {
    const int64_t n = n_blocks * 36;
    hipLaunchKernelGGL(q4k_decode_order_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint8_t *)src, (uint8_t *)dst, n_blocks);
    return MH_LAUNCH_OK("q4k_decode_order");
}

// the attention flags in force (option "attn_flags", else the default): bit 0 XCD placement of a K/V group's heads, bit 1 two-stage key fetch (un-pipelined kernel),
// bit 2 keep the un-pipelined kernel, bit 3 dec_qkv warms the L2 with the cache rows, bits 4..7 weight-warming workgroups (decode_warm_table)
int decode_attn_flags() { return option(OPT_ATTN_FLAGS) >= 0 ? option(OPT_ATTN_FLAGS) : 91; }
// the table of warming regions of every layer (host side; the engine uploads it once): which rows the launches behind layer li's attention will stream, and how they
// fall onto those launches' workgroups.  flags = the attention flags: bit 4 gate|up, bit 5 down, bit 6 o-projection, bit 7 the next layer's q|k|v.  Returns 0 when no layer has
// a region (the caller then leaves DecodeCtx::warm_tab null).
int decode_warm_table(const DecodeCtx &c, const DecodeLayer *layers, int n_layers, int flags, WeightWarm *out) {
    int any = 0;
    for (int li = 0; li < n_layers; ++li) {
        const DecodeLayer &L = layers[li];
        WeightWarm ww{};
        if (L.Wgu_raw && L.Wdown_raw && c.H % 256 == 0 && c.I % 256 == 0) {
            const int nbH = c.H / 256, nbI = c.I / 256;
            if ((flags & 16) && c.I % (GUB_PAIRS * GUB_WPB) == 0 && c.I / (GUB_PAIRS * GUB_WPB) <= 8 * WeightWarm::GROUPS) {      // dec_gateup_blk: workgroup g = pairs [g * 35, + 35)
                const int per = GUB_PAIRS * GUB_WPB * nbH * 144, cnt = c.I / (GUB_PAIRS * GUB_WPB);
                ww.base[ww.n] = L.Wgu_raw; ww.bytes[ww.n] = per; ww.count[ww.n] = cnt; ++ww.n;
                ww.base[ww.n] = L.Wgu_raw + (int64_t)c.I * nbH * 144; ww.bytes[ww.n] = per; ww.count[ww.n] = cnt; ++ww.n;
            }
            if ((flags & 64) && L.Wo && (c.heads * c.D) % 256 == 0 && (c.heads * c.D) / 256 <= 8 && c.H % (PJ_ROWS1 * PJ_WPB1) == 0) {      // dec_proj (o): workgroup g = rows [g * 16, + 16) of the decode-order rows
                const int per = PJ_ROWS1 * PJ_WPB1 * ((c.heads * c.D) / 256) * 144, cnt = c.H / (PJ_ROWS1 * PJ_WPB1);
                if (cnt <= 8 * WeightWarm::GROUPS) { ww.base[ww.n] = L.Wo; ww.bytes[ww.n] = per; ww.count[ww.n] = cnt; ++ww.n; }
            }
            if ((flags & 128) && li + 1 < n_layers && nbH <= 8 && layers[li + 1].qkv_N % (QKV_ROWS * QKV_WPB) == 0) {      // the next layer's dec_qkv: workgroup g = rows [g * 16, + 16)
                const int per = QKV_ROWS * QKV_WPB * nbH * 144, cnt = layers[li + 1].qkv_N / (QKV_ROWS * QKV_WPB);
                if (cnt <= 8 * WeightWarm::GROUPS && ww.n < WeightWarm::MAXR) { ww.base[ww.n] = layers[li + 1].Wqkv; ww.bytes[ww.n] = per; ww.count[ww.n] = cnt; ++ww.n; }
            }
            if ((flags & 32) && ww.n < WeightWarm::MAXR) {      // dec_proj_blk (down): workgroup g = rows [g * rpw, + rpw)
                const int rpw = pjb_rows_per_wg(c.H, nbI * 256), cnt = (c.H + rpw - 1) / rpw;
                if (cnt <= 8 * WeightWarm::GROUPS && c.H % rpw == 0) { ww.base[ww.n] = L.Wdown_raw; ww.bytes[ww.n] = rpw * nbI * 144; ww.count[ww.n] = cnt; ++ww.n; }
            }
        }
        any += ww.n;
        out[li] = ww;
    }
    return any;
}

// whether case 1 of decode_kernel_launch carries the o-projection (the same conditions as there)
template <int D>
static size_t merged_o_lds(const DecodeCtx &c) {
    constexpr int NP_ = DEC_PIPE_NT / 64 - ((D / 2 + 63) / 64) - 1;
    return std::max(pipe_lds_bytes<D, D / 2>(c.cache_limit, NP_), fused_lds_bytes<1, 2>(c.heads * c.D, false, 8));
}
bool decode_merges_o(const DecodeCtx &c) {
    if (!c.merge_o || !c.attn_pairs || (c.attn_flags & 4) || c.cache_limit > 2048 || (c.D != 128 && c.D != 64)) return false;
    const int ds_env = option(OPT_ATTN_DS) > 0 ? option(OPT_ATTN_DS) : 0;
    if (!(ds_env == 0 || ds_env == 2)) return false;
    if ((c.heads * c.D) > 2048 || (c.heads * c.D) % 256) return false;      // the projection's register form (dec_proj_kernel<1, ..>: rows of at most eight super-blocks)
    return (c.D == 128 ? merged_o_lds<128>(c) : merged_o_lds<64>(c)) <= (size_t)(160 * 1024 - 2 * c.D * 2 - 64);
}
// ... and the q|k|v projection in front of it (merge_o = 3): hidden sizes the projection's register form covers, a q|k|v grid that keeps the XCD columns of the launch
template <int D>
static size_t merged_front_lds(const DecodeCtx &c) { return std::max(merged_o_lds<D>(c), fused_lds_bytes<1, 2>(c.H, false, 8)); }
static bool decode_merges_front(const DecodeCtx &c, const DecodeLayer &L, int li) {
    // (not layer 0: its q|k|v role also writes the embedding row that the o-projection adds back -- a plain store another workgroup of the same launch would read)
    if (li == 0 || c.merge_o < 3 || !c.qkv_pairs || !decode_merges_o(c) || c.H > 2048 || c.H % 256) return false;
    const int grid_q = ((L.qkv_N + 1) / 2 + 7) / 8;
    if (grid_q & 7) return false;
    return (c.D == 128 ? merged_front_lds<128>(c) : merged_front_lds<64>(c)) <= (size_t)(160 * 1024 - 2 * c.D * 2 - 64);
}
// ... and the previous layer's down projection in front of those (merge_o = 4): layer li's down projection carries layer li + 1's q|k|v + attention + o-projection
template <int D>
static size_t merged_chain_lds(const DecodeCtx &c) { return std::max(merged_front_lds<D>(c), pjb_lds_bytes(c.I, pjb_rows_per_wg(c.H, c.I))); }
static bool decode_merges_chain(const DecodeCtx &c, const DecodeLayer *layers, int li) {
    if (c.merge_o < 4 || !c.x_pairs || li < 0 || li + 1 >= c.n_layers || c.I % 256) return false;
    DecodeCtx f = c;
    f.merge_o = 3;
    if (!decode_merges_front(f, layers[li + 1], li + 1)) return false;
    const int NS = (c.I / 256 + 7) / 8, rpw = pjb_rows_per_wg(c.H, c.I), grid_d = (c.H + rpw - 1) / rpw;
    const int pjb_min_ns = option(OPT_PJB_MIN_NS) >= 0 ? option(OPT_PJB_MIN_NS) : 3;
    if (!(NS >= pjb_min_ns && NS <= 5 && layers[li].Wdown_raw && option(OPT_NO_PJB) <= 0 && c.H >= rpw) || (grid_d & 7)) return false;
    return (c.D == 128 ? merged_chain_lds<128>(c) : merged_chain_lds<64>(c)) <= (size_t)(160 * 1024 - 2 * c.D * 2 - 64);
}
int decode_kernel_launch(const DecodeCtx &c, const DecodeLayer *layers, int li, int which, hipStream_t st) {
    if ((c.D != 128 && c.D != 64) || c.heads % c.kv_heads) return MLLM_HIP_ERR_SHAPE;
    float *x = c.x0, *t = c.x1;
    const DecodeLayer &L = layers[li];
    int rc = 0;
    switch (which) {
    case 0:
    {
        if (decode_merges_chain(c, layers, li - 1)) return MLLM_HIP_OK;      // this layer's q|k|v + attention + o-projection rode in the previous layer's down-projection launch
        if (decode_merges_front(c, L, li)) {
            uint16_t *kl = c.kslab + (size_t)li * c.cache_limit * c.kv_heads * c.D, *vl = c.vslab + (size_t)li * c.kv_heads * c.D * c.vt_ld;
            const int flags = c.attn_flags, ds = 2;
            const int grid_a0 = (flags & 1) ? dec_attn_grid(c.heads, c.kv_heads, ds) : c.heads * ds, attn_groups = grid_a0 / 8;
            const WeightWarm *ww = (flags & 1) && !(flags & 4) && c.warm_tab ? c.warm_tab + li : nullptr;
            const int grid_attn = grid_a0 + (ww ? 8 * std::max(8, WeightWarm::GROUPS - attn_groups) : 0);      // the attention's workgroups and, behind them, the warming ones
            const int grid_q = ((L.qkv_N + 1) / 2 + 7) / 8;
            const int rows = 1, Ko = c.heads * c.D, grid_o = ((c.H + rows - 1) / rows + 7) / 8;
            KvWarm kw{nullptr, nullptr, nullptr, c.kv_heads, c.D, c.kv_heads * c.D, c.vt_ld, c.cache_limit};
            if ((flags & 8) && (flags & 1) && c.kv_heads <= 8) { kw.kslab = kl; kw.vslab = vl; }
            const QkvFront F{x, x, c.emb_qs, c.emb_d, L.in_norm, L.Wqkv, L.bqkv, c.qkv_pairs + (size_t)li * L.qkv_N, c.eps, c.vocab, L.qkv_N, c.H, grid_q, kw};
            const OProjRole P{L.Wo, x, t, c.attn_pairs + (size_t)li * Ko, c.poll_err, c.H, Ko, grid_attn, rows};
            const size_t plds = c.D == 128 ? merged_front_lds<128>(c) : merged_front_lds<64>(c);
#define FRONT_CASE(DD, EMB)                                                                                                                                   \
    {                                                                                                                                                         \
        rc = allow_lds(dec_qkv_attn_oproj_kernel<DD, 2, EMB>, plds);                                                                                          \
        if (rc) return rc;                                                                                                                                    \
        hipLaunchKernelGGL((dec_qkv_attn_oproj_kernel<DD, 2, EMB>), dim3(grid_q + grid_attn + grid_o), dim3(DEC_PIPE_NT), plds, st, c.state, c.cur_sin, c.cur_cos, kl, vl, c.heads, \
                           c.kv_heads, c.cache_limit, c.vt_ld, flags, attn_groups, ww, F, P);                                                                 \
    }
            if (c.D == 128) FRONT_CASE(128, false) else FRONT_CASE(64, false)
#undef FRONT_CASE
            return MH_LAUNCH_OK("dec_qkv_attn_oproj");
        }
        // bit 3 of the attention flags (set by default): dec_qkv warms the L2 with the layer's cache rows for the attention launch that follows
        const int aflags = c.attn_flags;
        KvWarm kw{nullptr, nullptr, nullptr, c.kv_heads, c.D, c.kv_heads * c.D, c.vt_ld, c.cache_limit};
        if ((aflags & 8) && (aflags & 1) && c.kv_heads <= 8) {
            kw.kslab = c.kslab + (size_t)li * c.cache_limit * c.kv_heads * c.D;
            kw.vslab = c.vslab + (size_t)li * c.kv_heads * c.D * c.vt_ld;
        }
        NS_DISPATCH(c.H, rc = launch_norm_gemv<NS>(c, L.in_norm, c.eps, L.Wqkv, L.bqkv, L.qkv_N, li == 0, x, x, c.qkv, st, kw));
        return rc;
    }
    case 1: {
        if (decode_merges_chain(c, layers, li - 1) || decode_merges_front(c, L, li)) return MLLM_HIP_OK;      // done inside the q|k|v launch (or the previous down-projection's)
        uint16_t *kl = c.kslab + (size_t)li * c.cache_limit * c.kv_heads * c.D, *vl = c.vslab + (size_t)li * c.kv_heads * c.D * c.vt_ld;
        const int nslots = decode_lds_slots(c.cache_limit, c.D, DEC_ATTN_NT, 2, true);
        const size_t lds = decode_lds_bytes(c.cache_limit, c.D, DEC_ATTN_NT, 2, nslots, true);
        // bit 0: XCD placement of a K/V group's heads, bit 1: two-stage key fetch (both neutral in time at T = 290..430, profiles/r02_attn_experiments.md;
        // the second keeps the fetched bytes near the algorithmic ones at short contexts)
        const int flags = c.attn_flags;
        const int ds_env = option(OPT_ATTN_DS) > 0 ? option(OPT_ATTN_DS) : 0;     // workgroups per head (1, 2 or 4); 0 = default
        const int ds = ds_env == 1 || ds_env == 2 || ds_env == 4 ? ds_env : 2;
        const dim3 grid((flags & 1) ? dec_attn_grid(c.heads, c.kv_heads, ds) : c.heads * ds);
        // the workgroups the attention does not need read this layer's gate|up (and o-projection) rows, so that the XCD L2s hold them when those launches arrive
        const int attn_groups = (int)grid.x / 8;
        const WeightWarm *ww = (flags & 1) && !(flags & 4) && c.warm_tab ? c.warm_tab + li : nullptr;
        // bit 2 of the flags (unset by default) keeps the un-pipelined kernel; caches beyond 2048 keys (more than 64 blocks: the carry is taken by one wave pass) stay on it too
        // option "merge_o": the o-projection rides in the attention's launch (K <= 2048: the register form dec_proj_kernel<1, 2, 8>); case 2 below is then a no-op
        if (decode_merges_o(c)) {
            const size_t plds = c.D == 128 ? merged_o_lds<128>(c) : merged_o_lds<64>(c);
            const int grid_attn = (int)grid.x + (ww ? 8 * std::max(8, WeightWarm::GROUPS - attn_groups) : 0);
            const int rows = c.merge_o == 2 ? 1 : 2;      // rows per wave of the projection role (1: twice the workgroups, on CUs the attention leaves idle anyway)
            const int K = c.heads * c.D, waves = (c.H + rows - 1) / rows;
            const OProjRole P{L.Wo, x, t, c.attn_pairs + (size_t)li * K, c.poll_err, c.H, K, grid_attn, rows};
#define MERGED_O_CASE(DD)                                                                                                                                      \
    {                                                                                                                                                         \
        rc = allow_lds(dec_attn_oproj_kernel<DD, 2>, plds);                                                                                                   \
        if (rc) return rc;                                                                                                                                    \
        hipLaunchKernelGGL((dec_attn_oproj_kernel<DD, 2>), dim3(grid_attn + (waves + 7) / 8), dim3(DEC_PIPE_NT), plds, st, c.state, c.qkv, c.cur_sin, c.cur_cos, kl, vl, c.heads, \
                           c.kv_heads, c.cache_limit, c.vt_ld, flags, attn_groups, ww, P);                                                                    \
    }
            if (c.D == 128) MERGED_O_CASE(128) else MERGED_O_CASE(64)
#undef MERGED_O_CASE
            return MH_LAUNCH_OK("dec_attn_oproj");
        }
        if (!(flags & 4) && c.cache_limit <= 2048 && (c.D == 128 || c.D == 64)) {
#define DEC_PIPE_CASE(DD, DSV)                                                                                                                            \
    {                                                                                                                                                     \
        constexpr int NP_ = DEC_PIPE_NT / 64 - ((DD / DSV + 63) / 64) - 1;                                                                                \
        const size_t plds = pipe_lds_bytes<DD, DD / DSV>(c.cache_limit, NP_);                                                                             \
        if (plds <= 160 * 1024 - 2 * DD * 2 - 64) {                                                                                                       \
            rc = allow_lds(dec_attn_pipe_kernel<DD, DSV>, plds);                                                                                          \
            if (rc) return rc;                                                                                                                            \
            hipLaunchKernelGGL((dec_attn_pipe_kernel<DD, DSV>), dim3(grid.x + (ww ? 8 * std::max(8, WeightWarm::GROUPS - attn_groups) : 0)), dim3(DEC_PIPE_NT), plds, st, c.state, c.qkv, c.cur_sin, c.cur_cos, kl, vl, c.fa_ws, \
                               c.heads, c.kv_heads, c.cache_limit, c.vt_ld, flags, attn_groups, ww);                                                      \
            return MH_LAUNCH_OK("dec_attn_pipe");                                                                                                         \
        }                                                                                                                                                 \
    }
            if (c.D == 128) { if (ds == 1) DEC_PIPE_CASE(128, 1) else if (ds == 2) DEC_PIPE_CASE(128, 2) else DEC_PIPE_CASE(128, 4) }
            else { if (ds == 1) DEC_PIPE_CASE(64, 1) else if (ds == 2) DEC_PIPE_CASE(64, 2) else DEC_PIPE_CASE(64, 4) }
#undef DEC_PIPE_CASE
        }
#define DEC_ATTN_CASE(DD, DSV)                                                                                                                          \
    {                                                                                                                                                   \
        rc = allow_lds(dec_attn_kernel<DD, DSV>, lds);                                                                                                  \
        if (rc) return rc;                                                                                                                              \
        hipLaunchKernelGGL((dec_attn_kernel<DD, DSV>), grid, dim3(DEC_ATTN_NT), lds, st, c.state, c.qkv, c.cur_sin, c.cur_cos, kl, vl, c.fa_ws, c.heads, \
                           c.kv_heads, c.cache_limit, c.vt_ld, nslots, flags);                                                                          \
    }
        if (c.D == 128) { if (ds == 1) DEC_ATTN_CASE(128, 1) else if (ds == 2) DEC_ATTN_CASE(128, 2) else DEC_ATTN_CASE(128, 4) }
        else { if (ds == 1) DEC_ATTN_CASE(64, 1) else if (ds == 2) DEC_ATTN_CASE(64, 2) else DEC_ATTN_CASE(64, 4) }
#undef DEC_ATTN_CASE
        return MH_LAUNCH_OK("dec_attn");
    }
    case 2:
        if (decode_merges_o(c)) return MLLM_HIP_OK;      // done inside the attention's launch 
        NS_DISPATCH(c.heads * c.D, rc = (launch_proj<NS>(L.Wo, L.Wo_raw, c.fa_ws, x, t, c.H, c.heads * c.D, st)));
        return rc;
    case 3:
        NS_DISPATCH(c.H, rc = launch_gateup<NS>(L, c, t, st));
        return rc;
    case 4:
        if (decode_merges_chain(c, layers, li)) {
            const int ln = li + 1;
            const DecodeLayer &Ln = layers[ln];
            uint16_t *kl = c.kslab + (size_t)ln * c.cache_limit * c.kv_heads * c.D, *vl = c.vslab + (size_t)ln * c.kv_heads * c.D * c.vt_ld;
            const int flags = c.attn_flags, ds = 2;
            const int grid_a0 = (flags & 1) ? dec_attn_grid(c.heads, c.kv_heads, ds) : c.heads * ds, attn_groups = grid_a0 / 8;
            const WeightWarm *ww = (flags & 1) && !(flags & 4) && c.warm_tab ? c.warm_tab + ln : nullptr;
            const int grid_attn = grid_a0 + (ww ? 8 * std::max(8, WeightWarm::GROUPS - attn_groups) : 0);
            const int grid_q = ((Ln.qkv_N + 1) / 2 + 7) / 8, rows = 1, Ko = c.heads * c.D, grid_o = ((c.H + rows - 1) / rows + 7) / 8;
            const int rpw = pjb_rows_per_wg(c.H, c.I), grid_d = (c.H + rpw - 1) / rpw;
            KvWarm kw{nullptr, nullptr, nullptr, c.kv_heads, c.D, c.kv_heads * c.D, c.vt_ld, c.cache_limit};
            if ((flags & 8) && (flags & 1) && c.kv_heads <= 8) { kw.kslab = kl; kw.vslab = vl; }
            const int cont = grid_q <= grid_d && option(OPT_CHAIN_CONT) != 0;      // the q|k|v role runs in the first down-projection workgroups once they are through (+1 %)
            const DownRole Dn{c.act, L.Wdown_raw, t, x, c.x_pairs + (size_t)li * c.H, c.H, c.I, rpw, grid_d, cont};
            const QkvFront F{nullptr, nullptr, c.emb_qs, c.emb_d, Ln.in_norm, Ln.Wqkv, Ln.bqkv, c.qkv_pairs + (size_t)ln * Ln.qkv_N, c.eps, c.vocab, Ln.qkv_N, c.H, grid_q, kw};
            const OProjRole P{Ln.Wo, nullptr, t, c.attn_pairs + (size_t)ln * Ko, c.poll_err, c.H, Ko, grid_attn, rows};
            const size_t plds = c.D == 128 ? merged_chain_lds<128>(c) : merged_chain_lds<64>(c);
#define CHAIN_CASE(DD, NSV)                                                                                                                                   \
    {                                                                                                                                                         \
        rc = allow_lds(dec_down_front_kernel<DD, 2, NSV>, plds);                                                                                              \
        if (rc) return rc;                                                                                                                                    \
        hipLaunchKernelGGL((dec_down_front_kernel<DD, 2, NSV>), dim3(grid_d + (cont ? 0 : grid_q) + grid_attn + grid_o), dim3(DEC_PIPE_NT), plds, st, c.state, c.cur_sin, c.cur_cos, kl, vl,      \
                           c.heads, c.kv_heads, c.cache_limit, c.vt_ld, flags, attn_groups, ww, Dn, F, P);                                                    \
    }
            const int NSd = (c.I / 256 + 7) / 8;
            if (c.D == 128) { if (NSd == 3) CHAIN_CASE(128, 3) else if (NSd == 4) CHAIN_CASE(128, 4) else CHAIN_CASE(128, 5) }
            else { if (NSd == 3) CHAIN_CASE(64, 3) else if (NSd == 4) CHAIN_CASE(64, 4) else CHAIN_CASE(64, 5) }
#undef CHAIN_CASE
            return MH_LAUNCH_OK("dec_down_front");
        }
        NS_DISPATCH(c.I, rc = (launch_proj<NS>(L.Wdown, L.Wdown_raw, c.act, t, x, c.H, c.I, st)));
        return rc;
    }
    return MLLM_HIP_ERR_ARG;
}

// argmax of the logits row with the engine's partials scratch: nparts workgroups, then the fold (prefill: into *out; decode: dec_next_kernel folds and advances the state)
static int argmax_parts_count(const DecodeCtx &c) { return std::max(1, std::min(c.max_parts, 128)); }
int argmax_row_launch(const DecodeCtx &c, const float *logits, int n, int *out, hipStream_t st) {
    const int np = argmax_parts_count(c);
    const int rc = argmax_parts_launch(logits, n, c.part_val, c.part_idx, np, st);
    return rc ? rc : argmax_final_launch(c.part_val, c.part_idx, np, out, st);
}
static int argmax_and_advance(const DecodeCtx &c, hipStream_t st) {
    const int np = argmax_parts_count(c);
    int rc = argmax_parts_launch(c.logits, c.vocab, c.part_val, c.part_idx, np, st);
    if (rc) return rc;
    hipLaunchKernelGGL(dec_next_kernel, dim3(1), dim3(256), 0, st, c.state, c.part_val, c.part_idx, np, c.tok_dev, c.history, c.rope_sin, c.rope_cos, c.cur_sin, c.cur_cos, c.D / 2,
                       c.cache_limit);
    return MH_LAUNCH_OK("dec_next");
}

// which slot (layer, kernel) of the step is a launch of its own, and of which kind (StepMarks): the merged forms leave the slots they swallow empty
static int step_slot_kind(const DecodeCtx &c, const DecodeLayer *layers, int li, int k) {
    const bool chained = decode_merges_chain(c, layers, li - 1), front = !chained && decode_merges_front(c, layers[li], li);
    switch (k) {
    case 0: return chained ? -1 : (front ? 6 : 0);
    case 1: return chained || front ? -1 : 1;
    case 2: return chained || front || decode_merges_o(c) ? -1 : 2;
    case 3: return 3;
    default: return decode_merges_chain(c, layers, li) ? 5 : 4;
    }
}
int decode_step_launch(const DecodeCtx &c, const DecodeLayer *layers, int n_layers, hipStream_t st, const StepMarks *marks) {
    float *x = c.x0;
    int rc = 0;
#define MARK(kind, after) do { if (marks && (rc = marks->mark(marks->user, kind, after))) return rc; } while (0)
    for (int li = 0; li < n_layers; ++li)
        for (int k = 0; k < 5; ++k) {
            const int kind = marks ? step_slot_kind(c, layers, li, k) : -1;
            if (kind >= 0) MARK(kind, 0);
            rc = decode_kernel_launch(c, layers, li, k, st);
            if (rc) return rc;
            if (kind >= 0) MARK(kind, 1);
        }
    if (c.Whead) {
        // Linear lm_head (LLaMA-style models): model.norm -> Q8_K -> Q4_K rows, the same fused kernel as the q|k|v projection; then argmax
        MARK(7, 0);
        NS_DISPATCH(c.H, rc = launch_norm_gemv<NS>(c, c.final_norm, c.final_eps, c.Whead, nullptr, c.vocab, false, x, x, c.logits, st));
        if (rc) return rc;
        MARK(7, 1);
        MARK(8, 0);
        rc = argmax_and_advance(c, st);
        if (rc) return rc;
        MARK(8, 1);
        return 0;
    }
    // tied lm_head + argmax
    if (c.H % 512 != 0 || c.H / 512 > 8) {
        // shapes the fused head kernel does not cover: the stand-alone launchers (same arithmetic), then advance the state
        MARK(7, 0);
        rc = mllm_hip_rmsnorm(x, c.final_norm, c.normed, nullptr, nullptr, nullptr, 1, c.H, c.final_eps, 0, st);
        if (!rc) rc = mllm_hip_quantize_q80(c.normed, c.x80_qs, c.x80_d, 1, c.H, st);
        if (!rc) rc = mllm_hip_linear_q40_q80(c.emb_qs, c.emb_d, nullptr, c.x80_qs, c.x80_d, c.logits, c.vocab, 1, c.vocab, c.H, st);
        if (rc) return rc;
        MARK(7, 1);
        MARK(8, 0);
        rc = argmax_and_advance(c, st);
        if (rc) return rc;
        MARK(8, 1);
        return 0;
    }
    const int head_wpc = option(OPT_HEAD_WPC) > 0 ? option(OPT_HEAD_WPC) : 8;   // waves per CU the row split aims at
    const int target_waves = 256 * head_wpc;
    int rpw = (c.vocab + target_waves - 1) / target_waves;
    rpw = ((rpw + 7) / 8) * 8;
    const int waves = (c.vocab + rpw - 1) / rpw, blocks = (waves + 3) / 4;
    if (blocks > c.max_parts) return MLLM_HIP_ERR_SHAPE;
    const size_t lds = (((size_t)c.H * 5 + (size_t)c.H / 32 * 4 + 15) & ~(size_t)15) + 4 * q40_tab_floats(c.H / 32) * sizeof(float);
#define HEAD_CASE(B) case B: rc = allow_lds(dec_head_kernel<B>, lds); if (rc) return rc; hipLaunchKernelGGL((dec_head_kernel<B>), dim3(blocks), dim3(256), lds, st, x, c.final_norm, c.final_eps, c.emb_qs, c.emb_d, c.logits, c.part_val, c.part_idx, c.vocab, c.H, rpw); break;
    MARK(7, 0);
    switch (c.H / 512) { HEAD_CASE(1) HEAD_CASE(2) HEAD_CASE(3) HEAD_CASE(4) HEAD_CASE(5) HEAD_CASE(6) HEAD_CASE(7) HEAD_CASE(8) }
#undef HEAD_CASE
    rc = MH_LAUNCH_OK("dec_head");
    if (rc) return rc;
    MARK(7, 1);
    MARK(8, 0);
    hipLaunchKernelGGL(dec_next_kernel, dim3(1), dim3(256), 0, st, c.state, c.part_val, c.part_idx, blocks, c.tok_dev, c.history, c.rope_sin, c.rope_cos, c.cur_sin, c.cur_cos,
                       c.D / 2, c.cache_limit);
    rc = MH_LAUNCH_OK("dec_next");
    if (rc) return rc;
    MARK(8, 1);
    return 0;
#undef MARK
}
}  // namespace mllm_hip

extern "C" int mllm_hip_row_fused_launch(const mllm_hip_row_fused *args, void *stream) {
    if (!args) return MLLM_HIP_ERR_ARG;
    return mllm_hip::row_fused_launch(*args, mllm_hip::as_stream(stream));
}
extern "C" int mllm_hip_row_fused_supported(const mllm_hip_row_fused *args) { return args ? mllm_hip::row_fused_supported(*args) : 0; }
