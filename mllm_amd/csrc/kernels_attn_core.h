// mllm_amd/csrc/kernels_attn_core.h -- device pieces of the reference-order attention shared by kernels_attn.hip (stand-alone
// launchers) and kernels_decode.hip (fused decode step).  See kernels_attn.hip for the evaluation order being reproduced.
#pragma once
#include "common.h"

namespace mllm_hip {

constexpr float FA_NEG = -3.402823466e+38f;   // std::numeric_limits<float>::lowest() (FlashAttention2.hpp:25)
constexpr int FA_KC = 256;                    // keys per chunk (one per thread)

template <bool F16>
__device__ __forceinline__ float kv_at(const void *p, int64_t i) {
    return F16 ? h2f(reinterpret_cast<const uint16_t *>(p)[i]) : reinterpret_cast<const float *>(p)[i];
}
template <int D, bool F16>
__device__ __forceinline__ void load_kv_row(float (&kr)[D], const void *base, int64_t off) {
    if (F16) {
        const uint4 *p = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(base) + off);
#pragma unroll
        for (int i = 0; i < D / 8; ++i) {
            const uint4 w = p[i];
            const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { kr[8 * i + 2 * e] = h2f((uint16_t)(ww[e] & 0xffff)); kr[8 * i + 2 * e + 1] = h2f((uint16_t)(ww[e] >> 16)); }
        }
    } else {
        const float4 *p = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(base) + off);
#pragma unroll
        for (int i = 0; i < D / 4; ++i) { const float4 w = p[i]; kr[4 * i] = w.x; kr[4 * i + 1] = w.y; kr[4 * i + 2] = w.z; kr[4 * i + 3] = w.w; }
    }
}
// mma0 of one (row, key): q row in LDS (broadcast reads), key row in registers
template <int D>
__device__ __forceinline__ float qk_dot(const float *q, const float (&kr)[D]) {
    float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < D / 8; ++i) {
        const float4 a = *reinterpret_cast<const float4 *>(q + 8 * i), b = *reinterpret_cast<const float4 *>(q + 8 * i + 4);
        l[0] = __fmaf_rn(a.x, kr[8 * i + 0], l[0]); l[1] = __fmaf_rn(a.y, kr[8 * i + 1], l[1]);
        l[2] = __fmaf_rn(a.z, kr[8 * i + 2], l[2]); l[3] = __fmaf_rn(a.w, kr[8 * i + 3], l[3]);
        l[4] = __fmaf_rn(b.x, kr[8 * i + 4], l[4]); l[5] = __fmaf_rn(b.y, kr[8 * i + 5], l[5]);
        l[6] = __fmaf_rn(b.z, kr[8 * i + 6], l[6]); l[7] = __fmaf_rn(b.w, kr[8 * i + 7], l[7]);
    }
    return ((l[0] + l[4]) + (l[1] + l[5])) + ((l[2] + l[6]) + (l[3] + l[7]));
}

// ------------------------------------------------------------------------------------------------------------------
// Sq == 1: __fa2_decode (:225-274 / :1346-1394) = the same recurrence with one key per tile.  One workgroup per head.
// pc: LDS array of (p_j, c_j) pairs for j < Sk.  q in LDS.  Returns through o_out[d] for tid < D.
// knew / vnew (optional, LDS fp16 rows) stand for key position `tnew` (the row this step appends).
// ------------------------------------------------------------------------------------------------------------------
template <int D, bool F16, int NT>
__device__ __forceinline__ void fa2_decode_head(const float *qs, const void *K, int64_t ldk, const void *V, int64_t ldv, int kvoff, int Sk, float2 *pc,
                                                float *wred /* [NT/64 + 2] */, const uint16_t *knew, const uint16_t *vnew, int tnew, float *o_out) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float scale = 1.0f / sqrtf((float)D);
    float carry = FA_NEG;
    for (int base = 0; base < Sk; base += NT) {
        const int j = base + tid;
        float s = FA_NEG;
        if (j < Sk) {
            float kr[D];
            if (knew && j == tnew) {
#pragma unroll
                for (int i = 0; i < D; ++i) kr[i] = h2f(knew[i]);
            } else {
                load_kv_row<D, F16>(kr, K, (int64_t)j * ldk + kvoff);
            }
            s = qk_dot<D>(qs, kr);
        }
        const float wincl = wave_scan_max(s);
        if (lane == 63) wred[wid] = wincl;
        __syncthreads();
        float before = carry;
        for (int w = 0; w < wid; ++w) before = fmaxf(before, wred[w]);
        const float incl = fmaxf(wincl, before);
        const float excl = wave_shift_up(incl, before);
        if (j < Sk) {
            const float cc = excl == incl ? 1.0f : glibc_expf((excl - incl) * scale);
            pc[j] = make_float2(glibc_expf((s - incl) * scale), cc);
        }
        float tot = carry;
        for (int w = 0; w < NT / 64; ++w) tot = fmaxf(tot, wred[w]);
        carry = tot;
        __syncthreads();
    }
    // sequential part: logsum (thread D) and o[d] (threads < D)
    if (tid < D) {
        float o = 0.0f;
        int j = 0;
        for (; j + 8 <= Sk; j += 8) {
            float vv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) vv[k] = (vnew && j + k == tnew) ? h2f(vnew[tid]) : kv_at<F16>(V, (int64_t)(j + k) * ldv + kvoff + tid);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float2 e = pc[j + k];
                const float cu = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(e.y)));
                if (cu != 1.0f) o = o * cu;
                o = __fmaf_rn(e.x, vv[k], o);
            }
        }
        for (; j < Sk; ++j) {
            const float v = (vnew && j == tnew) ? h2f(vnew[tid]) : kv_at<F16>(V, (int64_t)j * ldv + kvoff + tid);
            const float2 e = pc[j];
            o = o * e.y;
            o = __fmaf_rn(e.x, v, o);
        }
        o_out[tid] = o;
    } else if (tid == ((D + 63) & ~63)) {   // first lane of the wave after the o-waves
        float l = 0.0f;
        for (int j = 0; j < Sk; ++j) { const float2 e = pc[j]; l = __fmaf_rn(l, e.y, e.x); }
        wred[NT / 64] = l;
    }
    __syncthreads();
    if (tid < D) o_out[tid] = o_out[tid] * (1.0f / wred[NT / 64]);
}

}  // namespace mllm_hip
