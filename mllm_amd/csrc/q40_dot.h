// mllm_amd/csrc/q40_dot.h -- the Q4_0 x Q8_0 row dot product (tied lm_head) in vec_dot_q4_0_q8_0's accumulation order
// (VecDotQ4.cpp:514-545); shared by gemv_q40_kernel (kernels_linear.hip) and dec_head_kernel (kernels_decode.hip).
#pragma once
#include "common.h"
#include "q4k_dot.h"

namespace mllm_hip {

template <int BPL>
struct Q40Act { int4 xa[BPL], xb[BPL]; float xd[BPL]; int4 ya8[BPL], yb8[BPL]; };   // ya8/yb8: 8 * (sum of the 4 bytes of each dword)

template <int BPL, int LPR>
__device__ __forceinline__ void q40_load_act(Q40Act<BPL> &A, const int8_t *xqs, const float *xdf, const uint16_t *xdh, int sub) {
#pragma unroll
    for (int b = 0; b < BPL; ++b) {
        const int blk = sub + LPR * b;
        A.xa[b] = *reinterpret_cast<const int4 *>(xqs + blk * 32);
        A.xb[b] = *reinterpret_cast<const int4 *>(xqs + blk * 32 + 16);
        A.xd[b] = xdf ? xdf[blk] : h2f(xdh[blk]);
        const int one = 0x01010101;
        A.ya8[b] = make_int4(8 * dot4(A.xa[b].x, one, 0), 8 * dot4(A.xa[b].y, one, 0), 8 * dot4(A.xa[b].z, one, 0), 8 * dot4(A.xa[b].w, one, 0));
        A.yb8[b] = make_int4(8 * dot4(A.xb[b].x, one, 0), 8 * dot4(A.xb[b].y, one, 0), 8 * dot4(A.xb[b].z, one, 0), 8 * dot4(A.xb[b].w, one, 0));
    }
}
// table of one wave: ts[rows][nblk][8] floats then td[rows][nblk], each row padded by 8 floats: with the bare pitches (8 * nblk and nblk floats, multiples of 64 and 16
// dwords at K = 1536) the eight rows a chain step reads at once fell into the same eight LDS banks -- SQ_LDS_BANK_CONFLICT was 12 us of dec_head's 35 (profiles/r03)
__host__ __device__ constexpr int q40_ts_stride(int nblk) { return nblk * 8 + 8; }
__host__ __device__ constexpr int q40_td_stride(int nblk) { return nblk + 8; }
template <int BPL, int LPR>
__device__ __forceinline__ void q40_emit(const uint4 (&q)[BPL], const uint16_t (&dw)[BPL], const Q40Act<BPL> &A, int sub, float *ts_row, float *td_row) {
#pragma unroll
    for (int b = 0; b < BPL; ++b) {
        const int blk = sub + LPR * b;
        float4 lo, hi;
        lo.x = (float)(dot4((int)(q[b].x & 0x0f0f0f0fu), A.xa[b].x, 0) - A.ya8[b].x);
        lo.y = (float)(dot4((int)(q[b].y & 0x0f0f0f0fu), A.xa[b].y, 0) - A.ya8[b].y);
        lo.z = (float)(dot4((int)(q[b].z & 0x0f0f0f0fu), A.xa[b].z, 0) - A.ya8[b].z);
        lo.w = (float)(dot4((int)(q[b].w & 0x0f0f0f0fu), A.xa[b].w, 0) - A.ya8[b].w);
        hi.x = (float)(dot4((int)((q[b].x >> 4) & 0x0f0f0f0fu), A.xb[b].x, 0) - A.yb8[b].x);
        hi.y = (float)(dot4((int)((q[b].y >> 4) & 0x0f0f0f0fu), A.xb[b].y, 0) - A.yb8[b].y);
        hi.z = (float)(dot4((int)((q[b].z >> 4) & 0x0f0f0f0fu), A.xb[b].z, 0) - A.yb8[b].z);
        hi.w = (float)(dot4((int)((q[b].w >> 4) & 0x0f0f0f0fu), A.xb[b].w, 0) - A.yb8[b].w);
        *reinterpret_cast<float4 *>(ts_row + blk * 8) = lo;
        *reinterpret_cast<float4 *>(ts_row + blk * 8 + 4) = hi;
        td_row[blk] = h2f(dw[b]) * A.xd[b];   // fp16(x.d) * fp16(y.d)
    }
}
// 8 lanes per row, 8 rows per wave: lane 8*rr + c walks class [0,4,2,6,1,5,3,7][c]; every lane of the row ends with the row's dot
__device__ __forceinline__ float q40_chain(const float *ts, const float *td, int nblk, int nrows, int lane) {
    const int rr = lane >> 3, c = lane & 7;
    const int cls = (c & 1) * 4 + bitrev2(c >> 1);
    float acc = 0.0f;
    if (rr < nrows) {
        const float *s = ts + (size_t)rr * q40_ts_stride(nblk) + cls, *d = td + (size_t)rr * q40_td_stride(nblk);
#pragma unroll 8
        for (int i = 0; i < nblk; ++i) acc = __fmaf_rn(d[i], s[i * 8], acc);
    }
    acc += MH_DPPF(0.0f, acc, DPP_QUAD_X1, 0xF);
    acc += MH_DPPF(0.0f, acc, DPP_QUAD_X2, 0xF);
    acc += MH_DPPF(0.0f, acc, DPP_HALF_MIRROR, 0xF);
    return acc;
}
constexpr size_t q40_tab_floats(int nblk) { return (size_t)8 * (q40_ts_stride(nblk) + q40_td_stride(nblk)); }   // per wave: 8 rows x (8 sums + 1 scale) per block, padded rows

}  // namespace mllm_hip
