// mllm_amd/csrc/q4k_dot.h -- the Q4_K x Q8_K row dot product of the GEMV kernels, in the reference's accumulation order.
//
// vec_dot_q4_K_q8_K (VecDotQ4.cpp:220-283, AVX2) keeps 8 fp32 lanes `acc` and 4 lanes `acc_m`:
//     acc[t]   = fma(y.d * fp16(x.d),     float(sumi[t]),  acc[t])    t = 4-byte column class of every 32-byte chunk
//     acc_m[u] = fma(-y.d * fp16(x.dmin), float(prod[u]),  acc_m[u])  prod[u] = mn[2u] q8s[2u] + mn[2u+1] q8s[2u+1]
// once per super-block, in super-block order, then hsum_float_8(acc) + ((acc_m0 + acc_m2) + (acc_m1 + acc_m3)).
// fp32 addition is not associative, so the kernels reproduce exactly these 12 chains:
//   * the 8 lanes that share a super-block split its 128 nibble bytes as (64-weight chunk j, 16-byte half tp); dword c of
//     a lane's 16 bytes is column class t = 4 tp + c.  Lanes {0,2,5,7} / {1,3,4,6} of a group hold tp = 0 / 1, so the two
//     DPP steps quad_perm[2,3,0,1] + row_half_mirror sum the four chunks of a class exactly (integers);
//   * every lane then converts ONE class sum (and lanes 0..3 one mins product) and drops (scale, value) into a per-wave LDS
//     table [row][super-block][12 slots]; 12 lanes per row walk the table in super-block order with one fmaf per entry;
//   * slots are ordered so that the reference's final horizontal adds are DPP quad_perm / half_mirror / row_shr steps.
#pragma once
#include "common.h"

namespace mllm_hip {

constexpr int Q4K_SLOTS = 12;

__device__ __forceinline__ int q4k_lane_j(int lane) { return (lane & 7) >> 1; }
__device__ __forceinline__ int q4k_lane_tp(int lane) { return ((lane & 7) ^ ((lane & 7) >> 2)) & 1; }
// byte offset of this lane's 16 nibble bytes inside the 128-byte qs of a super-block (and /2 of its activation offset)
__device__ __forceinline__ int q4k_lane_qoff(int lane) { return 32 * q4k_lane_j(lane) + 16 * q4k_lane_tp(lane); }
__device__ __forceinline__ int bitrev2(int v) { return ((v & 1) << 1) | (v >> 1); }

template <int NSTEPS>
struct Q4KAct {   // this lane's slice of the Q8_K activation row, per wave step
    int4 xa[NSTEPS], xb[NSTEPS];
    float xd[NSTEPS];
    int mq0[NSTEPS], mq1[NSTEPS];
    bool valid[NSTEPS];
};

// q8s32: int32 sums of 32 consecutive q8 values (8 per super-block)
template <int NSTEPS>
__device__ __forceinline__ void q4k_load_act(Q4KAct<NSTEPS> &A, const int8_t *qs, const float *d, const int *q8s32, int nb, int lane) {
    const int g = lane >> 3, u = lane & 3, j = q4k_lane_j(lane), tp = q4k_lane_tp(lane);
#pragma unroll
    for (int st = 0; st < NSTEPS; ++st) {
        const int blk = st * 8 + g;
        A.valid[st] = blk < nb;
        const int b = A.valid[st] ? blk : 0;
        A.xa[st] = *reinterpret_cast<const int4 *>(qs + b * 256 + 64 * j + 16 * tp);
        A.xb[st] = *reinterpret_cast<const int4 *>(qs + b * 256 + 64 * j + 32 + 16 * tp);
        A.xd[st] = d[b];
        A.mq0[st] = q8s32[b * 8 + 2 * u];
        A.mq1[st] = q8s32[b * 8 + 2 * u + 1];
    }
}
// same from the block_q8_K planes in global memory (bsums: int16 sums of 16)
template <int NSTEPS>
__device__ __forceinline__ void q4k_load_act_planes(Q4KAct<NSTEPS> &A, const int8_t *qs, const float *d, const int16_t *bsums, int nb, int lane) {
    const int g = lane >> 3, u = lane & 3, j = q4k_lane_j(lane), tp = q4k_lane_tp(lane);
#pragma unroll
    for (int st = 0; st < NSTEPS; ++st) {
        const int blk = st * 8 + g;
        A.valid[st] = blk < nb;
        const int b = A.valid[st] ? blk : 0;
        A.xa[st] = *reinterpret_cast<const int4 *>(qs + b * 256 + 64 * j + 16 * tp);
        A.xb[st] = *reinterpret_cast<const int4 *>(qs + b * 256 + 64 * j + 32 + 16 * tp);
        A.xd[st] = d[b];
        const int16_t *bs = bsums + b * 16 + 4 * u;
        A.mq0[st] = (int)bs[0] + (int)bs[1];
        A.mq1[st] = (int)bs[2] + (int)bs[3];
    }
}

// (scale, value) table entries of one (row, wave step): hdr = first 16 bytes of the super-block, q = this lane's nibbles
template <int NSTEPS>
__device__ __forceinline__ void q4k_emit(const uint4 hdr, const uint4 q, const Q4KAct<NSTEPS> &A, int st, int lane, float2 *row_tab /* [NSTEPS*8][12] */) {
    const int g = lane >> 3, r = lane & 7, j = q4k_lane_j(lane), tp = q4k_lane_tp(lane), u = r & 3;
    const float d = h2f((uint16_t)(hdr.x & 0xffff)), dmin = h2f((uint16_t)(hdr.x >> 16));
    uint32_t sc8[2], mn8[2];
    unpack_q4k_scales(hdr.y, hdr.z, hdr.w, sc8, mn8);
    const int sc_lo = byte_of(sc8, 2 * j), sc_hi = byte_of(sc8, 2 * j + 1);
    int s0 = sc_lo * dot4((int)(q.x & 0x0f0f0f0fu), A.xa[st].x, 0) + sc_hi * dot4((int)((q.x >> 4) & 0x0f0f0f0fu), A.xb[st].x, 0);
    int s1 = sc_lo * dot4((int)(q.y & 0x0f0f0f0fu), A.xa[st].y, 0) + sc_hi * dot4((int)((q.y >> 4) & 0x0f0f0f0fu), A.xb[st].y, 0);
    int s2 = sc_lo * dot4((int)(q.z & 0x0f0f0f0fu), A.xa[st].z, 0) + sc_hi * dot4((int)((q.z >> 4) & 0x0f0f0f0fu), A.xb[st].z, 0);
    int s3 = sc_lo * dot4((int)(q.w & 0x0f0f0f0fu), A.xa[st].w, 0) + sc_hi * dot4((int)((q.w >> 4) & 0x0f0f0f0fu), A.xb[st].w, 0);
    s0 += MH_DPP(0, s0, DPP_QUAD_X2, 0xF); s1 += MH_DPP(0, s1, DPP_QUAD_X2, 0xF); s2 += MH_DPP(0, s2, DPP_QUAD_X2, 0xF); s3 += MH_DPP(0, s3, DPP_QUAD_X2, 0xF);
    s0 += MH_DPP(0, s0, DPP_HALF_MIRROR, 0xF); s1 += MH_DPP(0, s1, DPP_HALF_MIRROR, 0xF);
    s2 += MH_DPP(0, s2, DPP_HALF_MIRROR, 0xF); s3 += MH_DPP(0, s3, DPP_HALF_MIRROR, 0xF);
    const int sel = j == 0 ? s0 : (j == 1 ? s1 : (j == 2 ? s2 : s3));    // class t = 4 tp + j
    const int prod = byte_of(mn8, 2 * u) * A.mq0[st] + byte_of(mn8, 2 * u + 1) * A.mq1[st];
    const float dy = A.xd[st] * d;              // y.d * fp16(x.d)        (VecDotQ4.cpp:228)
    const float dm = (-A.xd[st]) * dmin;      // -y.d * fp16(x.dmin)    (:229)
    if (A.valid[st]) {
        float2 *e = row_tab + (st * 8 + g) * Q4K_SLOTS;
        e[2 * bitrev2(j) + tp] = make_float2(dy, (float)sel);
        if (r < 4) e[8 + bitrev2(u)] = make_float2(dm, (float)prod);
    }
}

// Walks the table of up to 4 rows: lane 16*rr + c (c < 12) owns chain c of row rr.  Returns, in lane 16*rr + 8, the dot
// product of row rr.  tab = [4][NBP][12] of this wave; nb super-blocks are live.
__device__ __forceinline__ float q4k_chain(const float2 *tab, int nbp, int nb, int nrows, int lane) {
    const int rr = lane >> 4, c = lane & 15;
    float acc = 0.0f;
    if (rr < nrows && c < Q4K_SLOTS) {
        const float2 *e = tab + (size_t)rr * nbp * Q4K_SLOTS + c;
#pragma unroll 6
        for (int i = 0; i < nb; ++i) {
            const float2 v = e[i * Q4K_SLOTS];
            acc = __fmaf_rn(v.x, v.y, acc);
        }
    }
    // slots 0..7 = classes [0,4,2,6,1,5,3,7]: hsum_float_8 = ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)); slots 8..11 = mins [0,2,1,3]
    acc += MH_DPPF(0.0f, acc, DPP_QUAD_X1, 0xF);
    acc += MH_DPPF(0.0f, acc, DPP_QUAD_X2, 0xF);
    const float full = acc + MH_DPPF(0.0f, acc, DPP_HALF_MIRROR, 0xF);
    acc = c < 8 ? full : acc;
    const float hs = MH_DPPF(0.0f, acc, 0x118 /* row_shr:8 */, 0xF);   // lane 8 of each row receives lane 0
    return hs + acc;
}

// wave-level fence between the table writes and the chain reads (other lanes' writes, same wave: LDS ops retire in order)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ROWS rows (already loaded: hdr/q per row and step) against the activation slice; out[rr] is wave-uniform.
template <int NSTEPS, int ROWS>
__device__ __forceinline__ void q4k_dot_rows(const uint4 (&hdr)[ROWS][NSTEPS], const uint4 (&q)[ROWS][NSTEPS], const Q4KAct<NSTEPS> &A, int nb, int lane,
                                             float2 *tab /* this wave's [min(ROWS,4)][NSTEPS*8][12] */, float (&out)[ROWS]) {
    constexpr int NBP = NSTEPS * 8;
#pragma unroll
    for (int p = 0; p < ROWS; p += 4) {
        if (p) wave_lds_fence();
#pragma unroll
        for (int rr = p; rr < (p + 4 < ROWS ? p + 4 : ROWS); ++rr)
#pragma unroll
            for (int st = 0; st < NSTEPS; ++st) q4k_emit<NSTEPS>(hdr[rr][st], q[rr][st], A, st, lane, tab + (size_t)(rr - p) * NBP * Q4K_SLOTS);
        wave_lds_fence();
        const float res = q4k_chain(tab, NBP, nb, ROWS - p < 4 ? ROWS - p : 4, lane);
#pragma unroll
        for (int rr = p; rr < (p + 4 < ROWS ? p + 4 : ROWS); ++rr)
            out[rr] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(res), 16 * (rr - p) + 8));
    }
}
// ---- decode-order weights (engine only) ------------------------------------------------------------------------------------------------
// The resident engine keeps a second copy of every decode Linear with the 128 nibble bytes of each super-block transposed
// (q4k_decode_order_kernel): new dword 4 t + j = old dword 8 j + t, i.e. the 16 bytes a lane loads are column class t of all four
// 64-weight chunks.  Lane (g = lane >> 3, t = lane & 7) then owns class t of super-block 8 st + g whole: the class sum needs no
// cross-lane step and no select, and the integer work is 8 v_dot4 + 8 24-bit mads per lane and row (the GEMV kernels are VALU-bound:
// 735 VALU instructions per wave for 4 rows before this layout, L2-resident weights ran no faster than cold ones).
__device__ __forceinline__ int bitrev3(int v) { return ((v & 1) << 2) | (v & 2) | (v >> 2); }
// Eight independent v_dot4_i32_i8 with a zero accumulator (the builtin picks the tied VOP2 form v_dot4c + one v_mov per dot).
// DOT results have a 3-wait-state hazard against a different VALU instruction that reads or overwrites them (LLVM's
// GCNHazardRecognizer: DotWriteDifferentVALURead / ...Write = 3 on gfx90a+); the compiler does not see through inline asm, so the
// block ends with s_nop 2 -- inside the block the eight results are independent, and the earlier ones are >= 3 instructions old.
__device__ __forceinline__ void dot4z_x8(int (&r)[8], const int (&a)[8], const int (&b)[8]) {
    asm("v_dot4_i32_i8 %0, %8, %16, 0\n\tv_dot4_i32_i8 %1, %9, %17, 0\n\tv_dot4_i32_i8 %2, %10, %18, 0\n\tv_dot4_i32_i8 %3, %11, %19, 0\n\t"
        "v_dot4_i32_i8 %4, %12, %20, 0\n\tv_dot4_i32_i8 %5, %13, %21, 0\n\tv_dot4_i32_i8 %6, %14, %22, 0\n\tv_dot4_i32_i8 %7, %15, %23, 0\n\ts_nop 2"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]),
          "v"(b[5]), "v"(b[6]), "v"(b[7]));
}
template <int NSTEPS>
struct Q4KActC {
    int xa[NSTEPS][4], xb[NSTEPS][4];   // q8 dwords of class t: low-nibble weights 64 j + 4 t .., high-nibble weights 64 j + 32 + 4 t ..
    float xd[NSTEPS];
    int mq0[NSTEPS], mq1[NSTEPS];
    bool valid[NSTEPS];
};
template <int NSTEPS>
__device__ __forceinline__ void q4kc_load_act(Q4KActC<NSTEPS> &A, const int8_t *qs, const float *d, const int *q8s32, int nb, int lane) {
    const int g = lane >> 3, t = lane & 7, u = lane & 3;
#pragma unroll
    for (int st = 0; st < NSTEPS; ++st) {
        const int blk = st * 8 + g;
        A.valid[st] = blk < nb;
        const int b = A.valid[st] ? blk : 0;
        const int *x = reinterpret_cast<const int *>(qs + b * 256) + t;
#pragma unroll
        for (int j = 0; j < 4; ++j) { A.xa[st][j] = x[16 * j]; A.xb[st][j] = x[16 * j + 8]; }
        A.xd[st] = d[b];
        A.mq0[st] = q8s32[b * 8 + 2 * u];
        A.mq1[st] = q8s32[b * 8 + 2 * u + 1];
    }
}
template <int NSTEPS>
__device__ __forceinline__ void q4kc_emit(const uint4 hdr, const uint4 q, const Q4KActC<NSTEPS> &A, int st, int lane, float2 *row_tab) {
    const int g = lane >> 3, t = lane & 7, u = lane & 3;
    const float d = h2f((uint16_t)(hdr.x & 0xffff)), dmin = h2f((uint16_t)(hdr.x >> 16));
    uint32_t sc8[2], mn8[2];
    unpack_q4k_scales(hdr.y, hdr.z, hdr.w, sc8, mn8);
    const uint32_t qw[4] = {q.x, q.y, q.z, q.w};
    int nib[8], xv[8], dd[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        nib[2 * j] = (int)(qw[j] & 0x0f0f0f0fu); nib[2 * j + 1] = (int)((qw[j] >> 4) & 0x0f0f0f0fu);
        xv[2 * j] = A.xa[st][j]; xv[2 * j + 1] = A.xb[st][j];
    }
    dot4z_x8(dd, nib, xv);
    int s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s = __mul24(byte_of(sc8, k), dd[k]) + s;
    const int prod = __mul24(byte_of(mn8, 2 * u), A.mq0[st]) + __mul24(byte_of(mn8, 2 * u + 1), A.mq1[st]);
    const float dy = A.xd[st] * d;              // y.d * fp16(x.d)        (VecDotQ4.cpp:228)
    const float dm = (-A.xd[st]) * dmin;      // -y.d * fp16(x.dmin)    (:229)
    if (A.valid[st]) {
        float2 *e = row_tab + (st * 8 + g) * Q4K_SLOTS;
        e[bitrev3(t)] = make_float2(dy, (float)s);
        if (t < 4) e[8 + bitrev2(u)] = make_float2(dm, (float)prod);
    }
}
template <int NSTEPS, int ROWS>
__device__ __forceinline__ void q4kc_dot_rows(const uint4 (&hdr)[ROWS][NSTEPS], const uint4 (&q)[ROWS][NSTEPS], const Q4KActC<NSTEPS> &A, int nb, int lane,
                                              float2 *tab, float (&out)[ROWS]) {
    constexpr int NBP = NSTEPS * 8;
#pragma unroll
    for (int p = 0; p < ROWS; p += 4) {
        if (p) wave_lds_fence();
#pragma unroll
        for (int rr = p; rr < (p + 4 < ROWS ? p + 4 : ROWS); ++rr)
#pragma unroll
            for (int st = 0; st < NSTEPS; ++st) q4kc_emit<NSTEPS>(hdr[rr][st], q[rr][st], A, st, lane, tab + (size_t)(rr - p) * NBP * Q4K_SLOTS);
        wave_lds_fence();
        const float res = q4k_chain(tab, NBP, nb, ROWS - p < 4 ? ROWS - p : 4, lane);
#pragma unroll
        for (int rr = p; rr < (p + 4 < ROWS ? p + 4 : ROWS); ++rr)
            out[rr] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(res), 16 * (rr - p) + 8));
    }
}
constexpr size_t q4k_tab_bytes(int nsteps, int rows) { return (size_t)(rows < 4 ? rows : 4) * nsteps * 8 * Q4K_SLOTS * sizeof(float2); }

}  // namespace mllm_hip
