// mllm_amd/csrc/kernels_linear.hip -- A1/A2/A5/A6: Linear / mat_mul for gfx950.
//
//   decode (M == 1): weight-streaming GEMV, HBM-bound.  One 64-lane wave owns an output row at a time; the 144-B
//     Q4_K super-blocks of the row are spread 8 lanes per block (one 16-B dwordx4 of nibbles per lane, so a wave
//     load covers 8 consecutive blocks = 1152 contiguous bytes), the Q8_K activation slice of each lane is loop
//     invariant and lives in registers, the 4x8-bit dot products run on v_dot4_i32_i8, integers are reduced exactly
//     with wavefront shuffles and only then scaled (int-exact per super-block like vec_dot_q4_K_q8_K).
//   prefill / vision (M >= 16): int8 MFMA GEMM (v_mfma_i32_32x32x32_i8): one MFMA per 32-wide sub-block so the 6-bit
//     sub-block scales can be applied to exact int32 partial sums; the `mins` term is a second MFMA whose B operand
//     is the broadcast 6-bit min (sum_e q8[e]*min_{sb(e)} == sum_j bsums[j]*min_{j/2}, VecDotQ4.cpp:318).
//   fp32 weights (patch-embed conv, fp32 models): f32-input MFMA (v_mfma_f32_32x32x2_f32, exact fp32 fma chain).
#include "common.h"

namespace mllm_hip {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------------------------
// Q4_K x Q8_K GEMV (M == 1)
// ------------------------------------------------------------------------------------------------------------------
struct Q4KPart { int i1, i2; };

__device__ __forceinline__ void q4k_lane_dot(const uint4 hdr, const uint4 q, const int4 xa, const int4 xb, const int q8s, const int r,
                                             int &i1, int &i2, float &d, float &dmin) {
    d = h2f((uint16_t)(hdr.x & 0xffff));
    dmin = h2f((uint16_t)(hdr.x >> 16));
    uint32_t sc8[2], mn8[2];
    unpack_q4k_scales(hdr.y, hdr.z, hdr.w, sc8, mn8);
    const int j = r >> 1;
    const int sc_lo = byte_of(sc8, 2 * j), sc_hi = byte_of(sc8, 2 * j + 1), mr = byte_of(mn8, r);
    int dl = dot4((int)(q.x & 0x0f0f0f0fu), xa.x, 0);
    dl = dot4((int)(q.y & 0x0f0f0f0fu), xa.y, dl);
    dl = dot4((int)(q.z & 0x0f0f0f0fu), xa.z, dl);
    dl = dot4((int)(q.w & 0x0f0f0f0fu), xa.w, dl);
    int dh = dot4((int)((q.x >> 4) & 0x0f0f0f0fu), xb.x, 0);
    dh = dot4((int)((q.y >> 4) & 0x0f0f0f0fu), xb.y, dh);
    dh = dot4((int)((q.z >> 4) & 0x0f0f0f0fu), xb.z, dh);
    dh = dot4((int)((q.w >> 4) & 0x0f0f0f0fu), xb.w, dh);
    i1 = sc_lo * dl + sc_hi * dh;
    i2 = mr * q8s;
}

// NSTEPS = ceil(K/2048): wave steps per row. ROWS = rows in flight per wave (memory-level parallelism).
template <int NSTEPS, int ROWS>
__global__ __launch_bounds__(256) void gemv_q4k_kernel(const uint8_t *__restrict__ W, const float *__restrict__ bias, const int8_t *__restrict__ xqs,
                                                       const float *__restrict__ xd, const int16_t *__restrict__ xbsums, void *__restrict__ y, int y_f16,
                                                       const float *__restrict__ residual, int N, int nb, int rows_per_wave) {
    const int lane = threadIdx.x & 63, g = lane >> 3, r = lane & 7;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int j = r >> 1, tp = r & 1;
    int4 xa[NSTEPS], xb[NSTEPS];
    float xdv[NSTEPS];
    int q8s[NSTEPS];
    bool valid[NSTEPS];
#pragma unroll
    for (int st = 0; st < NSTEPS; ++st) {
        const int blk = st * 8 + g;
        valid[st] = blk < nb;
        const int b = valid[st] ? blk : 0;
        xa[st] = *reinterpret_cast<const int4 *>(xqs + b * 256 + 64 * j + 16 * tp);
        xb[st] = *reinterpret_cast<const int4 *>(xqs + b * 256 + 64 * j + 32 + 16 * tp);
        xdv[st] = valid[st] ? xd[b] : 0.0f;
        q8s[st] = (int)xbsums[b * 16 + 2 * r] + (int)xbsums[b * 16 + 2 * r + 1];
    }
    const int row0 = wave * rows_per_wave;
    const int row1 = min(N, row0 + rows_per_wave);
    for (int row = row0; row < row1; row += ROWS) {
        uint4 hdr[ROWS][NSTEPS], q[ROWS][NSTEPS];
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            const int rw = min(row + rr, row1 - 1);
#pragma unroll
            for (int st = 0; st < NSTEPS; ++st) {
                const int blk = valid[st] ? st * 8 + g : 0;
                const uint8_t *wb = W + ((int64_t)rw * nb + blk) * 144;
                hdr[rr][st] = *reinterpret_cast<const uint4 *>(wb);
                q[rr][st] = *reinterpret_cast<const uint4 *>(wb + 16 + 16 * r);
            }
        }
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            float acc = 0.0f;
#pragma unroll
            for (int st = 0; st < NSTEPS; ++st) {
                int i1, i2;
                float d, dmin;
                q4k_lane_dot(hdr[rr][st], q[rr][st], xa[st], xb[st], q8s[st], r, i1, i2, d, dmin);
                i1 = group8_sum(i1);
                i2 = group8_sum(i2);
                // d = y.d * fp16(x.d) ; dmin = y.d * fp16(x.dmin)  (VecDotQ4.cpp:228-229)
                const float p = __fmaf_rn(__fmul_rn(xdv[st], d), (float)i1, -__fmul_rn(__fmul_rn(xdv[st], dmin), (float)i2));
                acc += valid[st] ? p : 0.0f;
            }
            acc = groups_total_lane63(acc);
            const int rw = row + rr;
            if (lane == 63 && rw < row1) {
                float v = acc;
                if (bias) v = __fadd_rn(v, bias[rw]);
                if (y_f16) reinterpret_cast<uint16_t *>(y)[rw] = f2h(v);
                else {
                    if (residual) v = __fadd_rn(v, residual[rw]);
                    reinterpret_cast<float *>(y)[rw] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Q4_0 (nibble/scale planes) x Q8_0 GEMV (M == 1): the tied lm_head (modeling_qwen2_vl.hpp:399 -> CPUmmFunction ->
// vec_dot_q4_0_q8_0, VecDotQ4.cpp:514-545). LPR lanes per row, BPL blocks per lane (K = 32*LPR*BPL), 64/LPR rows per wave pass.
// ------------------------------------------------------------------------------------------------------------------
template <int BPL, int LPR>
__global__ __launch_bounds__(256) void gemv_q40_kernel(const uint8_t *__restrict__ Wqs, const uint16_t *__restrict__ Wd, const float *__restrict__ bias,
                                                       const int8_t *__restrict__ xqs, const uint16_t *__restrict__ xd, float *__restrict__ y, int N,
                                                       int rows_per_wave) {
    constexpr int RPW = 64 / LPR;  // rows per wave pass
    const int lane = threadIdx.x & 63, sub = lane % LPR, rsel = lane / LPR;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nblk = BPL * LPR;
    int4 xa[BPL], xb[BPL];
    float xdv[BPL];
    int xs8[BPL];
#pragma unroll
    for (int b = 0; b < BPL; ++b) {
        const int blk = sub + LPR * b;
        xa[b] = *reinterpret_cast<const int4 *>(xqs + blk * 32);
        xb[b] = *reinterpret_cast<const int4 *>(xqs + blk * 32 + 16);
        xdv[b] = h2f(xd[blk]);
        const int one = 0x01010101;
        int s = dot4(xa[b].x, one, 0); s = dot4(xa[b].y, one, s); s = dot4(xa[b].z, one, s); s = dot4(xa[b].w, one, s);
        s = dot4(xb[b].x, one, s); s = dot4(xb[b].y, one, s); s = dot4(xb[b].z, one, s); s = dot4(xb[b].w, one, s);
        xs8[b] = 8 * s;
    }
    const int row0 = wave * rows_per_wave, row1 = min(N, row0 + rows_per_wave);
    for (int base = row0; base < row1; base += 2 * RPW) {
        uint4 q[2][BPL];
        uint16_t dw[2][BPL];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int rw = min(base + RPW * u + rsel, row1 - 1);
#pragma unroll
            for (int b = 0; b < BPL; ++b) {
                const int64_t bi = (int64_t)rw * nblk + sub + LPR * b;
                q[u][b] = *reinterpret_cast<const uint4 *>(Wqs + bi * 16);
                dw[u][b] = Wd[bi];
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float acc = 0.0f;
#pragma unroll
            for (int b = 0; b < BPL; ++b) {
                int i = dot4((int)(q[u][b].x & 0x0f0f0f0fu), xa[b].x, 0);
                i = dot4((int)(q[u][b].y & 0x0f0f0f0fu), xa[b].y, i);
                i = dot4((int)(q[u][b].z & 0x0f0f0f0fu), xa[b].z, i);
                i = dot4((int)(q[u][b].w & 0x0f0f0f0fu), xa[b].w, i);
                i = dot4((int)((q[u][b].x >> 4) & 0x0f0f0f0fu), xb[b].x, i);
                i = dot4((int)((q[u][b].y >> 4) & 0x0f0f0f0fu), xb[b].y, i);
                i = dot4((int)((q[u][b].z >> 4) & 0x0f0f0f0fu), xb[b].z, i);
                i = dot4((int)((q[u][b].w >> 4) & 0x0f0f0f0fu), xb[b].w, i);
                i -= xs8[b];  // (nib - 8) * q8 summed
                acc = __fmaf_rn(__fmul_rn(h2f(dw[u][b]), xdv[b]), (float)i, acc);
            }
            acc = LPR == 16 ? group16_sum(acc) : group8_sum(acc);
            const int rw = base + RPW * u + rsel;
            if (sub == 0 && rw < row1) y[rw] = bias ? __fadd_rn(acc, bias[rw]) : acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Q4_K x Q8_K GEMM (M >= 16) on int8 MFMA. Workgroup = 4 waves (2 along M x 2 along N); wave tile = 32 (M) x 64 (N).
// v_mfma_i32_32x32x32_i8 operand maps: lane l (row/col = l&31, h = l>>5) holds k = 16h .. 16h+15 of its A row / B column
// (16 int8 = 4 VGPRs); C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*h.
// A row m = activation row (q8 plane), B column n = weight row: one MFMA covers one 32-wide Q4_K sub-block.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_q4k_kernel(const uint8_t *__restrict__ W, const float *__restrict__ bias, const int8_t *__restrict__ xqs,
                                                       const float *__restrict__ xd, void *__restrict__ y, int y_f16, int64_t ldy,
                                                       const float *__restrict__ residual, int M, int N, int K) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int nb = K >> 8;
    const int m0 = blockIdx.y * 64 + (wid >> 1) * 32;
    const int n0 = blockIdx.x * 128 + (wid & 1) * 64;
    if (m0 >= M) return;
    const int am = min(m0 + col, M - 1);                       // A row of this lane (clamped: rows >= M are never stored)
    const int8_t *arow = xqs + (int64_t)am * K + 16 * h;
    const uint8_t *wrow[2];
    bool nvalid[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = n0 + nt * 32 + col;
        nvalid[nt] = n < N;
        wrow[nt] = W + (int64_t)min(n, N - 1) * nb * 144;
    }
    v16f facc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) facc[nt][i] = 0.0f;

    for (int blk = 0; blk < nb; ++blk) {
        v4i a[8];
#pragma unroll
        for (int sb = 0; sb < 8; ++sb) a[sb] = *reinterpret_cast<const v4i *>(arow + blk * 256 + sb * 32);
        float dxr[16];  // y.d of the 16 rows this lane's accumulators belong to
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rm = min(m0 + (i & 3) + 8 * (i >> 2) + 4 * h, M - 1);
            dxr[i] = xd[(int64_t)rm * nb + blk];
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const uint8_t *wb = wrow[nt] + (int64_t)blk * 144;
            const uint4 hdr = *reinterpret_cast<const uint4 *>(wb);
            const float dw = h2f((uint16_t)(hdr.x & 0xffff)), dmw = h2f((uint16_t)(hdr.x >> 16));
            uint32_t sc8[2], mn8[2];
            unpack_q4k_scales(hdr.y, hdr.z, hdr.w, sc8, mn8);
            v16i acc1, acc2;
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc1[i] = 0; acc2[i] = 0; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4 q = *reinterpret_cast<const uint4 *>(wb + 16 + 32 * j + 16 * h);
                v4i blo, bhi;
                blo[0] = (int)(q.x & 0x0f0f0f0fu); blo[1] = (int)(q.y & 0x0f0f0f0fu); blo[2] = (int)(q.z & 0x0f0f0f0fu); blo[3] = (int)(q.w & 0x0f0f0f0fu);
                bhi[0] = (int)((q.x >> 4) & 0x0f0f0f0fu); bhi[1] = (int)((q.y >> 4) & 0x0f0f0f0fu);
                bhi[2] = (int)((q.z >> 4) & 0x0f0f0f0fu); bhi[3] = (int)((q.w >> 4) & 0x0f0f0f0fu);
                v16i zero;
#pragma unroll
                for (int i = 0; i < 16; ++i) zero[i] = 0;
                const int s0 = byte_of(sc8, 2 * j), s1 = byte_of(sc8, 2 * j + 1);
                const int mb0 = byte_of(mn8, 2 * j) * 0x01010101, mb1 = byte_of(mn8, 2 * j + 1) * 0x01010101;
                v4i bm0 = {mb0, mb0, mb0, mb0}, bm1 = {mb1, mb1, mb1, mb1};
                const v16i p0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[2 * j], blo, zero, 0, 0, 0);
                const v16i p1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[2 * j + 1], bhi, zero, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[2 * j], bm0, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[2 * j + 1], bm1, acc2, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 16; ++i) acc1[i] += s0 * p0[i] + s1 * p1[i];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __fmaf_rn(__fmul_rn(dxr[i], dw), (float)acc1[i], -__fmul_rn(__fmul_rn(dxr[i], dmw), (float)acc2[i]));
                facc[nt][i] += p;
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = n0 + nt * 32 + col;
        if (!nvalid[nt]) continue;
        const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (m >= M) continue;
            float v = facc[nt][i];
            if (bias) v = __fadd_rn(v, bv);
            if (y_f16) reinterpret_cast<uint16_t *>(y)[(int64_t)m * ldy + n] = f2h(v);
            else {
                if (residual) v = __fadd_rn(v, residual[(int64_t)m * ldy + n]);
                reinterpret_cast<float *>(y)[(int64_t)m * ldy + n] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// fp32 GEMM y = x W^T (+bias) on v_mfma_f32_32x32x2_f32: A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]; exact fp32 fma
// chain in k order. One wave per 32x32 tile, 4 waves (2x2) per workgroup. K % 8 == 0.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float *__restrict__ W, const float *__restrict__ bias, const float *__restrict__ x,
                                                       float *__restrict__ y, int64_t ldy, int M, int N, int K) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 64 + (wid >> 1) * 32, n0 = blockIdx.x * 64 + (wid & 1) * 32;
    if (m0 >= M || n0 >= N) return;
    const float *ar = x + (int64_t)min(m0 + col, M - 1) * K;
    const float *br = W + (int64_t)min(n0 + col, N - 1) * K;
    v16f acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    for (int k = 0; k < K; k += 8) {
        const float4 a0 = *reinterpret_cast<const float4 *>(ar + k), a1 = *reinterpret_cast<const float4 *>(ar + k + 4);
        const float4 b0 = *reinterpret_cast<const float4 *>(br + k), b1 = *reinterpret_cast<const float4 *>(br + k + 4);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a0.y : a0.x, h ? b0.y : b0.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a0.w : a0.z, h ? b0.w : b0.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a1.y : a1.x, h ? b1.y : b1.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a1.w : a1.z, h ? b1.w : b1.z, acc, 0, 0, 0);
    }
    const int n = n0 + col;
    if (n >= N) return;
    const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int m = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (m < M) y[(int64_t)m * ldy + n] = bias ? __fadd_rn(acc[i], bv) : acc[i];
    }
}

static int launch_gemv_q4k(const void *W, const float *bias, const int8_t *xqs, const float *xd, const int16_t *xbsums, void *y, int y_f16,
                           const float *residual, int N, int K, hipStream_t st) {
    const int nb = K / 256;
    const int nsteps = (nb + 7) / 8;
    // Little's law: ~6.3 TB/s x ~2 us of HBM latency = ~50 KB in flight per CU.  Every wave issues all the loads of its
    // rows up front (ROWS x NSTEPS KiB), so give each wave one batch of rows and put as many waves as the register budget
    // admits on the chip (<= 28 per CU at <= 72 VGPRs) before letting a wave loop over a second batch.
    const int target_waves = 256 * 24;
    int rows_per_wave = (N + target_waves - 1) / target_waves;
#define GEMV_CASE(NS, RW)                                                                                                       \
    case NS: {                                                                                                                  \
        rows_per_wave = ((rows_per_wave + RW - 1) / RW) * RW;                                                                   \
        const int waves = (N + rows_per_wave - 1) / rows_per_wave;                                                              \
        hipLaunchKernelGGL((gemv_q4k_kernel<NS, RW>), dim3((waves + 3) / 4), dim3(256), 0, st, (const uint8_t *)W, bias, xqs, xd, \
                           xbsums, y, y_f16, residual, N, nb, rows_per_wave);                                                   \
    } break;
    switch (nsteps) {
        GEMV_CASE(1, 4)
        GEMV_CASE(2, 2)
        GEMV_CASE(3, 2)
        GEMV_CASE(4, 1)
        GEMV_CASE(5, 1)
        GEMV_CASE(6, 1)
    default: return MLLM_HIP_ERR_SHAPE;
    }
#undef GEMV_CASE
    return MH_LAUNCH_OK("gemv_q4k");
}
}  // namespace mllm_hip

using namespace mllm_hip;

extern "C" int mllm_hip_linear_q4k_q8k(const void *W, const float *bias, const int8_t *xqs, const float *xd, const int16_t *xbsums, void *y,
                                       int y_dtype, int64_t ldy, const float *residual, int M, int N, int K, void *stream) {
    if (K % 256 != 0 || K <= 0 || N <= 0) return MLLM_HIP_ERR_SHAPE;
    if (y_dtype != MLLM_HIP_F32 && y_dtype != MLLM_HIP_F16) return MLLM_HIP_ERR_DTYPE;
    if (M <= 0) return MLLM_HIP_OK;
    const int y_f16 = y_dtype == MLLM_HIP_F16;
    hipStream_t st = as_stream(stream);
    if (M < 16) {
        const int nb = K / 256;
        for (int m = 0; m < M; ++m) {
            void *ym = y_f16 ? (void *)((uint16_t *)y + (int64_t)m * ldy) : (void *)((float *)y + (int64_t)m * ldy);
            int rc = launch_gemv_q4k(W, bias, xqs + (int64_t)m * K, xd + (int64_t)m * nb, xbsums + (int64_t)m * (K / 16), ym, y_f16,
                                     residual ? residual + (int64_t)m * ldy : nullptr, N, K, st);
            if (rc) return rc;
        }
        return MLLM_HIP_OK;
    }
    dim3 grid((N + 127) / 128, (M + 63) / 64);
    hipLaunchKernelGGL(gemm_q4k_kernel, grid, dim3(256), 0, st, (const uint8_t *)W, bias, xqs, xd, y, y_f16, ldy, residual, M, N, K);
    return MH_LAUNCH_OK("gemm_q4k");
}

extern "C" int mllm_hip_linear_q40_q80(const uint8_t *Wqs, const uint16_t *Wd, const float *bias, const int8_t *xqs, const uint16_t *xd,
                                       float *y, int64_t ldy, int M, int N, int K, void *stream) {
    if (K % 256 != 0 || K <= 0 || N <= 0) return MLLM_HIP_ERR_SHAPE;
    // 16 lanes per row when K is a multiple of 512 (<= 8 blocks per lane), else 8 lanes per row (K multiple of 256)
    const int lpr = (K % 512 == 0 && K / 512 <= 8) ? 16 : 8;
    const int bpl = K / 32 / lpr;
    if (bpl > 8) return MLLM_HIP_ERR_SHAPE;
    hipStream_t st = as_stream(stream);
    const int target_waves = 256 * 8, rpp = 2 * (64 / lpr);
    int rows_per_wave = (N + target_waves - 1) / target_waves;
    rows_per_wave = ((rows_per_wave + rpp - 1) / rpp) * rpp;
    const int waves = (N + rows_per_wave - 1) / rows_per_wave;
    for (int m = 0; m < M; ++m) {
        const int8_t *xq = xqs + (int64_t)m * K;
        const uint16_t *xdd = xd + (int64_t)m * (K / 32);
        float *ym = y + (int64_t)m * ldy;
#define Q40_CASE(B, L) if (bpl == B && lpr == L) hipLaunchKernelGGL((gemv_q40_kernel<B, L>), dim3((waves + 3) / 4), dim3(256), 0, st, Wqs, Wd, bias, xq, xdd, ym, N, rows_per_wave);
        Q40_CASE(1, 16) Q40_CASE(2, 16) Q40_CASE(3, 16) Q40_CASE(4, 16) Q40_CASE(5, 16) Q40_CASE(6, 16) Q40_CASE(7, 16) Q40_CASE(8, 16)
        Q40_CASE(1, 8) Q40_CASE(3, 8) Q40_CASE(5, 8) Q40_CASE(7, 8)
#undef Q40_CASE
        int rc = MH_LAUNCH_OK("gemv_q40");
        if (rc) return rc;
    }
    return MLLM_HIP_OK;
}

extern "C" int mllm_hip_linear_f32(const float *W, const float *bias, const float *x, float *y, int64_t ldy, int M, int N, int K, void *stream) {
    if (K % 8 != 0 || N <= 0) return MLLM_HIP_ERR_SHAPE;
    if (M <= 0) return MLLM_HIP_OK;
    dim3 grid((N + 63) / 64, (M + 63) / 64);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, as_stream(stream), W, bias, x, y, ldy, M, N, K);
    return MH_LAUNCH_OK("gemm_f32");
}
extern "C" int mllm_hip_patch_gemm_f32(const float *patches, const float *W, const float *bias, float *out, int N, int KK, int OC, void *stream) {
    return mllm_hip_linear_f32(W, bias, patches, out, OC, N, OC, KK, stream);
}

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
extern "C" size_t mllm_hip_linear_workspace_bytes(int wdtype, int M, int K) {
    if (wdtype == MLLM_HIP_Q4_K) return align256((size_t)M * K) + align256((size_t)M * (K / 256) * 4) + align256((size_t)M * (K / 16) * 2);
    if (wdtype == MLLM_HIP_Q4_0) return align256((size_t)M * K) + align256((size_t)M * (K / 32) * 2);
    return 0;
}
// CPULinear::execute (op/CPULinear.cpp:98-234): quantise x to the weight's vec_dot_type, dot, bias.
// For Q4_0 weights `W` must be the nibble plane followed (at a 256-B aligned offset N*K/2) by the fp16 scale plane.
extern "C" int mllm_hip_linear(const void *W, int wdtype, const float *bias, const float *x, void *y, int y_dtype, int64_t ldy, int M, int N,
                               int K, void *workspace, void *stream) {
    if (wdtype == MLLM_HIP_F32) {
        if (y_dtype != MLLM_HIP_F32) return MLLM_HIP_ERR_DTYPE;
        return mllm_hip_linear_f32((const float *)W, bias, x, (float *)y, ldy, M, N, K, stream);
    }
    if (!workspace) return MLLM_HIP_ERR_ARG;
    uint8_t *ws = (uint8_t *)workspace;
    if (wdtype == MLLM_HIP_Q4_K) {
        int8_t *qs = (int8_t *)ws;
        float *d = (float *)(ws + align256((size_t)M * K));
        int16_t *bs = (int16_t *)(ws + align256((size_t)M * K) + align256((size_t)M * (K / 256) * 4));
        int rc = mllm_hip_quantize_q8k(x, qs, d, bs, M, K, stream);
        if (rc) return rc;
        return mllm_hip_linear_q4k_q8k(W, bias, qs, d, bs, y, y_dtype, ldy, nullptr, M, N, K, stream);
    }
    if (wdtype == MLLM_HIP_Q4_0) {
        if (y_dtype != MLLM_HIP_F32) return MLLM_HIP_ERR_DTYPE;
        int8_t *qs = (int8_t *)ws;
        uint16_t *d = (uint16_t *)(ws + align256((size_t)M * K));
        int rc = mllm_hip_quantize_q80(x, qs, d, M, K, stream);
        if (rc) return rc;
        const uint8_t *Wqs = (const uint8_t *)W;
        const uint16_t *Wd = (const uint16_t *)(Wqs + align256((size_t)N * K / 2));
        return mllm_hip_linear_q40_q80(Wqs, Wd, bias, qs, d, (float *)y, ldy, M, N, K, stream);
    }
    return MLLM_HIP_ERR_DTYPE;
}
