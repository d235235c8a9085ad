// mllm_amd/csrc/kernels_elem.hip -- the HBM-bound families of the hot path for gfx950:
//   A4  activation quantisation (Q8_K / Q8_0 planes)         A9/A18 RMSNorm / LayerNorm (+ fused A4)
//   A14 SiLU, silu*up, A18 GELU/QuickGELU LUT, A20 add/mul    A8 embedding gather (Q4_0 planes)
//   A10/A11/A19 rotary apply (+ fp16 KV-slab store = A12 append), softmax, index_put, argmax
// All are row- or element-parallel: one 64-lane wave owns one 256-value quant block (4 consecutive values per lane,
// 16-B coalesced loads), reductions are wavefront shuffles, nothing is staged through LDS except cross-wave sums.
#include <algorithm>
#include <vector>
#include <cmath>
#include <cstring>

#include "common.h"
#include "decode_launch.h"

namespace mllm_hip {

// ------------------------------------------------------------------------------------------------------------------
// A4 core: one wave quantises one 256-value block held as 4 consecutive values per lane.
// quantize_row_q8_K_reference (ggml QuantizeQ8.cpp:216-251): max = x[first j with largest |x|]; iscale = -128/max;
// q = min(127, nearest_int(iscale*x)); bsums over 16; d = 1/iscale.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_quant_q8k(float4 v, int lane, int8_t *qs_blk, float *d_out, int16_t *bsums_blk) {
    unsigned abits;
    const float mx = q8k_first_max(v, abits);      // the first element (in element order) that attains amax -- the reference's strict `>` scan keeps the first
    const float iscale = abits ? __fdiv_rn(-128.0f, mx) : 0.0f, dd = abits ? __fdiv_rn(1.0f, iscale) : 0.0f;      // an all-zero block: every q = nearest_int(0 * x) = 0
    uint32_t b[4];
    q8k_round4(v, iscale, b);
    const uint32_t packed = q8k_bytes(b);
    reinterpret_cast<uint32_t *>(qs_blk)[lane] = packed;
    const int s = group4_sum(q8k_sum4(packed));
    if ((lane & 3) == 0) bsums_blk[lane >> 2] = (int16_t)s;
    if (lane == 0) *d_out = dd;
}

// Same quantisation, written straight into the activation-side operand layout of the Q4_K GEMM (kernels_linear.hip: Ax / Am / Ad of
// q8k_prepack_kernel), so that prefill needs no separate pack pass.  Lane l holds values 4l..4l+3 of block (m, i): chunk j = l >> 4,
// sub-block half = (l >> 3) & 1, column class t = l & 7 -> 4 fp16 at fragment (t, p = j >> 1, h = j & 1), element half*4.
// Rows m >= M of the last 32-row tile are written as zeros (live == false).
typedef _Float16 v4h_t __attribute__((ext_vector_type(4)));
#ifndef QP_WAVES
#define QP_WAVES 8      // rows per workgroup of the quantise + pack launches (a 32-row tile is 4 workgroups)
#endif
__device__ __forceinline__ void wave_quant_pack(float4 v, int lane, bool live, uint8_t *pack, size_t tb, int nb, int m, int i) {
    unsigned abits;
    const float mx = q8k_first_max(v, abits);
    const bool nz = live && abits != 0;
    const float iscale = nz ? __fdiv_rn(-128.0f, mx) : 0.0f, dd = nz ? __fdiv_rn(1.0f, iscale) : 0.0f;
    uint32_t b[4];
    q8k_round4(v, iscale, b);
    const f32x2_t unmagic = {-12582912.0f, -12582912.0f};
    const f32x2_t q01 = f32x2_t{__uint_as_float(b[0]), __uint_as_float(b[1])} + unmagic, q23 = f32x2_t{__uint_as_float(b[2]), __uint_as_float(b[3])} + unmagic;      // q as floats, exactly
    const size_t tile = (size_t)(m >> 5) * nb + i;
    const int mi = m & 31, j = lane >> 4, hf = (lane >> 3) & 1, t = lane & 7;
    v4h_t o;
    o[0] = (_Float16)q01.x; o[1] = (_Float16)q01.y; o[2] = (_Float16)q23.x; o[3] = (_Float16)q23.y;
    *reinterpret_cast<v4h_t *>(pack + ((((tile * 8 + t) * 2 + (j >> 1)) * 64 + (j & 1) * 32 + mi) * 16) + hf * 8) = o;
    // q8s[k] = sum of the 32 values of 8-lane group k; lane 16u gathers (q8s[2u], q8s[2u+1]) -> (even part, low bit) pairs of Am
    const int s8 = group8_sum(q8k_sum4(q8k_bytes(b)));
    const int s8n = MH_DPP(0, s8, 0x108 /* row_shl:8 */, 0xF);
    if ((lane & 15) == 0) {
        v4h_t mo;
        mo[0] = (_Float16)(float)(s8 & ~1); mo[1] = (_Float16)(float)(s8 & 1);
        mo[2] = (_Float16)(float)(s8n & ~1); mo[3] = (_Float16)(float)(s8n & 1);
        *reinterpret_cast<v4h_t *>(pack + tb * Q4KP_W_PER_BLK + ((tile * 4 + (lane >> 4)) * 32 + mi) * 8) = mo;
    }
    if (lane == 0) *reinterpret_cast<float *>(pack + tb * (Q4KP_W_PER_BLK + Q4KP_M_PER_BLK) + (tile * 32 + mi) * 4) = dd;
}
__global__ __launch_bounds__(64 * QP_WAVES) void quantize_q8k_pack_kernel(const float *__restrict__ x, uint8_t *__restrict__ pack, int M, int nb) {
    const int lane = threadIdx.x & 63;
    const int64_t blk = (int64_t)blockIdx.x * QP_WAVES + (threadIdx.x >> 6);
    const int Mp = (M + 31) & ~31;
    if (blk >= (int64_t)Mp * nb) return;
    // the waves of a workgroup take CONSECUTIVE ROWS of one 256-block: in the packed operand a row owns 16 bytes of every 512-byte fragment row, so the
    // workgroup's stores fill whole 64-byte lines together (with consecutive blocks of one row per workgroup every store touched a line of its own)
    const int mi = (int)(blk & 31), i = (int)((blk >> 5) % nb), m = (int)((blk >> 5) / nb) * 32 + mi;
    const bool live = m < M;
    float4 v = make_float4(0, 0, 0, 0);
    if (live) v = reinterpret_cast<const float4 *>(x + ((int64_t)m * nb + i) * 256)[lane];
    wave_quant_pack(v, lane, live, pack, q4kp_tile_blocks(M, nb), nb, m, i);
}
__global__ __launch_bounds__(256) void quantize_q8k_kernel(const float *__restrict__ x, int8_t *__restrict__ qs, float *__restrict__ d,
                                                           int16_t *__restrict__ bsums, int64_t n_blocks) {
    const int lane = threadIdx.x & 63;
    const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= n_blocks) return;
    const float4 v = reinterpret_cast<const float4 *>(x + blk * 256)[lane];
    wave_quant_q8k(v, lane, qs + blk * 256, d + blk, bsums + blk * 16);
}

// quantize_row_q8_0_reference (ggml QuantizeQ8.cpp:32-55): d = amax/127 (stored fp16), q = roundf(x * (1/d)).
// One lane per 32-block would serialise; use 8 lanes per block (4 values each), 8 blocks per wave.
__global__ __launch_bounds__(256) void quantize_q80_kernel(const float *__restrict__ x, int8_t *__restrict__ qs, uint16_t *__restrict__ d, int64_t n_blocks) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t blk = wave * 8 + (lane >> 3);
    const bool ok = blk < n_blocks;
    float4 v = make_float4(0, 0, 0, 0);
    if (ok) v = reinterpret_cast<const float4 *>(x + blk * 32)[lane & 7];
    float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    amax = group8_max(amax);
    const float dd = __fdiv_rn(amax, 127.0f);
    const float id = dd != 0.0f ? __fdiv_rn(1.0f, dd) : 0.0f;
    if (!ok) return;
    const int q0 = (int)roundf(__fmul_rn(v.x, id)), q1 = (int)roundf(__fmul_rn(v.y, id));
    const int q2 = (int)roundf(__fmul_rn(v.z, id)), q3 = (int)roundf(__fmul_rn(v.w, id));
    const uint32_t packed = (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
    reinterpret_cast<uint32_t *>(qs + blk * 32)[lane & 7] = packed;
    if ((lane & 7) == 0) d[blk] = f2h(dd);
}

// ------------------------------------------------------------------------------------------------------------------
// A18 LayerNorm statistics in the reference's order (op/CPULayerNorm.cpp:49-88): sum += x[d] and ssq = fma(c, c, ssq) are
// sequential fp32 chains over the row (GCC cannot reassociate them; it does contract the second into an fma), so a row is
// one lane's work.  A workgroup takes LN_ROWS rows: all 256 threads stream the rows through LDS in coalesced chunks (double
// buffered), lanes 0..LN_ROWS-1 of wave 0 walk their row with ds_read_b128.  stats[row] = (mean, sqrtf(ssq/dim + eps)).
// ------------------------------------------------------------------------------------------------------------------
constexpr int LN_ROWS = 16, LN_CH = 256, LN_PITCH = LN_CH + 4;
__global__ __launch_bounds__(256) void ln_stats_kernel(const float *__restrict__ x, float *__restrict__ stats, int M, int dim, float eps) {
    __shared__ __attribute__((aligned(16))) float buf[2][LN_ROWS * LN_PITCH];
    __shared__ float mean_s[LN_ROWS];
    const int tid = threadIdx.x, row0 = blockIdx.x * LN_ROWS;
    const int nch = (dim + LN_CH - 1) / LN_CH;
    float mean = 0.0f;
    for (int pass = 0; pass < 2; ++pass) {
        float acc = 0.0f;
        float4 stage[4];
        auto fetch = [&](int c) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + 256 * i, r = idx >> 6, c4 = idx & 63;
                const int col = c * LN_CH + 4 * c4, row = min(row0 + r, M - 1);
                const float *p = x + (int64_t)row * dim + col;
                float4 v = make_float4(0, 0, 0, 0);
                if (col + 3 < dim) v = *reinterpret_cast<const float4 *>(p);
                else { if (col < dim) v.x = p[0]; if (col + 1 < dim) v.y = p[1]; if (col + 2 < dim) v.z = p[2]; }
                stage[i] = v;
            }
        };
        auto park = [&](int b) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + 256 * i, r = idx >> 6, c4 = idx & 63;
                float4 v = stage[i];
                if (pass == 1) {          // the centring c = x - mean is order-free: all 256 threads do it while parking, the walker keeps only its fma chain
                    const float mr = mean_s[r];
                    v.x = v.x - mr; v.y = v.y - mr; v.z = v.z - mr; v.w = v.w - mr;
                }
                *reinterpret_cast<float4 *>(&buf[b][r * LN_PITCH + 4 * c4]) = v;
            }
        };
        fetch(0);
        park(0);
        __syncthreads();
        for (int c = 0; c < nch; ++c) {
            if (c + 1 < nch) fetch(c + 1);
            if (tid < LN_ROWS) {
                const float *rowp = &buf[c & 1][tid * LN_PITCH];
                const int n = min(LN_CH, dim - c * LN_CH);
                int k = 0;
                // 16 values per step, all four LDS reads issued before the dependent chain (the chain never waits on LDS latency)
                if (pass == 0) {
                    for (; k + 16 <= n; k += 16) {
                        const float4 v0 = *reinterpret_cast<const float4 *>(rowp + k), v1 = *reinterpret_cast<const float4 *>(rowp + k + 4);
                        const float4 v2 = *reinterpret_cast<const float4 *>(rowp + k + 8), v3 = *reinterpret_cast<const float4 *>(rowp + k + 12);
                        acc = acc + v0.x; acc = acc + v0.y; acc = acc + v0.z; acc = acc + v0.w;
                        acc = acc + v1.x; acc = acc + v1.y; acc = acc + v1.z; acc = acc + v1.w;
                        acc = acc + v2.x; acc = acc + v2.y; acc = acc + v2.z; acc = acc + v2.w;
                        acc = acc + v3.x; acc = acc + v3.y; acc = acc + v3.z; acc = acc + v3.w;
                    }
                    for (; k < n; ++k) acc = acc + rowp[k];
                } else {
                    for (; k + 16 <= n; k += 16) {
                        const float4 v0 = *reinterpret_cast<const float4 *>(rowp + k), v1 = *reinterpret_cast<const float4 *>(rowp + k + 4);
                        const float4 v2 = *reinterpret_cast<const float4 *>(rowp + k + 8), v3 = *reinterpret_cast<const float4 *>(rowp + k + 12);
                        const float cc[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};   // already centred
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc = __fmaf_rn(cc[e], cc[e], acc);
                    }
                    for (; k < n; ++k) { const float cc = rowp[k]; acc = __fmaf_rn(cc, cc, acc); }
                }
            }
            if (c + 1 < nch) park((c + 1) & 1);
            __syncthreads();
        }
        if (pass == 0) {
            if (tid < LN_ROWS) mean_s[tid] = acc / (float)dim;
            __syncthreads();
            mean = tid < LN_ROWS ? mean_s[tid] : 0.0f;
        } else if (tid < LN_ROWS && row0 + tid < M) {
            stats[2 * (row0 + tid)] = mean;
            stats[2 * (row0 + tid) + 1] = sqrtf(acc / (float)dim + eps);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// A9 RMSNorm (op/CPURMSNorm.cpp:31-136) / A18 LayerNorm (op/CPULayerNorm.cpp:49-88), one 256-thread workgroup per row.
// Optional fused A4: the normalised row is quantised to q8k planes in the same pass (wave w owns blocks w, w+4, ...).
// ------------------------------------------------------------------------------------------------------------------
template <bool LAYERNORM>
__global__ __launch_bounds__(256) void norm_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b,
                                                   float *__restrict__ y, int8_t *__restrict__ qs, float *__restrict__ qd,
                                                   int16_t *__restrict__ bsums, int dim, float eps, int add_unit_offset,
                                                   const float *__restrict__ stats, uint8_t *__restrict__ pack, int M) {
    __shared__ double red[8];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (pack && row >= M) {   // padding rows of the last 32-row tile of the packed operand: zeros
        for (int blk = wid; blk < (dim >> 8); blk += 4) wave_quant_pack(make_float4(0, 0, 0, 0), lane, false, pack, q4kp_tile_blocks(M, dim >> 8), dim >> 8, row, blk);
        return;
    }
    const float *xr = x + (int64_t)row * dim;
    float mean = 0.0f, inv = 0.0f;
    if (!LAYERNORM) {
        double ss = 0.0;
        for (int d = tid; d < dim; d += 256) { const float v = xr[d]; ss += (double)v * (double)v; }
        ss = wave_sum_d(ss);
        if (lane == 0) red[wid] = ss;
        __syncthreads();
        ss = red[0] + red[1] + red[2] + red[3];
        const float m = (float)(ss / (double)dim);
        inv = __fdiv_rn(1.0f, sqrtf(__fadd_rn(m, eps)));
    } else {
        mean = stats[2 * row];
        inv = stats[2 * row + 1];   // "rms" of the reference: the divisor
    }
    const bool quant = qs != nullptr;
    if (quant || pack || (dim & 255) == 0) {
        const int nblk = dim >> 8;
        for (int blk = wid; blk < nblk; blk += 4) {
            const int d0 = blk * 256 + lane * 4;
            const float4 v = *reinterpret_cast<const float4 *>(xr + d0);
            const float4 ww = *reinterpret_cast<const float4 *>(w + d0);
            float4 o;
            if (!LAYERNORM) {
                const float w0 = add_unit_offset ? 1.0f + ww.x : ww.x, w1 = add_unit_offset ? 1.0f + ww.y : ww.y;
                const float w2 = add_unit_offset ? 1.0f + ww.z : ww.z, w3 = add_unit_offset ? 1.0f + ww.w : ww.w;
                o.x = __fmul_rn(__fmul_rn(v.x, inv), w0);
                o.y = __fmul_rn(__fmul_rn(v.y, inv), w1);
                o.z = __fmul_rn(__fmul_rn(v.z, inv), w2);
                o.w = __fmul_rn(__fmul_rn(v.w, inv), w3);
            } else {
                float4 bb = make_float4(0, 0, 0, 0);
                if (b) bb = *reinterpret_cast<const float4 *>(b + d0);
                o.x = __fdiv_rn(__fmul_rn(ww.x, __fsub_rn(v.x, mean)), inv);
                o.y = __fdiv_rn(__fmul_rn(ww.y, __fsub_rn(v.y, mean)), inv);
                o.z = __fdiv_rn(__fmul_rn(ww.z, __fsub_rn(v.z, mean)), inv);
                o.w = __fdiv_rn(__fmul_rn(ww.w, __fsub_rn(v.w, mean)), inv);
                if (b) { o.x = __fadd_rn(o.x, bb.x); o.y = __fadd_rn(o.y, bb.y); o.z = __fadd_rn(o.z, bb.z); o.w = __fadd_rn(o.w, bb.w); }
            }
            if (y) *reinterpret_cast<float4 *>(y + (int64_t)row * dim + d0) = o;
            if (quant) {
                const int64_t gblk = (int64_t)row * nblk + blk;
                wave_quant_q8k(o, lane, qs + gblk * 256, qd + gblk, bsums + gblk * 16);
            }
            if (pack) wave_quant_pack(o, lane, true, pack, q4kp_tile_blocks(M, nblk), nblk, row, blk);
        }
    } else {
        for (int d = tid; d < dim; d += 256) {
            float o;
            if (!LAYERNORM) o = __fmul_rn(__fmul_rn(xr[d], inv), add_unit_offset ? 1.0f + w[d] : w[d]);
            else { o = __fdiv_rn(__fmul_rn(w[d], __fsub_rn(xr[d], mean)), inv); if (b) o = __fadd_rn(o, b[d]); }
            y[(int64_t)row * dim + d] = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// A18 LayerNorm as ONE launch for rows of NCH * 256 values (the towers' 1024 / 1280): a wave owns a row from fetch to packed GEMM operand -- no LDS, no barrier.
// The wave fetches its row into registers (NCH x 16 bytes per lane, all requests out at once; lane l holds values 4l..4l+3 of each 256-block -- the layout the Q8_K
// quantiser and the GEMM operand packer want) and, through its own strip of LDS, a second copy in the walk layout: lane l holds the 4 NCH CONSECUTIVE values behind
// 4 NCH l.  The reference's `sum += x[d]` is one chain over the row in index order.  Here the accumulator TRAVELS: a hop moves all lanes' accumulators one lane up
// (`v_mov_b32_dpp wave_ror:1`), then every lane adds its own values, so after hop h lane h holds exactly the reference's partial sum through its last value (what the other
// lanes compute meanwhile is overwritten by what arrives) and after 64 hops lane 63 holds the row's.  One dependent instruction per value plus one hop per 4 NCH values, no
// LDS read and no wait in the chain's way.  `ssq = fma(c, c, ssq)` the same way.  Then the wave normalises, quantises and packs its row from its registers.
// Measured on the tower's 1024 x 1280 rows: ln_stats_kernel + norm_kernel<true> 25.2 us; this kernel with lanes 0..3 of one wave walking four LDS-parked rows (reads five
// ahead, one per four links) 19.7 us; with a wave-uniform chain fed by `v_readlane_b32` 23.8 us; the travelling accumulator hopping every four values (the pack layout,
// no LDS at all) 17.6 us -- a wave-wide DPP hop costs about three plain links --; this form 15.7 us: 2,560 links at 9.6 cycles, against about 8 for a bare dependent fp32 add.
// ------------------------------------------------------------------------------------------------------------------
constexpr int LNF_ROWS = 4;      // waves per workgroup (nothing is shared between them)
constexpr int DPP_WAVE_ROR1 = 0x13C;
// one pass over the row in the walk layout (lane l holds values 4 NCH l .. 4 NCH l + 4 NCH - 1): 64 hops, each the hop itself plus 4 NCH dependent links
template <int NCH, bool SQ>
__device__ __forceinline__ float lnf_walk(const float4 (&u)[NCH]) {
    float acc = 0.0f;      // lane 63's zero is what lane 0 takes in on the first hop
#pragma unroll 4      // (one hop per trip: 17.0 us, four: 15.7, sixteen: 15.6)
    for (int h = 0; h < 64; ++h) {
        acc = MH_DPPF(0.0f, acc, DPP_WAVE_ROR1, 0xF);      // lane l + 1 takes lane l's accumulator, lane 0 lane 63's
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (SQ) { acc = __fmaf_rn(u[j].x, u[j].x, acc); acc = __fmaf_rn(u[j].y, u[j].y, acc); acc = __fmaf_rn(u[j].z, u[j].z, acc); acc = __fmaf_rn(u[j].w, u[j].w, acc); }
            else { acc = __fadd_rn(acc, u[j].x); acc = __fadd_rn(acc, u[j].y); acc = __fadd_rn(acc, u[j].z); acc = __fadd_rn(acc, u[j].w); }
        }
    }
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 63));
}
template <int NCH>
__global__ __launch_bounds__(64 * LNF_ROWS) void ln_fused_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b, float *__restrict__ y,
                                                                 int8_t *__restrict__ qs, float *__restrict__ qd, int16_t *__restrict__ bsums, uint8_t *__restrict__ pack,
                                                                 int M, float eps) {
    constexpr int dim = NCH * 256;
    extern __shared__ __attribute__((aligned(16))) char lnf_smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, row = blockIdx.x * LNF_ROWS + wid;
    const size_t tb = pack ? q4kp_tile_blocks(M, NCH) : 0;
    if (row >= M) {      // padding rows of the last 32-row tile of the packed operand: zeros
        if (pack)
            for (int blk = 0; blk < NCH; ++blk) wave_quant_pack(make_float4(0, 0, 0, 0), lane, false, pack, tb, NCH, row, blk);
        return;
    }
    const float *xr = x + (int64_t)row * dim;
    float4 v[NCH], u[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) v[c] = *reinterpret_cast<const float4 *>(xr + c * 256 + lane * 4);
    // the same row once more in the walk layout, through the wave's own LDS strip (only this wave touches it: no barrier, the wave's LDS operations complete in order)
    float *mine = reinterpret_cast<float *>(lnf_smem) + wid * dim;
#pragma unroll
    for (int c = 0; c < NCH; ++c) *reinterpret_cast<float4 *>(mine + c * 256 + lane * 4) = v[c];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own stores have completed; nothing may be moved across
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < NCH; ++j) u[j] = *reinterpret_cast<const float4 *>(mine + 4 * NCH * lane + 4 * j);
    const float mean = lnf_walk<NCH, false>(u) / (float)dim;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        v[c].x = __fsub_rn(v[c].x, mean); v[c].y = __fsub_rn(v[c].y, mean); v[c].z = __fsub_rn(v[c].z, mean); v[c].w = __fsub_rn(v[c].w, mean);
        u[c].x = __fsub_rn(u[c].x, mean); u[c].y = __fsub_rn(u[c].y, mean); u[c].z = __fsub_rn(u[c].z, mean); u[c].w = __fsub_rn(u[c].w, mean);
    }
    const float sd = sqrtf(lnf_walk<NCH, true>(u) / (float)dim + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int d0 = c * 256 + lane * 4;
        const float4 ww = *reinterpret_cast<const float4 *>(w + d0);
        float4 o;
        o.x = __fdiv_rn(__fmul_rn(ww.x, v[c].x), sd);
        o.y = __fdiv_rn(__fmul_rn(ww.y, v[c].y), sd);
        o.z = __fdiv_rn(__fmul_rn(ww.z, v[c].z), sd);
        o.w = __fdiv_rn(__fmul_rn(ww.w, v[c].w), sd);
        if (b) {
            const float4 bb = *reinterpret_cast<const float4 *>(b + d0);
            o.x = __fadd_rn(o.x, bb.x); o.y = __fadd_rn(o.y, bb.y); o.z = __fadd_rn(o.z, bb.z); o.w = __fadd_rn(o.w, bb.w);
        }
        if (y) *reinterpret_cast<float4 *>(y + (int64_t)row * dim + d0) = o;
        if (qs) {
            const int64_t gblk = (int64_t)row * NCH + c;
            wave_quant_q8k(o, lane, qs + gblk * 256, qd + gblk, bsums + gblk * 16);
        }
        if (pack) wave_quant_pack(o, lane, true, pack, tb, NCH, row, c);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// A14: x/(1+exp(-x)) with the reference's AVX2 polynomial expf, one fp32 lane of it
// (compute/ActivationFunction.hpp:96-134 mllm_v_expf, :137-146 mllm_v_silu): same constants, same fma placement.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float v_expf(float x) {
    const float r = 0x1.8p23f;
    const float z = __fmaf_rn(x, 0x1.715476p+0f, r);
    const float n = __fsub_rn(z, r);
    const float b = __fmaf_rn(-n, 0x1.7f7d1cp-20f, __fmaf_rn(-n, 0x1.62e4p-1f, x));
    const uint32_t e = __float_as_uint(z) << 23;
    const float k = __uint_as_float(e + __float_as_uint(1.0f));
    const bool c = fabsf(n) > 126.0f;
    const float u = __fmul_rn(b, b);
    const float j = __fmaf_rn(__fmaf_rn(__fmaf_rn(0x1.0e4020p-7f, b, 0x1.573e2ep-5f), u, __fmaf_rn(0x1.555e66p-3f, b, 0x1.fffdb6p-2f)), u,
                              __fmul_rn(0x1.ffffecp-1f, b));
    if (!c) return __fmaf_rn(j, k, k);
    const uint32_t g = (n <= 0.0f) ? 0x82000000u : 0u;
    const float s1 = __uint_as_float(g + 0x7f000000u);
    const float s2 = __uint_as_float(e - g);
    if (fabsf(n) > 192.0f) return __fmul_rn(s1, s1);
    return __fmul_rn(__fmaf_rn(s2, j, s2), s1);
}
__device__ __forceinline__ float silu_ref(float x) { return __fdiv_rn(x, __fadd_rn(1.0f, v_expf(__fsub_rn(0.0f, x)))); }

__global__ __launch_bounds__(256) void silu_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * 1024;
    for (; i + 3 < n; i += stride) {
        float4 v = *reinterpret_cast<const float4 *>(x + i);
        v.x = silu_ref(v.x); v.y = silu_ref(v.y); v.z = silu_ref(v.z); v.w = silu_ref(v.w);
        *reinterpret_cast<float4 *>(y + i) = v;
    }
    if (i < n) for (int64_t j = i; j < n && j < i + 4; ++j) y[j] = silu_ref(x[j]);
}
// CPUSiLU::execute calls mllm_vec_silu_f32 per (b, h, s) ROW (op/CPUSiLU.cpp:35-47): the dim % 8 trailing values of every row take the scalar mllm_silu_f32 = x / (1 + expf(-x)) with
// libm's expf (ActivationFunction.cpp:24-26), the rest the polynomial.  Rows of a multiple of 8 (every model on the hot path) have no such tail: silu_kernel above.
__global__ __launch_bounds__(256) void silu_rows_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t rows, int dim) {
    __shared__ uint64_t tab[32];
    expf_tab_store(tab, expf_tab_fetch());
    __syncthreads();
    const int full = dim & ~7;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * dim; i += (int64_t)gridDim.x * 256) {
        const float v = x[i];
        y[i] = (int)(i % dim) < full ? silu_ref(v) : __fdiv_rn(v, __fadd_rn(1.0f, glibc_expf(-v, tab)));
    }
}
// silu(gate)*up of QWen2MLP (modeling_qwen2_vl.hpp:205-208) on a fused [M][2I] gate|up buffer
__global__ __launch_bounds__(256) void silu_mul_kernel(const float *__restrict__ gu, float *__restrict__ y, int M, int I) {
    const int64_t total = (int64_t)M * I / 4;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t e = t * 4;
        const int m = (int)(e / I), c = (int)(e % I);
        const float4 g = *reinterpret_cast<const float4 *>(gu + (int64_t)m * 2 * I + c);
        const float4 u = *reinterpret_cast<const float4 *>(gu + (int64_t)m * 2 * I + I + c);
        float4 o;
        o.x = __fmul_rn(silu_ref(g.x), u.x); o.y = __fmul_rn(silu_ref(g.y), u.y);
        o.z = __fmul_rn(silu_ref(g.z), u.z); o.w = __fmul_rn(silu_ref(g.w), u.w);
        *reinterpret_cast<float4 *>(y + e) = o;
    }
}
// The same packing producer with the activation folded in (prefill only): the fp32 value that act_lut / silu_mul would have stored is
// computed in registers and quantised at once -- one launch and one fp32 round trip through HBM less per MLP.
//   ACT 1: v = lut[f16(x)] (GELU / QuickGELU LUT, A18);  ACT 2: v = silu(gate) * up on a fused [M][2 K] gate|up row (A14 + F_TTMUL)
template <int ACT>
__global__ __launch_bounds__(64 * QP_WAVES) void quantize_q8k_pack_act_kernel(const float *__restrict__ x, const uint16_t *__restrict__ lut, uint8_t *__restrict__ pack,
                                                                    int M, int nb) {
    const int lane = threadIdx.x & 63;
    const int64_t blk = (int64_t)blockIdx.x * QP_WAVES + (threadIdx.x >> 6);
    const int Mp = (M + 31) & ~31;
    if (blk >= (int64_t)Mp * nb) return;
    const int mi = (int)(blk & 31), i = (int)((blk >> 5) % nb), m = (int)((blk >> 5) / nb) * 32 + mi;      // consecutive rows of one block per workgroup (see above)
    const bool live = m < M;
    float4 v = make_float4(0, 0, 0, 0);
    if (live) {
        if (ACT == 1) {
            const float4 a = reinterpret_cast<const float4 *>(x + ((int64_t)m * nb + i) * 256)[lane];
            v = make_float4(h2f(lut[f2h(a.x)]), h2f(lut[f2h(a.y)]), h2f(lut[f2h(a.z)]), h2f(lut[f2h(a.w)]));
        } else {
            const int64_t K = (int64_t)nb * 256;
            const float4 g = reinterpret_cast<const float4 *>(x + (int64_t)m * 2 * K + i * 256)[lane];
            const float4 u = reinterpret_cast<const float4 *>(x + (int64_t)m * 2 * K + K + i * 256)[lane];
            v = make_float4(__fmul_rn(silu_ref(g.x), u.x), __fmul_rn(silu_ref(g.y), u.y), __fmul_rn(silu_ref(g.z), u.z), __fmul_rn(silu_ref(g.w), u.w));
        }
    }
    wave_quant_pack(v, lane, live, pack, q4kp_tile_blocks(M, nb), nb, m, i);
}

// A18: y = f16->f32(lut[f32->f16(x)])  (mllm_vec_gelu_f32 / mllm_vec_gelu_quick_f32, ggml Quantize.hpp:92-131)
__global__ __launch_bounds__(256) void act_lut_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t n, const uint16_t *__restrict__ lut) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = h2f(lut[f2h(x[i])]);
}
template <int OP>
__global__ __launch_bounds__(256) void binary_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ y, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * 1024;
    for (; i + 3 < n; i += stride) {
        const float4 p = *reinterpret_cast<const float4 *>(a + i), q = *reinterpret_cast<const float4 *>(b + i);
        float4 o;
        if (OP == 0) { o.x = __fadd_rn(p.x, q.x); o.y = __fadd_rn(p.y, q.y); o.z = __fadd_rn(p.z, q.z); o.w = __fadd_rn(p.w, q.w); }
        else { o.x = __fmul_rn(p.x, q.x); o.y = __fmul_rn(p.y, q.y); o.z = __fmul_rn(p.z, q.z); o.w = __fmul_rn(p.w, q.w); }
        *reinterpret_cast<float4 *>(y + i) = o;
    }
    if (i < n) for (int64_t j = i; j < n && j < i + 4; ++j) y[j] = OP == 0 ? __fadd_rn(a[j], b[j]) : __fmul_rn(a[j], b[j]);
}

// CPUSoftMax (op/CPUSoftMax.cpp:28-65 -> mllm_vec_soft_max_f32, ActivationFunction.cpp:29-80), one wave per row, every bit the reference's:
// the columns go through in chunks of 8 -- y = v_expf(x - max) per lane, the chunk's sum in the AVX2 hsum order ((v4+v0)+(v6+v2)) + ((v5+v1)+(v7+v3)) -- and the
// chunk sums are added to the row sum one after the other; the < 8 trailing columns use libm's expf (glibc_expf) and join the sum one by one; y *= 1 / sum.
__global__ __launch_bounds__(64) void softmax_kernel(const float *__restrict__ x, float *__restrict__ y, int n, const int *__restrict__ valid) {
    __shared__ uint64_t tab[32];
    expf_tab_store(tab, expf_tab_fetch());
    __syncthreads();
    const int row = blockIdx.x, lane = threadIdx.x;
    const int v = valid ? valid[row] : n;
    const float *xr = x + (int64_t)row * n;
    float *yr = y + (int64_t)row * n;
    float mx = -INFINITY;
    for (int i = lane; i < v; i += 64) mx = fmaxf(mx, xr[i]);
    mx = wave_max(mx);
    const int nfull = v & ~7;
    float sum = 0.0f;
    for (int base = 0; base < nfull; base += 64) {      // eight chunks per pass, chunk g in lanes 8g .. 8g+7
        const int i = base + lane;
        float e = 0.0f;
        if (i < nfull) { e = v_expf(__fsub_rn(xr[i], mx)); yr[i] = e; }
        float t = __fadd_rn(e, __shfl_xor(e, 4));       // lanes 0..3 of the chunk: v[l+4] + v[l]
        t = __fadd_rn(t, __shfl_xor(t, 2));             // lane 0: t0 + t2, lane 1: t1 + t3
        t = __fadd_rn(t, __shfl_xor(t, 1));             // lane 0: (t0 + t2) + (t1 + t3)
        const int chunks = min(8, (nfull - base) >> 3);
        for (int g = 0; g < chunks; ++g) sum = __fadd_rn(sum, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), g * 8)));
    }
    const int ntail = v - nfull;
    if (ntail > 0) {
        float e = 0.0f;
        if (lane < ntail) { e = glibc_expf(__fsub_rn(xr[nfull + lane], mx), tab); yr[nfull + lane] = e; }
        for (int g = 0; g < ntail; ++g) sum = __fadd_rn(sum, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), g)));
    }
    const float inv = __fdiv_rn(1.0f, sum);
    // a lane rescales the columns it wrote itself (the same lane -> column map as above): no cross-lane visibility is needed
    for (int i = lane; i < nfull; i += 64) yr[i] = __fmul_rn(yr[i], inv);
    if (lane < ntail) yr[nfull + lane] = __fmul_rn(yr[nfull + lane], inv);
    for (int i = v + lane; i < n; i += 64) yr[i] = 0.0f;
}

// ---- the same softmax for ONE long row (a vocabulary: top-p sampling), spread over the chip.  What is order-free runs everywhere -- the maximum, y = v_expf(x - max) and
// the sum of every 8-chunk in its hsum order --; the row sum is still the reference's: the chunk sums added one after the other, then the trailing columns.  One wave does
// that with a travelling accumulator (lane l holds 16 consecutive chunk sums, a wave_ror hop between lanes: ln_fused_kernel's walk): 19 k dependent adds instead of the
// 1.9 ms the single-wave kernel above spends on a 151,936-wide row.
__global__ __launch_bounds__(256) void softmax_row_max_kernel(const float *__restrict__ x, int n, float *__restrict__ part) {
    __shared__ float sv[4];
    float mx = -INFINITY;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) mx = fmaxf(mx, x[i]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
}
__global__ __launch_bounds__(256) void softmax_row_exp_kernel(const float *__restrict__ x, float *__restrict__ y, int n, const float *__restrict__ part, int nparts,
                                                              float *__restrict__ chunk_sums) {
    __shared__ uint64_t tab[32];
    expf_tab_store(tab, expf_tab_fetch());
    float mx = -INFINITY;
    for (int i = 0; i < nparts; ++i) mx = fmaxf(mx, part[i]);
    __syncthreads();
    const int nfull = n & ~7;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ((n + 255) & ~255); i += gridDim.x * 256) {
        float e = 0.0f;
        if (i < nfull) { e = v_expf(__fsub_rn(x[i], mx)); y[i] = e; }
        else if (i < n) y[i] = glibc_expf(__fsub_rn(x[i], mx), tab);      // the < 8 trailing columns: libm's expf
        float t = __fadd_rn(e, __shfl_xor(e, 4));
        t = __fadd_rn(t, __shfl_xor(t, 2));
        t = __fadd_rn(t, __shfl_xor(t, 1));
        if ((threadIdx.x & 7) == 0 && i < nfull) chunk_sums[i >> 3] = t;
    }
}
constexpr int DPP_WAVE_ROR1_SM = 0x13C;
__global__ __launch_bounds__(64) void softmax_row_sum_kernel(const float *__restrict__ chunk_sums, int nchunks, const float *__restrict__ y, int n, float *__restrict__ inv_out) {
    const int lane = threadIdx.x;
    float acc = 0.0f;
    for (int base = 0; base < nchunks; base += 1024) {
        float u[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { const int c = base + 16 * lane + j; u[j] = c < nchunks ? chunk_sums[c] : 0.0f; }      // + 0.0f behind the end leaves the sum as it is
#pragma unroll 4
        for (int h = 0; h < 64; ++h) {
            acc = MH_DPPF(0.0f, acc, DPP_WAVE_ROR1_SM, 0xF);
#pragma unroll
            for (int j = 0; j < 16; ++j) acc = __fadd_rn(acc, u[j]);
        }
    }
    float sum = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 63));
    for (int i = n & ~7; i < n; ++i) sum = __fadd_rn(sum, y[i]);
    if (lane == 0) *inv_out = __fdiv_rn(1.0f, sum);
}
__global__ __launch_bounds__(256) void softmax_row_scale_kernel(float *__restrict__ y, int n, const float *__restrict__ inv) {
    const float s = *inv;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) y[i] = __fmul_rn(y[i], s);
}

__global__ __launch_bounds__(256) void index_put_rows_kernel(float *__restrict__ dst, const float *__restrict__ value, const int *__restrict__ idx, int dim) {
    const int r = blockIdx.x;
    const float4 *src = reinterpret_cast<const float4 *>(value + (int64_t)r * dim);
    float4 *out = reinterpret_cast<float4 *>(dst + (int64_t)idx[r] * dim);
    for (int i = threadIdx.x; i < dim / 4; i += 256) out[i] = src[i];
}

// the same splice with the indices as the reference hands them over -- a Tensor of floats (CPUIndexPutFunc.hpp:85-92: `(int)replace_idx->dataAt<float>`); rows whose
// destination lies outside [0, n_dst_rows) are skipped
__global__ __launch_bounds__(256) void index_put_rows_fidx_kernel(float *__restrict__ dst, int n_dst_rows, const float *__restrict__ value, const float *__restrict__ idx, int dim) {
    const int r = blockIdx.x, d = (int)idx[r];
    if (d < 0 || d >= n_dst_rows) return;
    const float4 *src = reinterpret_cast<const float4 *>(value + (int64_t)r * dim);
    float4 *out = reinterpret_cast<float4 *>(dst + (int64_t)d * dim);
    for (int i = threadIdx.x; i < dim / 4; i += 256) out[i] = src[i];
}

// dst[r][c] = src[r][c] for a [rows][cols] window of two pitched buffers (cols, both pitches and both bases multiples of 4 floats): the column split of a fused
// projection (efficient_split, compute/Split.hpp as called by CPUSplitFunc.hpp:145-172)
__global__ __launch_bounds__(256) void copy_2d_f32_kernel(const float *__restrict__ src, int64_t lds, float *__restrict__ dst, int64_t ldd, int rows, int cols4) {
    const int64_t n = (int64_t)rows * cols4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols4, c = i - r * cols4;
        reinterpret_cast<float4 *>(dst + r * ldd)[c] = reinterpret_cast<const float4 *>(src + r * lds)[c];
    }
}

// out[c][r] = in[r][c]: 32 x 32 tiles through LDS (pitch 33: conflict-free both ways), coalesced on both sides
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float *__restrict__ in, float *__restrict__ out, int rows, int cols) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(int64_t)(r0 + i) * cols + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) out[(int64_t)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

// first-maximum argmax of one row (std::max_element semantics, processing_qwen2_vl.hpp:284-289), single workgroup
__global__ __launch_bounds__(1024) void argmax_kernel(const float *__restrict__ x, int n, int *__restrict__ out) {
    __shared__ float sv[16];
    __shared__ int si[16];
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 1024) { const float v = x[i]; if (v > best) { best = v; bi = i; } }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = __shfl_xor(best, m, 64);
        const int oi = __shfl_xor(bi, m, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        *out = bi;
    }
}

// first-maximum argmax of a row spread over the chip: workgroup b scans its slice and leaves (value, index); dec_next_kernel (decode) or argmax_final_kernel folds the
// partials with the same tie rule (equal values: the smaller index).  The single-workgroup mllm_hip_argmax took 47 us on the 151,936 logits -- 7 % of a Qwen1.5-0.5B token.
__global__ __launch_bounds__(256) void argmax_parts_kernel(const float *__restrict__ x, int n, float *__restrict__ part_val, int *__restrict__ part_idx) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int per = (((n + (int)gridDim.x - 1) / (int)gridDim.x) + 3) & ~3;
    const int lo = blockIdx.x * per, hi = min(n, lo + per);
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int i = lo + (int)threadIdx.x; i < hi; i += 256) { const float v = x[i]; if (v > best) { best = v; besti = i; } }      // a thread's indices ascend: strict > keeps the first
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
        part_val[blockIdx.x] = best;
        part_idx[blockIdx.x] = besti;
    }
}
__global__ __launch_bounds__(64) void argmax_final_kernel(const float *__restrict__ part_val, const int *__restrict__ part_idx, int nparts, int *__restrict__ out) {
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int i = threadIdx.x; i < nparts; i += 64) {
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
    if (threadIdx.x == 0) *out = besti;
}

// N2: the candidate set of _LlmTextGenerateTopkSamplingMethod::generate (mllm/Generate.cpp:56-67: std::partial_sort of the (logit, index)
// pairs by descending logit, first k kept), on device so that k values travel instead of the whole logits row.  k rounds of a first-
// maximum argmax over the entries not yet taken: descending values, equal values by ascending index (the reference's order among equal
// logits is whatever its heap leaves -- not specified).  Single workgroup; k <= 64.
__global__ __launch_bounds__(1024) void topk_kernel(const float *__restrict__ x, int n, int k, float *__restrict__ out_val, int *__restrict__ out_idx) {
    __shared__ float sv[16];
    __shared__ int si[16];
    __shared__ int taken[64];
    for (int r = 0; r < k; ++r) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = threadIdx.x; i < n; i += 1024) {
            const float v = x[i];
            if (v > best || (bi == 0x7fffffff && !(v != v))) {      // first candidate of this thread, or strictly larger (NaNs never win)
                bool free = true;
                for (int t = 0; t < r; ++t) free = free && taken[t] != i;
                if (free) { best = v; bi = i; }
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float ov = __shfl_xor(best, m, 64);
            const int oi = __shfl_xor(bi, m, 64);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; }
        }
        if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 16; ++w)
                if (si[w] != 0x7fffffff && (bi == 0x7fffffff || sv[w] > best || (sv[w] == best && si[w] < bi))) { best = sv[w]; bi = si[w]; }
            taken[r] = bi;
            out_val[r] = best;
            out_idx[r] = bi;
        }
        __syncthreads();
    }
}

// A8: dequantize_row_q4_0 (ggml QuantizeQ4.cpp:74-93) of row ids[s] from the nibble/scale planes: y = (nib-8)*d
__global__ __launch_bounds__(256) void embedding_q40_kernel(const float *__restrict__ ids, const uint8_t *__restrict__ Wqs, const uint16_t *__restrict__ Wd,
                                                            float *__restrict__ out, int hidden, int vocab) {
    const int s = blockIdx.x;
    int id = (int)ids[s];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const uint8_t *q = Wqs + (int64_t)id * (hidden / 2);
    const uint16_t *dd = Wd + (int64_t)id * (hidden / 32);
    float *o = out + (int64_t)s * hidden;
    for (int t = threadIdx.x; t < hidden / 2; t += 256) {
        const int blk = t >> 4, j = t & 15;
        const float d = h2f(dd[blk]);
        const uint8_t b = q[t];
        o[blk * 32 + j] = __fmul_rn((float)((int)(b & 0xF) - 8), d);
        o[blk * 32 + j + 16] = __fmul_rn((float)((int)(b >> 4) - 8), d);
    }
}

__global__ __launch_bounds__(256) void repack_q40_kernel(const uint8_t *__restrict__ raw, uint8_t *__restrict__ qs, uint16_t *__restrict__ d, int64_t n_blocks) {
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n_blocks * 16; t += (int64_t)gridDim.x * 256) {
        const int64_t blk = t >> 4;
        const int j = (int)(t & 15);
        const uint8_t *src = raw + blk * 18;
        qs[blk * 16 + j] = src[2 + j];
        if (j == 0) d[blk] = (uint16_t)src[0] | ((uint16_t)src[1] << 8);
    }
}

// A10/A11/A19 rope_hf rotate (CPUMultimodalRoPE.cpp:153-221). The reference is built with GCC -O2 -mfma, whose default
// contraction turns `a*c - b*s` into fma(a, c, -(b*s)) and `a*s + b*c` into fma(a, s, b*c).
template <bool OUT_F16>
__global__ __launch_bounds__(256) void rope_apply_kernel(const float *__restrict__ x, int64_t ldx, const float *__restrict__ sin_t, const float *__restrict__ cos_t,
                                                         int ld_tab, void *__restrict__ out, int64_t ldo, int S, int H, int D, int period) {
    // period > 0: row s takes table row s % period (the images of a vision pass share one table: one launch rotates them all)
    const int half = D >> 1;
    const int64_t total = (int64_t)S * H * half;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int d = (int)(t % half);
        const int h = (int)((t / half) % H);
        const int s = (int)(t / ((int64_t)half * H));
        const int ts = period > 0 ? s % period : s;
        const float a = x[(int64_t)s * ldx + h * D + d], b = x[(int64_t)s * ldx + h * D + d + half];
        const float sv = sin_t[(int64_t)ts * ld_tab + d], cv = cos_t[(int64_t)ts * ld_tab + d];
        const float v1 = __fmaf_rn(a, cv, -__fmul_rn(b, sv));
        const float v2 = __fmaf_rn(a, sv, __fmul_rn(b, cv));
        const int64_t o = (int64_t)s * ldo + h * D + d;
        if (OUT_F16) { reinterpret_cast<uint16_t *>(out)[o] = f2h(v1); reinterpret_cast<uint16_t *>(out)[o + half] = f2h(v2); }
        else { reinterpret_cast<float *>(out)[o] = v1; reinterpret_cast<float *>(out)[o + half] = v2; }
    }
}
// transposed variant: out[c * ldo + s] (the engine's V slab keeps one row per (kv head, dim) so a decode lane reads its dim contiguously in keys)
__global__ __launch_bounds__(256) void store_f16_t_kernel(const float *__restrict__ x, int64_t ldx, uint16_t *__restrict__ out, int64_t ldo, int S, int n) {
    const int64_t total = (int64_t)S * n;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int c = (int)(t / S), s_ = (int)(t % S);
        out[(int64_t)c * ldo + s_] = f2h(x[(int64_t)s_ * ldx + c]);
    }
}
// the three steps between the fused q|k|v projection and the prefill attention in one launch: q rotated in place (fp32), k rotated into its fp16 slab rows (KVCache append,
// CPUKVCache.cpp:253-275), v into the transposed fp16 slab -- the arithmetic of rope_apply_kernel / store_f16_t_kernel, element for element
__global__ __launch_bounds__(256) void qkv_rope_append_kernel(float *__restrict__ qkv, int64_t ldq, const float *__restrict__ sin_t, const float *__restrict__ cos_t, int ld_tab,
                                                              uint16_t *__restrict__ kout, int64_t ldk, uint16_t *__restrict__ vout, int64_t ldv, int S, int Hq, int Hkv, int D) {
    const int half = D >> 1;
    const int64_t nq = (int64_t)S * Hq * half, nk = (int64_t)S * Hkv * half, nv = (int64_t)S * Hkv * D;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < nq + nk + nv; t += (int64_t)gridDim.x * 256) {
        if (t < nq + nk) {
            const bool isk = t >= nq;
            const int64_t u = isk ? t - nq : t;
            const int H = isk ? Hkv : Hq;
            const int d = (int)(u % half), h = (int)((u / half) % H), s_ = (int)(u / ((int64_t)half * H));
            float *x = qkv + (int64_t)s_ * ldq + (isk ? Hq * D : 0) + h * D + d;
            const float a = x[0], b = x[half];
            const float sv = sin_t[(int64_t)s_ * ld_tab + d], cv = cos_t[(int64_t)s_ * ld_tab + d];
            const float v1 = __fmaf_rn(a, cv, -__fmul_rn(b, sv)), v2 = __fmaf_rn(a, sv, __fmul_rn(b, cv));
            if (isk) { uint16_t *o = kout + (int64_t)s_ * ldk + h * D + d; o[0] = f2h(v1); o[half] = f2h(v2); }
            else { x[0] = v1; x[half] = v2; }
        } else {
            const int64_t u = t - nq - nk;
            const int c = (int)(u / S), s_ = (int)(u % S);
            vout[(int64_t)c * ldv + s_] = f2h(qkv[(int64_t)s_ * ldq + (Hq + Hkv) * D + c]);
        }
    }
}
// RoPE(q) -> q_out, RoPE(k) -> k_out and its fp16 image in the K slab rows, v -> fp16 V slab rows: four Ops of the reference's attention block (two RoPE layers, two KVCache
// layers) that the adapter's lazy window hands over together; rope_apply_kernel's and store_f16's arithmetic (the fp16 k is the rounding of the stored fp32 k_out)
__global__ __launch_bounds__(256) void rope2_store2_kernel(const float *__restrict__ q, const float *__restrict__ sin_q, const float *__restrict__ cos_q, int ld_tab_q,
                                                           float *__restrict__ q_out, int Hq, const float *__restrict__ k, const float *__restrict__ sin_k,
                                                           const float *__restrict__ cos_k, int ld_tab_k, float *__restrict__ k_out, uint16_t *__restrict__ k16,
                                                           const float *__restrict__ v, uint16_t *__restrict__ v16, int Hkv, int S, int D) {
    // blockIdx.y = position; x walks the (Hq + Hkv) * D / 2 rotation pairs and the Hkv * D values of v of that position (32-bit index arithmetic: a decode step is one position)
    const int half = D >> 1, s_ = blockIdx.y;
    const int nq = Hq * half, nk = Hkv * half, nv = Hkv * D;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < nq + nk + nv; t += gridDim.x * 256) {
        if (t < nq + nk) {
            const bool isk = t >= nq;
            const int u = isk ? t - nq : t, H = isk ? Hkv : Hq;
            const int h = u / half, d = u - h * half;
            const int64_t o = (int64_t)s_ * H * D + h * D + d;
            const float *x = isk ? k : q;
            const float a = x[o], b = x[o + half];
            const float sv = isk ? sin_k[(int64_t)s_ * ld_tab_k + d] : sin_q[(int64_t)s_ * ld_tab_q + d], cv = isk ? cos_k[(int64_t)s_ * ld_tab_k + d] : cos_q[(int64_t)s_ * ld_tab_q + d];
            const float v1 = __fmaf_rn(a, cv, -__fmul_rn(b, sv)), v2 = __fmaf_rn(a, sv, __fmul_rn(b, cv));
            if (isk) { k_out[o] = v1; k_out[o + half] = v2; k16[o] = f2h(v1); k16[o + half] = f2h(v2); }
            else { q_out[o] = v1; q_out[o + half] = v2; }
        } else {
            const int64_t u = (int64_t)s_ * nv + (t - nq - nk);
            v16[u] = f2h(v[u]);
        }
    }
}
// the same for B sequences that each append ONE token to their own slabs (batched decode): row b of qkv, rotary row b, destination = sequence b's slabs at its position t_b
__global__ __launch_bounds__(256) void seqs_rope_append_kernel(float *__restrict__ qkv, int64_t ldq, const float *__restrict__ sin_t, const float *__restrict__ cos_t, int ld_tab,
                                                               const SeqKV *__restrict__ seqs, int64_t layer_k_off, int64_t layer_v_off, int64_t ldk, int64_t ldv, int Hq, int Hkv,
                                                               int D) {
    const int b = blockIdx.y, half = D >> 1;
    const SeqKV sq = seqs[b];
    uint16_t *kout = sq.k + layer_k_off + (int64_t)sq.t * ldk, *vout = sq.v + layer_v_off + sq.t;
    float *row = qkv + (int64_t)b * ldq;
    const int nq = Hq * half, nk = Hkv * half, nv = Hkv * D;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < nq + nk + nv; t += gridDim.x * 256) {
        if (t < nq + nk) {
            const bool isk = t >= nq;
            const int u = isk ? t - nq : t;
            const int d = u % half, h = u / half;
            float *x = row + (isk ? Hq * D : 0) + h * D + d;
            const float a = x[0], bb = x[half];
            const float sv = sin_t[(int64_t)b * ld_tab + d], cv = cos_t[(int64_t)b * ld_tab + d];
            const float v1 = __fmaf_rn(a, cv, -__fmul_rn(bb, sv)), v2 = __fmaf_rn(a, sv, __fmul_rn(bb, cv));
            if (isk) { uint16_t *o = kout + h * D + d; o[0] = f2h(v1); o[half] = f2h(v2); }
            else { x[0] = v1; x[half] = v2; }
        } else {
            const int c = t - nq - nk;
            vout[(int64_t)c * ldv] = f2h(row[(Hq + Hkv) * D + c]);
        }
    }
}
__global__ __launch_bounds__(256) void store_f16_kernel(const float *__restrict__ x, int64_t ldx, uint16_t *__restrict__ out, int64_t ldo, int S, int n) {
    const int64_t total = (int64_t)S * n;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int s = (int)(t / n), c = (int)(t % n);
        out[(int64_t)s * ldo + c] = f2h(x[(int64_t)s * ldx + c]);
    }
}
// conv2d receptive-field gather: image given in logical (h, c, w) order -> rows [oh*ow][c][kh][kw], the flattening the
// reference uses for both kernel and receptive field (Convolution.cpp:8-33, :45-60), so weights stay [OC][C][kh][kw]
// CHW: the image in the MEMORY order of the reference's image Tensor [B, head = H, sequence = C, dimension = W] (BSHD memory = [C][H][W]: what a Tensor uploaded by
// Backend::copy_from_host holds); same rows out
template <bool CHW>
__global__ __launch_bounds__(256) void im2patch_kernel(const float *__restrict__ img, float *__restrict__ patches, int H, int C, int W, int p) {
    const int ow = W / p, KK = p * C * p;
    const int64_t total = (int64_t)(H / p) * ow * KK;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int k = (int)(t % KK);
        const int pix = (int)(t / KK);
        const int kw = k % p, kh = (k / p) % p, c = k / (p * p);
        const int oy = pix / ow, ox = pix % ow;
        patches[t] = CHW ? img[((int64_t)c * H + (oy * p + kh)) * W + ox * p + kw] : img[((int64_t)(oy * p + kh) * C + c) * W + ox * p + kw];
    }
}

static inline int grid_for(int64_t n_items, int per_block, int cap = 4096) {
    int64_t g = (n_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
}  // namespace mllm_hip

using namespace mllm_hip;

extern "C" int mllm_hip_quantize_q8k(const float *x, int8_t *qs, float *d, int16_t *bsums, int M, int K, void *stream) {
    if (K % 256 != 0 || M < 0) return MLLM_HIP_ERR_SHAPE;
    const int64_t nb = (int64_t)M * (K / 256);
    if (nb == 0) return MLLM_HIP_OK;
    hipLaunchKernelGGL(quantize_q8k_kernel, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, as_stream(stream), x, qs, d, bsums, nb);
    return MH_LAUNCH_OK("quantize_q8k");
}
extern "C" int mllm_hip_quantize_q80(const float *x, int8_t *qs, uint16_t *d, int M, int K, void *stream) {
    if (K % 32 != 0 || M < 0) return MLLM_HIP_ERR_SHAPE;
    const int64_t nb = (int64_t)M * (K / 32);
    if (nb == 0) return MLLM_HIP_OK;
    hipLaunchKernelGGL(quantize_q80_kernel, dim3((unsigned)((nb + 31) / 32)), dim3(256), 0, as_stream(stream), x, qs, d, nb);
    return MH_LAUNCH_OK("quantize_q80");
}
extern "C" int mllm_hip_rmsnorm(const float *x, const float *w, float *y, int8_t *qs, float *d, int16_t *bsums, int M, int dim,
                                float eps, int add_unit_offset, void *stream) {
    if (M <= 0) return MLLM_HIP_OK;
    if (qs && (dim % 256 != 0 || !d || !bsums)) return MLLM_HIP_ERR_SHAPE;
    if (!qs && !y) return MLLM_HIP_ERR_ARG;
    hipLaunchKernelGGL(norm_kernel<false>, dim3(M), dim3(256), 0, as_stream(stream), x, w, (const float *)nullptr, y, qs, d, bsums, dim, eps, add_unit_offset,
                       (const float *)nullptr, (uint8_t *)nullptr, M);
    return MH_LAUNCH_OK("rmsnorm");
}
static int layernorm_impl(const float *x, const float *w, const float *b, float *y, int8_t *qs, float *d, int16_t *bsums, uint8_t *pack,
                          int M, int dim, float eps, void *stream) {
    if (M <= 0) return MLLM_HIP_OK;
    if ((qs || pack) && dim % 256 != 0) return MLLM_HIP_ERR_SHAPE;
    if (qs && (!d || !bsums)) return MLLM_HIP_ERR_SHAPE;
    if (!qs && !y && !pack) return MLLM_HIP_ERR_ARG;
    if (dim % 256 == 0 && dim / 256 <= 8 && option(OPT_NO_LNF) <= 0) {      // rows of 256 .. 2048 values: the whole op in one launch
        const int rows = pack ? (M + 31) & ~31 : M;
        const dim3 grid((rows + LNF_ROWS - 1) / LNF_ROWS), block(64 * LNF_ROWS);
        const size_t lds = (size_t)LNF_ROWS * dim * sizeof(float);
#define LNF_CASE(N) case N: hipLaunchKernelGGL(ln_fused_kernel<N>, grid, block, lds, as_stream(stream), x, w, b, y, qs, d, bsums, pack, M, eps); break;
        switch (dim / 256) { LNF_CASE(1) LNF_CASE(2) LNF_CASE(3) LNF_CASE(4) LNF_CASE(5) LNF_CASE(6) LNF_CASE(7) LNF_CASE(8) }
#undef LNF_CASE
        return MH_LAUNCH_OK("layernorm_fused");
    }
    // per-row (mean, rms) scratch: grown on demand, owned by the library (stream-ordered use only)
    static float *stats = nullptr;
    static int stats_rows = 0;
    if (M > stats_rows) {
        if (stats) MH_CHECK(hipFree(stats));
        stats_rows = ((M + 1023) / 1024) * 1024;
        MH_CHECK(hipMalloc(&stats, (size_t)stats_rows * 2 * sizeof(float)));
    }
    hipLaunchKernelGGL(ln_stats_kernel, dim3((M + LN_ROWS - 1) / LN_ROWS), dim3(256), 0, as_stream(stream), x, stats, M, dim, eps);
    int rc = MH_LAUNCH_OK("ln_stats");
    if (rc) return rc;
    hipLaunchKernelGGL(norm_kernel<true>, dim3(pack ? (M + 31) & ~31 : M), dim3(256), 0, as_stream(stream), x, w, b, y, qs, d, bsums, dim, eps, 0, (const float *)stats,
                       pack, M);
    return MH_LAUNCH_OK("layernorm");
}
extern "C" int mllm_hip_layernorm(const float *x, const float *w, const float *b, float *y, int8_t *qs, float *d, int16_t *bsums,
                                  int M, int dim, float eps, void *stream) {
    return layernorm_impl(x, w, b, y, qs, d, bsums, nullptr, M, dim, eps, stream);
}
// norm / quantiser variants whose Q8_K output is the packed activation operand of mllm_hip_linear_q4kp_packed (`xpack` of
// mllm_hip_q4k_prepack_bytes(M, dim) bytes); y (fp32) optional
extern "C" int mllm_hip_layernorm_packed(const float *x, const float *w, const float *b, float *y, void *xpack, int M, int dim, float eps, void *stream) {
    if (!xpack) return MLLM_HIP_ERR_ARG;
    return layernorm_impl(x, w, b, y, nullptr, nullptr, nullptr, (uint8_t *)xpack, M, dim, eps, stream);
}
extern "C" int mllm_hip_rmsnorm_packed(const float *x, const float *w, float *y, void *xpack, int M, int dim, float eps, int add_unit_offset, void *stream) {
    if (M <= 0) return MLLM_HIP_OK;
    if (!xpack || dim % 256 != 0) return MLLM_HIP_ERR_SHAPE;
    hipLaunchKernelGGL(norm_kernel<false>, dim3((M + 31) & ~31), dim3(256), 0, as_stream(stream), x, w, (const float *)nullptr, y, (int8_t *)nullptr, (float *)nullptr,
                       (int16_t *)nullptr, dim, eps, add_unit_offset, (const float *)nullptr, (uint8_t *)xpack, M);
    return MH_LAUNCH_OK("rmsnorm_packed");
}
extern "C" int mllm_hip_quantize_q8k_packed(const float *x, void *xpack, int M, int K, void *stream) {
    if (K % 256 != 0 || M < 0 || !xpack) return MLLM_HIP_ERR_SHAPE;
    if (M == 0) return MLLM_HIP_OK;
    const int nb = K / 256;
    const int64_t blocks = (int64_t)((M + 31) & ~31) * nb;
    hipLaunchKernelGGL(quantize_q8k_pack_kernel, dim3((unsigned)((blocks + QP_WAVES - 1) / QP_WAVES)), dim3(64 * QP_WAVES), 0, as_stream(stream), x, (uint8_t *)xpack, M, nb);
    return MH_LAUNCH_OK("quantize_q8k_packed");
}
extern "C" int mllm_hip_quantize_q8k_packed_act(const float *x, const uint16_t *lut, void *xpack, int M, int K, void *stream) {
    if (K % 256 != 0 || M < 0 || !xpack || !lut) return MLLM_HIP_ERR_SHAPE;
    if (M == 0) return MLLM_HIP_OK;
    const int nb = K / 256;
    const int64_t blocks = (int64_t)((M + 31) & ~31) * nb;
    hipLaunchKernelGGL(quantize_q8k_pack_act_kernel<1>, dim3((unsigned)((blocks + QP_WAVES - 1) / QP_WAVES)), dim3(64 * QP_WAVES), 0, as_stream(stream), x, lut, (uint8_t *)xpack, M, nb);
    return MH_LAUNCH_OK("quantize_q8k_packed_act");
}
extern "C" int mllm_hip_quantize_q8k_packed_silu_mul(const float *gu, void *xpack, int M, int I, void *stream) {
    if (I % 256 != 0 || M < 0 || !xpack) return MLLM_HIP_ERR_SHAPE;
    if (M == 0) return MLLM_HIP_OK;
    const int nb = I / 256;
    const int64_t blocks = (int64_t)((M + 31) & ~31) * nb;
    hipLaunchKernelGGL(quantize_q8k_pack_act_kernel<2>, dim3((unsigned)((blocks + QP_WAVES - 1) / QP_WAVES)), dim3(64 * QP_WAVES), 0, as_stream(stream), gu, (const uint16_t *)nullptr, (uint8_t *)xpack, M, nb);
    return MH_LAUNCH_OK("quantize_q8k_packed_silu_mul");
}
extern "C" int mllm_hip_debug_ln_stats(const float *x, float *stats, int M, int dim, float eps, void *stream) {
    hipLaunchKernelGGL(ln_stats_kernel, dim3((M + LN_ROWS - 1) / LN_ROWS), dim3(256), 0, as_stream(stream), x, stats, M, dim, eps);
    return MH_LAUNCH_OK("ln_stats");
}
extern "C" int mllm_hip_silu(const float *x, float *y, int64_t n, void *stream) {
    if (n <= 0) return MLLM_HIP_OK;
    hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, as_stream(stream), x, y, n);
    return MH_LAUNCH_OK("silu");
}
extern "C" int mllm_hip_silu_rows(const float *x, float *y, int64_t rows, int dim, void *stream) {
    if (rows <= 0 || dim <= 0) return MLLM_HIP_OK;
    if (dim % 8 == 0) return mllm_hip_silu(x, y, rows * dim, stream);
    hipLaunchKernelGGL(silu_rows_kernel, dim3(grid_for(rows * dim, 256)), dim3(256), 0, as_stream(stream), x, y, rows, dim);
    return MH_LAUNCH_OK("silu_rows");
}
extern "C" int mllm_hip_silu_mul(const float *gu, float *y, int M, int I, void *stream) {
    if (I % 4 != 0) return MLLM_HIP_ERR_SHAPE;
    if (M <= 0) return MLLM_HIP_OK;
    hipLaunchKernelGGL(silu_mul_kernel, dim3(grid_for((int64_t)M * I / 4, 256)), dim3(256), 0, as_stream(stream), gu, y, M, I);
    return MH_LAUNCH_OK("silu_mul");
}
extern "C" int mllm_hip_build_act_luts(uint16_t *gelu_host, uint16_t *quickgelu_host) {
    // init_table_gelu_f16 / init_table_gelu_quick_f16 (ggml Quantize.hpp:74-131): libm tanhf/expf on the host, like the reference
    for (int i = 0; i < (1 << 16); ++i) {
        const _Float16 hv = *reinterpret_cast<const _Float16 *>(&(const uint16_t &)(uint16_t)i);
        const float f = (float)hv;
        if (gelu_host) {
            const _Float16 g = (_Float16)(0.5f * f * (1.0f + tanhf(0.79788456080286535587989211986876f * f * (1.0f + 0.044715f * f * f))));
            memcpy(&gelu_host[i], &g, 2);
        }
        if (quickgelu_host) {
            const _Float16 q = (_Float16)(f * (1.0f / (1.0f + expf(-1.702f * f))));
            memcpy(&quickgelu_host[i], &q, 2);
        }
    }
    return MLLM_HIP_OK;
}
extern "C" int mllm_hip_act_lut(const float *x, float *y, int64_t n, const uint16_t *lut, void *stream) {
    if (n <= 0) return MLLM_HIP_OK;
    hipLaunchKernelGGL(act_lut_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), x, y, n, lut);
    return MH_LAUNCH_OK("act_lut");
}
extern "C" int mllm_hip_add(const float *a, const float *b, float *y, int64_t n, void *stream) {
    if (n <= 0) return MLLM_HIP_OK;
    hipLaunchKernelGGL(binary_kernel<0>, dim3(grid_for(n, 1024)), dim3(256), 0, as_stream(stream), a, b, y, n);
    return MH_LAUNCH_OK("add");
}
extern "C" int mllm_hip_mul(const float *a, const float *b, float *y, int64_t n, void *stream) {
    if (n <= 0) return MLLM_HIP_OK;
    hipLaunchKernelGGL(binary_kernel<1>, dim3(grid_for(n, 1024)), dim3(256), 0, as_stream(stream), a, b, y, n);
    return MH_LAUNCH_OK("mul");
}
extern "C" int mllm_hip_softmax(const float *x, float *y, int rows, int n, const int *valid, void *stream) {
    if (rows <= 0) return MLLM_HIP_OK;
    hipStream_t st = as_stream(stream);
    if (rows == 1 && !valid && n >= 16384) {      // one long row: the order-free parts over the chip, the row sum by one wave (same bits as softmax_kernel)
        constexpr int NP = 128;
        const int nchunks = n >> 3;
        float *scr = nullptr;
        MH_CHECK(hipMallocAsync((void **)&scr, ((size_t)NP + 4 + nchunks) * 4, st));
        float *part = scr, *inv = scr + NP, *cs = scr + NP + 4;
        int rc = MLLM_HIP_OK;
        hipLaunchKernelGGL(softmax_row_max_kernel, dim3(NP), dim3(256), 0, st, x, n, part);
        rc = MH_LAUNCH_OK("softmax_row_max");
        if (!rc) { hipLaunchKernelGGL(softmax_row_exp_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, x, y, n, (const float *)part, NP, cs); rc = MH_LAUNCH_OK("softmax_row_exp"); }
        if (!rc) { hipLaunchKernelGGL(softmax_row_sum_kernel, dim3(1), dim3(64), 0, st, (const float *)cs, nchunks, (const float *)y, n, inv); rc = MH_LAUNCH_OK("softmax_row_sum"); }
        if (!rc) { hipLaunchKernelGGL(softmax_row_scale_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, y, n, (const float *)inv); rc = MH_LAUNCH_OK("softmax_row_scale"); }
        const hipError_t e = hipFreeAsync(scr, st);
        if (e != hipSuccess && !rc) { set_error("hipFreeAsync", e, __FILE__, __LINE__); rc = MLLM_HIP_ERR_HIP; }
        return rc;
    }
    hipLaunchKernelGGL(softmax_kernel, dim3(rows), dim3(64), 0, as_stream(stream), x, y, n, valid);
    return MH_LAUNCH_OK("softmax");
}
extern "C" int mllm_hip_index_put_rows(float *dst, const float *value, const int *idx, int n_rows, int dim, void *stream) {
    if (dim % 4 != 0) return MLLM_HIP_ERR_SHAPE;
    if (n_rows <= 0) return MLLM_HIP_OK;
    hipLaunchKernelGGL(index_put_rows_kernel, dim3(n_rows), dim3(256), 0, as_stream(stream), dst, value, idx, dim);
    return MH_LAUNCH_OK("index_put_rows");
}
extern "C" int mllm_hip_index_put_rows_fidx(float *dst, int n_dst_rows, const float *value, const float *idx, int n_rows, int dim, void *stream) {
    if (dim % 4 != 0 || n_dst_rows < 0) return MLLM_HIP_ERR_SHAPE;
    if (n_rows <= 0) return MLLM_HIP_OK;
    if (!dst || !value || !idx) return MLLM_HIP_ERR_ARG;
    hipLaunchKernelGGL(index_put_rows_fidx_kernel, dim3(n_rows), dim3(256), 0, as_stream(stream), dst, n_dst_rows, value, idx, dim);
    return MH_LAUNCH_OK("index_put_rows_fidx");
}
extern "C" int mllm_hip_copy_2d_f32(const float *src, int64_t lds, float *dst, int64_t ldd, int rows, int cols, void *stream) {
    if (rows < 0 || cols < 0 || cols % 4 || lds % 4 || ldd % 4 || lds < cols || ldd < cols) return MLLM_HIP_ERR_SHAPE;
    if (rows == 0 || cols == 0) return MLLM_HIP_OK;
    if (!src || !dst || ((uintptr_t)src | (uintptr_t)dst) % 16) return MLLM_HIP_ERR_ARG;
    const int64_t n = (int64_t)rows * (cols / 4);
    hipLaunchKernelGGL(copy_2d_f32_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, as_stream(stream), src, lds, dst, ldd, rows, cols / 4);
    return MH_LAUNCH_OK("copy_2d_f32");
}
extern "C" int mllm_hip_transpose_f32(const float *x, float *y, int rows, int cols, void *stream) {
    if (rows <= 0 || cols <= 0) return MLLM_HIP_ERR_SHAPE;
    if (!x || !y || x == y) return MLLM_HIP_ERR_ARG;
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, as_stream(stream), x, y, rows, cols);
    return MH_LAUNCH_OK("transpose_f32");
}
namespace mllm_hip {
int argmax_parts_launch(const float *x, int n, float *part_val, int *part_idx, int nparts, hipStream_t st) {
    hipLaunchKernelGGL(argmax_parts_kernel, dim3(nparts), dim3(256), 0, st, x, n, part_val, part_idx);
    return MH_LAUNCH_OK("argmax_parts");
}
int argmax_final_launch(const float *part_val, const int *part_idx, int nparts, int *out, hipStream_t st) {
    hipLaunchKernelGGL(argmax_final_kernel, dim3(1), dim3(64), 0, st, part_val, part_idx, nparts, out);
    return MH_LAUNCH_OK("argmax_final");
}
}  // namespace mllm_hip
extern "C" int mllm_hip_argmax(const float *x, int n, int *out_index, void *stream) {
    if (n <= 0) return MLLM_HIP_ERR_SHAPE;
    hipStream_t st = as_stream(stream);
    if (n < 16384) {      // short rows: one workgroup
        hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, st, x, n, out_index);
        return MH_LAUNCH_OK("argmax");
    }
    // long rows (a vocabulary): partials over the chip in stream-ordered scratch, then the fold
    constexpr int NP = 128;
    void *scr = nullptr;
    MH_CHECK(hipMallocAsync(&scr, NP * 8, st));
    int rc = argmax_parts_launch(x, n, (float *)scr, (int *)((char *)scr + NP * 4), NP, st);
    if (!rc) rc = argmax_final_launch((const float *)scr, (const int *)((char *)scr + NP * 4), NP, out_index, st);
    const hipError_t e = hipFreeAsync(scr, st);
    if (e != hipSuccess && !rc) { set_error("hipFreeAsync", e, __FILE__, __LINE__); rc = MLLM_HIP_ERR_HIP; }
    return rc;
}
namespace mllm_hip {
// the same selection over a slice, and over candidates that carry their own ids (ids == nullptr: the position is the id): workgroup b takes positions
// [b * per, (b + 1) * per) and leaves its k best as (value, id), best first; slots it cannot fill get id 0x7fffffff, which no later stage accepts as a candidate.
// The k best of the row are among the k best of every slice, and both stages order equal values by ascending id, so two stages select what the single workgroup does.
template <int NT>
__global__ __launch_bounds__(NT) void topk_slice_kernel(const float *__restrict__ x, const int *__restrict__ ids, int n, int per, int k, float *__restrict__ out_val,
                                                        int *__restrict__ out_idx) {
    // the slice's values live in LDS for the k rounds; a selected value is replaced by NaN (never a candidate), as is the value of an unfilled slot of the stage before
    extern __shared__ __attribute__((aligned(16))) char tk_smem[];
    float *vals = reinterpret_cast<float *>(tk_smem);      // [per]
    __shared__ float sv[NT / 64];
    __shared__ int si[NT / 64], sp[NT / 64];
    const int lo = blockIdx.x * per, cnt = max(0, min(n, lo + per) - lo);
    float *ov_ = out_val + (int64_t)blockIdx.x * k;
    int *oi_ = out_idx + (int64_t)blockIdx.x * k;
    for (int i = threadIdx.x; i < cnt; i += NT) vals[i] = (ids && ids[lo + i] == 0x7fffffff) ? __int_as_float(0x7fc00000) : x[lo + i];
    __syncthreads();
    for (int r = 0; r < k; ++r) {
        float best = -INFINITY;
        int bi = 0x7fffffff, bp = -1;
        for (int i = threadIdx.x; i < cnt; i += NT) {
            const float v = vals[i];
            if (v != v) continue;
            const int id = ids ? ids[lo + i] : lo + i;
            if (bi == 0x7fffffff || v > best || (v == best && id < bi)) { best = v; bi = id; bp = i; }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float ov = __shfl_xor(best, m, 64);
            const int oi = __shfl_xor(bi, m, 64), op = __shfl_xor(bp, m, 64);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; bp = op; }
        }
        if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; sp[threadIdx.x >> 6] = bp; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < NT / 64; ++w)
                if (si[w] != 0x7fffffff && (bi == 0x7fffffff || sv[w] > best || (sv[w] == best && si[w] < bi))) { best = sv[w]; bi = si[w]; bp = sp[w]; }
            if (bp >= 0) vals[bp] = __int_as_float(0x7fc00000);
            ov_[r] = best;
            oi_[r] = bi;
        }
        __syncthreads();
    }
}
}  // namespace mllm_hip
extern "C" int mllm_hip_topk(const float *x, int n, int k, float *out_val, int *out_idx, void *stream) {
    if (n <= 0 || k <= 0 || k > 64 || k > n) return MLLM_HIP_ERR_SHAPE;
    hipStream_t st = as_stream(stream);
    if (n < 16384) {
        hipLaunchKernelGGL(topk_kernel, dim3(1), dim3(1024), 0, st, x, n, k, out_val, out_idx);
        return MH_LAUNCH_OK("topk");
    }
    // a vocabulary row: 128 slices leave their k best each, one workgroup picks the row's k best among those (k rounds over 128 k candidates instead of over n values)
    constexpr int NP = 128;
    const int per = (n + NP - 1) / NP;
    void *scr = nullptr;
    MH_CHECK(hipMallocAsync(&scr, (size_t)NP * k * 8, st));
    float *cv = (float *)scr;
    int *ci = (int *)(cv + (size_t)NP * k);
    if (per > 8192) { (void)hipFreeAsync(scr, st); hipLaunchKernelGGL(topk_kernel, dim3(1), dim3(1024), 0, st, x, n, k, out_val, out_idx); return MH_LAUNCH_OK("topk"); }      // rows beyond a million values
    hipLaunchKernelGGL(topk_slice_kernel<256>, dim3(NP), dim3(256), (size_t)per * 4, st, x, (const int *)nullptr, n, per, k, cv, ci);
    int rc = MH_LAUNCH_OK("topk_slices");
    if (!rc) {
        hipLaunchKernelGGL(topk_slice_kernel<1024>, dim3(1), dim3(1024), (size_t)NP * k * 4, st, (const float *)cv, (const int *)ci, NP * k, NP * k, k, out_val, out_idx);
        rc = MH_LAUNCH_OK("topk_final");
    }
    const hipError_t e = hipFreeAsync(scr, st);
    if (e != hipSuccess && !rc) { set_error("hipFreeAsync", e, __FILE__, __LINE__); rc = MLLM_HIP_ERR_HIP; }
    return rc;
}
// Host part of the same method (Generate.cpp:69-87): softmax with temperature over the k candidates, in the reference's mixed float /
// double arithmetic with the host's libm exp (the reference runs it on the host too), then the renormalisation by the float sum.
extern "C" int mllm_hip_topk_probs_host(const float *top_val, int k, float temperature, float *probs) {
    if (k <= 0 || !top_val || !probs) return MLLM_HIP_ERR_ARG;
    int am = 0;
    for (int i = 1; i < k; ++i) if (top_val[i] > top_val[am]) am = i;      // std::max_element: first maximum
    const double max_logit = top_val[am];
    double sum_exp = 0.f;
    for (int i = 0; i < k; ++i) {
        probs[i] = exp((top_val[i] - max_logit) / temperature);
        sum_exp += probs[i];
    }
    for (int i = 0; i < k; ++i) probs[i] /= sum_exp;
    double acc = 0.0;
    for (int i = 0; i < k; ++i) acc += probs[i];
    const float fsum = acc;
    for (int i = 0; i < k; ++i) probs[i] /= fsum;
    return MLLM_HIP_OK;
}
extern "C" int mllm_hip_embedding_q40(const float *ids, const uint8_t *Wqs, const uint16_t *Wd, float *out, int S, int hidden, int vocab, void *stream) {
    if (hidden % 32 != 0) return MLLM_HIP_ERR_SHAPE;
    if (S <= 0) return MLLM_HIP_OK;
    hipLaunchKernelGGL(embedding_q40_kernel, dim3(S), dim3(256), 0, as_stream(stream), ids, Wqs, Wd, out, hidden, vocab);
    return MH_LAUNCH_OK("embedding_q40");
}
extern "C" int mllm_hip_repack_q40(const void *raw_blocks, uint8_t *qs, uint16_t *d, int64_t n_blocks, void *stream) {
    if (n_blocks <= 0) return MLLM_HIP_OK;
    hipLaunchKernelGGL(repack_q40_kernel, dim3(grid_for(n_blocks * 16, 256, 65535)), dim3(256), 0, as_stream(stream), (const uint8_t *)raw_blocks, qs, d, n_blocks);
    return MH_LAUNCH_OK("repack_q40");
}
extern "C" int mllm_hip_rope_apply(const float *x, int64_t ldx, const float *sin_t, const float *cos_t, int ld_tab, void *out,
                                   int out_dtype, int64_t ldo, int S, int H, int D, void *stream) {
    if (D % 2 != 0) return MLLM_HIP_ERR_SHAPE;
    if (S <= 0) return MLLM_HIP_OK;
    const int g = grid_for((int64_t)S * H * (D / 2), 256);
    if (out_dtype == MLLM_HIP_F16)
        hipLaunchKernelGGL(rope_apply_kernel<true>, dim3(g), dim3(256), 0, as_stream(stream), x, ldx, sin_t, cos_t, ld_tab, out, ldo, S, H, D, 0);
    else if (out_dtype == MLLM_HIP_F32)
        hipLaunchKernelGGL(rope_apply_kernel<false>, dim3(g), dim3(256), 0, as_stream(stream), x, ldx, sin_t, cos_t, ld_tab, out, ldo, S, H, D, 0);
    else return MLLM_HIP_ERR_DTYPE;
    return MH_LAUNCH_OK("rope_apply");
}
extern "C" int mllm_hip_rope2_store2(const float *q, const float *sin_q, const float *cos_q, int ld_tab_q, float *q_out, int Hq, const float *k, const float *sin_k, const float *cos_k,
                                     int ld_tab_k, float *k_out, uint16_t *k16, const float *v, uint16_t *v16, int Hkv, int S, int D, void *stream) {
    if (D % 2 != 0 || Hq <= 0 || Hkv <= 0) return MLLM_HIP_ERR_SHAPE;
    if (S <= 0) return MLLM_HIP_OK;
    const int per = (Hq + Hkv) * (D / 2) + Hkv * D;
    hipLaunchKernelGGL(rope2_store2_kernel, dim3((per + 255) / 256, S), dim3(256), 0, as_stream(stream), q, sin_q, cos_q, ld_tab_q, q_out, Hq, k, sin_k, cos_k, ld_tab_k, k_out, k16, v, v16, Hkv,
                       S, D);
    return MH_LAUNCH_OK("rope2_store2");
}
namespace mllm_hip {
// fp32 in place over `rows` rows whose table row is (row % period): the images of one vision pass in one launch
int rope_apply_periodic(float *x, int64_t ldx, const float *sin_t, const float *cos_t, int ld_tab, int rows, int period, int H, int D, hipStream_t st) {
    if (rows <= 0) return MLLM_HIP_OK;
    hipLaunchKernelGGL(rope_apply_kernel<false>, dim3(grid_for((int64_t)rows * H * (D / 2), 256)), dim3(256), 0, st, x, ldx, sin_t, cos_t, ld_tab, (void *)x, ldx, rows, H, D, period);
    return MH_LAUNCH_OK("rope_apply_periodic");
}
}  // namespace mllm_hip
extern "C" int mllm_hip_qkv_rope_append(float *qkv, int64_t ldq, const float *sin_t, const float *cos_t, int ld_tab, uint16_t *k_rows, int64_t ldk, uint16_t *v_t, int64_t ldv,
                                       int S, int Hq, int Hkv, int D, void *stream) {
    if (D <= 0 || D % 2 || Hq <= 0 || Hkv <= 0 || ldq < (int64_t)(Hq + 2 * Hkv) * D) return MLLM_HIP_ERR_SHAPE;
    if (S <= 0) return MLLM_HIP_OK;
    if (!qkv || !sin_t || !cos_t || !k_rows || !v_t) return MLLM_HIP_ERR_ARG;
    const int64_t n = (int64_t)S * (Hq + Hkv) * (D / 2) + (int64_t)S * Hkv * D;
    hipLaunchKernelGGL(qkv_rope_append_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), qkv, ldq, sin_t, cos_t, ld_tab, k_rows, ldk, v_t, ldv, S, Hq, Hkv, D);
    return MH_LAUNCH_OK("qkv_rope_append");
}
namespace mllm_hip {
int seqs_rope_append_launch(float *qkv, int64_t ldq, const float *sin_t, const float *cos_t, int ld_tab, const SeqKV *seqs_dev, int64_t layer_k_off, int64_t layer_v_off,
                            int64_t ldk, int64_t ldvt, int B, int Hq, int Hkv, int D, hipStream_t st) {
    if (B <= 0) return MLLM_HIP_OK;
    if (D % 2 || !qkv || !seqs_dev) return MLLM_HIP_ERR_ARG;
    const int n = (Hq + Hkv) * (D / 2) + Hkv * D;
    hipLaunchKernelGGL(seqs_rope_append_kernel, dim3((n + 255) / 256, B), dim3(256), 0, st, qkv, ldq, sin_t, cos_t, ld_tab, seqs_dev, layer_k_off, layer_v_off, ldk, ldvt, Hq, Hkv, D);
    return MH_LAUNCH_OK("seqs_rope_append");
}
}  // namespace mllm_hip
extern "C" int mllm_hip_store_f16(const float *x, int64_t ldx, uint16_t *out, int64_t ldo, int S, int n, void *stream) {
    if (S <= 0 || n <= 0) return MLLM_HIP_OK;
    hipLaunchKernelGGL(store_f16_kernel, dim3(grid_for((int64_t)S * n, 256)), dim3(256), 0, as_stream(stream), x, ldx, out, ldo, S, n);
    return MH_LAUNCH_OK("store_f16");
}
extern "C" int mllm_hip_store_f16_t(const float *x, int64_t ldx, uint16_t *out, int64_t ldo, int S, int n, void *stream) {
    if (S <= 0 || n <= 0) return MLLM_HIP_OK;
    hipLaunchKernelGGL(store_f16_t_kernel, dim3(grid_for((int64_t)S * n, 256)), dim3(256), 0, as_stream(stream), x, ldx, out, ldo, S, n);
    return MH_LAUNCH_OK("store_f16_t");
}
extern "C" int mllm_hip_im2patch_hcw(const float *img, float *patches, int H, int C, int W, int p, void *stream) {
    if (H % p || W % p) return MLLM_HIP_ERR_SHAPE;
    hipLaunchKernelGGL(im2patch_kernel<false>, dim3(grid_for((int64_t)H * C * W, 256)), dim3(256), 0, as_stream(stream), img, patches, H, C, W, p);
    return MH_LAUNCH_OK("im2patch_hcw");
}
extern "C" int mllm_hip_im2patch_chw(const float *img, float *patches, int H, int C, int W, int p, void *stream) {
    if (H % p || W % p) return MLLM_HIP_ERR_SHAPE;
    hipLaunchKernelGGL(im2patch_kernel<true>, dim3(grid_for((int64_t)H * C * W, 256)), dim3(256), 0, as_stream(stream), img, patches, H, C, W, p);
    return MH_LAUNCH_OK("im2patch_chw");
}

// ---- host-side rotary tables: the reference's own libm formulas ------------------------------------------------------
extern "C" int mllm_hip_rope_table_hf(float base, int dim, int n_pos, float *sin_host, float *cos_host) {
    // CPURoPE.cpp:22-31 (theta in double -> float), :100-128 (HF half-split table, both halves filled)
    const int half = dim / 2;
    for (int i = 0; i < half; ++i) {
        const float theta = (float)(1.0 / pow((double)base, 2.0 * i / dim));
        for (int s = 0; s < n_pos; ++s) {
            const float v = (float)s * theta;
            sin_host[(size_t)s * dim + i] = sin_host[(size_t)s * dim + i + half] = sinf(v);
            cos_host[(size_t)s * dim + i] = cos_host[(size_t)s * dim + i + half] = cosf(v);
        }
    }
    return MLLM_HIP_OK;
}
extern "C" int mllm_hip_rope_table_hf_llama3(float base, int dim, int n_pos, float factor, float low_freq_factor, float high_freq_factor, float original_max_pos,
                                             float *sin_host, float *cos_host) {
    // CPURoPE.cpp:33-71 (_compute_llama3_theta: Llama-3.x frequency scaling; the blend contracts to one fma in the reference build), then the HF table of :100-128
    if (dim <= 0 || dim % 2 || n_pos <= 0 || !sin_host || !cos_host || factor == 0.0f || low_freq_factor == 0.0f || high_freq_factor == low_freq_factor) return MLLM_HIP_ERR_ARG;
    const int half = dim / 2;
    const float low_freq_wavelen = original_max_pos / low_freq_factor, high_freq_wavelen = original_max_pos / high_freq_factor;
    for (int i = 0; i < half; ++i) {
        float theta = (float)(1.0 / pow((double)base, 2.0 * i / dim));
        const float wavelen = (float)(2 * M_PI / theta);
        if (wavelen > low_freq_wavelen) {
            theta /= factor;
        } else if (wavelen >= high_freq_wavelen && wavelen <= low_freq_wavelen) {
            const float smooth = (original_max_pos / wavelen - low_freq_factor) / (high_freq_factor - low_freq_factor);
            theta = fmaf(1 - smooth, theta / factor, smooth * theta);
        }
        for (int s = 0; s < n_pos; ++s) {
            const float v = (float)s * theta;
            sin_host[(size_t)s * dim + i] = sin_host[(size_t)s * dim + i + half] = sinf(v);
            cos_host[(size_t)s * dim + i] = cos_host[(size_t)s * dim + i + half] = cosf(v);
        }
    }
    return MLLM_HIP_OK;
}
extern "C" int mllm_hip_mrope_table(float base, int dim, const float *pos, int S, const int *section, int n_section, float *sin_host, float *cos_host) {
    // CPUMultimodalRoPE.cpp:26-36 theta, :84-118 per-axis sin/cos, :37-82 stitch by mrope_section; tables [S][dim/2]
    const int half = dim / 2;
    int c0 = 0;
    for (int j = 0; j < n_section; ++j) {
        const int axis = j % 3;
        for (int c = c0; c < c0 + section[j] && c < half; ++c) {
            const float theta = (float)(1.0 / pow((double)base, 2.0 * c / dim));
            for (int s = 0; s < S; ++s) {
                const float v = theta * pos[(size_t)axis * S + s];
                sin_host[(size_t)s * half + c] = sinf(v);
                cos_host[(size_t)s * half + c] = cosf(v);
            }
        }
        c0 += section[j];
    }
    return MLLM_HIP_OK;
}
// VISIONROPE's output (CPUVisionRoPE.cpp:19-28 inv_freq by float pow, :56-103 (h, w) per patch in merge-block order, :29-55 angle = pos * inv_freq): the angle
// table `[t*h*w][rot_dim]`, h angles in columns [0, rot_dim/2), w angles behind them
extern "C" int mllm_hip_vision_rope_angles(int t, int h, int w, int merge, int rot_dim, float *angles_host) {
    const int q = rot_dim / 2;
    float inv[256];
    if (q <= 0 || q > 256 || merge <= 0 || t < 0 || h < 0 || w < 0) return MLLM_HIP_ERR_SHAPE;
    if (!angles_host) return MLLM_HIP_ERR_ARG;
    for (int i = 0; i < q; ++i) inv[i] = 1.0f / powf(10000.0f, (2.0f * i) / (float)rot_dim);
    const int nhb = h / merge, nwb = w / merge;
    size_t p = 0;
    for (int ti = 0; ti < t; ++ti)
        for (int bh = 0; bh < nhb; ++bh)
            for (int bw = 0; bw < nwb; ++bw)
                for (int jh = 0; jh < merge; ++jh)
                    for (int jw = 0; jw < merge; ++jw, ++p) {
                        const int ph = bh * merge + jh, pw = bw * merge + jw;
                        for (int i = 0; i < q; ++i) {
                            angles_host[p * rot_dim + i] = (float)ph * inv[i];
                            angles_host[p * rot_dim + q + i] = (float)pw * inv[i];
                        }
                    }
    return MLLM_HIP_OK;
}
// CPUVisionRoPEFunc.hpp:21-60 evaluates std::sin / std::cos of the angle per use: tabulated here.  Tables [t*h*w][rot_dim]
extern "C" int mllm_hip_vision_rope_table(int t, int h, int w, int merge, int rot_dim, float *sin_host, float *cos_host) {
    if (!sin_host || !cos_host) return MLLM_HIP_ERR_ARG;
    const int q = rot_dim / 2;
    if (q <= 0 || q > 256 || merge <= 0 || t < 0 || h < 0 || w < 0) return MLLM_HIP_ERR_SHAPE;
    // an angle is (float)position * inv[i] with position < max(h, w): max(h, w) * q distinct values (640 for a 448 x 448 image) instead of t * h * w * rot_dim libm calls (82 k,
    // about 2 ms of the caller's thread) -- the same float goes into the same sinf / cosf, so the tables are the same bits
    float inv[256];
    for (int i = 0; i < q; ++i) inv[i] = 1.0f / powf(10000.0f, (2.0f * i) / (float)rot_dim);
    const int np = std::max(h, w);
    std::vector<float> sv((size_t)np * q), cv((size_t)np * q);
    for (int pos = 0; pos < np; ++pos)
        for (int i = 0; i < q; ++i) { const float a = (float)pos * inv[i]; sv[(size_t)pos * q + i] = sinf(a); cv[(size_t)pos * q + i] = cosf(a); }
    const int nhb = h / merge, nwb = w / merge;
    size_t p = 0;
    for (int ti = 0; ti < t; ++ti)
        for (int bh = 0; bh < nhb; ++bh)
            for (int bw = 0; bw < nwb; ++bw)
                for (int jh = 0; jh < merge; ++jh)
                    for (int jw = 0; jw < merge; ++jw, ++p) {
                        const int ph = bh * merge + jh, pw = bw * merge + jw;
                        for (int i = 0; i < q; ++i) {
                            sin_host[p * rot_dim + i] = sv[(size_t)ph * q + i]; cos_host[p * rot_dim + i] = cv[(size_t)ph * q + i];
                            sin_host[p * rot_dim + q + i] = sv[(size_t)pw * q + i]; cos_host[p * rot_dim + q + i] = cv[(size_t)pw * q + i];
                        }
                    }
    return MLLM_HIP_OK;
}
