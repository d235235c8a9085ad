// mllm_amd/csrc/kernels_linear.hip -- A1/A2/A5/A6: Linear / mat_mul for gfx950.
//
//   decode (M == 1): weight-streaming GEMV, HBM-bound.  One 64-lane wave owns an output row at a time; the 144-B
//     Q4_K super-blocks of the row are spread 8 lanes per block (one 16-B dwordx4 of nibbles per lane, so a wave
//     load covers 8 consecutive blocks = 1152 contiguous bytes), the Q8_K activation slice of each lane is loop
//     invariant and lives in registers, the 4x8-bit dot products run on v_dot4_i32_i8, integers are reduced exactly
//     with wavefront shuffles and only then scaled (int-exact per super-block like vec_dot_q4_K_q8_K).
//   prefill / vision (M >= 16): class-decomposed exact GEMM on v_mfma_f32_32x32x16_f16 (gemm_q4k_kernel below): per column class of
//     vec_dot_q4_K_q8_K's AVX2 lanes the integer sum of (nibble * 6-bit scale) x q8 -- all operands and partial sums exact in fp16 / fp32 --
//     comes out of two MFMAs on pre-packed fp16 operands; the mins term sum_j mn_j * bsum_j is one more MFMA per pair of sub-blocks;
//     the reference's per-super-block fp32 chain step acc = fma(d_x d_w, sum, acc) then runs on the VALU, so the result is bit-identical.
//   fp32 weights (patch-embed conv, fp32 models): gemm_f32_mfma_kernel -- vec_dot_fp32's 32 chains as 32 accumulator tiles of v_mfma_f32_16x16x4_f32 -- and, for
//   shapes it does not take (M < 16, K < 128, K % 4), gemm_f32_kernel: the 32 chains + ordered leftovers on the VALU.
#include <type_traits>
#include <atomic>

#include "common.h"
#include "q4k_dot.h"
#include "q40_dot.h"

namespace mllm_hip {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------------------------
// Q4_K x Q8_K GEMV (M == 1): q4k_dot.h holds the arithmetic (reference accumulation order, bit-exact)
// ------------------------------------------------------------------------------------------------------------------
// NSTEPS = ceil(K/2048): wave steps per row. ROWS = rows in flight per wave (memory-level parallelism).
template <int NSTEPS, int ROWS>
__global__ __launch_bounds__(256) void gemv_q4k_kernel(const uint8_t *__restrict__ W, const float *__restrict__ bias, const int8_t *__restrict__ xqs,
                                                       const float *__restrict__ xd, const int16_t *__restrict__ xbsums, void *__restrict__ y, int y_f16,
                                                       const float *__restrict__ residual, int N, int nb, int rows_per_wave, int M, int64_t ldy) {
    // M activation rows (M < 16: batched decode, short prefills) meet every weight row while it sits in registers: the weights are streamed ONCE per launch, the
    // activation planes of row m (a few KiB, L1 / L2 resident) are re-read per batch of weight rows.  Each (m, n) dot is the M == 1 arithmetic, unchanged.
    __shared__ float2 tab_all[4 * (ROWS < 4 ? ROWS : 4) * NSTEPS * 8 * Q4K_SLOTS];
    const int lane = threadIdx.x & 63, g = lane >> 3, wid = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wid;
    float2 *tab = tab_all + wid * (ROWS < 4 ? ROWS : 4) * NSTEPS * 8 * Q4K_SLOTS;
    Q4KAct<NSTEPS> A;
    if (M == 1) q4k_load_act_planes<NSTEPS>(A, xqs, xd, xbsums, nb, lane);
    const int qoff = q4k_lane_qoff(lane);
    const int row0 = wave * rows_per_wave;
    const int row1 = min(N, row0 + rows_per_wave);
    for (int row = row0; row < row1; row += ROWS) {
        uint4 hdr[ROWS][NSTEPS], q[ROWS][NSTEPS];
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            const int rw = min(row + rr, row1 - 1);
#pragma unroll
            for (int st = 0; st < NSTEPS; ++st) {
                const int blk = A.valid[st] ? st * 8 + g : 0;
                const uint8_t *wb = W + ((int64_t)rw * nb + blk) * 144;
                hdr[rr][st] = *reinterpret_cast<const uint4 *>(wb);
                q[rr][st] = *reinterpret_cast<const uint4 *>(wb + 16 + qoff);
            }
        }
        for (int m = 0; m < M; ++m) {
            if (M > 1) q4k_load_act_planes<NSTEPS>(A, xqs + (int64_t)m * nb * 256, xd + (int64_t)m * nb, xbsums + (int64_t)m * nb * 16, nb, lane);
            float out[ROWS];
            wave_lds_fence();   // the previous iteration's chain reads are done before the table is overwritten
            q4k_dot_rows<NSTEPS, ROWS>(hdr, q, A, nb, lane, tab, out);
            if (lane == 0) {
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) {
                    const int rw = row + rr;
                    if (rw < row1) {
                        float v = out[rr];
                        if (bias) v = v + bias[rw];
                        if (y_f16) reinterpret_cast<uint16_t *>(y)[(int64_t)m * ldy + rw] = f2h(v);
                        else {
                            if (residual) v = v + residual[(int64_t)m * ldy + rw];
                            reinterpret_cast<float *>(y)[(int64_t)m * ldy + rw] = v;
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Q4_0 (nibble/scale planes) x Q8_0 GEMV (M == 1): the tied lm_head (modeling_qwen2_vl.hpp:399 -> CPUmmFunction ->
// vec_dot_q4_0_q8_0, VecDotQ4.cpp:514-545): per 32-block, acc[t] = fma(fp16(x.d) * fp16(y.d), float(s_t), acc[t]) with
// s_t the dot of bytes 4t..4t+3 of (nibbles - 8) (low nibbles = bytes 0..15, high = 16..31), blocks in order, then
// hsum_float_8.  LPR lanes per row, BPL blocks per lane (K = 32*LPR*BPL): every lane forms the 8 class sums of its blocks,
// parks them in a per-wave LDS table [row][block][8] and 8 lanes per row walk the table in block order (q40_chain).
// ------------------------------------------------------------------------------------------------------------------
template <int BPL, int LPR>
__global__ __launch_bounds__(256) void gemv_q40_kernel(const uint8_t *__restrict__ Wqs, const uint16_t *__restrict__ Wd, const float *__restrict__ bias,
                                                       const int8_t *__restrict__ xqs, const uint16_t *__restrict__ xd, float *__restrict__ y, int N,
                                                       int rows_per_wave, int M, int64_t ldy) {
    extern __shared__ __attribute__((aligned(16))) char q40_smem[];
    constexpr int RPW = 64 / LPR;            // rows per load pass
    constexpr int NPASS = 8 / RPW;           // load passes per 8-row chain pass
    constexpr int nblk = BPL * LPR;
    const int lane = threadIdx.x & 63, sub = lane % LPR, rsel = lane / LPR, wid = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wid;
    float *ts = reinterpret_cast<float *>(q40_smem) + wid * q40_tab_floats(nblk), *td = ts + 8 * q40_ts_stride(nblk);
    Q40Act<BPL> A;      // (M rows: as in gemv_q4k_kernel, the weight rows of a pass stay in registers while every activation row meets them)
    if (M == 1) q40_load_act<BPL, LPR>(A, xqs, nullptr, xd, sub);
    const int row0 = wave * rows_per_wave, row1 = min(N, row0 + rows_per_wave);
    for (int base = row0; base < row1; base += 8) {
        uint4 q[NPASS][BPL];
        uint16_t dw[NPASS][BPL];
#pragma unroll
        for (int u = 0; u < NPASS; ++u) {
            const int rw = min(base + RPW * u + rsel, row1 - 1);
#pragma unroll
            for (int b = 0; b < BPL; ++b) {
                const int64_t bi = (int64_t)rw * nblk + sub + LPR * b;
                q[u][b] = *reinterpret_cast<const uint4 *>(Wqs + bi * 16);
                dw[u][b] = Wd[bi];
            }
        }
        for (int m = 0; m < M; ++m) {
            if (M > 1) q40_load_act<BPL, LPR>(A, xqs + (int64_t)m * nblk * 32, nullptr, xd + (int64_t)m * nblk, sub);
            wave_lds_fence();
#pragma unroll
            for (int u = 0; u < NPASS; ++u) {
                const int rl = RPW * u + rsel;
                q40_emit<BPL, LPR>(q[u], dw[u], A, sub, ts + (size_t)rl * q40_ts_stride(nblk), td + (size_t)rl * q40_td_stride(nblk));
            }
            wave_lds_fence();
            const float acc = q40_chain(ts, td, nblk, min(8, row1 - base), lane);
            const int rw = base + (lane >> 3);
            if ((lane & 7) == 0 && rw < row1) y[(int64_t)m * ldy + rw] = bias ? acc + bias[rw] : acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Q4_K x Q8_K GEMM (M >= 16) in the reference's accumulation order, on v_mfma_f32_32x32x16_f16.
//
// Per (activation row m, weight row n) the reference keeps 8 + 4 fp32 chains over the super-blocks (q4k_dot.h).  The
// integer partial sums that feed a chain step are exact in fp32 whatever order they are added in, so they are formed on
// the matrix cores: for column class t of super-block i,  sumi[t] = sum_{sb < 8} sc[sb] * dot4(q4[sb][4t..], q8[sb][4t..])
// is a K = 32 dot product of (q4 * sc) (<= 945, exact in fp16) with q8 (exact in fp16) = two 32x32x16 MFMAs with a zero C;
// the mins product prod[u] = mn[2u] q8s[2u] + mn[2u+1] q8s[2u+1] is one more MFMA per u with q8s split into an even part
// (|.| <= 4064, even: exact in fp16) and its low bit.  The fp32 chain step acc = fma(d_m d_n, sum, acc) then runs on the
// VALU over the 16 accumulator registers, once per class / mins lane.
// Operands are pre-swizzled so every MFMA operand is one coalesced 16-byte (8-byte for mins) load per lane:
//   weights  q4k_prepack: Bw[n/32][i][t][p][h][n%32][8 f16]  (k = 4 low nibbles * sc[2j], 4 high nibbles * sc[2j+1], j = 2p+h)
//                         Bm[n/32][i][u][n%32][4 f16] = (mn[2u], mn[2u], mn[2u+1], mn[2u+1]);  Bd[n/32][i][n%32] = (d, dmin)
//   acts     q8k_prepack: Ax[m/32][i][t][p][h][m%32][8 f16]  (q8[64j+4t+b], q8[64j+32+4t+b]);
//                         Am[m/32][i][u][m%32][4 f16] = (even(q8s[2u]), q8s[2u]&1, even(q8s[2u+1]), q8s[2u+1]&1);  Ad[m/32][i][m%32] = y.d
// A = activations (C rows = m), B = weights (C cols = n).  Lane l: col = l & 31, h = l >> 5, C reg r -> row (r&3) + 8 (r>>2) + 4h.
// ------------------------------------------------------------------------------------------------------------------
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));


// Weight side of the GEMM: the nibbles stay nibbles in HBM and in LDS (0.72 B per weight) and are expanded to the fp16 MFMA operand nibble * scale in
// registers (q4_expand below).  Per (32-row tile, super-block i):
//   Bq [class pair tp = t >> 1][lane = 32 h + n % 32][4 dwords (t & 1, p)]   one dword = the 8 weights of fragment (t, p) of that lane: the bytes
//        qs[32 j + 4 t .. + 3], j = 2 p + h, nibbles re-ordered so that two masks and one shift split them into fp16 pairs (see q4_expand):
//        bits 0-3 lo0, 4-7 lo2, 8-11 hi0, 12-15 hi2, 16-19 lo1, 20-23 lo3, 24-27 hi1, 28-31 hi3   (lo_b / hi_b = low / high nibble of byte b)
//   Bs [lane][p]            fp16 pair (sc[2 j], sc[2 j + 1]), j = 2 p + h
//   Bm [u][n % 32][4 f16]   (mn[2u], mn[2u], mn[2u+1], mn[2u+1]);   Bd [n % 32] = (d, dmin)
// one thread per (row n, super-block i, chunk j)
__global__ __launch_bounds__(256) void q4k_prepack_kernel(const uint8_t *__restrict__ W, uint8_t *__restrict__ out, int N, int nb) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int j = (int)(gid & 3);
    const int64_t ni = gid >> 2;
    const int i = (int)(ni % nb);
    const int n = (int)(ni / nb);
    if (n >= ((N + 31) / 32) * 32) return;
    const size_t tb = q4kp_tile_blocks(N, nb);
    uint32_t *Bq = reinterpret_cast<uint32_t *>(out);
    uint32_t *Bs = reinterpret_cast<uint32_t *>(out + tb * Q4KW_Q_PER_BLK);
    v4h *Bm = reinterpret_cast<v4h *>(out + tb * (Q4KW_Q_PER_BLK + Q4KW_S_PER_BLK));
    float2 *Bd = reinterpret_cast<float2 *>(out + tb * (Q4KW_Q_PER_BLK + Q4KW_S_PER_BLK + Q4KW_M_PER_BLK));
    const size_t tile = (size_t)(n >> 5) * nb + i;
    const int nin = n & 31, p = j >> 1, h = j & 1;
    const bool live = n < N;
    const uint8_t *wb = W + ((int64_t)(live ? n : 0) * nb + i) * 144;
    const uint4 hdr = *reinterpret_cast<const uint4 *>(wb);
    uint32_t sc8[2], mn8[2];
    unpack_q4k_scales(hdr.y, hdr.z, hdr.w, sc8, mn8);
    const uint4 qa = *reinterpret_cast<const uint4 *>(wb + 16 + 32 * j), qb = *reinterpret_cast<const uint4 *>(wb + 16 + 32 * j + 16);
    const uint32_t qw[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const uint32_t w = live ? qw[t] : 0u;
        // byte b of w = lo_b | hi_b << 4
        const uint32_t o = (w & 0xFu) | ((w >> 16) & 0xFu) << 4 | ((w >> 4) & 0xFu) << 8 | ((w >> 20) & 0xFu) << 12 | ((w >> 8) & 0xFu) << 16 |
                           ((w >> 24) & 0xFu) << 20 | ((w >> 12) & 0xFu) << 24 | ((w >> 28) & 0xFu) << 28;
        Bq[((tile * 4 + (t >> 1)) * 64 + h * 32 + nin) * 4 + (t & 1) * 2 + p] = o;
    }
    {
        const _Float16 s0 = (_Float16)(live ? (float)byte_of(sc8, 2 * j) : 0.0f), s1 = (_Float16)(live ? (float)byte_of(sc8, 2 * j + 1) : 0.0f);
        Bs[(tile * 64 + h * 32 + nin) * 2 + p] = (uint32_t)__builtin_bit_cast(uint16_t, s0) | (uint32_t)__builtin_bit_cast(uint16_t, s1) << 16;
    }
    // mins of u = j: (mn[2u], mn[2u], mn[2u+1], mn[2u+1]); scales of the block by j == 0
    v4h mo;
    mo[0] = mo[1] = (_Float16)(live ? (float)byte_of(mn8, 2 * j) : 0.0f);
    mo[2] = mo[3] = (_Float16)(live ? (float)byte_of(mn8, 2 * j + 1) : 0.0f);
    Bm[(tile * 4 + j) * 32 + nin] = mo;
    if (j == 0) Bd[tile * 32 + nin] = live ? make_float2(h2f((uint16_t)(hdr.x & 0xffff)), h2f((uint16_t)(hdr.x >> 16))) : make_float2(0.0f, 0.0f);
}

__global__ __launch_bounds__(256) void q8k_prepack_kernel(const int8_t *__restrict__ xqs, const float *__restrict__ xd, const int16_t *__restrict__ xbsums,
                                                          uint8_t *__restrict__ out, int M, int nb) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int j = (int)(gid & 3);
    const int64_t mi = gid >> 2;
    const int i = (int)(mi % nb);
    const int m = (int)(mi / nb);
    if (m >= ((M + 31) / 32) * 32) return;
    const size_t tb = q4kp_tile_blocks(M, nb);
    v8h *Ax = reinterpret_cast<v8h *>(out);
    v4h *Am = reinterpret_cast<v4h *>(out + tb * Q4KP_W_PER_BLK);
    float *Ad = reinterpret_cast<float *>(out + tb * (Q4KP_W_PER_BLK + Q4KP_M_PER_BLK));
    const size_t tile = (size_t)(m >> 5) * nb + i;
    const int min_ = m & 31, p = j >> 1, h = j & 1;
    const bool live = m < M;
    const int8_t *xb = xqs + ((int64_t)(live ? m : 0) * nb + i) * 256 + 64 * j;
    const uint4 la = *reinterpret_cast<const uint4 *>(xb), lb = *reinterpret_cast<const uint4 *>(xb + 16);
    const uint4 ha = *reinterpret_cast<const uint4 *>(xb + 32), hb = *reinterpret_cast<const uint4 *>(xb + 48);
    const uint32_t lo[8] = {la.x, la.y, la.z, la.w, lb.x, lb.y, lb.z, lb.w}, hi[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        v8h o;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            o[bb] = (_Float16)(live ? (float)(int8_t)((lo[t] >> (8 * bb)) & 0xff) : 0.0f);
            o[4 + bb] = (_Float16)(live ? (float)(int8_t)((hi[t] >> (8 * bb)) & 0xff) : 0.0f);
        }
        Ax[((tile * 8 + t) * 2 + p) * 64 + h * 32 + min_] = o;
    }
    const int16_t *bs = xbsums + ((int64_t)(live ? m : 0) * nb + i) * 16 + 4 * j;
    const int q0 = live ? (int)bs[0] + (int)bs[1] : 0, q1 = live ? (int)bs[2] + (int)bs[3] : 0;
    v4h mo;
    mo[0] = (_Float16)(float)(q0 & ~1); mo[1] = (_Float16)(float)(q0 & 1);
    mo[2] = (_Float16)(float)(q1 & ~1); mo[3] = (_Float16)(float)(q1 & 1);
    Am[(tile * 4 + j) * 32 + min_] = mo;
    if (j == 0) Ad[tile * 32 + min_] = live ? xd[(int64_t)m * nb + i] : 0.0f;
}

// Workgroup = 256 threads = one 32 (m) x 64 (n) tile: two 32 x 32 MFMA tiles (nw = wid >> 1), TWO waves per tile -- the 12 chains of a tile would
// fill a wave's registers (and spill into AGPRs, which the VALU chain step cannot address), so wave ch of a tile owns classes {2ch, 2ch+1, 4+2ch, 5+2ch}
// and mins lanes {2ch, 2ch+1} (96 accumulation registers), and the pair meets once, after the K loop, for the reference's final adds
// ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)), (m0+m2)+(m1+m3).  A workgroup keeps one wave per SIMD and 64 KiB of LDS, so TWO workgroups share a CU: they
// are not in step with each other, and one's LDS reads / VALU chain steps / prologue / epilogue run under the other's MFMAs and operand stream
// (the 512-thread, 64 x 64 form of round 1 had all eight waves in lock step behind one barrier: every phase of a half-step exposed, 1 workgroup per CU).
// Operands travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: the packed fragments are already lane-linear, one 1-KiB fragment = one wave
// instruction) into a ring of four half-super-block slots, requested three half-steps ahead of the MFMAs that read them (counted s_waitcnt vmcnt(N),
// raw s_barrier; the DMA never passes through VGPRs).  A half-step = 4 column classes; every wave issues exactly four DMAs for it.
// Slot = 16 fragments of 1 KiB, fragment c written by wave c & 3 as its DMA number c >> 2:
//   c 0..7    activation fragments (class c >> 1 of the half, k-half p = c & 1), fp16
//   c 8..11   weight nibbles: n-tile (c - 8) >> 1, class pair (c - 8) & 1        (4 B per lane and fragment; expanded in registers)
//   c 12..14  even half: [Ad | Bd n0 | Bd n1], [Bs n0 | Bs n1], -- ;   odd half: Am, Bm n0, Bm n1
//   c 15      unused (keeps the DMA count per wave uniform)
constexpr int GQ_SLOT = 16384, GQ_SLOTS = 4, GQ_LDS = GQ_SLOT * GQ_SLOTS;
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
// (q & mask) | magic in one VALU instruction (the compiler prefers v_and + v_or with literals; gfx9 VOP3 reads one SGPR, so the magic sits in a VGPR)
__device__ __forceinline__ uint32_t and_or(uint32_t q, uint32_t mask_s, uint32_t magic_v) {
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(q), "s"(mask_s), "v"(magic_v));
    return r;
}
// one dword of re-ordered nibbles -> the 8 fp16 values (lo0..lo3) * s[0], (hi0..hi3) * s[1] of an MFMA B fragment.  0x6400 | n = 1024 + n and
// 0x5400 | n << 4 = 64 + n as fp16; fma(1024 + n, s, -1024 s) = n s exactly (one rounding of an exactly representable value, n s <= 945).
__device__ __forceinline__ v8h q4_expand(uint32_t q, v2h s, v2h o1024, v2h o64, uint32_t m0f, uint32_t mf0, uint32_t k1024, uint32_t k64) {
    const uint32_t t = q >> 8;
    const v2h d0 = __builtin_bit_cast(v2h, and_or(q, m0f, k1024)), d1 = __builtin_bit_cast(v2h, and_or(q, mf0, k64));
    const v2h d2 = __builtin_bit_cast(v2h, and_or(t, m0f, k1024)), d3 = __builtin_bit_cast(v2h, and_or(t, mf0, k64));
    const v2h w0 = __builtin_elementwise_fma(d0, (v2h){s[0], s[0]}, (v2h){o1024[0], o1024[0]});
    const v2h w1 = __builtin_elementwise_fma(d1, (v2h){s[0], s[0]}, (v2h){o64[0], o64[0]});
    const v2h w2 = __builtin_elementwise_fma(d2, (v2h){s[1], s[1]}, (v2h){o1024[1], o1024[1]});
    const v2h w3 = __builtin_elementwise_fma(d3, (v2h){s[1], s[1]}, (v2h){o64[1], o64[1]});
    return (v8h){w0[0], w0[1], w1[0], w1[1], w2[0], w2[1], w3[0], w3[1]};
}
#if defined(MLLM_HIP_STAMPS)   // diagnosis build (scratch/stamps.sh): per-wave shader-clock stamps of the first GQ_ST_HS half-steps, kept in LDS, dumped by every 37th workgroup
constexpr int GQ_ST_HS = 12, GQ_ST_N = 8, GQ_ST_WG = 64;
__device__ unsigned long long g_gq_stamps[GQ_ST_WG * (4 * GQ_ST_HS * GQ_ST_N + 8)];
#define GQS(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0 && st_hs < GQ_ST_HS) st_lds[(wid * GQ_ST_HS + st_hs) * GQ_ST_N + (i)] = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GQS(i) do { } while (0)
#endif
// diagnosis builds (scratch/gemm_variants.sh): each switch removes one component of the half-step; the results are wrong, only the time is read
#if defined(GQ_NO_RETIRE)
constexpr int GQ_RET = 1;
#else
constexpr int GQ_RET = 16;
#endif
#if defined(GQ_NO_MFMA)
#define GQ_MFMA(a, b, c) (c)
#else
#define GQ_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif
#if defined(GQ_NO_EXPAND)
#define GQ_EXPAND(q, s, o1, o2) __builtin_bit_cast(v8h, (u32x4){q, q, q, q})
#else
#define GQ_EXPAND(q, s, o1, o2) q4_expand(q, s, o1, o2, m0f, mf0, k1024, k64)
#endif
__global__ __launch_bounds__(256, 2) void gemm_q4k_kernel(const uint8_t *__restrict__ Wp, const uint8_t *__restrict__ Xp, const float *__restrict__ bias,
                                                          void *__restrict__ y, int y_f16, int64_t ldy, const float *__restrict__ residual, int M, int N,
                                                          int nb, int m_fastest) {
    extern __shared__ __attribute__((aligned(16))) char ring[];
    // Tile of this workgroup.  Workgroups go to the XCDs round-robin by their linear id, and every XCD's L2 has to pull what its workgroups read: with the
    // n-tile pair as the fastest index an XCD sees 1/8 of the weights and ALL activations, with the m-tile as the fastest 1/8 of the activations and all
    // weights -- the launcher picks the order that re-reads the smaller operand eight times (FETCH_SIZE of the ViT fc2: 95 MB per launch with its 11 MB of
    // activations pulled by all eight L2s).
    const int GXt = (N + 63) / 64, MTt = (M + 31) / 32;
    const int tile_x = m_fastest ? (int)blockIdx.x / MTt : (int)blockIdx.x % GXt, tile_y = m_fastest ? (int)blockIdx.x % MTt : (int)blockIdx.x / GXt;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: the DMA bookkeeping below stays scalar
#if defined(MLLM_HIP_STAMPS)
    __shared__ unsigned long long st_lds[4 * GQ_ST_HS * GQ_ST_N];
    int st_hs = 0;
    unsigned long long st_rt[6];
    st_rt[0] = __builtin_amdgcn_s_memrealtime();
    st_rt[2] = 0;
    for (int i = threadIdx.x; i < 4 * GQ_ST_HS * GQ_ST_N; i += 256) st_lds[i] = 0;
#endif
#if defined(GQ_STAGGER)   // diagnosis: the second workgroup of every CU starts GQ_STAGGER x 64 cycles late (first wave of 512 workgroups only)
    if ((((int)blockIdx.x >> 8) & 1) && (int)blockIdx.x < 512) {
#pragma unroll 1
        for (int i = 0; i < GQ_STAGGER; i += 100) __builtin_amdgcn_s_sleep(100);
    }
#endif
    const int col = lane & 31, h = lane >> 5;
    const int NT = (N + 31) / 32;
    const int nw = wid >> 1, ch = wid & 1;
    const int mt = tile_y, nt = tile_x * 2 + nw;          // mt < ceil(M / 32) by the grid
    const bool active = nt < NT;
    const size_t tbw = q4kp_tile_blocks(N, nb), tbx = q4kp_tile_blocks(M, nb);
    // DMA sources (n-tiles clamped: a workgroup at the edge copies a valid tile twice and does not use the copy)
    const int nts0 = min(tile_x * 2, NT - 1), nts1 = min(tile_x * 2 + 1, NT - 1);
    const uint8_t *XpM = Xp + tbx * Q4KP_W_PER_BLK, *XpD = XpM + tbx * Q4KP_M_PER_BLK;
    const uint8_t *WpS = Wp + tbw * Q4KW_Q_PER_BLK, *WpM = WpS + tbw * Q4KW_S_PER_BLK, *WpD = WpM + tbw * Q4KW_M_PER_BLK;
    const unsigned ring0 = (unsigned)(size_t)ring;   // LDS byte address of the ring
    // A lone wave issues about one instruction per 4-8 cycles whatever its kind, so everything about the stream is set up once: this wave's two
    // activation fragments come from sa0 / sa1 + voffA (8 KiB further per half-step), its nibble fragment from sq + voffB (2 KiB further), its aux
    // fragment from per-lane pointers that advance by a per-lane stride per block.  One asm block per half-step (M0 saved and restored once).
    auto uniform_ptr = [](const uint8_t *ptr) __attribute__((always_inline)) {
        const uint32_t plo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)ptr), phi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)ptr >> 32));
        return reinterpret_cast<const uint8_t *>(((uint64_t)phi << 32) | (uint64_t)plo);   // (readfirstlane returns int: no sign extension into the high half)
    };
    const uint8_t *sa0 = uniform_ptr(Xp + (size_t)mt * nb * Q4KP_W_PER_BLK + (size_t)wid * 1024);
    const uint8_t *sa1 = uniform_ptr(Xp + (size_t)mt * nb * Q4KP_W_PER_BLK + (size_t)(4 + wid) * 1024);
    const uint8_t *sq = uniform_ptr(Wp + (size_t)(nw ? nts1 : nts0) * nb * Q4KW_Q_PER_BLK + (size_t)ch * 1024);
    const uint8_t *pe = Xp + lane * 16, *po = Xp + lane * 16;     // aux sources: even / odd half (lanes without a job re-read a valid address into an unused KiB)
    unsigned se = 0, so = 0;
    if (wid == 0) {
        pe = lane < 8    ? XpD + (size_t)mt * nb * 128 + lane * 16
             : lane < 24 ? WpD + (size_t)nts0 * nb * 256 + (lane - 8) * 16
             : lane < 40 ? WpD + (size_t)nts1 * nb * 256 + (lane - 24) * 16
                         : Xp + lane * 16;
        se = lane < 8 ? 128u : (lane < 40 ? 256u : 0u);
        po = XpM + (size_t)mt * nb * 1024 + lane * 16;
        so = 1024u;
    } else if (wid == 1) {
        pe = WpS + (size_t)(lane < 32 ? nts0 : nts1) * nb * 512 + (lane & 31) * 16;
        se = 512u;
        po = WpM + (size_t)nts0 * nb * 1024 + lane * 16;
        so = 1024u;
    } else if (wid == 2) {
        po = WpM + (size_t)nts1 * nb * 1024 + lane * 16;
        so = 1024u;
    }
    unsigned voffA = (unsigned)lane * 16u, voffB = (unsigned)lane * 16u;
    unsigned islot = ring0;                   // LDS address of the slot the next issue fills
    auto issue = [&](int hb) __attribute__((always_inline)) {
        const unsigned d0 = __builtin_amdgcn_readfirstlane(islot + (unsigned)wid * 1024u);
        const uint8_t *pa = hb == 0 ? pe : po;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\t"
                     "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                     "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %6\n\t"
                     "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voffA), "v"(voffB), "s"(d0), "s"(sa0), "s"(sa1), "s"(sq), "v"(pa) : "memory", "scc");
        voffA += 8192u;
        voffB += 2048u;
        islot = islot + (unsigned)GQ_SLOT == ring0 + (unsigned)GQ_LDS ? ring0 : islot + (unsigned)GQ_SLOT;
        if (hb == 0) pe += se; else po += so;
    };
    const int TS = 2 * nb;
    issue(0);                 // the first three half-steps are on their way before the accumulators are even cleared
    issue(1);                 // TS >= 2 always (nb >= 1)
    if (TS > 2) issue(0);
    __builtin_amdgcn_sched_barrier(0);
#if defined(MLLM_HIP_STAMPS)
    st_rt[1] = __builtin_amdgcn_s_memrealtime();
#endif
    v16f acc[4], accm[2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) accm[u][r] = 0.0f;
    v16f zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.0f;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // Software pipeline over the half-steps: a half-step requests its LDS operands, then *retires the previous half-step's class sums*
    // (the VALU chain step acc = fma(dd, c, acc), which covers the LDS latency), expands its nibbles, then issues its own MFMAs, whose results are
    // read one barrier later -- no wave waits on an MFMA it has just issued.  dd starts at 0 so the first retire is fma(0, 0, 0).
    float dd[16], dm[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dd[r] = dm[r] = 0.0f;
    v16f cp0 = zero, cp1 = zero;
    v2h sc[2] = {{0, 0}, {0, 0}}, o1024[2] = {{0, 0}, {0, 0}}, o64[2] = {{0, 0}, {0, 0}};   // this lane's sub-block scale pairs (p = 0, 1) of the block, and -1024 s, -64 s
    const uint32_t k1024 = 0x64006400u, k64 = 0x54005400u;
    const uint32_t m0f = __builtin_amdgcn_readfirstlane(0x000F000F), mf0 = __builtin_amdgcn_readfirstlane(0x00F000F0);
    // per-lane LDS read addresses inside slot 0; the compute slot's offset cycles through the ring
    const unsigned lA = ring0 + (unsigned)(ch * 4096 + lane * 16), lQ = ring0 + (unsigned)((8 + 2 * nw + ch) * 1024 + lane * 16);
    const unsigned lX = ring0 + (unsigned)(12288 + 16 * h), lW = ring0 + (unsigned)(12288 + 128 + nw * 256 + col * 8), lS = ring0 + (unsigned)(13312 + nw * 512 + lane * 8);
    const unsigned lAm = ring0 + (unsigned)(12288 + col * 8 + 2 * ch * 256), lBm = ring0 + (unsigned)(13312 + nw * 1024 + col * 8 + 2 * ch * 256);
    unsigned cslot = 0;
    // one half-step; HB = which half of the block, REM = half-steps still to come after this one, clipped to 3 (3 = steady state)
    auto step = [&](auto HB, auto REM) __attribute__((always_inline)) {
        constexpr int hb = decltype(HB)::value, rem = decltype(REM)::value;
        GQS(0);
        // this wave's DMA of the half-step has landed when at most the (up to two) younger half-steps are outstanding: 4 per half-step
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GQS(1);
#if !defined(GQ_NO_BAR)
        __builtin_amdgcn_s_barrier();      // everyone's part of the slot has landed; everyone is done with the previous slot
#endif
        asm volatile("" ::: "memory");
        GQS(2);
#if defined(MLLM_HIP_STAMPS)
        if (st_rt[2] == 0) st_rt[2] = __builtin_amdgcn_s_memrealtime();
#endif
#if !defined(GQ_NO_DMA)
        if (rem >= 3) issue(1 - hb);
#endif
        GQS(3);
        if (active) {
            typedef __attribute__((address_space(3))) const char *lds_cp;
            const unsigned cs = __builtin_amdgcn_readfirstlane(cslot);
            // LDS requests in the order of use: the 4 nibble dwords of this wave's fragments (+ the block's scale pairs), its 4 activation fragments
            // (classes 2ch, 2ch+1 of the half, k-halves 0 / 1), then the block scales (even half) or the mins operands (odd half)
            const u32x4 qv = *reinterpret_cast<__attribute__((address_space(3))) const u32x4 *>((lds_cp)(size_t)(lQ + cs));
            u32x2 sraw;
            if (hb == 0) sraw = *reinterpret_cast<__attribute__((address_space(3))) const u32x2 *>((lds_cp)(size_t)(lS + cs));
            const __attribute__((address_space(3))) v8h *A = reinterpret_cast<__attribute__((address_space(3))) const v8h *>((lds_cp)(size_t)(lA + cs));
            const v8h a0 = A[0], a2 = A[128], a1 = A[64], a3 = A[192];
            f32x2 dwv;
            f32x4 dx[4];
            v4h a4[2], b4[2];                                  // mins operands (k = 0..3 of the fragment; lanes 32..63 carry k = 8..15 = zeros)
            if (hb == 0) {
                dwv = *reinterpret_cast<__attribute__((address_space(3))) const f32x2 *>((lds_cp)(size_t)(lW + cs));
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) dx[g4] = *reinterpret_cast<__attribute__((address_space(3))) const f32x4 *>((lds_cp)(size_t)(lX + cs) + 32 * g4);   // rows 8 g4 + 4h + (0..3)
            } else {
                const __attribute__((address_space(3))) v4h *Am = reinterpret_cast<__attribute__((address_space(3))) const v4h *>((lds_cp)(size_t)(lAm + cs));
                const __attribute__((address_space(3))) v4h *Bm = reinterpret_cast<__attribute__((address_space(3))) const v4h *>((lds_cp)(size_t)(lBm + cs));
#pragma unroll
                for (int k = 0; k < 2; ++k) { a4[k] = Am[k * 32]; b4[k] = Bm[k * 32]; }   // all lanes read (no wait under a predicate)
            }
            GQS(4);
            // The matrix pipe runs 32 cycles per MFMA and takes 8 of them from the wave's issue; the VALU work of the half-step is spread BETWEEN the MFMAs
            // (sched_group_barrier below) so that the pipe runs under it: retire cp0 of the previous half-step | expand b0 | MFMA | retire cp1 | expand b2 |
            // MFMA | expand b1 | MFMA | expand b3 | MFMA | new block scales (even half) or the mins chain step (odd half).
            // (hb == 0 retires the odd half of the previous block, still under that block's dd.)
#pragma unroll
            for (int r = 0; r < GQ_RET; ++r) acc[2 * (1 - hb)][r] = __fmaf_rn(dd[r], cp0[r], acc[2 * (1 - hb)][r]);
            if (hb == 0) {
                const v2h n1024 = {(_Float16)-1024.0f, (_Float16)-1024.0f}, n64 = {(_Float16)-64.0f, (_Float16)-64.0f};
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const unsigned sv = p == 0 ? sraw.x : sraw.y;    // (bit_cast of a variably indexed vector element miscompiles with this clang: both p read element 0)
                    sc[p] = __builtin_bit_cast(v2h, sv);
                    o1024[p] = sc[p] * n1024;                        // exact: 1024 * 63 = 64512 < 65504
                    o64[p] = sc[p] * n64;
                }
            }
            v16f cm0, cm1;
            if (hb == 1) {
                v8h am[2], bm[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const u32x2 ua = __builtin_bit_cast(u32x2, a4[k]), ub = __builtin_bit_cast(u32x2, b4[k]);
                    const u32x4 wa = {h == 0 ? ua[0] : 0u, h == 0 ? ua[1] : 0u, 0u, 0u}, wb = {h == 0 ? ub[0] : 0u, h == 0 ? ub[1] : 0u, 0u, 0u};
                    am[k] = __builtin_bit_cast(v8h, wa); bm[k] = __builtin_bit_cast(v8h, wb);
                }
                cm0 = GQ_MFMA(am[0], bm[0], zero);
                cm1 = GQ_MFMA(am[1], bm[1], zero);
            }
            // fragment order (class k, k-half p): qv = (k0 p0, k0 p1, k1 p0, k1 p1)
            const v8h b0 = GQ_EXPAND(qv[0], sc[0], o1024[0], o64[0]);
            cp0 = GQ_MFMA(a0, b0, zero);
#pragma unroll
            for (int r = 0; r < GQ_RET; ++r) acc[2 * (1 - hb) + 1][r] = __fmaf_rn(dd[r], cp1[r], acc[2 * (1 - hb) + 1][r]);
            const v8h b2 = GQ_EXPAND(qv[2], sc[0], o1024[0], o64[0]);
            cp1 = GQ_MFMA(a2, b2, zero);
            const v8h b1 = GQ_EXPAND(qv[1], sc[1], o1024[1], o64[1]);
            cp0 = GQ_MFMA(a1, b1, cp0);
            const v8h b3 = GQ_EXPAND(qv[3], sc[1], o1024[1], o64[1]);
            cp1 = GQ_MFMA(a3, b3, cp1);
            if (hb == 0) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dd[4 * g4 + e] = dx[g4][e] * dwv[0];        // y.d * fp16(x.d)
                        dm[4 * g4 + e] = (-dx[g4][e]) * dwv[1];     // -y.d * fp16(x.dmin)
                    }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    accm[0][r] = __fmaf_rn(dm[r], cm0[r], accm[0][r]);
                    accm[1][r] = __fmaf_rn(dm[r], cm1[r], accm[1][r]);
                }
            }
        }
        GQS(6);
#if defined(MLLM_HIP_STAMPS)
        ++st_hs;
#endif
        cslot = cslot + (unsigned)GQ_SLOT == (unsigned)GQ_LDS ? 0u : cslot + (unsigned)GQ_SLOT;
    };
    using std::integral_constant;
#pragma unroll 1
    for (int i = 0; i + 2 < nb; ++i) {                    // blocks 0 .. nb-3: both halves in the steady state
        step(integral_constant<int, 0>{}, integral_constant<int, 3>{});
        step(integral_constant<int, 1>{}, integral_constant<int, 3>{});
    }
    if (nb >= 2) {                                        // block nb-2: its odd half has nothing left to request
        step(integral_constant<int, 0>{}, integral_constant<int, 3>{});
        step(integral_constant<int, 1>{}, integral_constant<int, 2>{});
    }
    step(integral_constant<int, 0>{}, integral_constant<int, 1>{});   // the last block
    step(integral_constant<int, 1>{}, integral_constant<int, 0>{});
    if (active) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[2][r] = __fmaf_rn(dd[r], cp0[r], acc[2][r]);
            acc[3][r] = __fmaf_rn(dd[r], cp1[r], acc[3][r]);
        }
    }
    // the pair meets through LDS (the ring is free now): wave ch finishes accumulator registers 8 ch .. 8 ch + 7 of the tile and hands the other
    // eight of its partial sums (a0+a4 | a2+a6, a1+a5 | a3+a7, m0 | m2, m1 | m3) to its partner -- IEEE adds commute, so which of the two waves
    // performs (a0+a4)+(a2+a6) does not matter.  Residual rows are requested together, ahead of the adds.
#if defined(MLLM_HIP_STAMPS)
    st_rt[3] = __builtin_amdgcn_s_memrealtime();
#endif
    float x[16], yv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { x[r] = acc[0][r] + acc[2][r]; yv[r] = acc[1][r] + acc[3][r]; }
    __syncthreads();
    float *xch = reinterpret_cast<float *>(ring) + (size_t)nw * 64 * 64 + lane;
    const int n = nt * 32 + col;
    auto finish = [&](auto R0) __attribute__((always_inline)) {
        constexpr int r0 = decltype(R0)::value, p0 = 8 - r0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = p0 + j;
            xch[r * 64] = x[r]; xch[(16 + r) * 64] = yv[r]; xch[(32 + r) * 64] = accm[0][r]; xch[(48 + r) * 64] = accm[1][r];
        }
        __syncthreads();
#if defined(MLLM_HIP_STAMPS)
        st_rt[4] = __builtin_amdgcn_s_memrealtime();
#endif
        if (!active || n >= N) return;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = r0 + j;
            const float hs = (x[r] + xch[r * 64]) + (yv[r] + xch[(16 + r) * 64]);
            v[j] = hs + ((accm[0][r] + xch[(32 + r) * 64]) + (accm[1][r] + xch[(48 + r) * 64]));
        }
        if (bias) {
            const float bv = bias[n];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] + bv;
        }
        const int mb = mt * 32 + 4 * h;
        if (y_f16) {
            uint16_t *yp = reinterpret_cast<uint16_t *>(y) + n;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = mb + ((r0 + j) & 3) + 8 * ((r0 + j) >> 2);
                if (m < M) yp[(int64_t)m * ldy] = f2h(v[j]);
            }
        } else {
            float *yp = reinterpret_cast<float *>(y) + n;
            if (residual) {
                float res[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) res[j] = residual[(int64_t)min(mb + ((r0 + j) & 3) + 8 * ((r0 + j) >> 2), M - 1) * ldy + n];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = v[j] + res[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = mb + ((r0 + j) & 3) + 8 * ((r0 + j) >> 2);
                if (m < M) yp[(int64_t)m * ldy] = v[j];
            }
        }
    };
    if (ch == 0) finish(std::integral_constant<int, 0>{}); else finish(std::integral_constant<int, 8>{});
#if defined(MLLM_HIP_STAMPS)
    {
        const int wg = blockIdx.x;
        if (wg % 37 == 0 && wg / 37 < GQ_ST_WG) {
            unsigned long long *dst = g_gq_stamps + (size_t)(wg / 37) * (4 * GQ_ST_HS * GQ_ST_N + 8);
            for (int i = threadIdx.x; i < 4 * GQ_ST_HS * GQ_ST_N; i += 256) dst[i] = st_lds[i];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the output stores have been acknowledged
            st_rt[5] = __builtin_amdgcn_s_memrealtime();
            if (threadIdx.x == 0) {
                unsigned long long *tail = dst + 4 * GQ_ST_HS * GQ_ST_N;
                for (int i = 0; i < 6; ++i) tail[i] = st_rt[i];
                tail[6] = __builtin_amdgcn_s_getreg(63492); tail[7] = __builtin_amdgcn_s_getreg(63508);   // HW_ID, XCC_ID
            }
        }
    }
#endif
}

#if defined(MLLM_HIP_STAMPS)
extern "C" int mllm_hip_debug_read_gemm_stamps(unsigned long long *host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gq_stamps), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#endif

// ------------------------------------------------------------------------------------------------------------------
// fp32 GEMM y = x W^T (+bias) in vec_dot_fp32's order (VecDotFP32.cpp:31-58): 32 fp32 chains per output (chain c takes
// k = 32 s + c), folded sum0+sum2, sum1+sum3, +, then lanes l/l+4, l/l+1, l/l+2 -- xor 16, 8, 4, 1, 2 over the chain
// index -- then the K % 32 leftovers by fma in order.  32 lanes = the 32 chains of a TM x TN block of outputs.
// ------------------------------------------------------------------------------------------------------------------
constexpr int F32_TM = 8, F32_TN = 4;
#define SWZ_XOR(v, k) __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x1f | ((k) << 10)))
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float *__restrict__ W, const float *__restrict__ bias, const float *__restrict__ x,
                                                       float *__restrict__ y, int64_t ldy, int M, int N, int K) {
    const int c = threadIdx.x & 31, gi = threadIdx.x >> 5;
    const int m0 = blockIdx.y * (2 * F32_TM) + (gi >> 2) * F32_TM, n0 = blockIdx.x * (4 * F32_TN) + (gi & 3) * F32_TN;
    const float *xr[F32_TM], *wr[F32_TN];
#pragma unroll
    for (int i = 0; i < F32_TM; ++i) xr[i] = x + (int64_t)min(m0 + i, M - 1) * K;
#pragma unroll
    for (int j = 0; j < F32_TN; ++j) wr[j] = W + (int64_t)min(n0 + j, N - 1) * K;
    float acc[F32_TM][F32_TN];
#pragma unroll
    for (int i = 0; i < F32_TM; ++i)
#pragma unroll
        for (int j = 0; j < F32_TN; ++j) acc[i][j] = 0.0f;
    const int np = K & ~31;
    for (int k = c; k < np; k += 32) {
        float xv[F32_TM], wv[F32_TN];
#pragma unroll
        for (int i = 0; i < F32_TM; ++i) xv[i] = xr[i][k];
#pragma unroll
        for (int j = 0; j < F32_TN; ++j) wv[j] = wr[j][k];
#pragma unroll
        for (int i = 0; i < F32_TM; ++i)
#pragma unroll
            for (int j = 0; j < F32_TN; ++j) acc[i][j] = __fmaf_rn(wv[j], xv[i], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < F32_TM; ++i)
#pragma unroll
        for (int j = 0; j < F32_TN; ++j) {
            float v = acc[i][j];
            v = v + SWZ_XOR(v, 16);
            v = v + SWZ_XOR(v, 8);
            v = v + SWZ_XOR(v, 4);
            v = v + SWZ_XOR(v, 1);
            v = v + SWZ_XOR(v, 2);
            acc[i][j] = v;
        }
    for (int k = np; k < K; ++k) {
        float xv[F32_TM], wv[F32_TN];
#pragma unroll
        for (int i = 0; i < F32_TM; ++i) xv[i] = xr[i][k];
#pragma unroll
        for (int j = 0; j < F32_TN; ++j) wv[j] = wr[j][k];
#pragma unroll
        for (int i = 0; i < F32_TM; ++i)
#pragma unroll
            for (int j = 0; j < F32_TN; ++j) acc[i][j] = __fmaf_rn(wv[j], xv[i], acc[i][j]);
    }
    // lane c stores output (i, j) = (c / TN, c % TN) of the block
#pragma unroll
    for (int i = 0; i < F32_TM; ++i)
#pragma unroll
        for (int j = 0; j < F32_TN; ++j)
            if (c == i * F32_TN + j && m0 + i < M && n0 + j < N) y[(int64_t)(m0 + i) * ldy + n0 + j] = bias ? acc[i][j] + bias[n0 + j] : acc[i][j];
}

// The same GEMM on the matrix cores.  vec_dot_fp32 keeps 32 chains per output (chain c takes k = c, c + 32, ...), and v_mfma_f32_16x16x4_f32 is an exact fp32 fma chain
// over its four k slots in slot order (scratch/mfma/test16.hip) -- so ONE accumulator tile per chain, fed k = 32 (4j + slot) + c at step j, advances chain c of 16 x 16
// outputs by four links per instruction, in the reference's order.  A lane (row l & 15, slot l >> 4) needs the 32 consecutive floats x[row][128 j + 32 slot ..] and
// w[row][...] of a step: they are the operands of the step's 32 MFMAs (one per chain).  Read straight from global memory each of a lane's eight 16-byte loads touches 64
// different lines (94 us for the patch embedding, the L1's tag rate), so a step's 128 columns of the workgroup's 32 + 32 rows are staged through LDS (coalesced 512-byte
// row pieces in, eight ds_read_b128 per lane and operand out, row pitch 132 floats: conflict-free), double-buffered, one barrier per step.  Wave = one 16 x 16 tile with
// 32 x 4 accumulator registers; workgroup = 2 x 2 tiles.  Behind the MFMA steps: the K / 32 % 4 whole links that are left (VALU, in the C layout), the chain fold in the
// order of the swizzle tree above (16, 8, 4, 1, 2), the K % 32 ordered leftovers, the bias.  Needs K % 4 == 0 and 16-byte aligned rows.
typedef float f32mm_v4 __attribute__((ext_vector_type(4)));
constexpr int F32MM_BLK = 1024 + 16;      // bytes of one staged pair of rows: 2 x 128 floats, their 128-byte lines interleaved (piece f of row 2p + b at slot 16 (f >> 3) + 8 b + (f & 7)), + 16
__device__ __forceinline__ void f32mm_glds16(const void *gsrc, unsigned lds_dst_in) {
    unsigned keep;
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane(lds_dst_in);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ __launch_bounds__(256, 2) void gemm_f32_mfma_kernel(const float *__restrict__ W, const float *__restrict__ bias, const float *__restrict__ x,
                                                               float *__restrict__ y, int64_t ldy, int M, int N, int K) {
    // staged rows 0..31: x rows m0b .. m0b+31, rows 32..63: w rows n0b .. n0b+31.  LDS-DMA lands a wave instruction's 64 x 16 bytes contiguously, so an instruction
    // takes the two rows of a pair with their 128-byte lines interleaved (eight adjacent lanes = one whole line) and the pairs are 16 bytes apart on top: the 16 rows a
    // ds_read_b128 of one column piece meets then lie in 16 different groups of four banks (pair p at 4p, the odd row 32 further: conflict-free).  No staging registers
    // (they would spill under 128 accumulators).  With 16-byte interleaving (adjacent lanes on alternating rows) the launch took 62 us: the loads did not coalesce.
    __shared__ __attribute__((aligned(16))) char stage[2][32 * F32MM_BLK];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, q = lane >> 4;
    const int m0b = blockIdx.y * 32, n0b = blockIdx.x * 32;
    const int m0 = m0b + (wid >> 1) * 16, n0 = n0b + (wid & 1) * 16;
    f32mm_v4 acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = f32mm_v4{0.0f, 0.0f, 0.0f, 0.0f};
    const int ni = K >> 5, nj = ni >> 2;
    const float *src[8];      // wave w lands pairs 8w .. 8w+7: lane l takes piece 8 (l >> 4) + (l & 7) of row 2 pair + ((l >> 3) & 1)
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int row = 2 * (8 * wid + g) + ((lane >> 3) & 1);
        src[g] = (row < 32 ? x + (int64_t)min(m0b + row, M - 1) * K : W + (int64_t)min(n0b + row - 32, N - 1) * K) + 4 * (8 * (lane >> 4) + (lane & 7));
    }
    const unsigned st0 = (unsigned)(size_t)&stage[0][0];
    auto land = [&](int j) {
        const unsigned dst = st0 + (unsigned)((j & 1) * 32 * F32MM_BLK + 8 * wid * F32MM_BLK);
#pragma unroll
        for (int g = 0; g < 8; ++g) f32mm_glds16(src[g] + 128 * j, dst + (unsigned)(g * F32MM_BLK));
    };
    if (nj > 0) land(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int rx = (wid >> 1) * 16 + l15, rw = 32 + (wid & 1) * 16 + l15;
    const int xoff = (rx >> 1) * F32MM_BLK + (rx & 1) * 128 + 256 * q, woff = (rw >> 1) * F32MM_BLK + (rw & 1) * 128 + 256 * q;      // piece 8q + e at + 16 e bytes
    for (int j = 0; j < nj; ++j) {
        const char *cur = stage[j & 1];
        if (j + 1 < nj) land(j + 1);
        // operands in two halves of 16 chains (64 operand registers on top of 128 accumulators do not fit two waves per SIMD)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            float4 xc[4], wc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                xc[g] = *reinterpret_cast<const float4 *>(cur + xoff + 16 * (4 * hf + g));
                wc[g] = *reinterpret_cast<const float4 *>(cur + woff + 16 * (4 * hf + g));
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = 16 * hf + 4 * g;
                acc[c0 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[g].x, wc[g].x, acc[c0 + 0], 0, 0, 0);
                acc[c0 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[g].y, wc[g].y, acc[c0 + 1], 0, 0, 0);
                acc[c0 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[g].z, wc[g].z, acc[c0 + 2], 0, 0, 0);
                acc[c0 + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[g].w, wc[g].w, acc[c0 + 3], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();      // step j + 1 has landed; nobody still reads step j's buffer when step j + 2 is requested into it
    }
    if (m0 >= M || n0 >= N) return;      // (after the last barrier: a tile wholly outside still helped staging)
    // from here on in the C layout: this lane owns outputs (m0 + 4 q + r, n0 + l15), r = 0..3
    const float *wn = W + (int64_t)min(n0 + l15, N - 1) * K;
    const float *xm[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) xm[r] = x + (int64_t)min(m0 + 4 * q + r, M - 1) * K;
    // (both tails request all their values first and then walk the links from registers: a loop of load -> fma pays a memory round trip per link -- with the
    // patch embedding's 24 leftover columns that was 50 of the launch's 62 us)
    for (int i = 4 * nj; i < ni; ++i) {      // whole links past the last full MFMA step
        float4 wt[8], xt[4][8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            wt[g] = *reinterpret_cast<const float4 *>(wn + 32 * i + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) xt[r][g] = *reinterpret_cast<const float4 *>(xm[r] + 32 * i + 4 * g);
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[4 * g + 0][r] = __fmaf_rn(wt[g].x, xt[r][g].x, acc[4 * g + 0][r]);
                acc[4 * g + 1][r] = __fmaf_rn(wt[g].y, xt[r][g].y, acc[4 * g + 1][r]);
                acc[4 * g + 2][r] = __fmaf_rn(wt[g].z, xt[r][g].z, acc[4 * g + 2][r]);
                acc[4 * g + 3][r] = __fmaf_rn(wt[g].w, xt[r][g].w, acc[4 * g + 3][r]);
            }
        }
    }
    // the fold's first level, then the requests for the K % 32 leftover columns (a multiple of four; with all 32 accumulator tiles still live they would not fit), then the rest
    f32mm_v4 s16[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) s16[c] = acc[c] + acc[c + 16];
#pragma unroll
    for (int c = 0; c < 16; ++c) asm volatile("" : "+v"(s16[c]));
    const int nl4 = (K - 32 * ni) >> 2;
    float4 wl[7], xl[4][7];
    if (nl4 > 0) {
#pragma unroll
        for (int g = 0; g < 7; ++g) {
            const int gg = min(g, nl4 - 1);
            wl[g] = *reinterpret_cast<const float4 *>(wn + 32 * ni + 4 * gg);
#pragma unroll
            for (int r = 0; r < 4; ++r) xl[r][g] = *reinterpret_cast<const float4 *>(xm[r] + 32 * ni + 4 * gg);
        }
    }
    f32mm_v4 v;
    {
        f32mm_v4 s8[8], s4[4];
#pragma unroll
        for (int c = 0; c < 8; ++c) s8[c] = s16[c] + s16[c + 8];
#pragma unroll
        for (int c = 0; c < 4; ++c) s4[c] = s8[c] + s8[c + 4];
        v = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    if (nl4 > 0) {
#pragma unroll
        for (int g = 0; g < 7; ++g) {
            if (g < nl4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = __fmaf_rn(wl[g].x, xl[r][g].x, v[r]);
                    v[r] = __fmaf_rn(wl[g].y, xl[r][g].y, v[r]);
                    v[r] = __fmaf_rn(wl[g].z, xl[r][g].z, v[r]);
                    v[r] = __fmaf_rn(wl[g].w, xl[r][g].w, v[r]);
                }
            }
        }
    }
    if (n0 + l15 < N) {
        const float bv = bias ? bias[n0 + l15] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 4 * q + r;
            if (m < M) y[(int64_t)m * ldy + n0 + l15] = bias ? v[r] + bv : v[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// A7, eager-attention form: gemm_fp32 / gemm_fp32_fp16 (compute/GemmFp.hpp:104-150, :233-283) per head on BHSD operands.  One thread per output element (the
// reference's order leaves every element its own chain over K); 64 consecutive columns per wave, so the row of B is one coalesced read and A's value a broadcast.
// Non-default path (attn_implementation = "eager"): written for parity, not tuned.
// ------------------------------------------------------------------------------------------------------------------
template <bool B16>
__global__ __launch_bounds__(256) void gemm_f32_bhsd_kernel(const float *__restrict__ a, const void *__restrict__ bv, float *__restrict__ c, int M, int N, int K) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6), h = blockIdx.z;
    if (i >= M || j >= N) return;
    const float *A = a + ((int64_t)h * M + i) * K;
    const float *B32 = reinterpret_cast<const float *>(bv) + (int64_t)h * K * N + j;
    const uint16_t *B16p = reinterpret_cast<const uint16_t *>(bv) + (int64_t)h * K * N + j;
    auto bval = [&](int k) -> float { return B16 ? h2f(B16p[(int64_t)k * N]) : B32[(int64_t)k * N]; };
    float acc = 0.0f;
    if (i < M - M % 8 && j < N - N % 8) {      // a full 8 x 8 tile: the AVX micro-kernel's chain, carried through C across the K blocks
        for (int k = 0; k < K; ++k) acc = __fmaf_rn(A[k], bval(k), acc);
    } else {                                   // edge tiles: the scalar path, one partial sum per 256-wide K block
        for (int k0 = 0; k0 < K; k0 += 256) {
            float sum = 0.0f;
            const int k1 = min(k0 + 256, K);
            for (int k = k0; k < k1; ++k) sum = __fmaf_rn(A[k], bval(k), sum);
            acc = __fadd_rn(acc, sum);
        }
    }
    c[((int64_t)h * M + i) * N + j] = acc;
}

static int launch_gemv_q4k(const void *W, const float *bias, const int8_t *xqs, const float *xd, const int16_t *xbsums, void *y, int y_f16,
                           const float *residual, int N, int K, hipStream_t st, int M = 1, int64_t ldy = 0) {
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
                           xbsums, y, y_f16, residual, N, nb, rows_per_wave, M, ldy);                                           \
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

extern "C" size_t mllm_hip_q4k_prepack_bytes(int rows, int K) { return K % 256 ? 0 : q4kp_bytes(rows, K); }     // activation side (fp16 fragments)
extern "C" size_t mllm_hip_q4k_wpack_bytes(int N, int K) { return K % 256 ? 0 : q4kw_bytes(N, K); }           // weight side (nibbles + scales)
extern "C" int mllm_hip_q4k_prepack(const void *W, int N, int K, void *out, void *stream) {
    if (K % 256 != 0 || K <= 0 || N <= 0) return MLLM_HIP_ERR_SHAPE;
    const int nb = K / 256;
    const int64_t threads = (int64_t)((N + 31) / 32) * 32 * nb * 4;
    hipLaunchKernelGGL(q4k_prepack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, as_stream(stream), (const uint8_t *)W, (uint8_t *)out, N, nb);
    return MH_LAUNCH_OK("q4k_prepack");
}
static int launch_gemm_packed(const void *Wpacked, const float *bias, const void *xpack, void *y, int y_dtype, int64_t ldy, const float *residual, int M,
                              int N, int K, hipStream_t st) {
    const int nb = K / 256;
    // the attribute is per device: one bit per device id, set once (a second mllm_hip_init(device) in the same process must not inherit device 0's flag)
    static std::atomic<uint64_t> attr_done{0};
    int dev = 0;
    MH_CHECK(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        MH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_q4k_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, GQ_LDS));
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    // 32 x 64 workgroup tiles, two workgroups per CU (64 KiB of LDS and one wave per SIMD each)
    const int64_t ntiles = (int64_t)((N + 63) / 64) * ((M + 31) / 32);
    if (ntiles > 0x7fffffff) return MLLM_HIP_ERR_SHAPE;
    // the operand that all eight XCD L2s re-read should be the smaller one: activations cost 2.16 B per element (fp16 fragments), weights 0.72 B
    const int order_env = option(OPT_GEMM_ORDER);      // 0 / 1 force n- / m-fastest (measurement)
    const int m_fastest = order_env >= 0 ? order_env : (q4kp_bytes(M, K) > q4kw_bytes(N, K) ? 1 : 0);
    hipLaunchKernelGGL(gemm_q4k_kernel, dim3((unsigned)ntiles), dim3(256), GQ_LDS, st, (const uint8_t *)Wpacked, (const uint8_t *)xpack, bias, y, y_dtype == MLLM_HIP_F16,
                       ldy, residual, M, N, nb, m_fastest);
    return MH_LAUNCH_OK("gemm_q4k");
}
// GEMM on pre-packed weights and activations already in packed form (mllm_hip_quantize_q8k_packed / _rmsnorm_packed / _layernorm_packed)
extern "C" int mllm_hip_linear_q4kp_packed(const void *Wpacked, const float *bias, const void *xpack, void *y, int y_dtype, int64_t ldy,
                                           const float *residual, int M, int N, int K, void *stream) {
    if (K % 256 != 0 || K <= 0 || N <= 0) return MLLM_HIP_ERR_SHAPE;
    if (y_dtype != MLLM_HIP_F32 && y_dtype != MLLM_HIP_F16) return MLLM_HIP_ERR_DTYPE;
    if (residual && y_dtype != MLLM_HIP_F32) return MLLM_HIP_ERR_DTYPE;      // the residual is an fp32 [M][N] buffer with y's pitch: it exists for fp32 outputs only
    if (M <= 0) return MLLM_HIP_OK;
    if (!Wpacked || !xpack) return MLLM_HIP_ERR_ARG;
    return launch_gemm_packed(Wpacked, bias, xpack, y, y_dtype, ldy, residual, M, N, K, as_stream(stream));
}
// GEMM on pre-packed weights; xpack = scratch of mllm_hip_q4k_prepack_bytes(M, K) bytes for the activation side
extern "C" int mllm_hip_linear_q4kp_q8k(const void *Wpacked, const float *bias, const int8_t *xqs, const float *xd, const int16_t *xbsums, void *xpack,
                                        void *y, int y_dtype, int64_t ldy, const float *residual, int M, int N, int K, void *stream) {
    if (K % 256 != 0 || K <= 0 || N <= 0) return MLLM_HIP_ERR_SHAPE;
    if (y_dtype != MLLM_HIP_F32 && y_dtype != MLLM_HIP_F16) return MLLM_HIP_ERR_DTYPE;
    if (residual && y_dtype != MLLM_HIP_F32) return MLLM_HIP_ERR_DTYPE;      // the residual is an fp32 [M][N] buffer with y's pitch: it exists for fp32 outputs only
    if (M <= 0) return MLLM_HIP_OK;
    if (!Wpacked || !xpack) return MLLM_HIP_ERR_ARG;
    const int nb = K / 256;
    hipStream_t st = as_stream(stream);
    const int64_t threads = (int64_t)((M + 31) / 32) * 32 * nb * 4;
    hipLaunchKernelGGL(q8k_prepack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, xqs, xd, xbsums, (uint8_t *)xpack, M, nb);
    int rc = MH_LAUNCH_OK("q8k_prepack");
    if (rc) return rc;
    return launch_gemm_packed(Wpacked, bias, xpack, y, y_dtype, ldy, residual, M, N, K, st);
}

extern "C" int mllm_hip_linear_q4k_q8k(const void *W, const float *bias, const int8_t *xqs, const float *xd, const int16_t *xbsums, void *y,
                                       int y_dtype, int64_t ldy, const float *residual, int M, int N, int K, void *stream) {
    if (K % 256 != 0 || K <= 0 || N <= 0) return MLLM_HIP_ERR_SHAPE;
    if (y_dtype != MLLM_HIP_F32 && y_dtype != MLLM_HIP_F16) return MLLM_HIP_ERR_DTYPE;
    if (residual && y_dtype != MLLM_HIP_F32) return MLLM_HIP_ERR_DTYPE;      // the residual is an fp32 [M][N] buffer with y's pitch: it exists for fp32 outputs only
    if (M <= 0) return MLLM_HIP_OK;
    const int y_f16 = y_dtype == MLLM_HIP_F16;
    hipStream_t st = as_stream(stream);
    if (M < 16) {
        // one launch: every weight row is fetched once and meets all M activation rows in registers (gemv_q4k_kernel's m loop)
        return launch_gemv_q4k(W, bias, xqs, xd, xbsums, y, y_f16, residual, N, K, st, M, ldy);
    }
    // raw Q4_K blocks with M >= 16: pack both sides into stream-ordered scratch, then the packed GEMM (callers that keep the
    // weights resident pre-pack once with mllm_hip_q4k_prepack and call mllm_hip_linear_q4kp_q8k)
    void *wp = nullptr, *xp = nullptr;
    MH_CHECK(hipMallocAsync(&wp, q4kw_bytes(N, K), st));
    hipError_t e = hipMallocAsync(&xp, q4kp_bytes(M, K), st);
    int rc = MLLM_HIP_OK;
    if (e != hipSuccess) { set_error("hipMallocAsync(xpack)", e, __FILE__, __LINE__); rc = MLLM_HIP_ERR_HIP; }
    if (!rc) rc = mllm_hip_q4k_prepack(W, N, K, wp, stream);
    if (!rc) rc = mllm_hip_linear_q4kp_q8k(wp, bias, xqs, xd, xbsums, xp, y, y_dtype, ldy, residual, M, N, K, stream);
    // both blocks are returned on every path (a failing free is reported only when nothing failed before it)
    for (void *p : {wp, xp}) {
        if (!p) continue;
        e = hipFreeAsync(p, st);
        if (e != hipSuccess && !rc) { set_error("hipFreeAsync", e, __FILE__, __LINE__); rc = MLLM_HIP_ERR_HIP; }
    }
    return rc;
}

extern "C" int mllm_hip_linear_q40_q80(const uint8_t *Wqs, const uint16_t *Wd, const float *bias, const int8_t *xqs, const uint16_t *xd,
                                       float *y, int64_t ldy, int M, int N, int K, void *stream) {
    if (K % 256 != 0 || K <= 0 || N <= 0) return MLLM_HIP_ERR_SHAPE;
    // 16 lanes per row when K is a multiple of 512 (<= 8 blocks per lane), else 8 lanes per row (K multiple of 256)
    const int lpr = (K % 512 == 0 && K / 512 <= 8) ? 16 : 8;
    const int bpl = K / 32 / lpr;
    if (bpl > 8) return MLLM_HIP_ERR_SHAPE;
    hipStream_t st = as_stream(stream);
    const int target_waves = 256 * 8, rpp = 8;
    const size_t lds = 4 * q40_tab_floats(K / 32) * sizeof(float);
    int rows_per_wave = (N + target_waves - 1) / target_waves;
    rows_per_wave = ((rows_per_wave + rpp - 1) / rpp) * rpp;
    const int waves = (N + rows_per_wave - 1) / rows_per_wave;
    // one launch per 15 activation rows: the table rows are fetched once per launch and meet every activation row in registers (gemv_q40_kernel's m loop)
    for (int m0 = 0; m0 < M; m0 += 15) {
        const int mc = M - m0 < 15 ? M - m0 : 15;
        const int8_t *xq = xqs + (int64_t)m0 * K;
        const uint16_t *xdd = xd + (int64_t)m0 * (K / 32);
        float *ym = y + (int64_t)m0 * ldy;
#define Q40_CASE(B, L) if (bpl == B && lpr == L) hipLaunchKernelGGL((gemv_q40_kernel<B, L>), dim3((waves + 3) / 4), dim3(256), lds, st, Wqs, Wd, bias, xq, xdd, ym, N, rows_per_wave, mc, ldy);
        Q40_CASE(1, 16) Q40_CASE(2, 16) Q40_CASE(3, 16) Q40_CASE(4, 16) Q40_CASE(5, 16) Q40_CASE(6, 16) Q40_CASE(7, 16) Q40_CASE(8, 16)
        Q40_CASE(1, 8) Q40_CASE(3, 8) Q40_CASE(5, 8) Q40_CASE(7, 8)
#undef Q40_CASE
        int rc = MH_LAUNCH_OK("gemv_q40");
        if (rc) return rc;
    }
    return MLLM_HIP_OK;
}

extern "C" int mllm_hip_gemm_f32_bhsd(const float *a, const void *b, int b_dtype, float *c, int heads, int M, int N, int K, void *stream) {
    if (heads < 0 || M < 0 || N < 0 || K <= 0) return MLLM_HIP_ERR_SHAPE;
    if (b_dtype != MLLM_HIP_F32 && b_dtype != MLLM_HIP_F16) return MLLM_HIP_ERR_DTYPE;
    if (heads == 0 || M == 0 || N == 0) return MLLM_HIP_OK;
    if (!a || !b || !c) return MLLM_HIP_ERR_ARG;
    if (heads > 65535 || (M + 3) / 4 > 65535) return MLLM_HIP_ERR_SHAPE;
    const dim3 grid((N + 63) / 64, (M + 3) / 4, heads);
    if (b_dtype == MLLM_HIP_F16) hipLaunchKernelGGL(gemm_f32_bhsd_kernel<true>, grid, dim3(256), 0, as_stream(stream), a, b, c, M, N, K);
    else hipLaunchKernelGGL(gemm_f32_bhsd_kernel<false>, grid, dim3(256), 0, as_stream(stream), a, b, c, M, N, K);
    return MH_LAUNCH_OK("gemm_f32_bhsd");
}
extern "C" int mllm_hip_linear_f32(const float *W, const float *bias, const float *x, float *y, int64_t ldy, int M, int N, int K, void *stream) {
    if (K <= 0 || N <= 0) return MLLM_HIP_ERR_SHAPE;
    if (M <= 0) return MLLM_HIP_OK;
    if (M >= 16 && K >= 128 && K % 4 == 0 && (((uintptr_t)W | (uintptr_t)x) & 15) == 0 && (M + 31) / 32 <= 65535) {
        hipLaunchKernelGGL(gemm_f32_mfma_kernel, dim3((N + 31) / 32, (M + 31) / 32), dim3(256), 0, as_stream(stream), W, bias, x, y, ldy, M, N, K);
        return MH_LAUNCH_OK("gemm_f32_mfma");
    }
    dim3 grid((N + 4 * F32_TN - 1) / (4 * F32_TN), (M + 2 * F32_TM - 1) / (2 * F32_TM));
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
        if (M == 1 && y_dtype == MLLM_HIP_F32) {      // one row: quantiser and GEMV in one launch (same arithmetic: the fused decode kernel)
            const int one = dec_linear_row_q4k(W, x, bias, (float *)y, N, K, as_stream(stream));
            if (one <= 0) return one;
        }
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
