// mllm_amd/csrc/kernels_attn.hip -- A13: flash_attention_2_forward for gfx950, in the reference's evaluation order.
//
// The reference (compute/FlashAttention2.hpp) walks key tiles of Bc = 4 (Bc = 1 when Sq == 1) with an online softmax; its
// result depends on that order (fp32 is not associative, expf of a different running max differs in the last bit), and the
// model amplifies last-bit differences through the Q8_K re-quantisation of every Linear.  So these kernels keep the order and
// spread across the chip only what is independent:
//   scores    s[r][j]: 8 fp32 chains per (row, key) (chain l takes d = 8i + l), folded ((l0+l4)+(l1+l5)) + ((l2+l6)+(l3+l7))
//             (mma0 :314-357 / :1432-1470, _mm256_hadd_ps :39-46)                         -- one thread per key, all rows
//   softmax   m' = max(m, tile) is a prefix maximum -> DPP scan over the tiles; c = expf((m - m') scale), p = expf((s - m') scale),
//             sum = ((p0+p1)+p2)+p3 (:430-465)                                             -- one lane per key tile
//   P V       o[r][d] = fma(p_j, v_j[d], o[r][d] * c) in key order, logsum = fma(logsum, c, sum) (:577-636; GCC contracts the
//             logsum update: vfmadd132ss in the built reference)                           -- one thread per (row tile, d)
// expf is glibc's (common.h:glibc_expf).  Causal masking follows :351-357 literally: tiles right of the diagonal are skipped,
// the tile whose row end meets its column end is masked j > i, nothing else is.  Leftover columns: Sk % 4 for fp32 K/V (:152),
// Sk % (Sk / 4) for fp16 K/V (:1277) -- both kept.
#include "common.h"
#include "decode_launch.h"
#include "kernels_attn_core.h"

namespace mllm_hip {

// ------------------------------------------------------------------------------------------------------------------
// Sq >= 4: __fa2_prefill_append with Br = Bc = 4.  One workgroup = one head x RT consecutive row tiles.
// ------------------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------------------
// Sq >= 4: __fa2_prefill_append with Br = Bc = 4, on the matrix cores.
//
// v_mfma_f32_32x32x2_f32 accumulates C = fma(a1, b1, fma(a0, b0, C)) -- an exact fp32 fma chain in k order (checked on the
// device: scratch/mfma/test.hip, 0 mismatches against fmaf chains) -- so the reference's chains map onto it unchanged:
//   scores   chain l of (row, key) takes d = 8 i + l, i ascending: one accumulator tile per l, operands (d = 8(2s+h)+l) for MFMA s;
//            wave w owns chains w and w+4 and adds them in registers (the fold's first level); the 4 sums meet in LDS and are folded
//            ((l0+l4)+(l1+l5)) + ((l2+l6)+(l3+l7));
//   softmax  thread (row, key tile): tile maximum, prefix maximum over the 8 tiles of the chunk (+ the carried maximum),
//            c = expf((m - m') scale), p = expf((s - m') scale), sum = ((p0+p1)+p2)+p3; logsum = fma(logsum, c, sum) per row;
//   P V      o[row][d] = fma(p_j, v_j[d], o[row][d]) in key order = MFMAs over key pairs; the rescale o *= c sits between key
//            tiles and is executed only for tiles in which some row's maximum moved (c is exactly 1.0f otherwise).
// One workgroup = one head x 32 query rows (8 row tiles); key chunks of 32 (8 key tiles); wave w of the P V phase owns dims
// 32w .. 32w+31.  K and V chunks are staged in LDS as fp32 (K pitch D+1, transposed-slab V pitch 33: conflict-free operand reads).
// ------------------------------------------------------------------------------------------------------------------
constexpr int FA_R = 32, FA_KCH = 32;
#ifndef FA_PVG
#define FA_PVG 2      // key tiles whose P V operands are requested together (4: one register too many for D = 80 under the three-workgroups-per-CU cap)
#endif
// diagnostic build only (-DFA_STAMPS, scratch/fa_stamps.py): s_memtime at the phase boundaries of chunks 4..7 of workgroup 0, wave 0
#ifdef FA_STAMPS
__device__ unsigned long long g_fa_stamps[64];
#define FA_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && chunk0 >= 4 * FA_KCH && chunk0 < 8 * FA_KCH) g_fa_stamps[(chunk0 / FA_KCH - 4) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FA_STAMP(i)
#endif
typedef float v16f_t __attribute__((ext_vector_type(16)));

// (head sizes up to 80 fit 168 registers and 44 KiB of LDS: three workgroups per CU, whose barrier / softmax / fetch phases run under each other's MFMAs)
template <int D, bool F16, bool VT = false>
__global__ __launch_bounds__(256, (D <= 80 ? 3 : 2)) void fa2_prefill_kernel(const float *__restrict__ Q, int64_t ldq, const void *__restrict__ K, int64_t ldk,
                                                          const void *__restrict__ V, int64_t ldv, float *__restrict__ O, int64_t ldo, int Sq, int Sk,
                                                          int sk_eff, int Hq, int Hkv, int causal, int64_t bq, int64_t bk, int64_t bv, int64_t bo) {
    // blockIdx.z = which of the independent (q, k, v, o) sets of the launch (the images of a vision pass); strides in elements
    Q += (int64_t)blockIdx.z * bq;
    O += (int64_t)blockIdx.z * bo;
    K = reinterpret_cast<const char *>(K) + (int64_t)blockIdx.z * bk * (F16 ? 2 : 4);
    V = reinterpret_cast<const char *>(V) + (int64_t)blockIdx.z * bv * (F16 ? 2 : 4);
    static_assert(D % 16 == 0 && D <= 128, "head dim");
    constexpr int NS = D / 16;              // MFMAs per score chain
    constexpr int KP = D + 4;               // K row pitch in LDS: a multiple of four floats, so a staged group is ONE 16-byte store (with the odd pitch D + 1 it was four
                                            // scalar stores); the per-key column reads of the score MFMAs then meet two lanes per bank instead of one, which costs them a cycle
    constexpr int VP = VT ? 33 : D;         // V: [key][D] as stored, or [d][33] from the transposed slab
    constexpr int NDT = (D + 31) / 32;      // 32-wide dim tiles of the output
    constexpr int KE = FA_KCH * D / 256;    // staged elements per thread and operand
    __shared__ float Ks[FA_KCH * KP];
    __shared__ float Vs[VT ? D * 33 : FA_KCH * D];
    __shared__ float Part[4 * 32 * 33];
    __shared__ float P[32 * 33];
    __shared__ float Cc[32 * 8], Sm[32 * 8];
    __shared__ float m_in[32], l_s[32];
    __shared__ int moved_any[8];
    __shared__ uint64_t etab[32];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, col = lane & 31, h = lane >> 5;
    // Workgroups go to the XCDs round-robin by linear id; all row blocks of a head read the same K / V, so with a head count that is a multiple of 8 the head becomes
    // the fastest index: head h then lives on XCD h % 8 and its K / V are pulled into ONE L2 instead of eight (FETCH_SIZE of the ViT block: 89 MB per launch before)
    int head = blockIdx.y, rb = blockIdx.x;
    if ((Hq & 7) == 0) { const int lin = blockIdx.x + gridDim.x * blockIdx.y; head = lin % Hq; rb = lin / Hq; }
    const int kvh = head / (Hq / Hkv);
    const int r0 = rb * FA_R;
    const int delta = Sk - Sq;
    const float scale = 1.0f / sqrtf((float)D);
    // this lane's query operands: row r0 + col, dims 8(2s+h) + l for the wave's two chains
    float qreg[2][NS];
    {
        const float *qrow = Q + (int64_t)min(r0 + col, Sq - 1) * ldq + head * D;
#pragma unroll
        for (int l2 = 0; l2 < 2; ++l2)
#pragma unroll
            for (int sI = 0; sI < NS; ++sI) qreg[l2][sI] = qrow[8 * (2 * sI + h) + wid + 4 * l2];
    }
    if (tid < 32) { m_in[tid] = FA_NEG; l_s[tid] = 0.0f; }
    expf_tab_store(etab, expf_tab_fetch());
    int klim = sk_eff;
    if (causal) klim = min(sk_eff, r0 + FA_R + delta + 4);   // tiles beyond are skipped for every row of this workgroup
    // P V on v_mfma_f32_16x16x4_f32 -- like the 32x32x2 form an exact fp32 fma chain in k order (scratch/mfma/test16.hip: 0 of 256 outputs differ
    // from sequential fmaf), but four keys per instruction and 40 instead of 2 x 64 cycles of dependent latency per key tile.  The 32 x D
    // output is 2 * D/16 tiles (row half rh, dim tile dt), tile id = rh + 2 dt, wave w owns ids w, w + 4, ...: independent chains that interleave.
    typedef float v4f_t __attribute__((ext_vector_type(4)));
    constexpr int ND16 = D / 16, NT16 = 2 * ND16, TPW = (NT16 + 3) / 4;
    const int c16 = lane & 15, q4 = lane >> 4;
    v4f_t oacc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) oacc[j] = v4f_t{0.0f, 0.0f, 0.0f, 0.0f};
    // staging registers (the next chunk is requested while the current one is consumed): four consecutive values per request --
    // four dims of a key row, or four keys of a dim row of the transposed V slab
    constexpr int KG = (FA_KCH * D / 4 + 255) / 256;     // 4-element groups per thread and operand
    float4 kst[KG], vst[KG];
    auto load4 = [&](const void *base, int64_t off) -> float4 {
        if (F16) {
            const uint2 w = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(base) + off);
            return make_float4(h2f((uint16_t)(w.x & 0xffff)), h2f((uint16_t)(w.x >> 16)), h2f((uint16_t)(w.y & 0xffff)), h2f((uint16_t)(w.y >> 16)));
        }
        return *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(base) + off);
    };
    // element offsets of this thread's groups in chunk 0; a chunk that lies wholly inside the Sk rows is reached by adding chunk0 * ld (one 64-bit add per
    // request instead of a clamp and 64-bit multiplies -- a lone wave pays for every instruction); only the ragged last chunk clamps its key rows
    int64_t koff[KG], voff[KG];
#pragma unroll
    for (int i = 0; i < KG; ++i) {
        const int e4 = min(tid + 256 * i, FA_KCH * D / 4 - 1);
        const int key = e4 / (D / 4), d4 = e4 - key * (D / 4);
        koff[i] = (int64_t)key * ldk + kvh * D + 4 * d4;
        if (VT) {
            const int dd = e4 / (FA_KCH / 4), k4 = e4 - dd * (FA_KCH / 4);
            voff[i] = (int64_t)(kvh * D + dd) * ldv + 4 * k4;
        } else {
            voff[i] = (int64_t)key * ldv + kvh * D + 4 * d4;
        }
    }
    auto fetch = [&](int chunk0) {
        if (chunk0 + FA_KCH <= Sk) {
            const int64_t ck = (int64_t)chunk0 * ldk, cv = VT ? (int64_t)chunk0 : (int64_t)chunk0 * ldv;
#pragma unroll
            for (int i = 0; i < KG; ++i) { kst[i] = load4(K, koff[i] + ck); vst[i] = load4(V, voff[i] + cv); }
            return;
        }
#pragma unroll
        for (int i = 0; i < KG; ++i) {
            const int e4 = min(tid + 256 * i, FA_KCH * D / 4 - 1);
            {
                const int key = e4 / (D / 4), d4 = e4 - key * (D / 4);
                kst[i] = load4(K, (int64_t)min(chunk0 + key, Sk - 1) * ldk + kvh * D + 4 * d4);
            }
            if (VT) {
                const int dd = e4 / (FA_KCH / 4), k4 = e4 - dd * (FA_KCH / 4);
                vst[i] = load4(V, (int64_t)(kvh * D + dd) * ldv + chunk0 + 4 * k4);
            } else {
                const int key = e4 / (D / 4), d4 = e4 - key * (D / 4);
                vst[i] = load4(V, (int64_t)min(chunk0 + key, Sk - 1) * ldv + kvh * D + 4 * d4);
            }
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int i = 0; i < KG; ++i) {
            const int e4 = tid + 256 * i;
            if (e4 < FA_KCH * D / 4) {
                {
                    const int key = e4 / (D / 4), d4 = e4 - key * (D / 4);
                    *reinterpret_cast<float4 *>(Ks + key * KP + 4 * d4) = kst[i];
                }
                if (VT) {
                    const int dd = e4 / (FA_KCH / 4), k4 = e4 - dd * (FA_KCH / 4);
                    float *vd = Vs + dd * 33 + 4 * k4;
                    vd[0] = vst[i].x; vd[1] = vst[i].y; vd[2] = vst[i].z; vd[3] = vst[i].w;
                } else {
                    *reinterpret_cast<float4 *>(Vs + 4 * e4) = vst[i];
                }
            }
        }
    };
    fetch(0);
    for (int chunk0 = 0; chunk0 < klim; chunk0 += FA_KCH) {
        FA_STAMP(0);
        __syncthreads();                       // the previous chunk's P V has finished with Ks / Vs / P
        FA_STAMP(1);
        park();
        if (tid < 8) moved_any[tid] = 0;
        __syncthreads();
        FA_STAMP(2);
        if (chunk0 + FA_KCH < klim) fetch(chunk0 + FA_KCH);
        // ---- scores: wave w owns chains w and w + 4 -- the pair the reference's fold adds first -- and hands their sum to the fold ------------------
        {
            v16f_t acc[2];
#pragma unroll
            for (int l2 = 0; l2 < 2; ++l2)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[l2][i] = 0.0f;
#pragma unroll
            for (int sI = 0; sI < NS; ++sI)
#pragma unroll
                for (int l2 = 0; l2 < 2; ++l2)
                    acc[l2] = __builtin_amdgcn_mfma_f32_32x32x2f32(qreg[l2][sI], Ks[col * KP + 8 * (2 * sI + h) + wid + 4 * l2], acc[l2], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) Part[(wid * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * 33 + col] = acc[0][i] + acc[1][i];
        }
        FA_STAMP(3);
        __syncthreads();
        FA_STAMP(4);
        // ---- fold the chains, mask, softmax of (row, key tile) ----------------------------------------------------------------------
        {
            const int row = tid >> 3, tile = tid & 7;
            const int c0 = chunk0 + 4 * tile;
            const int tr0 = r0 + (row & ~3), nr = min(4, Sq - tr0);
            const int nc = min(4, sk_eff - c0);
            const bool live = nc > 0 && nr > 0 && !(causal && (c0 - delta > tr0 + nr - 1));
            const bool diag = causal && (tr0 + nr == c0 + nc - delta);
            float s4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float *pp = Part + row * 33 + 4 * tile + e;
                float v = (pp[0 * 1056] + pp[1 * 1056]) + (pp[2 * 1056] + pp[3 * 1056]);     // pp[w] = l_w + l_(w+4)
                if (diag && e > (row & 3)) v = FA_NEG;
                s4[e] = v;
            }
            FA_STAMP(8);
            float tm = FA_NEG;
            if (live) {
                tm = s4[0];
                if (nc > 1) tm = fmaxf(tm, s4[1]);
                if (nc > 2) tm = fmaxf(tm, s4[2]);
                if (nc > 3) tm = fmaxf(tm, s4[3]);
            }
            // prefix maximum over the 8 tiles of this row (8 consecutive lanes), then the carried maximum
            float sc = tm;
            { const float t = MH_DPPF(sc, sc, 0x111, 0xF); sc = tile >= 1 ? fmaxf(sc, t) : sc; }
            { const float t = MH_DPPF(sc, sc, 0x112, 0xF); sc = tile >= 2 ? fmaxf(sc, t) : sc; }
            { const float t = MH_DPPF(sc, sc, 0x114, 0xF); sc = tile >= 4 ? fmaxf(sc, t) : sc; }
            const float carry = m_in[row];
            const float incl = fmaxf(sc, carry);
            const float prev = MH_DPPF(incl, incl, 0x111, 0xF);
            const float excl = tile == 0 ? carry : prev;
            FA_STAMP(9);
            float p4[4] = {0.0f, 0.0f, 0.0f, 0.0f}, cc = 1.0f, sum = 0.0f;
            if (live) {
                if (excl != incl) { cc = glibc_expf((excl - incl) * scale, etab); moved_any[tile] = 1; }
                // all four columns unconditionally (independent chains for the scheduler); columns past the end hold finite scores of
                // the clamped last key row and are zeroed afterwards
#pragma unroll
                for (int e = 0; e < 4; ++e) p4[e] = glibc_expf((s4[e] - incl) * scale, etab);
#pragma unroll
                for (int e = 1; e < 4; ++e) p4[e] = e < nc ? p4[e] : 0.0f;
                sum = ((p4[0] + p4[1]) + p4[2]) + p4[3];
            }
            FA_STAMP(10);
#pragma unroll
            for (int e = 0; e < 4; ++e) P[row * 33 + 4 * tile + e] = p4[e];
            Cc[row * 8 + tile] = cc;
            Sm[row * 8 + tile] = sum;
            __builtin_amdgcn_wave_barrier();
            if (tile == 7) m_in[row] = incl;
        }
        FA_STAMP(5);
        __syncthreads();
        FA_STAMP(6);
        // ---- logsum per row (32 lanes of the last wave), P V per dim tile ----------------------------------------------------------------
        if (wid == 3 && lane < 32) {
            float l = l_s[lane];
#pragma unroll
            for (int t = 0; t < 8; ++t) l = __fmaf_rn(l, Cc[lane * 8 + t], Sm[lane * 8 + t]);
            l_s[lane] = l;
        }
        // which key tiles saw some row's maximum move: one 8-lane LDS read, folded into a wave-uniform mask (the per-tile `if (moved_any[t])` paid an LDS round trip in front
        // of every tile's MFMAs)
        const unsigned moved_mask = (unsigned)__builtin_amdgcn_readfirstlane((int)__ballot(lane < 8 && moved_any[lane & 7] != 0));
        // the MFMA operands of FA_PVG key tiles at a time are requested before the first of their MFMAs (left inside the tile loop, each tile's reads sat behind the previous
        // tile's rescale branch and every MFMA waited out an LDS round trip: 2,300 cycles per chunk for 24 MFMAs -- scratch/fa_stamps.py); the next group's reads travel
        // while this group's MFMAs run
        const int rh0 = wid & 1;      // every tile of this wave has the same row half: tile ids w, w + 4, ...
#pragma unroll
        for (int hh = 0; hh < 8 / FA_PVG; ++hh) {
            float pop[FA_PVG], vop[FA_PVG][TPW];
#pragma unroll
            for (int tt = 0; tt < FA_PVG; ++tt) {
                const int key = 4 * (FA_PVG * hh + tt) + q4;
                pop[tt] = P[(16 * rh0 + c16) * 33 + key];
#pragma unroll
                for (int j = 0; j < TPW; ++j) {
                    const int dd = 16 * (min(wid + 4 * j, NT16 - 1) >> 1) + c16;
                    vop[tt][j] = VT ? Vs[dd * 33 + key] : Vs[key * D + dd];
                }
            }
#pragma unroll
            for (int tt = 0; tt < FA_PVG; ++tt) {
                const int t = FA_PVG * hh + tt;
                if ((moved_mask >> t) & 1u) {
#pragma unroll
                    for (int j = 0; j < TPW; ++j) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) oacc[j][i] = oacc[j][i] * Cc[(16 * rh0 + 4 * q4 + i) * 8 + t];
                    }
                }
                // a wave whose last tile id is past the end repeats the last tile (its result is not stored): no control flow around the MFMAs,
                // which would make the compiler shuttle every accumulator through AGPR copies at each branch
#pragma unroll
                for (int j = 0; j < TPW; ++j) oacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(pop[tt], vop[tt][j], oacc[j], 0, 0, 0);
            }
        }
        FA_STAMP(7);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int tile = wid + 4 * j, rh = tile & 1, dt = tile >> 1;
        if (tile < NT16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * rh + 4 * q4 + i;
                if (r0 + row < Sq) O[(int64_t)(r0 + row) * ldo + head * D + 16 * dt + c16] = oacc[j][i] * (1.0f / l_s[row]);
            }
        }
    }
}

template <int D, bool F16, int NT, bool VT>
__global__ __launch_bounds__(NT) void fa2_decode_kernel(const float *__restrict__ Q, const void *__restrict__ K, int64_t ldk, const void *__restrict__ V,
                                                        int64_t ldv, float *__restrict__ O, int Sk, const int *__restrict__ sk_dev, int cap, int nslots, int Hq,
                                                        int Hkv) {
    extern __shared__ __attribute__((aligned(16))) char fa_smem[];
    const int head = blockIdx.x, kvh = head / (Hq / Hkv);
    DecodePrefetch<D, F16, NT, VT> P;
    fa2_decode_prefetch<D, F16, NT, VT>(P, K, ldk, V, ldv, kvh * D, cap, nslots);
    if (sk_dev) Sk = min(*sk_dev, cap);
    const DecodeLds L = carve_decode(fa_smem, cap, D, NT, nslots);
    if (threadIdx.x < D) L.qs[threadIdx.x] = Q[head * D + threadIdx.x];
    __syncthreads();
    fa2_decode_head<D, F16, NT, VT>(L, P, K, ldk, V, ldv, kvh * D, kvh * D, Sk, cap, nullptr, nullptr, -1);
    if (threadIdx.x < D) O[head * D + threadIdx.x] = L.ob[threadIdx.x];
}
// ONE decode position through the five Ops of the reference's attention block in one launch (mllm_hip_fa2_decode_step; integration/hip's lazy window): RoPE(q), RoPE(k), the
// fp16 appends of the rotated k and of v to their cache slabs, and F_FA2 over the T + 1 keys.  A workgroup per query head rotates its q (and stores the RoPE Op's fp32 output),
// rotates its group's k and rounds k / v to fp16 in LDS -- the first head of a group also stores the RoPE Op's k output and the two slab rows -- and hands those LDS rows to
// fa2_decode_head as key T, so no workgroup depends on another's stores.  rope_apply_kernel's, store_f16's and fa2_decode_kernel's arithmetic, element for element.
struct AttnStep {
    const float *q_raw, *k_raw, *v_raw, *sin_q, *cos_q, *sin_k, *cos_k;
    float *q_out, *k_out, *O;
    uint16_t *kslab, *vslab;      // [T + 1][Hkv * D] fp16
    int T, Hq, Hkv, nslots;
};
template <int D, int NT>
__global__ __launch_bounds__(NT) void fa2_decode_step_kernel(const AttnStep A) {
    constexpr int HALF = D / 2;
    extern __shared__ __attribute__((aligned(16))) char fa_smem[];
    __shared__ __attribute__((aligned(16))) uint16_t knew[D];
    __shared__ __attribute__((aligned(16))) uint16_t vnew[D];
    const int head = blockIdx.x, gsize = A.Hq / A.Hkv, kvh = head / gsize, tid = threadIdx.x, KVD = A.Hkv * D, cap = A.T + 1;
    float qa = 0.0f, qb = 0.0f, sn = 0.0f, cs = 0.0f;
    if (tid < HALF) { const float *qp = A.q_raw + head * D; qa = qp[tid]; qb = qp[tid + HALF]; sn = A.sin_q[tid]; cs = A.cos_q[tid]; }
    else if (tid < D) { const float *kp = A.k_raw + kvh * D; qa = kp[tid - HALF]; qb = kp[tid]; sn = A.sin_k[tid - HALF]; cs = A.cos_k[tid - HALF]; }
    else if (tid < 2 * D) qa = A.v_raw[kvh * D + (tid - D)];
    __builtin_amdgcn_sched_barrier(0);
    DecodePrefetch<D, true, NT, false> P;
    fa2_decode_prefetch<D, true, NT, false>(P, A.kslab, KVD, A.vslab, KVD, kvh * D, cap, A.nslots);
    __builtin_amdgcn_sched_barrier(0);
    const DecodeLds L = carve_decode(fa_smem, cap, D, NT, A.nslots);
    const bool writer = head == kvh * gsize;
    if (tid < HALF) {
        const float v1 = __fmaf_rn(qa, cs, -__fmul_rn(qb, sn)), v2 = __fmaf_rn(qa, sn, __fmul_rn(qb, cs));
        L.qs[tid] = v1; L.qs[tid + HALF] = v2;
        A.q_out[head * D + tid] = v1; A.q_out[head * D + tid + HALF] = v2;
    } else if (tid < D) {
        const float v1 = __fmaf_rn(qa, cs, -__fmul_rn(qb, sn)), v2 = __fmaf_rn(qa, sn, __fmul_rn(qb, cs));
        knew[tid - HALF] = f2h(v1); knew[tid] = f2h(v2);
        if (writer) { A.k_out[kvh * D + tid - HALF] = v1; A.k_out[kvh * D + tid] = v2; }
    } else if (tid < 2 * D) {
        vnew[tid - D] = f2h(qa);
    }
    __syncthreads();
    if (writer && tid < D) {
        A.kslab[(int64_t)A.T * KVD + kvh * D + tid] = knew[tid];
        A.vslab[(int64_t)A.T * KVD + kvh * D + tid] = vnew[tid];
    }
    fa2_decode_head<D, true, NT, false>(L, P, A.kslab, KVD, A.vslab, KVD, kvh * D, kvh * D, cap, cap, knew, vnew, A.T);
    if (tid < D) A.O[head * D + tid] = L.ob[tid];
}
template <int D>
static int launch_fa2_step(AttnStep A, hipStream_t st, bool dry) {
    constexpr int NT = 1024;
    const int cap = A.T + 1;
    A.nslots = decode_lds_slots(cap, D, NT, 2, false);
    const size_t lds = decode_lds_bytes(cap, D, NT, 2, A.nslots, false);
    if (lds > 160 * 1024) return MLLM_HIP_ERR_SHAPE;
    if (dry) return MLLM_HIP_OK;
    auto kern = fa2_decode_step_kernel<D, NT>;
    if (lds > 48 * 1024) MH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(A.Hq), dim3(NT), lds, st, A);
    return MH_LAUNCH_OK("fa2_decode_step");
}
static int fa2_step(const float *q_raw, const float *sin_q, const float *cos_q, float *q_out, const float *k_raw, const float *sin_k, const float *cos_k, float *k_out, const float *v_raw,
                    uint16_t *kslab, uint16_t *vslab, int T, float *O, int Hq, int Hkv, int D, hipStream_t st, bool dry) {
    if (T < 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || (Hkv * D) % 8) return MLLM_HIP_ERR_SHAPE;
    if (!dry && (!q_raw || !k_raw || !v_raw || !sin_q || !cos_q || !sin_k || !cos_k || !q_out || !k_out || !kslab || !vslab || !O)) return MLLM_HIP_ERR_ARG;
    const AttnStep A{q_raw, k_raw, v_raw, sin_q, cos_q, sin_k, cos_k, q_out, k_out, O, kslab, vslab, T, Hq, Hkv, 0};
    switch (D) {
    case 64: return launch_fa2_step<64>(A, st, dry);
    case 128: return launch_fa2_step<128>(A, st, dry);
    default: return MLLM_HIP_ERR_SHAPE;
    }
}
// the same for B sequences in one launch (batched decode): blockIdx.y = sequence, whose slabs and key count come from its descriptor; fp16 K rows, transposed fp16 V
template <int D, int NT>
__global__ __launch_bounds__(NT) void fa2_decode_seqs_kernel(const float *__restrict__ Q, int64_t ldq, const SeqKV *__restrict__ seqs, int64_t layer_k_off, int64_t layer_v_off,
                                                             int64_t ldk, int64_t ldv, float *__restrict__ O, int64_t ldo, int cap, int nslots, int Hq, int Hkv) {
    extern __shared__ __attribute__((aligned(16))) char fa_smem[];
    const int head = blockIdx.x, kvh = head / (Hq / Hkv), b = blockIdx.y;
    const SeqKV sq = seqs[b];
    const void *K = sq.k + layer_k_off, *V = sq.v + layer_v_off;
    DecodePrefetch<D, true, NT, true> P;
    fa2_decode_prefetch<D, true, NT, true>(P, K, ldk, V, ldv, kvh * D, cap, nslots);
    const int Sk = min(sq.t + 1, cap);
    const DecodeLds L = carve_decode(fa_smem, cap, D, NT, nslots);
    if (threadIdx.x < D) L.qs[threadIdx.x] = Q[(int64_t)b * ldq + head * D + threadIdx.x];
    __syncthreads();
    fa2_decode_head<D, true, NT, true>(L, P, K, ldk, V, ldv, kvh * D, kvh * D, Sk, cap, nullptr, nullptr, -1);
    if (threadIdx.x < D) O[(int64_t)b * ldo + head * D + threadIdx.x] = L.ob[threadIdx.x];
}
template <int D>
static int launch_fa2_seqs(const float *q, int64_t ldq, const SeqKV *seqs_dev, int64_t layer_k_off, int64_t layer_v_off, int64_t ldk, int64_t ldvt, float *o, int64_t ldo, int B,
                           int Hq, int Hkv, int cap, hipStream_t st) {
    constexpr int NT = 1024;
    const int nslots = decode_lds_slots(cap, D, NT, 2, true);
    const size_t lds = decode_lds_bytes(cap, D, NT, 2, nslots, true);
    if (lds > 160 * 1024) return MLLM_HIP_ERR_SHAPE;
    auto kern = fa2_decode_seqs_kernel<D, NT>;
    if (lds > 48 * 1024) MH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(Hq, B), dim3(NT), lds, st, q, ldq, seqs_dev, layer_k_off, layer_v_off, ldk, ldvt, o, ldo, cap, nslots, Hq, Hkv);
    return MH_LAUNCH_OK("fa2_decode_seqs");
}
int seqs_fa2_decode_launch(const float *q, int64_t ldq, const SeqKV *seqs_dev, int64_t layer_k_off, int64_t layer_v_off, int64_t ldk, int64_t ldvt, float *o, int64_t ldo, int B,
                           int Hq, int Hkv, int D, int cap, hipStream_t st) {
    if (B <= 0) return MLLM_HIP_OK;
    if (!q || !o || !seqs_dev || Hq <= 0 || Hkv <= 0 || Hq % Hkv || (ldk % 8) || (ldvt % 8)) return MLLM_HIP_ERR_ARG;
    switch (D) {
    case 64: return launch_fa2_seqs<64>(q, ldq, seqs_dev, layer_k_off, layer_v_off, ldk, ldvt, o, ldo, B, Hq, Hkv, cap, st);
    case 128: return launch_fa2_seqs<128>(q, ldq, seqs_dev, layer_k_off, layer_v_off, ldk, ldvt, o, ldo, B, Hq, Hkv, cap, st);
    default: return MLLM_HIP_ERR_SHAPE;
    }
}
}  // namespace mllm_hip

using namespace mllm_hip;

extern "C" int mllm_hip_fa2_decode_step(const float *q_raw, const float *sin_q, const float *cos_q, float *q_out, const float *k_raw, const float *sin_k, const float *cos_k, float *k_out,
                                        const float *v_raw, uint16_t *kslab, uint16_t *vslab, int T, float *O, int Hq, int Hkv, int D, void *stream) {
    return fa2_step(q_raw, sin_q, cos_q, q_out, k_raw, sin_k, cos_k, k_out, v_raw, kslab, vslab, T, O, Hq, Hkv, D, as_stream(stream), false);
}
extern "C" int mllm_hip_fa2_decode_step_supported(int T, int Hq, int Hkv, int D) {
    return fa2_step(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, T, nullptr, Hq, Hkv, D, nullptr, true) == MLLM_HIP_OK ? 1 : 0;
}
extern "C" size_t mllm_hip_fa2_workspace_bytes(int Sq, int Hq, int D, int max_sk) {
    (void)Sq; (void)Hq; (void)D; (void)max_sk;
    return 256;   // the kernels keep their state in LDS; a token allocation keeps callers' bookkeeping uniform
}

template <int D, bool F16, bool VT = false>
static int launch_fa2(const float *Q, int64_t ldq, const void *K, int64_t ldk, const void *V, int64_t ldv, float *O, int64_t ldo, int Sq, int Sk,
                      int Hq, int Hkv, int causal, const int *sk_dev, int sk_max, hipStream_t st, int nbatch = 1, int64_t bq = 0, int64_t bk = 0, int64_t bv = 0,
                      int64_t bo = 0) {
    constexpr int NT = 1024;
    auto decode_row = [&](const float *q, float *o, int sk, const int *skd, int skm) -> int {
        constexpr int ELT = F16 ? 2 : 4;
        const int nslots = decode_lds_slots(skm, D, NT, ELT, VT);
        const size_t lds = decode_lds_bytes(skm, D, NT, ELT, nslots, VT);
        if (lds > 160 * 1024) return MLLM_HIP_ERR_SHAPE;
        auto kern = fa2_decode_kernel<D, F16, NT, VT>;
        if (lds > 48 * 1024) MH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(Hq), dim3(NT), lds, st, q, K, ldk, V, ldv, o, sk, skd, skm, nslots, Hq, Hkv);
        return MH_LAUNCH_OK("fa2_decode");
    };
    if (nbatch > 1 && Sq < 4) return MLLM_HIP_ERR_SHAPE;      // (the batched form is the prefill kernel's)
    if (Sq == 1) return decode_row(Q, O, Sk, sk_dev, sk_dev ? sk_max : Sk);
    if (sk_dev) return MLLM_HIP_ERR_ARG;
    if (Sq < 4) {
        // Br = Bc = 1 (CPUFlashAttention2Func.hpp:71-72): row r of __fa2_prefill_append sees keys j <= r + (Sk - Sq) when causal,
        // all keys otherwise, one key per tile -- the decode recurrence per row
        for (int r = 0; r < Sq; ++r) {
            const int sk_r = causal ? min(Sk, r + (Sk - Sq) + 1) : Sk;
            int rc = decode_row(Q + (int64_t)r * ldq, O + (int64_t)r * ldo, sk_r, nullptr, sk_r);
            if (rc) return rc;
        }
        return MLLM_HIP_OK;
    }
    const int Tc = Sk / 4;
    const int left = F16 ? (Tc ? Sk % Tc : 0) : Sk % 4;
    const int sk_eff = Tc * 4 + left;
    constexpr int R = FA_R;
    hipLaunchKernelGGL((fa2_prefill_kernel<D, F16, VT>), dim3((Sq + R - 1) / R, Hq, nbatch), dim3(256), 0, st, Q, ldq, K, ldk, V, ldv, O, ldo, Sq, Sk, sk_eff, Hq,
                       Hkv, causal, bq, bk, bv, bo);
    return MH_LAUNCH_OK("fa2_prefill");
}

extern "C" int mllm_hip_fa2(const float *Q, int64_t ldq, const void *K, int64_t ldk, const void *V, int64_t ldv, int kv_dtype, float *O,
                            int64_t ldo, int Sq, int Sk, int Hq, int Hkv, int D, int causal, const int *sk_dev, void *workspace, void *stream) {
    (void)workspace;
    if (Sq <= 0 || Sk <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0) return MLLM_HIP_ERR_SHAPE;
    if (kv_dtype != MLLM_HIP_F16 && kv_dtype != MLLM_HIP_F32) return MLLM_HIP_ERR_DTYPE;
    // 16-byte vector loads of K rows
    if ((ldk % 8) || (ldv % 8)) return MLLM_HIP_ERR_SHAPE;
    hipStream_t st = as_stream(stream);
    const bool f16 = kv_dtype == MLLM_HIP_F16;
#define FA2_CASE(DD)                                                                                                                        \
    case DD:                                                                                                                                \
        return f16 ? launch_fa2<DD, true>(Q, ldq, K, ldk, V, ldv, O, ldo, Sq, Sk, Hq, Hkv, causal, sk_dev, Sk, st)                          \
                   : launch_fa2<DD, false>(Q, ldq, K, ldk, V, ldv, O, ldo, Sq, Sk, Hq, Hkv, causal, sk_dev, Sk, st);
    switch (D) {
        FA2_CASE(16)
        FA2_CASE(64)
        FA2_CASE(80)
        FA2_CASE(128)
    default: return MLLM_HIP_ERR_SHAPE;
    }
#undef FA2_CASE
}

// nbatch independent attentions of the same geometry in ONE launch (the images of a vision pass: 512 workgroups per 448-pixel image leave the third workgroup slot
// of a CU empty): set b reads q / k / v / o at element offsets b * bq / bk / bv / bo.  Sq >= 4 (the prefill recurrence), fp32 or fp16 K / V rows.
extern "C" int mllm_hip_fa2_batch(const float *Q, int64_t ldq, const void *K, int64_t ldk, const void *V, int64_t ldv, int kv_dtype, float *O, int64_t ldo, int Sq, int Sk,
                                  int Hq, int Hkv, int D, int causal, int nbatch, int64_t bq, int64_t bk, int64_t bv, int64_t bo, void *stream) {
    if (Sq < 4 || Sk <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0 || nbatch <= 0 || nbatch > 65535) return MLLM_HIP_ERR_SHAPE;
    if (kv_dtype != MLLM_HIP_F16 && kv_dtype != MLLM_HIP_F32) return MLLM_HIP_ERR_DTYPE;
    if ((ldk % 8) || (ldv % 8) || (bk % 8) || (bv % 8)) return MLLM_HIP_ERR_SHAPE;
    if (!Q || !K || !V || !O) return MLLM_HIP_ERR_ARG;
    hipStream_t st = as_stream(stream);
    const bool f16 = kv_dtype == MLLM_HIP_F16;
#define FA2_CASE(DD)                                                                                                                                    \
    case DD:                                                                                                                                            \
        return f16 ? launch_fa2<DD, true>(Q, ldq, K, ldk, V, ldv, O, ldo, Sq, Sk, Hq, Hkv, causal, nullptr, Sk, st, nbatch, bq, bk, bv, bo)             \
                   : launch_fa2<DD, false>(Q, ldq, K, ldk, V, ldv, O, ldo, Sq, Sk, Hq, Hkv, causal, nullptr, Sk, st, nbatch, bq, bk, bv, bo);
    switch (D) {
        FA2_CASE(16)
        FA2_CASE(64)
        FA2_CASE(80)
        FA2_CASE(128)
    default: return MLLM_HIP_ERR_SHAPE;
    }
#undef FA2_CASE
}

// Prefill on the engine's own KV layout: K rows fp16 [Sk][Hkv*D] (ldk), V transposed fp16 [Hkv*D][ldvt] (kernels_decode.hip
// reads the same slab).  Same arithmetic as mllm_hip_fa2 with kv_dtype fp16.
extern "C" int mllm_hip_fa2_vt(const float *Q, int64_t ldq, const void *K, int64_t ldk, const void *Vt, int64_t ldvt, float *O, int64_t ldo, int Sq, int Sk,
                               int Hq, int Hkv, int D, int causal, void *stream) {
    if (Sq <= 0 || Sk <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0 || (ldk % 8) || (ldvt % 8)) return MLLM_HIP_ERR_SHAPE;
    hipStream_t st = as_stream(stream);
    switch (D) {
    case 64: return launch_fa2<64, true, true>(Q, ldq, K, ldk, Vt, ldvt, O, ldo, Sq, Sk, Hq, Hkv, causal, nullptr, Sk, st);
    case 128: return launch_fa2<128, true, true>(Q, ldq, K, ldk, Vt, ldvt, O, ldo, Sq, Sk, Hq, Hkv, causal, nullptr, Sk, st);
    default: return MLLM_HIP_ERR_SHAPE;
    }
}

#ifdef FA_STAMPS
extern "C" int mllm_hip_debug_read_fa_stamps(unsigned long long *host) { return hipMemcpyFromSymbol(host, HIP_SYMBOL(mllm_hip::g_fa_stamps), 64 * 8) == hipSuccess ? 0 : -1; }
#endif
