// mllm_amd/csrc/kernels_attn_core.h -- device pieces of the reference-order attention shared by kernels_attn.hip (stand-alone
// launchers) and kernels_decode.hip (fused decode step).  See kernels_attn.hip for the evaluation order being reproduced.
#pragma once
#include "common.h"
#ifndef STAMP
#define STAMP(i)
#endif
#ifndef STAMPT
#define STAMPT(i, t)
#endif
#ifndef STAMPV
#define STAMPV(i, v)
#endif
#ifndef STAMPB
#define STAMPB(i)
#endif

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

// mma0 of one key straight from memory (no register image of the key row): same chains, same order
template <int D, bool F16>
__device__ __forceinline__ float qk_dot_row(const float *q, const void *base, int64_t off) {
    float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (F16) {
        const uint4 *p = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(base) + off);
#pragma unroll
        for (int i = 0; i < D / 8; ++i) {
            const uint4 w = p[i];
            const float4 a = *reinterpret_cast<const float4 *>(q + 8 * i), b = *reinterpret_cast<const float4 *>(q + 8 * i + 4);
            l[0] = __fmaf_rn(a.x, h2f((uint16_t)(w.x & 0xffff)), l[0]); l[1] = __fmaf_rn(a.y, h2f((uint16_t)(w.x >> 16)), l[1]);
            l[2] = __fmaf_rn(a.z, h2f((uint16_t)(w.y & 0xffff)), l[2]); l[3] = __fmaf_rn(a.w, h2f((uint16_t)(w.y >> 16)), l[3]);
            l[4] = __fmaf_rn(b.x, h2f((uint16_t)(w.z & 0xffff)), l[4]); l[5] = __fmaf_rn(b.y, h2f((uint16_t)(w.z >> 16)), l[5]);
            l[6] = __fmaf_rn(b.z, h2f((uint16_t)(w.w & 0xffff)), l[6]); l[7] = __fmaf_rn(b.w, h2f((uint16_t)(w.w >> 16)), l[7]);
        }
    } else {
        const float4 *p = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(base) + off);
#pragma unroll
        for (int i = 0; i < D / 8; ++i) {
            const float4 k0 = p[2 * i], k1 = p[2 * i + 1];
            const float4 a = *reinterpret_cast<const float4 *>(q + 8 * i), b = *reinterpret_cast<const float4 *>(q + 8 * i + 4);
            l[0] = __fmaf_rn(a.x, k0.x, l[0]); l[1] = __fmaf_rn(a.y, k0.y, l[1]); l[2] = __fmaf_rn(a.z, k0.z, l[2]); l[3] = __fmaf_rn(a.w, k0.w, l[3]);
            l[4] = __fmaf_rn(b.x, k1.x, l[4]); l[5] = __fmaf_rn(b.y, k1.y, l[5]); l[6] = __fmaf_rn(b.z, k1.z, l[6]); l[7] = __fmaf_rn(b.w, k1.w, l[7]);
        }
    }
    return ((l[0] + l[4]) + (l[1] + l[5])) + ((l[2] + l[6]) + (l[3] + l[7]));
}

// ------------------------------------------------------------------------------------------------------------------
// Sq == 1: __fa2_decode (:225-274 / :1346-1394) = the same recurrence with one key per tile.  One workgroup (NT threads) per head.
//   A  scores: one thread per key (8 chains over the key row, q broadcast from LDS);
//   B  m_j is a prefix maximum: DPP scan + wave totals; c_j = expf((m_{j-1} - m_j) scale) (exactly 1 when the maximum does not
//      move -- the usual case, flagged in a bit mask), p_j = expf((s_j - m_j) scale) -> LDS;
//   C  o[d] = fma(p_j, v_j[d], o[d] * c_j) and logsum = fma(logsum, c_j, p_j) are sequential in j: lanes d = 0..D-1 (and one
//      lane for logsum) walk the keys.  The walk must never wait on memory, so ALL threads stream V through an LDS ring of
//      FA_VSLOTS chunks of FA_VCH keys ([key][d], the rows as they lie in the slab): the first chunks are requested before
//      phase A and land while the scores are computed; later chunks are fetched FA_VSLOTS chunks ahead of the walk.
// knew / vnew (optional, LDS fp16 rows) stand for key position `tnew` (the row this step appends; other workgroups of the
// GQA group must not depend on the slab row being visible yet).
// LDS carve (DecodeLds): p[cap] c[cap] floats, cmask[cap/8] bytes, qs[D] ob[D] floats, wred[NT/64+2], V ring nslots*FA_VCH*D (fp16 or fp32)
// ------------------------------------------------------------------------------------------------------------------
constexpr int FA_VCH = 128;
struct DecodeLds {
    float *p, *c, *qs, *ob, *wred;
    uint64_t *etab;      // glibc_expf's 2^(i/32) table
    uint8_t *cmask;
    char *vring;
    int nslots;
};
__host__ __device__ static inline size_t decode_lds_fixed(int cap, int D, int NT) {
    const size_t capr = (size_t)((cap + 63) & ~63);
    return capr * 8 + capr / 8 + (size_t)(2 * D + NT / 64 + 2) * 4 + 64 + 256;
}
__host__ __device__ static inline size_t decode_lds_bytes(int cap, int D, int NT, int elt, int nslots, bool vt = false) {
    return ((decode_lds_fixed(cap, D, NT) + 15) & ~(size_t)15) + (size_t)nslots * (vt ? (size_t)D * (FA_VCH * 2 + 16) : (size_t)FA_VCH * D * elt);
}
static inline int decode_lds_slots(int cap, int D, int NT, int elt, bool vt) {
    int nslots = 4;
    while (nslots > 1 && decode_lds_bytes(cap, D, NT, elt, nslots, vt) > 160 * 1024) --nslots;
    return nslots;
}
__device__ __forceinline__ DecodeLds carve_decode(char *smem, int cap, int D, int NT, int nslots) {
    const size_t capr = (size_t)((cap + 63) & ~63);
    DecodeLds L;
    L.p = reinterpret_cast<float *>(smem);
    L.c = L.p + capr;
    L.cmask = reinterpret_cast<uint8_t *>(L.c + capr);
    L.qs = reinterpret_cast<float *>(smem + ((capr * 8 + capr / 8 + 15) & ~(size_t)15));
    L.ob = L.qs + D;
    L.wred = L.ob + D;
    L.etab = reinterpret_cast<uint64_t *>((reinterpret_cast<uintptr_t>(L.wred + NT / 64 + 2) + 15) & ~(uintptr_t)15);
    L.vring = smem + ((decode_lds_fixed(cap, D, NT) + 15) & ~(size_t)15);
    L.nslots = nslots;
    return L;
}

// VT: V is the engine's transposed fp16 slab, element (key j, dim d) at V[(kvoff + d) * ldv + j] (rows padded by >= 128 keys).
// The ring chunk is then [D][FA_VCH keys] (pitch FA_VPITCH bytes): parking is a plain 16-byte copy and the walker lane of dim d
// reads 8 keys per ds_read_b128; with the reference layout ([key][D] rows) the walker reads one fp16 / fp32 per key.
constexpr int FA_VPITCH = FA_VCH * 2 + 16;

// DV (transposed slab only): the value dims THIS workgroup walks.  The decode kernel splits a head's D dims over D / DV workgroups: every one of them computes
// the scores (it needs all of K) but fetches, parks and walks only its DV rows of V -- the front half of the kernel is bound by what one CU can pull from HBM.
template <int D, bool F16, int NT, bool VT, int DV = D>
struct DecodeGeom {
    static_assert(DV == D || VT, "a dim slice needs the transposed slab");
    static constexpr int ELT = F16 ? 2 : 4;
    static constexpr int ROWV = VT ? FA_VCH * 2 / 16 : D * ELT / 16;   // 16-byte vectors per ring row
    static constexpr int CHV = VT ? DV * ROWV : FA_VCH * ROWV;         // vectors per chunk
    static constexpr int VPT = (CHV + NT - 1) / NT;                    // vectors per thread per chunk
    static constexpr size_t SLOT = VT ? (size_t)DV * FA_VPITCH : (size_t)FA_VCH * D * ELT;
};
// What a thread requests from memory before anything else in the kernel (speculatively: rows below `cap` always exist): its half of
// the key row of pass 0 and its vectors of the first V chunks.  The loads fly while the prologue (rotary, barriers) runs.
template <int D, bool F16, int NT, bool VT, int DV = D>
struct DecodePrefetch {
    uint4 v[4][DecodeGeom<D, F16, NT, VT, DV>::VPT];
    uint2 k[F16 ? D / 8 : 1];     // direct form: this lane's half of its key row (strided 8-byte pieces)
    uint4 kc[VT ? D / 16 : 1];    // staged form: coalesced 16-byte pieces of the pass-0 key rows, parked in the (not yet used) V ring
    uint64_t etab;                // this lane's entry of glibc_expf's table (requested with the other early loads, stored to LDS later)
};
// pass-0 keys staged through LDS: rows [NT/2][D fp16] at pitch D*2+16 bytes in the ring area; needs the full 4-slot ring
template <int D>
__host__ __device__ constexpr int fa_kpitch() { return D * 2 + 16; }
template <int D, int NT>
__host__ __device__ constexpr bool fa_kstage_fits() { return (size_t)(NT / 2) * fa_kpitch<D>() <= (size_t)4 * D * (FA_VCH * 2 + 16); }
template <int D, bool F16, int NT, bool VT, int DV = D>
__device__ __forceinline__ void fa2_decode_fetch_v(uint4 *dst, const void *V, int64_t ldv, int kvoff, int ch, int cap) {
    using G = DecodeGeom<D, F16, NT, VT, DV>;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < G::VPT; ++i) {
        const int vi = tid + NT * i, row = vi / G::ROWV, part = vi % G::ROWV;
        if (vi < G::CHV) {
            if (VT) dst[i] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(V) + (int64_t)(kvoff + row) * ldv + ch * FA_VCH + part * 8);
            else dst[i] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(V) + ((int64_t)min(ch * FA_VCH + row, cap - 1) * ldv + kvoff) * G::ELT + part * 16);
        }
    }
}
// `part`: 0 = everything (speculative: rows below `cap` always exist), 1 = only the first NT/8 keys of the staged form (speculative), 2 = the rest of
// the staged form, limited to keys below `nkeys` (issued once the key count is known: at short contexts the fixed-size speculative fetch read up to 40x
// the bytes the step needs)
template <int D, bool F16, int NT, bool VT, int DV = D>
__device__ __forceinline__ void fa2_decode_prefetch(DecodePrefetch<D, F16, NT, VT, DV> &P, const void *K, int64_t ldk, const void *V, int64_t ldv, int kvoff,
                                                    int cap, int nslots, int part = 0, int nkeys = 0) {
    if (part != 2) P.etab = expf_tab_fetch();
    if (VT && nslots == 4 && fa_kstage_fits<D, NT>()) {
        // coalesced: one wave instruction = 1 KiB of consecutive key rows (the strided per-lane form below costs one cache line per lane)
        constexpr int ROWK = D * 2 / 16;
        constexpr int NSPEC = (D / 16) / 4 > 0 ? (D / 16) / 4 : 1;     // vectors per thread of the speculative part: NT * NSPEC / ROWK keys
#pragma unroll
        for (int i = 0; i < D / 16; ++i) {
            if (part == 1 && i >= NSPEC) continue;
            if (part == 2 && i < NSPEC) continue;
            const int vi = threadIdx.x + NT * i, key = min(vi / ROWK, cap - 1), part16 = vi % ROWK;
            if (part == 2 && key >= nkeys) continue;
            P.kc[i] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(K) + (int64_t)key * ldk + kvoff + part16 * 8);
        }
    } else if (F16 && part != 2) {
        const int j = min((int)threadIdx.x >> 1, cap - 1), hf = threadIdx.x & 1;
        const uint16_t *kp = reinterpret_cast<const uint16_t *>(K) + (int64_t)j * ldk + kvoff + 4 * hf;
#pragma unroll
        for (int i = 0; i < D / 8; ++i) P.k[i] = *reinterpret_cast<const uint2 *>(kp + 8 * i);
    }
    (void)V; (void)ldv;   // the V chunks are requested once Sk is known (fa2_decode_head): they are not needed before phase C
}

// ring slot of chunk ch: the usual 4-slot ring needs no division
__device__ __forceinline__ int fa_slot(int ch, int nslots) { return nslots == 4 ? (ch & 3) : ch % nslots; }

// voff: first V row of this workgroup (kvoff + its dim slice); vnew points at the slice's first element
template <int D, bool F16, int NT, bool VT = false, int DV = D>
__device__ __forceinline__ void fa2_decode_head(const DecodeLds &L, DecodePrefetch<D, F16, NT, VT, DV> &P, const void *K, int64_t ldk, const void *V,
                                                int64_t ldv, int kvoff, int voff, int Sk, int cap, const uint16_t *knew, const uint16_t *vnew, int tnew) {
    static_assert(!VT || F16, "the transposed slab is fp16");
    using G = DecodeGeom<D, F16, NT, VT, DV>;
    constexpr int VPT = G::VPT, ROWV = G::ROWV, CHV = G::CHV;
    constexpr size_t SLOT = G::SLOT;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float scale = 1.0f / sqrtf((float)D);
    const int nkv = (VT && vnew) ? Sk - 1 : Sk;     // keys whose V comes from memory (VT: the appended token is the last key, walked from vnew)
    const int nch = (nkv + FA_VCH - 1) / FA_VCH;
    const int npre = min(nch, L.nslots);
    auto park_chunk = [&](int ch, const uint4 (&src)[VPT]) {
        char *slot = L.vring + (size_t)fa_slot(ch, L.nslots) * SLOT;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int vi = tid + NT * i, row = vi / ROWV, part = vi % ROWV;
            if (vi < CHV) {
                uint4 v = src[i];
                if (!VT && F16 && vnew && ch * FA_VCH + row == tnew) v = reinterpret_cast<const uint4 *>(vnew)[part];
                *reinterpret_cast<uint4 *>(slot + (VT ? (size_t)row * FA_VPITCH + part * 16 : (size_t)vi * 16)) = v;
            }
        }
    };
    STAMP(0);
    const uint64_t etab_v = P.etab;
    const bool kstaged = VT && L.nslots == 4 && fa_kstage_fits<D, NT>();
    if (kstaged) {
        constexpr int ROWK = D * 2 / 16;
#pragma unroll
        for (int i = 0; i < (VT ? D / 16 : 1); ++i) {
            const int vi = tid + NT * i;
            *reinterpret_cast<uint4 *>(L.vring + (size_t)(vi / ROWK) * fa_kpitch<D>() + (vi % ROWK) * 16) = P.kc[i];
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (s < npre) fa2_decode_fetch_v<D, F16, NT, VT, DV>(P.v[s], V, ldv, voff, s, cap);
    if (kstaged) __syncthreads();
    // ---- A + B: two lanes per key (chains l = 0..3 and 4..7), NT/2 keys per pass ------------------------------------------------------
    float carry = FA_NEG;
    for (int base = 0; base < Sk; base += NT / 2) {
        const int j = base + (tid >> 1), hf = tid & 1;
        float s = FA_NEG;
        if (j < Sk) {
            float l[4] = {0, 0, 0, 0};
            const float *q = L.qs + 4 * hf;
            if (F16) {
                uint2 kw[D / 8];
                if (knew && j == tnew) {
#pragma unroll
                    for (int i = 0; i < D / 8; ++i) kw[i] = *reinterpret_cast<const uint2 *>(knew + 8 * i + 4 * hf);
                } else if (base == 0 && kstaged) {
                    const char *kr = L.vring + (size_t)(tid >> 1) * fa_kpitch<D>() + 8 * hf;
#pragma unroll
                    for (int i = 0; i < D / 8; ++i) kw[i] = *reinterpret_cast<const uint2 *>(kr + 16 * i);
                } else if (base == 0) {
#pragma unroll
                    for (int i = 0; i < D / 8; ++i) kw[i] = P.k[i];
                } else {
                    const uint16_t *kp = reinterpret_cast<const uint16_t *>(K) + (int64_t)j * ldk + kvoff + 4 * hf;
#pragma unroll
                    for (int i = 0; i < D / 8; ++i) kw[i] = *reinterpret_cast<const uint2 *>(kp + 8 * i);
                }
#pragma unroll
                for (int i = 0; i < D / 8; ++i) {
                    const float4 a = *reinterpret_cast<const float4 *>(q + 8 * i);
                    l[0] = __fmaf_rn(a.x, h2f((uint16_t)(kw[i].x & 0xffff)), l[0]); l[1] = __fmaf_rn(a.y, h2f((uint16_t)(kw[i].x >> 16)), l[1]);
                    l[2] = __fmaf_rn(a.z, h2f((uint16_t)(kw[i].y & 0xffff)), l[2]); l[3] = __fmaf_rn(a.w, h2f((uint16_t)(kw[i].y >> 16)), l[3]);
                }
            } else {
                const float *kp = reinterpret_cast<const float *>(K) + (int64_t)j * ldk + kvoff + 4 * hf;
#pragma unroll
                for (int i = 0; i < D / 8; ++i) {
                    const float4 w = *reinterpret_cast<const float4 *>(kp + 8 * i);
                    const float4 a = *reinterpret_cast<const float4 *>(q + 8 * i);
                    l[0] = __fmaf_rn(a.x, w.x, l[0]); l[1] = __fmaf_rn(a.y, w.y, l[1]); l[2] = __fmaf_rn(a.z, w.z, l[2]); l[3] = __fmaf_rn(a.w, w.w, l[3]);
                }
            }
            // _mm256_hadd_ps: ((l0+l4)+(l1+l5)) + ((l2+l6)+(l3+l7)); the partner lane holds the other half of the chains
            const float r0 = l[0] + MH_DPPF(0.0f, l[0], DPP_QUAD_X1, 0xF), r1 = l[1] + MH_DPPF(0.0f, l[1], DPP_QUAD_X1, 0xF);
            const float r2 = l[2] + MH_DPPF(0.0f, l[2], DPP_QUAD_X1, 0xF), r3 = l[3] + MH_DPPF(0.0f, l[3], DPP_QUAD_X1, 0xF);
            s = (r0 + r1) + (r2 + r3);
        }
        STAMP(2);
        // prefix maximum over keys = over lane pairs: scan the wave (both lanes of a pair hold the same s), then the wave totals
        const float wincl = wave_scan_max(s);
        if (lane == 63) L.wred[wid] = wincl;
        if (base == 0) expf_tab_store(L.etab, etab_v);
        __syncthreads();
        float before = carry, tot = carry;
#pragma unroll
        for (int w4 = 0; w4 < NT / 256; ++w4) {
            const float4 t4 = *reinterpret_cast<const float4 *>(L.wred + 4 * w4);
            const float tv[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                tot = fmaxf(tot, tv[e]);
                before = (4 * w4 + e < wid) ? fmaxf(before, tv[e]) : before;
            }
        }
        const float incl = fmaxf(wincl, before);
        // value before this KEY: lane pair (2k, 2k+1) -> inclusive value of lane 2k-1
        const float up1 = wave_shift_up(incl, before);
        const float up2 = wave_shift_up(up1, before);
        const float excl = hf ? up2 : up1;
        const bool moved = j < Sk && excl != incl;
        // one expf per lane: the even lane of a key's pair takes p, the odd lane c (expf(0) is exactly 1.0f when the maximum stayed)
        if (j < Sk) {
            const float e = glibc_expf(((hf ? excl : s) - incl) * scale, L.etab);
            (hf ? L.c : L.p)[j] = e;
        }
        // one mask bit per key: even lanes' ballot bits, compacted
        unsigned long long mv = __ballot(moved && hf == 0);
        mv = (mv | (mv >> 1)) & 0x3333333333333333ull; mv = (mv | (mv >> 2)) & 0x0f0f0f0f0f0f0f0full; mv = (mv | (mv >> 4)) & 0x00ff00ff00ff00ffull;
        mv = (mv | (mv >> 8)) & 0x0000ffff0000ffffull; mv = (mv | (mv >> 16)) & 0x00000000ffffffffull;
        if (lane < 4 && base + (wid << 5) < ((Sk + 31) & ~31)) L.cmask[((base + (wid << 5)) >> 3) + lane] = (uint8_t)(mv >> (8 * lane));
        carry = tot;
        __syncthreads();
    }
    STAMP(3);
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (s < npre) park_chunk(s, P.v[s]);
    __syncthreads();
    STAMP(4);
    // ---- C ----------------------------------------------------------------------------------------------------------------------
    // c_j is stored as exactly 1.0f when the maximum did not move, so "o * c" and "fma(logsum, c, p)" may be evaluated for every key
    // (x * 1.0f == x, fma(x, 1.0f, p) == x + p bit for bit); the mask only spares the common step the LDS reads of c.
    float o = 0.0f, lsum = 0.0f;
    const bool walker = tid < DV, summer = tid == ((DV + 63) & ~63);
    uint4 vref[VPT];
    for (int ch = 0; ch < nch; ++ch) {
        const bool refill = ch + L.nslots < nch;
        if (refill) fa2_decode_fetch_v<D, F16, NT, VT, DV>(vref, V, ldv, voff, ch + L.nslots, cap);
        const int j0 = ch * FA_VCH, n = min(FA_VCH, nkv - j0);
        // the chunk's 128 mask bits, wave-uniform in SGPRs (one LDS read per chunk instead of one per step on the critical path)
        unsigned long long chunk_mask_lo, chunk_mask_hi;
        {
            typedef unsigned int u32x4m __attribute__((ext_vector_type(4)));
            const u32x4m mw = *reinterpret_cast<const u32x4m *>(L.cmask + (j0 >> 3));
            chunk_mask_lo = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(mw[0]) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(mw[1]) << 32);
            chunk_mask_hi = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(mw[2]) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(mw[3]) << 32);
        }
        if (walker && VT) {
            // transposed ring, 16 keys per step, LDS reads issued TWO steps ahead of the fma chain that consumes them (a three-stage register
            // ring: raw fp16 pairs + the 16 p's + the mask), so the dependent chain never waits on LDS latency
            const char *row = L.vring + (size_t)fa_slot(ch, L.nslots) * SLOT + (size_t)tid * FA_VPITCH;
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            struct Stage { u32x4 v0, v1; f32x4 p[4]; };
            Stage S0, S1, S2;
            const unsigned long long cm_lo = chunk_mask_lo, cm_hi = chunk_mask_hi;
            auto rd = [&](int k16, Stage &S) {
                const int kk = min(k16, FA_VCH - 16);
                S.v0 = *reinterpret_cast<const u32x4 *>(row + kk * 2);
                S.v1 = *reinterpret_cast<const u32x4 *>(row + kk * 2 + 16);
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) S.p[q4] = *reinterpret_cast<const f32x4 *>(L.p + j0 + kk + 4 * q4);
            };
            auto step = [&](int k16, const Stage &S) {
                const unsigned w[8] = {S.v0[0], S.v0[1], S.v0[2], S.v0[3], S.v1[0], S.v1[1], S.v1[2], S.v1[3]};
                const int m16 = (int)(((k16 & 64) ? cm_hi : cm_lo) >> (k16 & 63)) & 0xffff;
                if (m16 == 0 && k16 + 16 <= n) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) o = __fmaf_rn(S.p[k >> 2][k & 3], h2f((uint16_t)((k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xffff))), o);
                } else if (k16 + 16 <= n) {
                    // some maximum moved inside these 16 keys: c is exactly 1.0f wherever it did not, so the rescale needs no select
                    f32x4 cq[4];
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) cq[q4] = *reinterpret_cast<const f32x4 *>(L.c + j0 + k16 + 4 * q4);
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        o = o * cq[k >> 2][k & 3];
                        o = __fmaf_rn(S.p[k >> 2][k & 3], h2f((uint16_t)((k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xffff))), o);
                    }
                } else {
                    f32x4 cq[4];
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) cq[q4] = *reinterpret_cast<const f32x4 *>(L.c + j0 + k16 + 4 * q4);
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const bool in = k16 + k < n;            // keys past the end: c = 1, p = 0, v = 0
                        const float vk = h2f((uint16_t)((k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xffff)));
                        o = o * (in ? cq[k >> 2][k & 3] : 1.0f);
                        o = __fmaf_rn(in ? S.p[k >> 2][k & 3] : 0.0f, in ? vk : 0.0f, o);
                    }
                }
            };
            rd(0, S0);
            rd(16, S1);
            for (int k16 = 0; k16 < n; k16 += 48) {
                rd(k16 + 32, S2);
                step(k16, S0);
                if (k16 + 16 < n) { rd(k16 + 48, S0); step(k16 + 16, S1); }
                if (k16 + 32 < n) { rd(k16 + 64, S1); step(k16 + 32, S2); }
            }
        } else if (walker) {
            // 16 keys per step, the next step's LDS reads issued before this step's fma chain
            const char *slot = L.vring + (size_t)fa_slot(ch, L.nslots) * SLOT;
            float va[16], vb[16];
            float4 pa[4], pb[4];
            int ma, mb;
            auto rd = [&](int k16, float (&vv)[16], float4 (&pp)[4], int &mm) {
                const int kk = min(k16, FA_VCH - 16);
                if (VT) {
                    const uint4 w0 = *reinterpret_cast<const uint4 *>(slot + (size_t)tid * FA_VPITCH + kk * 2);
                    const uint4 w1 = *reinterpret_cast<const uint4 *>(slot + (size_t)tid * FA_VPITCH + kk * 2 + 16);
                    const uint32_t w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) { vv[2 * e] = h2f((uint16_t)(w[e] & 0xffff)); vv[2 * e + 1] = h2f((uint16_t)(w[e] >> 16)); }
                } else {
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int jl = kk + k;
                        vv[k] = F16 ? (float)reinterpret_cast<const _Float16 *>(slot)[jl * D + tid] : reinterpret_cast<const float *>(slot)[jl * D + tid];
                    }
                }
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) pp[q4] = *reinterpret_cast<const float4 *>(L.p + j0 + kk + 4 * q4);
                mm = *reinterpret_cast<const uint16_t *>(L.cmask + ((j0 + kk) >> 3));
            };
            auto step = [&](int k16, const float (&vv)[16], const float4 (&pp)[4], int mm) {
                const float ps[16] = {pp[0].x, pp[0].y, pp[0].z, pp[0].w, pp[1].x, pp[1].y, pp[1].z, pp[1].w,
                                      pp[2].x, pp[2].y, pp[2].z, pp[2].w, pp[3].x, pp[3].y, pp[3].z, pp[3].w};
                const int m16 = __builtin_amdgcn_readfirstlane(mm);
                if (m16 == 0 && k16 + 16 <= n) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) o = __fmaf_rn(ps[k], vv[k], o);
                } else {
                    float4 cq[4];
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) cq[q4] = *reinterpret_cast<const float4 *>(L.c + j0 + k16 + 4 * q4);
                    const float cs[16] = {cq[0].x, cq[0].y, cq[0].z, cq[0].w, cq[1].x, cq[1].y, cq[1].z, cq[1].w,
                                          cq[2].x, cq[2].y, cq[2].z, cq[2].w, cq[3].x, cq[3].y, cq[3].z, cq[3].w};
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const bool in = k16 + k < n;            // keys past the end: c = 1, p = 0, v = 0 (LDS / padding may hold anything)
                        o = o * (in ? cs[k] : 1.0f);
                        o = __fmaf_rn(in ? ps[k] : 0.0f, in ? vv[k] : 0.0f, o);
                    }
                }
            };
            rd(0, va, pa, ma);
            for (int k16 = 0; k16 < n; k16 += 32) {
                rd(k16 + 16, vb, pb, mb);
                step(k16, va, pa, ma);
                if (k16 + 16 < n) {
                    rd(k16 + 32, va, pa, ma);
                    step(k16 + 16, vb, pb, mb);
                }
            }
        }
        else if (summer) {
            // logsum = fma(logsum, c, p).  The last chunk also takes the appended key.  The p's (and the mask) of the next 16 keys are
            // read before this step's dependent adds, so the lane pays the LDS latency once per chunk instead of once per step.
            const int ns = ch == nch - 1 ? Sk - j0 : n;
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            struct SStage { f32x4 p[4]; };
            SStage A, B;
            auto rd = [&](int k16, SStage &S) {
                const int kk = min(k16, ((ns + 15) & ~15) - 16);
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) S.p[q4] = *reinterpret_cast<const f32x4 *>(L.p + j0 + kk + 4 * q4);
            };
            auto step = [&](int k16, const SStage &S) {
                const int m16 = (int)(((k16 & 64) ? chunk_mask_hi : chunk_mask_lo) >> (k16 & 63)) & 0xffff;
                if (m16 == 0 && k16 + 16 <= ns) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) lsum = lsum + S.p[k >> 2][k & 3];
                } else {
                    f32x4 cq[4];
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) cq[q4] = *reinterpret_cast<const f32x4 *>(L.c + j0 + k16 + 4 * q4);
                    if (k16 + 16 <= ns) {
#pragma unroll
                        for (int k = 0; k < 16; ++k) lsum = __fmaf_rn(lsum, cq[k >> 2][k & 3], S.p[k >> 2][k & 3]);
                    } else {
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            const bool in = k16 + k < ns;
                            lsum = __fmaf_rn(lsum, in ? cq[k >> 2][k & 3] : 1.0f, in ? S.p[k >> 2][k & 3] : 0.0f);
                        }
                    }
                }
            };
            if (ns > 0) rd(0, A);
            for (int k16 = 0; k16 < ns; k16 += 32) {
                rd(k16 + 16, B);
                step(k16, A);
                if (k16 + 16 < ns) { rd(k16 + 32, A); step(k16 + 16, B); }
            }
        }
        if (ch == nch - 1) { STAMPT(5, 0); STAMPT(6, DV > 64 ? 64 : 0); STAMPT(7, (DV + 63) & ~63); }
        if (nch > L.nslots) __syncthreads();   // only a ring that gets refilled needs the walkers and the parking threads to meet per chunk (T <= nslots * 128 keys: never)
        if (refill) park_chunk(ch + L.nslots, vref);
    }
    if (walker && VT && vnew) {   // the appended token: rescale (always a multiply in the reference), then its value row
        o = o * L.c[Sk - 1];
        o = __fmaf_rn(L.p[Sk - 1], h2f(vnew[tid]), o);
    }
    if (summer) {
        if (nch == 0) lsum = __fmaf_rn(lsum, L.c[0], L.p[0]);   // Sk == 1 with the appended key only
        L.wred[NT / 64] = lsum;
    }
    __syncthreads();
    if (walker) L.ob[tid] = o * (1.0f / L.wred[NT / 64]);
    __syncthreads();
}
// ==================================================================================================================
// The same decode step with its two halves overlapped (round 3).  fa2_decode_head above runs A (scores of all keys) -> B (prefix maximum, p, c) ->
// park V -> C (the sequential walk): the walk, bound by one wave's dependent fma chain, starts when the last score is done.  Here a workgroup is a small
// pipeline over BLOCKS of FP_B = 32 keys:
//   producer waves (all but the walkers and the logsum wave)   block b of wave w = b mod NP: the block's key rows and its V^T slice are requested from memory
//       (coalesced 16-byte pieces, both speculatively at kernel entry for the first round), the key rows staged in the wave's own LDS region, two lanes per key
//       compute the score (the chains and the pair adds of phase A), the wave scans its block maximum and publishes it; once every earlier block has published,
//       the carry is their maximum and p_j / c_j / the moved-bits follow exactly as in phase B; the V^T slice is parked over the (now dead) key rows of the
//       region and the block is flagged READY.
//   walker wave(s)    lane d walks o[d] = fma(p_j, v_j[d], o[d] * c_j) over the keys in order, block after block as they become READY (three-stage register ring of
//       16-key steps as above); after a block it publishes `walked`, which lets the producer of block b + NP reuse the region.
//   logsum lane        the same walk over p / c only.
// Nothing in the arithmetic changes -- every score, every maximum, every expf argument and the order of the dependent fma chain are those of fa2_decode_head -- so
// the results are bit-identical; what changes is that the walk of key 0 starts as soon as block 0 is through, while the other blocks' scores are still being computed.
// Hand-offs are words in LDS written after `s_waitcnt lgkmcnt(0)` (LDS is one memory per CU: no cache to invalidate); the only workgroup barriers are the one behind
// the rotary prologue and the one in front of the final normalisation.
// ==================================================================================================================
constexpr int FP_B = 32;
template <int D, int DV>
struct PipeGeom {
    static constexpr int KPITCH = D * 2 + 16;            // bytes per staged key row: conflict-free 8-byte reads by (key, half) lanes
    static constexpr int VPITCH = FP_B * 2 + 16;         // bytes per V^T row of a block (32 keys fp16 + pad): conflict-free 16-byte reads by lane = dim
    static constexpr int KBYTES = FP_B * KPITCH, VBYTES = DV * VPITCH;
    static constexpr int REGION = ((KBYTES > VBYTES ? KBYTES : VBYTES) + 15) & ~15;
    static constexpr int ROWK = D * 2 / 16;              // 16-byte pieces per key row
    static constexpr int NKV = FP_B * ROWK / 64;         // key pieces per lane per block
    static constexpr int ROWV = FP_B * 2 / 16;           // 16-byte pieces per V^T row of a block (4)
    static constexpr int NVV = (DV * ROWV + 63) / 64;    // V pieces per lane per block
};
// hand-off words live in LDS and are read / written with explicit ds_ instructions (a volatile access through the generic pointer is lowered to flat_load ... sc0 sc1)
__device__ __forceinline__ uint32_t lds_ld_u32(const uint32_t *p) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)p) : "memory");
    return v;
}
// every LDS store this wave issued before is complete (and visible to the other waves of the CU) before the word is written
__device__ __forceinline__ void lds_publish_u32(uint32_t *p, uint32_t v) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" : : "v"((uint32_t)(uintptr_t)p), "v"(v) : "memory");
}
__device__ __forceinline__ uint64_t lds_ld_u64(const uint32_t *p) {
    uint64_t v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)p) : "memory");
    return v;
}
// one more block READY: bit `bit` of the 64-bit word at p (two dwords), set after every LDS store this wave issued before has completed
__device__ __forceinline__ void lds_publish_bit(uint32_t *p, int bit) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_or_b32 %0, %1" : : "v"((uint32_t)(uintptr_t)(p + (bit >> 5))), "v"(1u << (bit & 31)) : "memory");
}
// a progress word nobody has to order anything behind (the reads it vouches for have already returned into registers)
__device__ __forceinline__ void lds_st_u32(uint32_t *p, uint32_t v) {
    asm volatile("ds_write_b32 %0, %1" : : "v"((uint32_t)(uintptr_t)p), "v"(v) : "memory");
}
struct PipeLds {
    float *p, *c, *tot, *qs, *ob, *lsum;
    uint32_t *flagT, *walked;
    uint2 *rm;      // per block {READY word, moved-bits of its 32 keys}: one 8-byte read gives a consumer both, the READY word first in program order of its stage
    uint64_t *etab;
    char *regions;
};
__host__ __device__ static inline int pipe_nblk(int cap) { return (cap + FP_B - 1) / FP_B + 1; }
__host__ __device__ static inline size_t pipe_lds_fixed(int cap, int D, int DV) {
    const size_t capr = (size_t)((cap + 63) & ~63), nb = (size_t)((pipe_nblk(cap) + 3) & ~3);
    return capr * 8 + nb * 16 + 16 + (size_t)(D + DV) * 4 + 16 + 256 + 64;      // p c | tot flagT rm(2) | walked | qs ob | lsum | etab | slack
}
template <int D, int DV>
__host__ __device__ static inline size_t pipe_lds_bytes(int cap, int np) { return ((pipe_lds_fixed(cap, D, DV) + 15) & ~(size_t)15) + (size_t)np * PipeGeom<D, DV>::REGION; }
__device__ __forceinline__ PipeLds carve_pipe(char *smem, int cap, int D, int DV) {
    const size_t capr = (size_t)((cap + 63) & ~63), nb = (size_t)((pipe_nblk(cap) + 3) & ~3);
    PipeLds L;
    L.p = reinterpret_cast<float *>(smem);
    L.c = L.p + capr;
    L.tot = reinterpret_cast<float *>(L.c + capr);
    L.flagT = reinterpret_cast<uint32_t *>(L.tot + nb);
    L.rm = reinterpret_cast<uint2 *>(L.flagT + nb);
    L.walked = reinterpret_cast<uint32_t *>(L.rm + nb);
    L.qs = reinterpret_cast<float *>(L.walked + 4);
    L.ob = L.qs + D;
    L.lsum = L.ob + DV;
    L.etab = reinterpret_cast<uint64_t *>((reinterpret_cast<uintptr_t>(L.lsum + 4) + 15) & ~(uintptr_t)15);
    L.regions = smem + ((pipe_lds_fixed(cap, D, DV) + 15) & ~(size_t)15);
    return L;
}
// what a producer lane holds of one block between the request and the LDS stores
typedef unsigned int pipe_u32x4 __attribute__((ext_vector_type(4)));      // a native vector: assignments stay loads / stores of <4 x i32> (HIP's uint4 struct copies as memcpy,
                                                                          // which kept the loop-carried block registers in scratch memory)
template <int D, int DV>
struct PipeRegs {
    pipe_u32x4 k[PipeGeom<D, DV>::NKV];
    pipe_u32x4 v[PipeGeom<D, DV>::NVV];
};
template <int D, int DV>
__device__ __forceinline__ void pipe_fetch_k(PipeRegs<D, DV> &R, const uint16_t *K, int64_t ldk, int kvoff, int b, int cap, int lane) {
    using G = PipeGeom<D, DV>;
#pragma unroll
    for (int i = 0; i < G::NKV; ++i) {
        const int vi = lane + 64 * i, key = min(b * FP_B + vi / G::ROWK, cap - 1), part = vi % G::ROWK;
        R.k[i] = *reinterpret_cast<const pipe_u32x4 *>(K + (int64_t)key * ldk + kvoff + part * 8);
    }
}
template <int D, int DV>
__device__ __forceinline__ void pipe_fetch_v(PipeRegs<D, DV> &R, const uint16_t *V, int64_t ldv, int voff, int b, int lane) {
    using G = PipeGeom<D, DV>;
#pragma unroll
    for (int i = 0; i < G::NVV; ++i) {
        const int vi = lane + 64 * i, row = vi / G::ROWV, part = vi % G::ROWV;
        if (vi < DV * G::ROWV) R.v[i] = *reinterpret_cast<const pipe_u32x4 *>(V + (int64_t)(voff + row) * ldv + b * FP_B + part * 8);      // rows are padded by >= 128 keys
    }
}

// NT threads: waves [0, NWK) walk the DV value dims, wave NWK holds the logsum lane, the rest produce.  `R` holds the first-round block of a producer wave, requested
// by the caller before its prologue (fa2_pipe_prefetch); `first_block` is that block's index (< 0: none).
template <int D, int DV, int NT>
__device__ __forceinline__ void fa2_decode_head_pipe(const PipeLds &L, PipeRegs<D, DV> &R, const uint16_t *K, int64_t ldk, const uint16_t *V, int64_t ldv, int kvoff,
                                                     int voff, int Sk, int cap, const uint16_t *knew, const uint16_t *vnew, int tnew) {
    using G = PipeGeom<D, DV>;
    constexpr int NWK = (DV + 63) / 64, NP = NT / 64 - NWK - 1;
    static_assert(NP >= 2, "the pipeline needs at least two producer waves");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float scale = 1.0f / sqrtf((float)D);
    // (every key's value, the appended token's included, is met inside its block: the producers patch vnew into the parked slice)
    const int nblkS = (Sk + FP_B - 1) / FP_B;                // blocks with scores
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    // READY words (L.rm[b].x, one per block).  A consumer keeps `known`, the count of leading blocks it has seen READY, and goes back to LDS only when it needs a block
    // beyond that: one wave-wide read of the next 64 words, whose leading ones extend `known`.  With the producers ahead that is a handful of polls per launch, so
    // the read that drains the wave's LDS queue is off the per-block path, and no inline-asm LDS access sits on it either (the compiler's lgkmcnt bookkeeping stays
    // exact, which is what keeps the read-ahead ahead).
    int known = 0;
    auto ensure = [&](int b) {
        while (b >= known) {
            asm volatile("" ::: "memory");
            const int idx = known + lane;
            const uint32_t f = idx < nblkS ? L.rm[idx].x : 0u;
            const unsigned long long ball = __ballot(f != 0);
            const int cnt = ball == ~0ull ? 64 : __builtin_ctzll(~ball);
            if (cnt == 0) __builtin_amdgcn_s_sleep(1);
            else known += cnt;
        }
        asm volatile("" ::: "memory");
    };
    if (wid > NWK) {
        // ---------------------------------------------------------------- producers ----------------------------------------------------------------
        const int pw = wid - NWK - 1;
        char *region = L.regions + (size_t)pw * G::REGION;
        const int key = lane >> 1, hf = lane & 1;
        for (int b = pw; b < nblkS; b += NP) {
            if (b != pw) {      // later rounds: request the block now (the first round's was requested at kernel entry), then wait for the walker to leave the region
                pipe_fetch_k<D, DV>(R, K, ldk, kvoff, b, cap, lane);
                pipe_fetch_v<D, DV>(R, V, ldv, voff, b, lane);
                while (true) {      // every walker wave has left block b - NP
                    bool free = true;
#pragma unroll
                    for (int w = 0; w < NWK; ++w) free = free && (int)lds_ld_u32(L.walked + w) >= b - NP + 1;
                    if (free) break;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            // key rows -> the region (pitch KPITCH), then each (key, half) lane reads its 8-byte pieces back
#pragma unroll
            for (int i = 0; i < G::NKV; ++i) {
                const int vi = lane + 64 * i;
                *reinterpret_cast<pipe_u32x4 *>(region + (size_t)(vi / G::ROWK) * G::KPITCH + (vi % G::ROWK) * 16) = R.k[i];
            }
            if (b == pw) STAMPT(6, (NWK + 1) * 64);
            const int j = b * FP_B + key;
            float s = FA_NEG;
            {
                float l[4] = {0, 0, 0, 0};
                const float *q = L.qs + 4 * hf;
                uint2 kw[D / 8];
                if (knew && j == tnew) {
#pragma unroll
                    for (int i = 0; i < D / 8; ++i) kw[i] = *reinterpret_cast<const uint2 *>(knew + 8 * i + 4 * hf);
                } else {
                    const char *kr = region + (size_t)key * G::KPITCH + 8 * hf;
#pragma unroll
                    for (int i = 0; i < D / 8; ++i) kw[i] = *reinterpret_cast<const uint2 *>(kr + 16 * i);
                }
#ifdef MLLM_HIP_STAMPS
                if (b == pw) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); STAMPT(11, (NWK + 1) * 64); }
#endif
#pragma unroll
                for (int i = 0; i < D / 8; ++i) {
                    const float4 a = *reinterpret_cast<const float4 *>(q + 8 * i);
                    l[0] = __fmaf_rn(a.x, h2f((uint16_t)(kw[i].x & 0xffff)), l[0]); l[1] = __fmaf_rn(a.y, h2f((uint16_t)(kw[i].x >> 16)), l[1]);
                    l[2] = __fmaf_rn(a.z, h2f((uint16_t)(kw[i].y & 0xffff)), l[2]); l[3] = __fmaf_rn(a.w, h2f((uint16_t)(kw[i].y >> 16)), l[3]);
                }
                // _mm256_hadd_ps: ((l0+l4)+(l1+l5)) + ((l2+l6)+(l3+l7)); the partner lane holds the other half of the chains
                const float r0 = l[0] + MH_DPPF(0.0f, l[0], DPP_QUAD_X1, 0xF), r1 = l[1] + MH_DPPF(0.0f, l[1], DPP_QUAD_X1, 0xF);
                const float r2 = l[2] + MH_DPPF(0.0f, l[2], DPP_QUAD_X1, 0xF), r3 = l[3] + MH_DPPF(0.0f, l[3], DPP_QUAD_X1, 0xF);
                if (j < Sk) s = (r0 + r1) + (r2 + r3);
            }
            if (b == pw) STAMPT(8, (NWK + 1) * 64);
            // the block's running maximum; its total goes out first, so that later blocks can take their carry while this one still computes its expf
            const float wincl = wave_scan_max(s);
            if (lane == 63) { L.tot[b] = wincl; lds_publish_u32(L.flagT + b, 1u); }
            float before = FA_NEG;
            if (b > 0) {
                const bool mine = lane < b;      // b <= 64: cap <= 2048 keys (the launcher keeps longer caches on fa2_decode_head)
                while (true) {
                    const uint32_t f = lds_ld_u32(L.flagT + (mine ? lane : 0));
                    if (__ballot(f != 0 || !mine) == ~0ull) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                const float t = mine ? L.tot[lane] : FA_NEG;
                before = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_scan_max(t)), 63));
            }
            if (b == pw) { STAMPT(9, (NWK + 1) * 64); STAMPT(12, (NWK + 5) * 64); }
            const float incl = fmaxf(wincl, before);
            const float up1 = wave_shift_up(incl, before);
            const float up2 = wave_shift_up(up1, before);
            const float excl = hf ? up2 : up1;
            const bool moved = j < Sk && excl != incl;
            if (j < Sk) {
                const float e = glibc_expf(((hf ? excl : s) - incl) * scale, L.etab);
                (hf ? L.c : L.p)[j] = e;
            } else {
                (hf ? L.c : L.p)[j] = hf ? 1.0f : 0.0f;      // behind the last key: o * 1 and + 0 * v leave the walk's accumulators as they are, so every block is walked as a full one
            }
            unsigned long long mv = __ballot(moved && hf == 0);
            mv = (mv | (mv >> 1)) & 0x3333333333333333ull; mv = (mv | (mv >> 2)) & 0x0f0f0f0f0f0f0f0full; mv = (mv | (mv >> 4)) & 0x00ff00ff00ff00ffull;
            mv = (mv | (mv >> 8)) & 0x0000ffff0000ffffull; mv = (mv | (mv >> 16)) & 0x00000000ffffffffull;
            if (lane == 0) L.rm[b].y = (uint32_t)mv;
            // the V^T slice over the dead key rows (every lane's key reads above have returned: their values were consumed by the score)
#pragma unroll
            for (int i = 0; i < G::NVV; ++i) {
                const int vi = lane + 64 * i;
                if (vi < DV * G::ROWV) *reinterpret_cast<pipe_u32x4 *>(region + (size_t)(vi / G::ROWV) * G::VPITCH + (vi % G::ROWV) * 16) = R.v[i];
            }
            // the appended token's value row is not in the slab yet (and another workgroup may be storing it right now): its column of the parked slice comes from vnew,
            // so the walk meets it inside its block like any other key (c and p of key Sk - 1 are the reference's rescale-then-add of the new token)
            if (tnew >= b * FP_B && tnew < (b + 1) * FP_B)
                for (int d = lane; d < DV; d += 64) *reinterpret_cast<uint16_t *>(region + (size_t)d * G::VPITCH + 2 * (tnew - b * FP_B)) = vnew[d];
            // (s_waitcnt lgkmcnt(0) covers the whole wave's stores: the wave executes it as one instruction)
            if (lane == 0) lds_publish_u32(&L.rm[b].x, 1u);
            if (b == pw) { STAMPT(10, (NWK + 1) * 64); STAMPT(13, (NWK + 5) * 64); }
        }
    } else if (wid < NWK) {
        // ---------------------------------------------------------------- walkers ------------------------------------------------------------------
        // One block (32 keys) per stage, the next block's LDS reads issued before this block's 32-long fma chain (two register stages): a lone wave issues about one
        // instruction per six cycles whatever its kind, so the walk is priced by instructions per key -- per block: 32 fma, 13 LDS reads, one READY / moved-bits
        // check and the loop, instead of that per 16 keys.
        float o = 0.0f;
        const int nblkV = nblkS;      // every block is walked as a full one (p = 0, c = 1 behind the last key; the appended key's value patched into its block)
        struct Stage { u32x4 v[4]; f32x4 p[8]; uint32_t m; };
        Stage SA, SB;
        const bool active = tid < DV;
        const char *row0 = L.regions + (size_t)(active ? tid : 0) * G::VPITCH;
        auto rd = [&](int b, Stage &S) {
            const char *row = row0 + (size_t)(b % NP) * G::REGION;
#pragma unroll
            for (int i = 0; i < 4; ++i) S.v[i] = *reinterpret_cast<const u32x4 *>(row + 16 * i);
#pragma unroll
            for (int q4 = 0; q4 < 8; ++q4) S.p[q4] = *reinterpret_cast<const f32x4 *>(L.p + FP_B * b + 4 * q4);
            S.m = L.rm[b].y;
        };
        // Order inside an iteration: (1) this block's stage is waited for -- `landed` names its last-requested registers, and LDS returns a wave's reads in order --,
        // (2) the next block's reads are issued, (3) the 32-long chain runs on registers that need no further wait.  In the common case (no maximum moved inside the
        // block, the block is full) (2) and (3) share one basic block and the 13 reads are interleaved with the dependent fmas (sched_group_barrier): a lone wave
        // issues in order, so an independent instruction placed between two dependent fmas issues in the shadow of the first one's latency; placed in front of the
        // chain it costs its own issue slots.  The chain's 6.6 cycles per key is then what a block costs.
        auto landed = [&](const Stage &S) { asm volatile("" : : "v"(S.p[7]), "v"(S.m) : "memory"); };
        auto chain32 = [&](const Stage &S) {
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const unsigned wk = S.v[k >> 3][(k >> 1) & 3];
                o = __fmaf_rn(S.p[k >> 2][k & 3], h2f((uint16_t)((k & 1) ? (wk >> 16) : (wk & 0xffff))), o);
            }
        };
        // the chain of a block in which some maximum moved: per 16 keys either the plain chain or (rescale, fma) per key -- c is exactly 1.0f wherever the maximum did not move
        auto rescaled = [&](const Stage &S, uint32_t m32, const f32x4 (&cq)[8]) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (((m32 >> (16 * h)) & 0xffff) == 0) {
#pragma unroll
                    for (int k = 16 * h; k < 16 * h + 16; ++k) {
                        const unsigned wk = S.v[k >> 3][(k >> 1) & 3];
                        o = __fmaf_rn(S.p[k >> 2][k & 3], h2f((uint16_t)((k & 1) ? (wk >> 16) : (wk & 0xffff))), o);
                    }
                } else {
#pragma unroll
                    for (int k = 16 * h; k < 16 * h + 16; ++k) {
                        const unsigned wk = S.v[k >> 3][(k >> 1) & 3];
                        o = o * cq[k >> 2][k & 3];
                        o = __fmaf_rn(S.p[k >> 2][k & 3], h2f((uint16_t)((k & 1) ? (wk >> 16) : (wk & 0xffff))), o);
                    }
                }
            }
        };
        auto turn = [&](int b, const Stage &Scur, Stage &Snext) {      // block b is in Scur; block b + 1 exists and goes to Snext
            landed(Scur);
            ensure(b + 1);
            const uint32_t m32 = (uint32_t)__builtin_amdgcn_readfirstlane(Scur.m);
            if (m32 == 0) {
                rd(b + 1, Snext);
                chain32(Scur);
#pragma unroll
                for (int i = 0; i < 13; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);      // two of the chain's fmas ...
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // ... then one of the next block's LDS reads
                }
                STAMPB(b);
                L.walked[wid] = (uint32_t)(b + 1);      // every lane stores the same word: neither a uniform branch nor an EXEC write (both are scheduling boundaries
                                                        // and would cut the block the interleave lives in)
            } else {
                // some maximum moved: the block's 32 c values and the next block's stage are requested together (21 reads, issued while the first of them travel).  (The
                // first form read c behind the next block's 13 reads and once per half, each time waiting out the LDS latency: 0.5 us per such block against 0.2.)
                asm volatile("" ::: "memory");      // keeps the arms from sharing a prefix: the common arm's reads must stay inside its block to be interleaved
                f32x4 cq[8];
#pragma unroll
                for (int q4 = 0; q4 < 8; ++q4) cq[q4] = *reinterpret_cast<const f32x4 *>(L.c + FP_B * b + 4 * q4);
                rd(b + 1, Snext);
                rescaled(Scur, m32, cq);
                STAMPB(b);
                L.walked[wid] = (uint32_t)(b + 1);
            }
        };
        auto last = [&](int b, const Stage &S) {      // the final block: nothing to request behind it
            landed(S);
            const uint32_t m32 = (uint32_t)__builtin_amdgcn_readfirstlane(S.m);
            if (m32 == 0) chain32(S);
            else {
                f32x4 cq[8];
#pragma unroll
                for (int q4 = 0; q4 < 8; ++q4) cq[q4] = *reinterpret_cast<const f32x4 *>(L.c + FP_B * b + 4 * q4);
                rescaled(S, m32, cq);
            }
            STAMPB(b);
        };
        ensure(0);
        rd(0, SA);
        STAMP(2);
        int b = 0;
        for (; b + 2 < nblkV; b += 2) {
            turn(b, SA, SB);
            turn(b + 1, SB, SA);
        }
        if (b + 1 < nblkV) { turn(b, SA, SB); last(b + 1, SB); }
        else last(b, SA);
        STAMP(3);
        if (active) L.ob[tid] = o;
    } else if (lane == 0) {
        // ---------------------------------------------------------------- logsum lane --------------------------------------------------------------
        float lsum = 0.0f;
        // the same shape as the walk: a block per stage, the next block's 9 reads interleaved with this block's 32 dependent adds
        struct SStage { f32x4 p[8]; uint32_t m; };
        SStage A, B;
        auto rd = [&](int b, SStage &S) {
#pragma unroll
            for (int q4 = 0; q4 < 8; ++q4) S.p[q4] = *reinterpret_cast<const f32x4 *>(L.p + FP_B * b + 4 * q4);
            S.m = L.rm[b].y;
        };
        auto landed = [&](const SStage &S) { asm volatile("" : : "v"(S.p[7]), "v"(S.m) : "memory"); };
        auto rescaled = [&](const SStage &S, const f32x4 (&cq)[8]) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (((S.m >> (16 * h)) & 0xffff) == 0) {
#pragma unroll
                    for (int k = 16 * h; k < 16 * h + 16; ++k) lsum = lsum + S.p[k >> 2][k & 3];
                } else {
#pragma unroll
                    for (int k = 16 * h; k < 16 * h + 16; ++k) lsum = __fmaf_rn(lsum, cq[k >> 2][k & 3], S.p[k >> 2][k & 3]);
                }
            }
        };
        auto turn = [&](int b, const SStage &Scur, SStage &Snext) {      // block b is in Scur; block b + 1 exists and goes to Snext
            landed(Scur);
            ensure(b + 1);
            if (Scur.m == 0) {
                rd(b + 1, Snext);
#pragma unroll
                for (int k = 0; k < 32; ++k) lsum = lsum + Scur.p[k >> 2][k & 3];
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            } else {      // some maximum moved: the block's c values and the next stage requested together (see the walker)
                asm volatile("" ::: "memory");
                f32x4 cq[8];
#pragma unroll
                for (int q4 = 0; q4 < 8; ++q4) cq[q4] = *reinterpret_cast<const f32x4 *>(L.c + FP_B * b + 4 * q4);
                rd(b + 1, Snext);
#pragma unroll
                for (int k = 0; k < 32; ++k) lsum = __fmaf_rn(lsum, cq[k >> 2][k & 3], Scur.p[k >> 2][k & 3]);      // c = 1.0f where nothing moved: lsum * 1 + p
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        };
        auto last = [&](int b, const SStage &S) {
            landed(S);
            if (S.m == 0) {
#pragma unroll
                for (int k = 0; k < 32; ++k) lsum = lsum + S.p[k >> 2][k & 3];
            } else {
                f32x4 cq[8];
#pragma unroll
                for (int q4 = 0; q4 < 8; ++q4) cq[q4] = *reinterpret_cast<const f32x4 *>(L.c + FP_B * b + 4 * q4);
                rescaled(S, cq);
            }
        };
        ensure(0);
        rd(0, A);
        int b = 0;
        for (; b + 2 < nblkS; b += 2) {
            turn(b, A, B);
            turn(b + 1, B, A);
        }
        if (b + 1 < nblkS) { turn(b, A, B); last(b + 1, B); }
        else last(b, A);
        *L.lsum = lsum;
        STAMPT(7, NWK * 64);
    }
    STAMP(4);
    __syncthreads();
    if (tid < DV) L.ob[tid] = L.ob[tid] * (1.0f / *L.lsum);
    __syncthreads();
}
}  // namespace mllm_hip
