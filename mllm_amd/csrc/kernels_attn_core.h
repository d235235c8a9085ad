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
}  // namespace mllm_hip
