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
#include "kernels_attn_core.h"

namespace mllm_hip {

// ------------------------------------------------------------------------------------------------------------------
// Sq >= 4: __fa2_prefill_append with Br = Bc = 4.  One workgroup = one head x RT consecutive row tiles.
// ------------------------------------------------------------------------------------------------------------------
constexpr int FA_VS = 64;   // keys per staged V sub-chunk
template <int D, bool F16, bool VT>
struct VStage {
    static constexpr int ELT = F16 ? 2 : 4;
    static constexpr int ROWV = VT ? FA_VS * 2 / 16 : D * ELT / 16;       // 16-byte vectors per staged row
    static constexpr int VEC = VT ? D * ROWV : FA_VS * ROWV;              // vectors per sub-chunk
    static constexpr int PITCH = FA_VS * 2 + 16;
};
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int D, bool F16, bool VT, int N>
__device__ __forceinline__ void vs_fetch(u32x4 (&stage)[N], const void *V, int64_t ldv, int kvh, int key0, int Sk) {
    using G = VStage<D, F16, VT>;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int vi = min((int)threadIdx.x + 256 * i, G::VEC - 1), row = vi / G::ROWV, part = vi % G::ROWV;
        if (VT) stage[i] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const uint16_t *>(V) + (int64_t)(kvh * D + row) * ldv + key0 + part * 8);
        else stage[i] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(V) + ((int64_t)min(key0 + row, Sk - 1) * ldv + kvh * D) * G::ELT + part * 16);
    }
}
template <int D, bool F16, bool VT, int N>
__device__ __forceinline__ void vs_park(const u32x4 (&stage)[N], char *buf) {
    using G = VStage<D, F16, VT>;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int vi = threadIdx.x + 256 * i, row = vi / G::ROWV, part = vi % G::ROWV;
        if (vi < G::VEC) *reinterpret_cast<u32x4 *>(buf + (VT ? (size_t)row * G::PITCH + part * 16 : (size_t)vi * 16)) = stage[i];
    }
}
template <int D>
struct PrefillCfg {
    static constexpr int RT = (256 / D) < 1 ? 1 : ((256 / D) > 4 ? 4 : (256 / D));   // row tiles per workgroup
    static constexpr int R = 4 * RT;
    static constexpr int SP = FA_KC + 4;   // pitch of the score rows
};

// VT: V is the transposed fp16 slab (element (key j, dim d) at V[(kvh*D + d) * ldv + j], ldv % 4 == 0, rows zero padded)
template <int D, bool F16, bool VT = false>
__global__ __launch_bounds__(256) void fa2_prefill_kernel(const float *__restrict__ Q, int64_t ldq, const void *__restrict__ K, int64_t ldk,
                                                          const void *__restrict__ V, int64_t ldv, float *__restrict__ O, int64_t ldo, int Sq, int Sk,
                                                          int sk_eff, int Hq, int Hkv, int causal) {
    using C = PrefillCfg<D>;
    constexpr int RT = C::RT, R = C::R, SP = C::SP;
    __shared__ __attribute__((aligned(16))) float qs[R * D];
    __shared__ __attribute__((aligned(16))) float S[R * SP];
    __shared__ __attribute__((aligned(16))) float Cc[RT * 64 * 4];
    __shared__ __attribute__((aligned(16))) float Sm[RT * 64 * 4];
    __shared__ float m_in[R], lfin[R];
    constexpr int ELT = F16 ? 2 : 4;
    constexpr int VS_ROWV = VT ? FA_VS * 2 / 16 : D * ELT / 16;       // 16-byte vectors per staged row
    constexpr int VS_VEC = VT ? D * VS_ROWV : FA_VS * VS_ROWV;        // vectors per sub-chunk
    constexpr int VSV = (VS_VEC + 255) / 256;                         // per thread
    constexpr int VS_PITCH = FA_VS * 2 + 16;
    constexpr int VS_BYTES = VT ? D * VS_PITCH : FA_VS * D * ELT;
    __shared__ __attribute__((aligned(16))) char vbuf[2 * VS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int head = blockIdx.y, kvh = head / (Hq / Hkv);
    const int r0 = blockIdx.x * R;
    const int delta = Sk - Sq;
    const float scale = 1.0f / sqrtf((float)D);
    for (int i = tid; i < R * D; i += 256) {
        const int r = i / D, d = i - r * D;
        qs[i] = Q[(int64_t)min(r0 + r, Sq - 1) * ldq + head * D + d];
    }
    if (tid < R) m_in[tid] = FA_NEG;
    // phase-C ownership: thread (g, d) accumulates rows 4g..4g+3 of dim d; threads d < 4 also carry logsum of row 4g + d
    const int g = tid / D, d = tid - g * D;
    const bool own = g < RT;
    float o[4] = {0.0f, 0.0f, 0.0f, 0.0f}, lsum = 0.0f;
    int klim = sk_eff;
    if (causal) klim = min(sk_eff, r0 + R + delta + 4);   // tiles beyond are skipped for every row of this workgroup
    __syncthreads();
    for (int chunk0 = 0; chunk0 < klim; chunk0 += FA_KC) {
        // ---- A: scores of key chunk0 + tid against the R rows -------------------------------------------------------
        {
            const int j = chunk0 + tid;
            float kr[D];
            load_kv_row<D, F16>(kr, K, (int64_t)min(j, Sk - 1) * ldk + kvh * D);
            const int c0 = j & ~3, nc = min(4, sk_eff - c0);
#pragma unroll 1
            for (int r = 0; r < R; ++r) {
                float s = qk_dot<D>(qs + r * D, kr);
                const int tr0 = r0 + (r & ~3), nr = min(4, Sq - tr0);
                if (causal && (tr0 + nr == c0 + nc - delta) && (j - c0) > (r & 3)) s = FA_NEG;
                S[r * SP + tid] = s;
            }
        }
        __syncthreads();
        // ---- B: per row, scan the 64 key tiles of the chunk ----------------------------------------------------------
        for (int r = wid; r < R; r += 4) {
            const int c0 = chunk0 + 4 * lane;
            const int tr0 = r0 + (r & ~3), nr = min(4, Sq - tr0);
            const int nc = min(4, sk_eff - c0);
            const bool live = nc > 0 && nr > 0 && !(causal && (c0 - delta > tr0 + nr - 1));
            const float4 s4 = *reinterpret_cast<const float4 *>(S + r * SP + 4 * lane);
            float tm = FA_NEG;
            if (live) {
                tm = s4.x;
                if (nc > 1) tm = fmaxf(tm, s4.y);
                if (nc > 2) tm = fmaxf(tm, s4.z);
                if (nc > 3) tm = fmaxf(tm, s4.w);
            }
            const float carry = m_in[r];
            const float incl = fmaxf(wave_scan_max(tm), carry);
            const float excl = wave_shift_up(incl, carry);
            float4 p4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            float cc = 1.0f, sum = 0.0f;
            if (live) {
                cc = excl == incl ? 1.0f : glibc_expf((excl - incl) * scale);
                p4.x = glibc_expf((s4.x - incl) * scale);
                if (nc > 1) p4.y = glibc_expf((s4.y - incl) * scale);
                if (nc > 2) p4.z = glibc_expf((s4.z - incl) * scale);
                if (nc > 3) p4.w = glibc_expf((s4.w - incl) * scale);
                sum = ((p4.x + p4.y) + p4.z) + p4.w;
            }
            *reinterpret_cast<float4 *>(S + r * SP + 4 * lane) = p4;
            Cc[((r >> 2) * 64 + lane) * 4 + (r & 3)] = cc;
            Sm[((r >> 2) * 64 + lane) * 4 + (r & 3)] = sum;
            const float last = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(incl), 63));
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) m_in[r] = last;
        }
        __syncthreads();
        // ---- C: rescale + P V in key order.  The walk must not wait on memory: V goes through LDS in sub-chunks of FA_VS keys (double
        // buffered; all 256 threads fetch sub-chunk s+1 while sub-chunk s is walked).  Layout in LDS: reference layout [key][D] as stored
        // (fp32 or fp16), transposed slab [D][FA_VS keys] fp16 (pitch VS_PITCH bytes).
        {
            const int ntl = min(64, (klim - chunk0 + 3) >> 2);
            const int nsub = (ntl * 4 + FA_VS - 1) / FA_VS;
            u32x4 stage[VSV];
            vs_fetch<D, F16, VT, VSV>(stage, V, ldv, kvh, chunk0, Sk);
#pragma unroll
            for (int sc = 0; sc < FA_KC / FA_VS; ++sc) {     // fully unrolled: `stage` stays in registers across the prefetch
                if (sc >= nsub) break;
                vs_park<D, F16, VT, VSV>(stage, vbuf + (size_t)(sc & 1) * VS_BYTES);
                __syncthreads();
                if (sc + 1 < nsub) vs_fetch<D, F16, VT, VSV>(stage, V, ldv, kvh, chunk0 + (sc + 1) * FA_VS, Sk);
                if (own) {
                    const char *vb = vbuf + (size_t)(sc & 1) * VS_BYTES;
                    const int t1 = min(ntl, (sc + 1) * (FA_VS / 4));
#pragma unroll 2
                    for (int tl = sc * (FA_VS / 4); tl < t1; ++tl) {
                        const int kl = 4 * tl - sc * FA_VS;     // first key of the tile inside the sub-chunk
                        const float4 c4 = *reinterpret_cast<const float4 *>(Cc + (g * 64 + tl) * 4);
                        float vv[4];
                        if (VT) {
                            const uint2 w = *reinterpret_cast<const uint2 *>(vb + (size_t)d * VS_PITCH + kl * 2);
                            vv[0] = h2f((uint16_t)(w.x & 0xffff)); vv[1] = h2f((uint16_t)(w.x >> 16));
                            vv[2] = h2f((uint16_t)(w.y & 0xffff)); vv[3] = h2f((uint16_t)(w.y >> 16));
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                vv[k] = F16 ? h2f(reinterpret_cast<const uint16_t *>(vb)[(kl + k) * D + d]) : reinterpret_cast<const float *>(vb)[(kl + k) * D + d];
                        }
                        const float cr[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float4 p = *reinterpret_cast<const float4 *>(S + (4 * g + r) * SP + 4 * tl);
                            float a = o[r] * cr[r];
                            a = __fmaf_rn(p.x, vv[0], a);
                            a = __fmaf_rn(p.y, vv[1], a);
                            a = __fmaf_rn(p.z, vv[2], a);
                            a = __fmaf_rn(p.w, vv[3], a);
                            o[r] = a;
                        }
                        if (d < 4) lsum = __fmaf_rn(lsum, Cc[(g * 64 + tl) * 4 + d], Sm[(g * 64 + tl) * 4 + d]);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (own && d < 4) lfin[4 * g + d] = lsum;
    __syncthreads();
    if (own) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = r0 + 4 * g + r;
            if (gr < Sq) O[(int64_t)gr * ldo + head * D + d] = o[r] * (1.0f / lfin[4 * g + r]);
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
    fa2_decode_head<D, F16, NT, VT>(L, P, K, ldk, V, ldv, kvh * D, Sk, cap, nullptr, nullptr, -1);
    if (threadIdx.x < D) O[head * D + threadIdx.x] = L.ob[threadIdx.x];
}
}  // namespace mllm_hip

using namespace mllm_hip;

extern "C" size_t mllm_hip_fa2_workspace_bytes(int Sq, int Hq, int D, int max_sk) {
    (void)Sq; (void)Hq; (void)D; (void)max_sk;
    return 256;   // the kernels keep their state in LDS; a token allocation keeps callers' bookkeeping uniform
}

template <int D, bool F16, bool VT = false>
static int launch_fa2(const float *Q, int64_t ldq, const void *K, int64_t ldk, const void *V, int64_t ldv, float *O, int64_t ldo, int Sq, int Sk,
                      int Hq, int Hkv, int causal, const int *sk_dev, int sk_max, hipStream_t st) {
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
    constexpr int R = PrefillCfg<D>::R;
    hipLaunchKernelGGL((fa2_prefill_kernel<D, F16, VT>), dim3((Sq + R - 1) / R, Hq), dim3(256), 0, st, Q, ldq, K, ldk, V, ldv, O, ldo, Sq, Sk, sk_eff, Hq,
                       Hkv, causal);
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
