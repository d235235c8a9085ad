// mllm_amd/csrc/kernels_attn.hip -- A13: flash_attention_2_forward for gfx950 (compute/FlashAttention2.hpp:2236-2284).
//
// Semantics kept from the reference: fp32 Q, fp16 (LLM KV cache) or fp32 (vision) K/V widened to fp32, fp32 accumulate,
// GQA kv_head = q_head / (Hq/Hkv) (:164), causal offset delta = Sk - Sq (:324), scores scaled inside the exponent
// p = expf((s - max) * scale) (:451-457), online max/sum.  The reference's tile-aligned causal quirk is NOT
// reproduced: masking here is exact per element.
//
// prefill (Sq > 1): exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, an fp32 fma chain -- no bf16 rounding of Q/K/V/P).
//   One wave owns 32 query rows.  S^T = K Q^T is computed with the key on the accumulator ROW and the query on the LANE
//   (A = K tile from LDS, B = Q from registers), so every lane owns one query's softmax state and the row reductions are
//   in-register plus one cross-half shuffle.  P then feeds O^T = V^T P directly as the B operand: k-step s of lane half
//   h is accumulator register s, i.e. key (s&3)+8(s>>2)+4h -- the A operand (V^T from LDS) is addressed with the same
//   key permutation (cdna_hip_programming.md §3 "accumulator tile as the next MFMA's operand").
// decode (Sq == 1): HBM/latency-bound KV stream; lane = key for the scores, lane = 2 output dims for P V, keys split over
//   waves and workgroups, partial (max, sum, out) merged by a second small kernel.  Sk can come from device memory so a
//   captured graph serves every step.
#include <cmath>

#include "common.h"

namespace mllm_hip {

typedef float v16f __attribute__((ext_vector_type(16)));

template <bool KV_F16>
__device__ __forceinline__ float ld_kv(const void *p, int64_t i) {
    if (KV_F16) return h2f(reinterpret_cast<const uint16_t *>(p)[i]);
    return reinterpret_cast<const float *>(p)[i];
}

// D = head dim (multiple of 2, <= 128), DT = ceil(D/32)
template <int D, bool KV_F16>
__global__ __launch_bounds__(256) void fa2_prefill_kernel(const float *__restrict__ Q, int64_t ldq, const void *__restrict__ K, int64_t ldk,
                                                          const void *__restrict__ V, int64_t ldv, float *__restrict__ O, int64_t ldo, int Sq, int Sk,
                                                          int Hq, int Hkv, int causal) {
    constexpr int DT = (D + 31) / 32;
    constexpr int KP = D + 1;       // K tile row stride (odd: conflict-free ds_read_b32 across keys)
    constexpr int VP = DT * 32;     // V tile row stride, zero padded
    __shared__ float Ks[32 * KP];
    __shared__ float Vs[32 * VP];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int qc = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, kvh = head / (Hq / Hkv);
    const int q_base = blockIdx.x * 128;
    const int qi = q_base + wid * 32 + qc;           // this lane's query row
    const int delta = Sk - Sq;
    const float scale = __fdiv_rn(1.0f, __fsqrt_rn((float)D));

    float qreg[D / 2];
    {
        const float *qp = Q + (int64_t)min(qi, Sq - 1) * ldq + head * D;
#pragma unroll
        for (int s = 0; s < D / 2; ++s) qreg[s] = qp[2 * s + h];
    }
    v16f o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    const int q_last = min(q_base + 127, Sq - 1);
    const int k_end = causal ? min(Sk, q_last + delta + 1) : Sk;
    for (int k0 = 0; k0 < k_end; k0 += 32) {
        __syncthreads();
        for (int e = tid; e < 32 * D; e += 256) {
            const int i = e / D, d = e - i * D;
            const int key = k0 + i;
            float kv = 0.0f, vv = 0.0f;
            if (key < Sk) {
                kv = ld_kv<KV_F16>(K, (int64_t)key * ldk + kvh * D + d);
                vv = ld_kv<KV_F16>(V, (int64_t)key * ldv + kvh * D + d);
            }
            Ks[i * KP + d] = kv;
            Vs[i * VP + d] = vv;
        }
        if (VP > D) for (int e = tid; e < 32 * (VP - D); e += 256) { const int i = e / (VP - D), d = D + e % (VP - D); Vs[i * VP + d] = 0.0f; }
        __syncthreads();

        v16f s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < D / 2; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[qc * KP + 2 * ks + h], qreg[ks], s, 0, 0, 0);
        // s[r] = score(key = k0 + (r&3)+8(r>>2)+4h, query = qi)
        float mloc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const bool ok = key < Sk && (!causal || key <= qi + delta);
            s[r] = ok ? s[r] : -INFINITY;
            mloc = fmaxf(mloc, s[r]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float m_use = m_new == -INFINITY ? 0.0f : m_new;
        const float alpha = m_run == -INFINITY ? 0.0f : expf(__fmul_rn(__fsub_rn(m_run, m_use), scale));
        float lsum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = expf(__fmul_rn(__fsub_rn(s[r], m_use), scale)); lsum += s[r]; }
        lsum += __shfl_xor(lsum, 32, 64);
        l_run = __fmaf_rn(l_run, alpha, lsum);
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[t][i] *= alpha;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const int key = (ks & 3) + 8 * (ks >> 2) + 4 * h;
                o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[key * VP + 32 * t + qc], s[ks], o[t], 0, 0, 0);
            }
    }
    if (qi < Sq) {
        float *op = O + (int64_t)qi * ldo + head * D;
        const float inv_l = l_run;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = 32 * t + 8 * g4 + 4 * h;
                if (d < D) {
                    float4 v;
                    v.x = __fdiv_rn(o[t][4 * g4 + 0], inv_l); v.y = __fdiv_rn(o[t][4 * g4 + 1], inv_l);
                    v.z = __fdiv_rn(o[t][4 * g4 + 2], inv_l); v.w = __fdiv_rn(o[t][4 * g4 + 3], inv_l);
                    *reinterpret_cast<float4 *>(op + d) = v;
                }
            }
    }
}

// ---- decode: partials over key splits ---------------------------------------------------------------------------------
// workspace layout per (head, split): [0] = max, [1] = sum, [2..2+D) = unnormalised out (fp32), stride WS_STRIDE floats
constexpr int WS_STRIDE = 136;

template <int D, bool KV_F16>
__global__ __launch_bounds__(256) void fa2_decode_partial_kernel(const float *__restrict__ Q, const void *__restrict__ K, int64_t ldk,
                                                                 const void *__restrict__ V, int64_t ldv, float *__restrict__ ws, int Sk_host,
                                                                 const int *__restrict__ sk_dev, int Hq, int Hkv, int nsplit) {
    __shared__ float qs[D];
    __shared__ float red[4][WS_STRIDE];
    __shared__ float ps[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int head = blockIdx.x, split = blockIdx.y, kvh = head / (Hq / Hkv);
    const int Sk = sk_dev ? *sk_dev : Sk_host;
    const float scale = __fdiv_rn(1.0f, __fsqrt_rn((float)D));
    if (tid < D) qs[tid] = Q[head * D + tid];
    __syncthreads();
    const int key = split * 256 + wid * 64 + lane;
    float s = -INFINITY;
    if (key < Sk) {
        float acc = 0.0f;
        if (KV_F16) {
            const uint4 *kp = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(K) + (int64_t)key * ldk + kvh * D);
#pragma unroll
            for (int c = 0; c < D / 8; ++c) {
                const uint4 u = kp[c];
                const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc = __fmaf_rn(qs[c * 8 + 2 * e], h2f((uint16_t)(w[e] & 0xffff)), acc);
                    acc = __fmaf_rn(qs[c * 8 + 2 * e + 1], h2f((uint16_t)(w[e] >> 16)), acc);
                }
            }
        } else {
            const float4 *kp = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(K) + (int64_t)key * ldk + kvh * D);
#pragma unroll
            for (int c = 0; c < D / 4; ++c) {
                const float4 u = kp[c];
                acc = __fmaf_rn(qs[c * 4], u.x, acc); acc = __fmaf_rn(qs[c * 4 + 1], u.y, acc);
                acc = __fmaf_rn(qs[c * 4 + 2], u.z, acc); acc = __fmaf_rn(qs[c * 4 + 3], u.w, acc);
            }
        }
        s = acc;
    }
    const float m_w = wave_max(s);
    const float m_use = m_w == -INFINITY ? 0.0f : m_w;
    const float p = key < Sk ? expf(__fmul_rn(__fsub_rn(s, m_use), scale)) : 0.0f;
    const float l_w = wave_sum(p);
    // P V: lane owns dims 2*lane, 2*lane+1; p_j is broadcast through LDS (a shuffle cannot read lanes masked off by D < 128)
    ps[wid][lane] = p;
    __syncthreads();
    float o0 = 0.0f, o1 = 0.0f;
    const int kbase = split * 256 + wid * 64;
    const int nk = min(64, Sk - kbase);
    if (2 * lane < D) {
        for (int j = 0; j < nk; ++j) {
            const float pj = ps[wid][j];
            const int64_t vo = (int64_t)(kbase + j) * ldv + kvh * D + 2 * lane;
            float v0, v1;
            if (KV_F16) {
                const uint32_t u = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint16_t *>(V) + vo);
                v0 = h2f((uint16_t)(u & 0xffff)); v1 = h2f((uint16_t)(u >> 16));
            } else {
                const float2 u = *reinterpret_cast<const float2 *>(reinterpret_cast<const float *>(V) + vo);
                v0 = u.x; v1 = u.y;
            }
            o0 = __fmaf_rn(pj, v0, o0);
            o1 = __fmaf_rn(pj, v1, o1);
        }
    }
    if (lane == 0) { red[wid][0] = m_w; red[wid][1] = l_w; }
    if (2 * lane < D) { red[wid][2 + 2 * lane] = o0; red[wid][3 + 2 * lane] = o1; }
    __syncthreads();
    // merge the 4 waves
    const float m_tot = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
    const float mt = m_tot == -INFINITY ? 0.0f : m_tot;
    float f[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) f[w] = red[w][0] == -INFINITY ? 0.0f : expf(__fmul_rn(__fsub_rn(red[w][0], mt), scale));
    float *out = ws + ((int64_t)head * nsplit + split) * WS_STRIDE;
    if (tid == 0) { out[0] = m_tot; out[1] = f[0] * red[0][1] + f[1] * red[1][1] + f[2] * red[2][1] + f[3] * red[3][1]; }
    if (tid < D) out[2 + tid] = f[0] * red[0][2 + tid] + f[1] * red[1][2 + tid] + f[2] * red[2][2 + tid] + f[3] * red[3][2 + tid];
}

template <int D>
__global__ __launch_bounds__(128) void fa2_decode_merge_kernel(const float *__restrict__ ws, float *__restrict__ O, int nsplit) {
    const int head = blockIdx.x, tid = threadIdx.x;
    const float scale = __fdiv_rn(1.0f, __fsqrt_rn((float)D));
    const float *base = ws + (int64_t)head * nsplit * WS_STRIDE;
    float m_tot = -INFINITY;
    for (int s = 0; s < nsplit; ++s) m_tot = fmaxf(m_tot, base[s * WS_STRIDE]);
    const float mt = m_tot == -INFINITY ? 0.0f : m_tot;
    float l = 0.0f, acc = 0.0f;
    for (int s = 0; s < nsplit; ++s) {
        const float ms = base[s * WS_STRIDE];
        const float f = ms == -INFINITY ? 0.0f : expf(__fmul_rn(__fsub_rn(ms, mt), scale));
        l = __fmaf_rn(f, base[s * WS_STRIDE + 1], l);
        if (tid < D) acc = __fmaf_rn(f, base[s * WS_STRIDE + 2 + tid], acc);
    }
    if (tid < D) O[head * D + tid] = __fdiv_rn(acc, l);
}
}  // namespace mllm_hip

using namespace mllm_hip;

extern "C" size_t mllm_hip_fa2_workspace_bytes(int Sq, int Hq, int D, int max_sk) {
    (void)Sq; (void)D;
    const int nsplit = (max_sk + 255) / 256;
    return (size_t)Hq * (nsplit > 0 ? nsplit : 1) * WS_STRIDE * sizeof(float);
}

template <int D, bool F16>
static int launch_fa2(const float *Q, int64_t ldq, const void *K, int64_t ldk, const void *V, int64_t ldv, float *O, int64_t ldo, int Sq, int Sk,
                      int Hq, int Hkv, int causal, const int *sk_dev, void *workspace, hipStream_t st) {
    if (Sq == 1) {
        if (!workspace) return MLLM_HIP_ERR_ARG;
        const int nsplit = (Sk + 255) / 256;  // with sk_dev, Sk is the upper bound (cache limit)
        hipLaunchKernelGGL((fa2_decode_partial_kernel<D, F16>), dim3(Hq, nsplit), dim3(256), 0, st, Q, K, ldk, V, ldv, (float *)workspace, Sk, sk_dev,
                           Hq, Hkv, nsplit);
        int rc = MH_LAUNCH_OK("fa2_decode_partial");
        if (rc) return rc;
        hipLaunchKernelGGL((fa2_decode_merge_kernel<D>), dim3(Hq), dim3(128), 0, st, (const float *)workspace, O, nsplit);
        return MH_LAUNCH_OK("fa2_decode_merge");
    }
    if (sk_dev) return MLLM_HIP_ERR_ARG;
    hipLaunchKernelGGL((fa2_prefill_kernel<D, F16>), dim3((Sq + 127) / 128, Hq), dim3(256), 0, st, Q, ldq, K, ldk, V, ldv, O, ldo, Sq, Sk, Hq, Hkv,
                       causal);
    return MH_LAUNCH_OK("fa2_prefill");
}

extern "C" int mllm_hip_fa2(const float *Q, int64_t ldq, const void *K, int64_t ldk, const void *V, int64_t ldv, int kv_dtype, float *O,
                            int64_t ldo, int Sq, int Sk, int Hq, int Hkv, int D, int causal, const int *sk_dev, void *workspace, void *stream) {
    if (Sq <= 0 || Sk <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0) return MLLM_HIP_ERR_SHAPE;
    if (kv_dtype != MLLM_HIP_F16 && kv_dtype != MLLM_HIP_F32) return MLLM_HIP_ERR_DTYPE;
    // 16-byte vector loads of K/V rows in the decode kernel
    if ((ldk % 8) || (ldv % 8)) return MLLM_HIP_ERR_SHAPE;
    hipStream_t st = as_stream(stream);
    const bool f16 = kv_dtype == MLLM_HIP_F16;
#define FA2_CASE(DD)                                                                                                                        \
    case DD:                                                                                                                                \
        return f16 ? launch_fa2<DD, true>(Q, ldq, K, ldk, V, ldv, O, ldo, Sq, Sk, Hq, Hkv, causal, sk_dev, workspace, st)                   \
                   : launch_fa2<DD, false>(Q, ldq, K, ldk, V, ldv, O, ldo, Sq, Sk, Hq, Hkv, causal, sk_dev, workspace, st);
    switch (D) {
        FA2_CASE(16)
        FA2_CASE(64)
        FA2_CASE(80)
        FA2_CASE(128)
    default: return MLLM_HIP_ERR_SHAPE;
    }
#undef FA2_CASE
}
