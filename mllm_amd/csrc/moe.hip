// mllm_amd/csrc/moe.hip -- SURVEY N4: a sparse-MoE feed-forward block composed, on the host side of the library, from the launchers of the other files.
// The block is the reference's MiniCPMMoE::Forward (mllm/models/minicpm_moe/modeling_minicpm_moe.hpp:52-105; the BailingMoE / SmallThinker blocks of models/ling and
// models/smallthinker route the same way): router Linear -> softmax -> Tensor::topk -> renormalise -> argsort / bincount routing -> per expert: gather its tokens
// (Tensor::clip(index, SEQUENCE)), gate / up / down MLP, scale each row by its routing weight, Tensor::scatter_add into the zero-initialised output.
// What runs where: everything with arithmetic on token rows on the device (launchers of kernels_linear / _elem / _n4); the routing itself -- S * k (value, expert)
// pairs -- crosses PCIe once, as the reference's own loop reads `tokens_per_expert` on the host per expert (:76-78).  The renormalisation of the k weights (a
// sequential fp32 sum, then one division: CPUSumFunc.hpp, CPUBinaryFunc.hpp) is done on those host copies with the same fp32 operations.
// Experts are visited in ascending order and a token meets an expert at most once, so the accumulation order of every output row is the reference's.
#include <algorithm>
#include <cstring>
#include <vector>

#include "common.h"

namespace mllm_hip {
namespace {
// y[r][:] *= w[r]  (CPUBinaryFunc F_TTMUL with a [R,1,1,1] right operand, modeling_minicpm_moe.hpp:86)
__global__ __launch_bounds__(256) void scale_rows_kernel(float *__restrict__ y, int64_t ld, const float *__restrict__ w, int D4) {
    const int r = blockIdx.x;
    const float s = w[r];
    float4 *row = reinterpret_cast<float4 *>(y + (int64_t)r * ld);
    for (int i = threadIdx.x; i < D4; i += 256) {
        float4 v = row[i];
        v.x = v.x * s; v.y = v.y * s; v.z = v.z * s; v.w = v.w * s;
        row[i] = v;
    }
}
struct Scratch {      // stream-ordered blocks, all returned on every path
    hipStream_t st;
    std::vector<void *> blocks;
    int rc = MLLM_HIP_OK;
    template <typename T> T *get(size_t n) {
        void *p = nullptr;
        if (rc) return nullptr;
        const hipError_t e = hipMallocAsync(&p, n ? n * sizeof(T) : 16, st);
        if (e != hipSuccess) { set_error("hipMallocAsync(moe scratch)", e, __FILE__, __LINE__); rc = MLLM_HIP_ERR_HIP; return nullptr; }
        blocks.push_back(p);
        return (T *)p;
    }
    void release() {
        for (void *p : blocks) {
            const hipError_t e = hipFreeAsync(p, st);
            if (e != hipSuccess && !rc) { set_error("hipFreeAsync", e, __FILE__, __LINE__); rc = MLLM_HIP_ERR_HIP; }
        }
        blocks.clear();
    }
};
}  // namespace
}  // namespace mllm_hip

using namespace mllm_hip;

extern "C" int mllm_hip_scale_rows(float *y, int64_t ld, const float *w, int R, int D, void *stream) {
    if (R < 0 || D <= 0 || D % 4 || ld % 4 || ld < D) return MLLM_HIP_ERR_SHAPE;
    if (R == 0) return MLLM_HIP_OK;
    if (!y || !w) return MLLM_HIP_ERR_ARG;
    hipLaunchKernelGGL(scale_rows_kernel, dim3(R), dim3(256), 0, as_stream(stream), y, ld, w, D / 4);
    return MH_LAUNCH_OK("scale_rows");
}

#define MOE_STEP(expr) do { if (!S.rc) { const int _r = (expr); if (_r) S.rc = _r; } } while (0)
#define MOE_HIP(expr) do { if (!S.rc) { const hipError_t _e = (expr); if (_e != hipSuccess) { set_error(#expr, _e, __FILE__, __LINE__); S.rc = MLLM_HIP_ERR_HIP; } } } while (0)

extern "C" int mllm_hip_moe_block(const float *x, float *out, int n_tok, int hidden, int inter, int n_experts, int per_tok, const void *router_q4k,
                                  const void *const *w1_q4k, const void *const *w3_q4k, const void *const *w2_q4k, void *stream) {
    if (n_tok < 0 || hidden <= 0 || inter <= 0 || hidden % 256 || inter % 256 || n_experts <= 0 || per_tok <= 0 || per_tok > n_experts) return MLLM_HIP_ERR_SHAPE;
    if (n_tok == 0) return MLLM_HIP_OK;
    if (!x || !out || !router_q4k || !w1_q4k || !w3_q4k || !w2_q4k) return MLLM_HIP_ERR_ARG;
    const int T = n_tok, H = hidden, I = inter, E = n_experts, k = per_tok;
    Scratch S;
    S.st = as_stream(stream);
    // activations as Q8_K planes (quantize_row_q8_K_reference before every Q4_K Linear, Matmul.cpp:77-120): K = hidden for x / the gathered rows, K = inter for the MLP's middle
    int8_t *qh = S.get<int8_t>((size_t)T * H); float *dh = S.get<float>((size_t)T * (H / 256)); int16_t *bh = S.get<int16_t>((size_t)T * (H / 16));
    int8_t *qi = S.get<int8_t>((size_t)T * I); float *di = S.get<float>((size_t)T * (I / 256)); int16_t *bi = S.get<int16_t>((size_t)T * (I / 16));
    float *scores = S.get<float>((size_t)T * E), *prob = S.get<float>((size_t)T * E);
    float *route = S.get<float>((size_t)4 * T * k);      // the four small routing buffers that cross PCIe
    float *topv = route, *topi = route ? route + (size_t)T * k : nullptr;
    float *xe = S.get<float>((size_t)T * H), *gu = S.get<float>((size_t)T * 2 * I), *act = S.get<float>((size_t)T * I), *ye = S.get<float>((size_t)T * H);
    float *idx_d = route ? route + (size_t)2 * T * k : nullptr, *w_d = route ? route + (size_t)3 * T * k : nullptr;
    // ---- router (:58-60) ----
    MOE_STEP(mllm_hip_quantize_q8k(x, qh, dh, bh, T, H, stream));
    MOE_STEP(mllm_hip_linear_q4k_q8k(router_q4k, nullptr, qh, dh, bh, scores, MLLM_HIP_F32, E, nullptr, T, E, H, stream));
    MOE_STEP(mllm_hip_softmax(scores, prob, T, E, nullptr, stream));
    MOE_STEP(mllm_hip_topk_rows(prob, E, topv, topi, T, E, k, stream));
    std::vector<float> hv((size_t)T * k), hi((size_t)T * k);
    MOE_HIP(hipMemcpyAsync(hv.data(), topv, hv.size() * 4, hipMemcpyDeviceToHost, S.st));
    MOE_HIP(hipMemcpyAsync(hi.data(), topi, hi.size() * 4, hipMemcpyDeviceToHost, S.st));
    MOE_HIP(hipMemsetAsync(out, 0, (size_t)T * H * 4, S.st));      // Tensor::zero_like (:73)
    MOE_HIP(hipStreamSynchronize(S.st));
    // ---- routing on the host copies (:61-69): renormalised weights, then the (token, slot) pairs of every expert in ascending pair order ----
    std::vector<float> idx_h((size_t)T * k), w_h((size_t)T * k);
    std::vector<int> first(E + 1, 0);
    if (!S.rc) {
        for (int t = 0; t < T; ++t) {
            float den = 0.0f;
            for (int d = 0; d < k; ++d) den = den + hv[(size_t)t * k + d];
            for (int d = 0; d < k; ++d) hv[(size_t)t * k + d] = hv[(size_t)t * k + d] / den;
        }
        std::vector<int> count(E, 0);
        for (size_t j = 0; j < hi.size(); ++j) { const int e = (int)hi[j]; if (e >= 0 && e < E) ++count[e]; }
        for (int e = 0; e < E; ++e) first[e + 1] = first[e] + count[e];
        std::vector<int> fill(first.begin(), first.end() - 1);
        for (size_t j = 0; j < hi.size(); ++j) {
            const int e = (int)hi[j];
            if (e < 0 || e >= E) continue;
            idx_h[fill[e]] = (float)(j / k);      // token_idxs = idxs / num_experts_per_tok (:69)
            w_h[fill[e]] = hv[j];                 // expert_weights.clip(exp_idx, SEQUENCE) (:85)
            ++fill[e];
        }
        MOE_HIP(hipMemcpyAsync(idx_d, idx_h.data(), idx_h.size() * 4, hipMemcpyHostToDevice, S.st));
        MOE_HIP(hipMemcpyAsync(w_d, w_h.data(), w_h.size() * 4, hipMemcpyHostToDevice, S.st));
    }
    // ---- experts, ascending (:74-90) ----
    for (int e = 0; e < E && !S.rc; ++e) {
        const int R = first[e + 1] - first[e];
        if (R == 0) continue;
        if (!w1_q4k[e] || !w3_q4k[e] || !w2_q4k[e]) { S.rc = MLLM_HIP_ERR_ARG; break; }
        const float *ie = idx_d + first[e], *we = w_d + first[e];
        MOE_STEP(mllm_hip_gather_rows(x, H, T, ie, xe, H, R, H, 0, stream));
        MOE_STEP(mllm_hip_quantize_q8k(xe, qh, dh, bh, R, H, stream));
        MOE_STEP(mllm_hip_linear_q4k_q8k(w1_q4k[e], nullptr, qh, dh, bh, gu, MLLM_HIP_F32, 2 * I, nullptr, R, I, H, stream));          // gate_proj -> columns [0, I)
        MOE_STEP(mllm_hip_linear_q4k_q8k(w3_q4k[e], nullptr, qh, dh, bh, gu + I, MLLM_HIP_F32, 2 * I, nullptr, R, I, H, stream));      // up_proj   -> columns [I, 2I)
        MOE_STEP(mllm_hip_silu_mul(gu, act, R, I, stream));
        MOE_STEP(mllm_hip_quantize_q8k(act, qi, di, bi, R, I, stream));
        MOE_STEP(mllm_hip_linear_q4k_q8k(w2_q4k[e], nullptr, qi, di, bi, ye, MLLM_HIP_F32, H, nullptr, R, H, I, stream));
        MOE_STEP(mllm_hip_scale_rows(ye, H, we, R, H, stream));
        MOE_STEP(mllm_hip_scatter_add_rows(out, H, T, ye, H, ie, R, H, stream));
    }
    S.release();
    {      // the host routing tables above are the source of two copies: they must outlive them
        const hipError_t e = hipStreamSynchronize(S.st);
        if (e != hipSuccess && !S.rc) { set_error("hipStreamSynchronize", e, __FILE__, __LINE__); S.rc = MLLM_HIP_ERR_HIP; }
    }
    return S.rc;
}
