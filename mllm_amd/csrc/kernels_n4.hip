// mllm_amd/csrc/kernels_n4.hip -- SURVEY section 8 row N4: the extra ops of the other model families, on the same C ABI.
//
//   mllm_hip_sliding_window_mask   SLIDINGWINDOWMASK (Mistral / Gemma style local attention)   backends/cpu/op/CPUSlidingWindowMask.cpp:30-58
//   mllm_hip_topk_rows             F_TOPK on DIMENSION (MoE routers)                           backends/cpu/op/CPUTopkFunc.hpp:48-70
//   mllm_hip_bincount              F_BINCOUNT (tokens per expert)                              backends/cpu/op/CPUBinCountFunc.hpp:20-35
//   mllm_hip_gather_rows           Tensor::clip(index, SEQUENCE) and F_FUYU_GATHER_EMBD        backends/cpu/op/CPUClipFunc.hpp:309-323, CPUFuyuGatherEmbdFunc.hpp:45-62
//   mllm_hip_scatter_add_rows      F_SCATTERADD on SEQUENCE (MoE combine)                      backends/cpu/op/CPUScatterAddFunc.hpp:38-52
//   mllm_hip_rope_table_ntk        NTKROPE table (MiniCPM3 / Phi-3 LongRoPE factors)           backends/cpu/op/CPUNTKRoPE.cpp:27-80   (host; the rotation is mllm_hip_rope_apply)
// Index tensors are fp32 on the device, as the reference's functions hand them over (Tensor holds floats); all results are bit-identical to the reference's
// (tests/golden/n4_ops.npz: the reference's own outputs).  These are HBM-bound byte movers except top-k, which is one wave per row.
#include <cfloat>
#include <climits>
#include <cmath>

#include "common.h"

namespace mllm_hip {

// scores [S][H][keys] (BSHD, D = keys): key d of query row s survives iff s - (window - 1) <= d <= s + (keys - S); everything else becomes lowest()
__global__ __launch_bounds__(256) void sliding_window_mask_kernel(const float *__restrict__ x, float *__restrict__ y, int S, int H, int keys, int window) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, total = (int64_t)S * H * keys;
    if (i >= total) return;
    const int d = (int)(i % keys), s = (int)(i / ((int64_t)H * keys));
    const int old = keys - S;
    const bool masked = S > 1 && (d > s + old || d < s - (window - 1));
    y[i] = masked ? -FLT_MAX : x[i];
}

// One wave per row.  The reference keeps the k largest (value, index) pairs in a min-heap and emits them in descending pair order, so among equal values the larger
// index comes first: round j picks the largest pair strictly below round j-1's pick.
__global__ __launch_bounds__(256) void topk_rows_kernel(const float *__restrict__ x, int64_t ldx, float *__restrict__ values, float *__restrict__ indices, int rows, int n,
                                                        int k) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + (int64_t)row * ldx;
    float lv = INFINITY;
    int li = INT_MAX;
    for (int j = 0; j < k; ++j) {
        float bv = -INFINITY;
        int bi = -1;
        for (int c = lane; c < n; c += 64) {
            const float v = xr[c];
            const bool below = v < lv || (v == lv && c < li);
            const bool better = bi < 0 || v > bv || (v == bv && c > bi);
            if (below && better) { bv = v; bi = c; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi > bi))) { bv = ov; bi = oi; }
        }
        if (lane == 0) {
            values[(int64_t)row * k + j] = bi >= 0 ? bv : 0.0f;
            indices[(int64_t)row * k + j] = (float)(bi >= 0 ? bi : 0);
        }
        lv = bv; li = bi;
    }
}

// counts[b] = number of ids whose integer part is b, 0 <= b < nbins (the reference sizes its output max + 1; the caller passes the bins it wants)
__global__ __launch_bounds__(256) void bincount_kernel(const float *__restrict__ ids, int n, float *__restrict__ counts, int nbins) {
    extern __shared__ int hist[];
    for (int b = threadIdx.x; b < nbins; b += 256) hist[b] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const int v = (int)ids[i];
        if (v >= 0 && v < nbins) atomicAdd(&hist[v], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += 256) counts[b] = (float)hist[b];
}

// out[r] = src[idx[r]]; with skip_negative (the Fuyu gather: out is the word-embedding buffer itself) rows whose index is negative are left alone
// (an index outside [0, n_src_rows) is never dereferenced: negative ones are the Fuyu form's "keep", anything else leaves the row untouched)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ src, int64_t lds, int n_src_rows, const float *__restrict__ idx, float *__restrict__ out, int64_t ldo,
                                                          int D, int skip_negative) {
    const int r = blockIdx.x;
    const int i = (int)idx[r];
    if (i < 0 || i >= n_src_rows) return;
    (void)skip_negative;
    const float *s = src + (int64_t)i * lds;
    float *o = out + (int64_t)r * ldo;
    for (int d = threadIdx.x; d < D; d += 256) o[d] = s[d];
}

// dst[idx[r]] += src[r] for r = 0 .. R-1 IN THAT ORDER: a thread owns a column and walks the rows, so a repeated destination accumulates exactly as the reference's loop does
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(float *__restrict__ dst, int64_t ldd, int n_dst_rows, const float *__restrict__ src, int64_t lds,
                                                               const float *__restrict__ idx, int R, int D) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    for (int r = 0; r < R; ++r) {
        const int i = (int)idx[r];
        if (i < 0 || i >= n_dst_rows) continue;      // out of range: skipped, never dereferenced
        float *p = dst + (int64_t)i * ldd + d;
        *p = *p + src[(int64_t)r * lds + d];
    }
}
}  // namespace mllm_hip

using namespace mllm_hip;

extern "C" int mllm_hip_sliding_window_mask(const float *x, float *y, int S, int H, int keys, int window, void *stream) {
    if (S < 0 || H <= 0 || keys <= 0 || window <= 0 || keys < S) return MLLM_HIP_ERR_SHAPE;
    if (S == 0) return MLLM_HIP_OK;
    if (!x || !y) return MLLM_HIP_ERR_ARG;
    const int64_t total = (int64_t)S * H * keys;
    hipLaunchKernelGGL(sliding_window_mask_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), x, y, S, H, keys, window);
    return MH_LAUNCH_OK("sliding_window_mask");
}
extern "C" int mllm_hip_topk_rows(const float *x, int64_t ldx, float *values, float *indices, int rows, int n, int k, void *stream) {
    if (rows < 0 || n <= 0 || k <= 0 || k > n) return MLLM_HIP_ERR_SHAPE;
    if (rows == 0) return MLLM_HIP_OK;
    if (!x || !values || !indices) return MLLM_HIP_ERR_ARG;
    hipLaunchKernelGGL(topk_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), x, ldx, values, indices, rows, n, k);
    return MH_LAUNCH_OK("topk_rows");
}
extern "C" int mllm_hip_bincount(const float *ids, int n, float *counts, int nbins, void *stream) {
    if (n < 0 || nbins <= 0 || nbins > 8192) return MLLM_HIP_ERR_SHAPE;
    if (!counts || (n > 0 && !ids)) return MLLM_HIP_ERR_ARG;
    hipLaunchKernelGGL(bincount_kernel, dim3(1), dim3(256), (size_t)nbins * sizeof(int), as_stream(stream), ids, n, counts, nbins);
    return MH_LAUNCH_OK("bincount");
}
extern "C" int mllm_hip_gather_rows(const float *src, int64_t lds, int n_src_rows, const float *idx, float *out, int64_t ldo, int R, int D, int skip_negative, void *stream) {
    if (R < 0 || D <= 0 || n_src_rows < 0) return MLLM_HIP_ERR_SHAPE;
    if (R == 0) return MLLM_HIP_OK;
    if (!src || !idx || !out) return MLLM_HIP_ERR_ARG;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(R), dim3(256), 0, as_stream(stream), src, lds, n_src_rows, idx, out, ldo, D, skip_negative);
    return MH_LAUNCH_OK("gather_rows");
}
extern "C" int mllm_hip_scatter_add_rows(float *dst, int64_t ldd, int n_dst_rows, const float *src, int64_t lds, const float *idx, int R, int D, void *stream) {
    if (R < 0 || D <= 0 || n_dst_rows < 0) return MLLM_HIP_ERR_SHAPE;
    if (R == 0) return MLLM_HIP_OK;
    if (!dst || !src || !idx) return MLLM_HIP_ERR_ARG;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((D + 255) / 256), dim3(256), 0, as_stream(stream), dst, ldd, n_dst_rows, src, lds, idx, R, D);
    return MH_LAUNCH_OK("scatter_add_rows");
}
extern "C" int mllm_hip_rope_table_ntk(float theta, int dim, int n_pos, int original_max_pos, const float *long_factor, const float *short_factor, float *sin_host,
                                       float *cos_host) {
    // CPUNTKRoPE.cpp:27-80: the table is built for max_position_embeddings positions, with the long factors iff that exceeds the original context; inv_freq divides by
    // dim (not dim / 2), as the reference does; angle = (s * (1 / ext)) * inv_freq in float; std::log(int) is the double overload, std::log(float) the float one
    if (dim <= 0 || dim % 2 || n_pos <= 0 || original_max_pos <= 1 || !long_factor || !short_factor || !sin_host || !cos_host) return MLLM_HIP_ERR_ARG;
    const int half = dim / 2;
    const float scale = (float)n_pos / (float)original_max_pos;
    const float scaling = (float)sqrt(1 + logf(scale) / log((double)original_max_pos));
    const float *ext = n_pos > original_max_pos ? long_factor : short_factor;
    for (int i = 0; i < half; ++i) {
        const float inv = 1.f / powf(theta, (float)i / (float)dim);
        const float rcp = 1.0f / ext[i];
        for (int s = 0; s < n_pos; ++s) {
            const float f = ((float)s * rcp) * inv;
            sin_host[(size_t)s * dim + i] = sin_host[(size_t)s * dim + i + half] = sinf(f) * scaling;
            cos_host[(size_t)s * dim + i] = cos_host[(size_t)s * dim + i + half] = cosf(f) * scaling;
        }
    }
    return MLLM_HIP_OK;
}
