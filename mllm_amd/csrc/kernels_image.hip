// mllm_amd/csrc/kernels_image.hip -- SURVEY N3: Qwen2-VL image preprocessing on the device.
//
// Qwen2VLImageProcessor::preprocess_images (mllm/models/qwen2_vl/processing_qwen2_vl.hpp:190-235) after the image decode: RescaleImage (x / 255, PreProcess.cpp:37-43) ->
// smart_resize (:84-109) -> ResizeImages(..., BICUBIC) = stb_image_resize2's cubic B-spline with edge clamp (PreProcess.cpp:84-154) -> NormalizeImages per channel
// (PreProcess.cpp:233-262) -> the frame doubled -> convertPatches (:119-177), giving the `[grid_h * grid_w][3 * 2 * 14 * 14]` rows the vision tower's patch embedding reads.
// The resize is third-party code vendored by the reference (third_party/stb/stb_image_resize2.h); its published algorithm -- per axis a gather whose coefficients are the
// kernel at the tap distance, normalised to 1, out-of-range taps folded into the edge pixel (:2821-2832, :3176-3500) -- is what is implemented: the coefficient tables are
// built on the host in the library's float arithmetic, the two gathers + normalise + patchify run as two kernels.  The library's SIMD loops fix an order of float additions
// that is not reproduced, so this path is held to the reference within a tolerance (tests/test_preprocess.py: 2e-5 absolute on values of magnitude <= 2.7), not bit for bit;
// oracle/oracle.py:qwen2vl_preprocess is the same restatement and the device result equals it exactly.
#include <cmath>
#include <map>
#include <vector>

#include "common.h"

using namespace mllm_hip;

namespace {

// smart_resize (processing_qwen2_vl.hpp:84-109)
bool smart_resize_host(int height, int width, int factor, int min_pixels, int max_pixels, int *hb, int *wb) {
    if (height <= 0 || width <= 0 || factor <= 0) return false;
    if (std::max(height, width) / static_cast<float>(std::min(height, width)) > 200) return false;      // MAX_RATIO
    const int64_t hw64 = (int64_t)height * width;
    if (hw64 > 0x7fffffff) return false;      // the reference forms height * width in int (processing_qwen2_vl.hpp:97,101): beyond that it has no defined result
    const int hw = (int)hw64;
    auto round_by_factor = [](int value, int f) { return ((value + f / 2) / f) * f; };
    auto floor_by_factor = [](float value, int f) { return static_cast<int>(std::floor(value / f)) * f; };
    auto ceil_by_factor = [](float value, int f) { return static_cast<int>(std::ceil(value / f)) * f; };
    int h_bar = std::max(factor, round_by_factor(height, factor)), w_bar = std::max(factor, round_by_factor(width, factor));
    if ((int64_t)h_bar * w_bar > max_pixels) {
        const float beta = std::sqrt(hw / static_cast<float>(max_pixels));
        h_bar = floor_by_factor(height / beta, factor);
        w_bar = floor_by_factor(width / beta, factor);
    } else if ((int64_t)h_bar * w_bar < min_pixels) {
        const float beta = std::sqrt(min_pixels / static_cast<float>(hw));
        h_bar = ceil_by_factor(height * beta, factor);
        w_bar = ceil_by_factor(width * beta, factor);
    }
    *hb = h_bar; *wb = w_bar;
    return h_bar > 0 && w_bar > 0;
}

float bspline(float x) {      // stbir__filter_cubic
    if (x < 0.0f) x = -x;
    if (x < 1.0f) return (4.0f + x * x * (3.0f * x - 6.0f)) / 6.0f;
    if (x < 2.0f) return (8.0f + x * (-12.0f + x * (6.0f - x))) / 6.0f;
    return 0.0f;
}

// per output index: first input pixel n0[o], tap count nt[o], coefficients coef[o * width + k]; returns the table width
int axis_taps(int in_size, int out_size, std::vector<int> &n0, std::vector<int> &nt, std::vector<float> &coef) {
    const float small = 1.0f / (1 << 20);
    const double ratio = (double)out_size / (double)in_size;
    const float scale = (float)ratio, inv_scale = (float)(1.0 / ratio);
    std::vector<std::map<int, float>> taps(out_size);
    if (scale >= 1.0f - small) {
        const float radius = 2.0f * scale;
        for (int n = 0; n < out_size; ++n) {
            const float oc = (float)n + 0.5f, centre = oc * inv_scale;
            const int first = (int)std::floor((oc - radius) * inv_scale + 0.5f), last = (int)std::floor((oc + radius) * inv_scale - 0.5f);
            for (int p = first; p <= last; ++p) {
                const float c = bspline(centre - ((float)p + 0.5f));
                if (!(c < small && c > -small)) taps[n][p] = c;
            }
        }
    } else {
        const float radius = 2.0f * inv_scale;
        const int margin = (int)std::ceil(2.0 * 2.0 / (double)scale) / 2 + 1;
        for (int p = -margin; p < in_size + margin; ++p) {
            const float ic = (float)p + 0.5f, out_centre = ic * scale;
            const int first = std::max(0, (int)std::floor((ic - radius) * scale + 0.5f)), last = std::min(out_size - 1, (int)std::floor((ic + radius) * scale - 0.5f));
            for (int o = first; o <= last; ++o) {
                const float c = bspline(((float)o + 0.5f) - out_centre) * scale;
                if (!(c < small && c > -small)) taps[o][p] = c;
            }
        }
    }
    std::vector<std::map<int, float>> folded(out_size);
    int width = 1;
    for (int o = 0; o < out_size; ++o) {
        float total = 0.0f;
        for (auto &kv : taps[o]) total = total + kv.second;
        const bool rescale = total < 1.0f - small || total > 1.0f + small;
        const float fs = 1.0f / total;
        for (auto &kv : taps[o]) {
            const float c = rescale ? kv.second * fs : kv.second;
            const int q = std::min(std::max(kv.first, 0), in_size - 1);      // STBIR_EDGE_CLAMP
            auto it = folded[o].find(q);
            if (it == folded[o].end()) folded[o][q] = c; else it->second = it->second + c;
        }
        if (folded[o].empty()) folded[o][std::min(std::max((int)(((float)o + 0.5f) * inv_scale), 0), in_size - 1)] = 1.0f;
        width = std::max(width, folded[o].rbegin()->first - folded[o].begin()->first + 1);
    }
    n0.assign(out_size, 0); nt.assign(out_size, 0); coef.assign((size_t)out_size * width, 0.0f);
    for (int o = 0; o < out_size; ++o) {
        n0[o] = folded[o].begin()->first;
        nt[o] = folded[o].rbegin()->first - n0[o] + 1;
        for (auto &kv : folded[o]) coef[(size_t)o * width + (kv.first - n0[o])] = kv.second;
    }
    return width;
}

// horizontal gather of rgb / 255: tmp[y][x][c] = sum_k coef[x][k] * (rgb[y][n0[x] + k][c] / 255), in tap order, multiply and add separately rounded
__global__ __launch_bounds__(256) void img_resize_h_kernel(const uint8_t *__restrict__ rgb, const int *__restrict__ n0, const int *__restrict__ nt, const float *__restrict__ coef,
                                                           int tw, float *__restrict__ tmp, int H, int W, int Wo) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)H * Wo * 3) return;
    const int c = (int)(gid % 3), x = (int)((gid / 3) % Wo), y = (int)(gid / (3 * (int64_t)Wo));
    const uint8_t *src = rgb + ((int64_t)y * W + n0[x]) * 3 + c;
    const float *cf = coef + (int64_t)x * tw;
    float acc = 0.0f;
    for (int k = 0; k < nt[x]; ++k) acc = __fadd_rn(acc, __fmul_rn(__fdiv_rn((float)src[3 * k], 255.0f), cf[k]));
    tmp[gid] = acc;
}
// vertical gather + (v - mean) / std + convertPatches' index map; both temporal slots of a patch row receive the frame
__global__ __launch_bounds__(256) void img_resize_v_patch_kernel(const float *__restrict__ tmp, const int *__restrict__ n0, const int *__restrict__ nt, const float *__restrict__ coef,
                                                                 int th, float *__restrict__ patches, int Ho, int Wo, float m0, float m1, float m2, float s0, float s1, float s2,
                                                                 int patch, int merge, int tps) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)Ho * Wo * 3) return;
    const int c = (int)(gid % 3), x = (int)((gid / 3) % Wo), y = (int)(gid / (3 * (int64_t)Wo));
    const float *src = tmp + ((int64_t)n0[y] * Wo + x) * 3 + c;
    const float *cf = coef + (int64_t)y * th;
    float acc = 0.0f;
    for (int k = 0; k < nt[y]; ++k) acc = __fadd_rn(acc, __fmul_rn(src[(int64_t)k * Wo * 3], cf[k]));
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    const float v = __fdiv_rn(__fsub_rn(acc, mean), sd);
    // row i = ((d1 * gwm + d2) * ms + d3) * ms + d4 with h = (d1 ms + d3) patch + d7, w = (d2 ms + d4) patch + d8; column j = ((c * tps + t) * patch + d7) * patch + d8
    const int gwm = Wo / patch / merge, ph = y / patch, d7 = y % patch, pw = x / patch, d8 = x % patch;
    const int d1 = ph / merge, d3 = ph % merge, d2 = pw / merge, d4 = pw % merge;
    const int64_t row = (((int64_t)d1 * gwm + d2) * merge + d3) * merge + d4, cols = (int64_t)3 * tps * patch * patch;
    for (int t = 0; t < tps; ++t) patches[row * cols + (((int64_t)c * tps + t) * patch + d7) * patch + d8] = v;
}

}  // namespace

extern "C" int mllm_hip_qwen2vl_preprocess_shape(int height, int width, int min_pixels, int max_pixels, int32_t *grid_thw) {
    int hb, wb;
    if (!grid_thw || !smart_resize_host(height, width, 28, min_pixels, max_pixels, &hb, &wb)) return MLLM_HIP_ERR_SHAPE;
    grid_thw[0] = 1; grid_thw[1] = hb / 14; grid_thw[2] = wb / 14;
    return MLLM_HIP_OK;
}

extern "C" int mllm_hip_qwen2vl_preprocess(const uint8_t *rgb_host, int height, int width, int min_pixels, int max_pixels, float *patches_dev, int32_t *grid_thw, void *stream) {
    if (!rgb_host || !patches_dev || !grid_thw) return MLLM_HIP_ERR_ARG;
    int Ho, Wo;
    if (!smart_resize_host(height, width, 28, min_pixels, max_pixels, &Ho, &Wo)) return MLLM_HIP_ERR_SHAPE;
    constexpr int patch = 14, merge = 2, tps = 2;
    if (Ho % (patch * merge) || Wo % (patch * merge)) return MLLM_HIP_ERR_SHAPE;
    hipStream_t st = as_stream(stream);
    std::vector<int> n0w, ntw, n0h, nth;
    std::vector<float> cw, ch;
    const int tw = axis_taps(width, Wo, n0w, ntw, cw), th = axis_taps(height, Ho, n0h, nth, ch);
    // one stream-ordered scratch: rgb | horizontal result | the two tap tables
    const size_t b_rgb = ((size_t)height * width * 3 + 255) & ~(size_t)255, b_tmp = (size_t)height * Wo * 3 * 4;
    const size_t b_tab = ((size_t)(2 * Wo + 2 * Ho) * 4 + cw.size() * 4 + ch.size() * 4 + 255) & ~(size_t)255;
    uint8_t *ws = nullptr;
    MH_CHECK(hipMallocAsync((void **)&ws, b_rgb + b_tmp + b_tab, st));
    uint8_t *d_rgb = ws;
    float *d_tmp = (float *)(ws + b_rgb);
    int *d_n0w = (int *)(ws + b_rgb + b_tmp), *d_ntw = d_n0w + Wo, *d_n0h = d_ntw + Wo, *d_nth = d_n0h + Ho;
    float *d_cw = (float *)(d_nth + Ho), *d_ch = d_cw + cw.size();
    // from here on every exit returns the scratch block: errors are collected, not returned on the spot
    int rc = MLLM_HIP_OK;
    auto step = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && !rc) { set_error(what, e, __FILE__, __LINE__); rc = MLLM_HIP_ERR_HIP; }
    };
    step(hipMemcpyAsync(d_rgb, rgb_host, (size_t)height * width * 3, hipMemcpyHostToDevice, st), "hipMemcpyAsync(rgb)");
    step(hipMemcpyAsync(d_n0w, n0w.data(), (size_t)Wo * 4, hipMemcpyHostToDevice, st), "hipMemcpyAsync(n0w)");
    step(hipMemcpyAsync(d_ntw, ntw.data(), (size_t)Wo * 4, hipMemcpyHostToDevice, st), "hipMemcpyAsync(ntw)");
    step(hipMemcpyAsync(d_n0h, n0h.data(), (size_t)Ho * 4, hipMemcpyHostToDevice, st), "hipMemcpyAsync(n0h)");
    step(hipMemcpyAsync(d_nth, nth.data(), (size_t)Ho * 4, hipMemcpyHostToDevice, st), "hipMemcpyAsync(nth)");
    step(hipMemcpyAsync(d_cw, cw.data(), cw.size() * 4, hipMemcpyHostToDevice, st), "hipMemcpyAsync(cw)");
    step(hipMemcpyAsync(d_ch, ch.data(), ch.size() * 4, hipMemcpyHostToDevice, st), "hipMemcpyAsync(ch)");
    step(hipStreamSynchronize(st), "hipStreamSynchronize");      // the host tables go out of scope (pageable memory: the copies above are staged, the sync makes that explicit)
    const int64_t n1 = (int64_t)height * Wo * 3, n2 = (int64_t)Ho * Wo * 3;
    if (!rc) {
        hipLaunchKernelGGL(img_resize_h_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, d_rgb, d_n0w, d_ntw, d_cw, tw, d_tmp, height, width, Wo);
        rc = MH_LAUNCH_OK("img_resize_h");
    }
    if (!rc) {
        // mean_ / std_ of Qwen2VLImageProcessor (processing_qwen2_vl.hpp:70-71)
        hipLaunchKernelGGL(img_resize_v_patch_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, d_tmp, d_n0h, d_nth, d_ch, th, patches_dev, Ho, Wo, 0.48145466f, 0.4578275f,
                           0.40821073f, 0.26862954f, 0.26130258f, 0.27577711f, patch, merge, tps);
        rc = MH_LAUNCH_OK("img_resize_v_patch");
    }
    step(hipFreeAsync(ws, st), "hipFreeAsync");
    if (rc) return rc;
    grid_thw[0] = 1; grid_thw[1] = Ho / patch; grid_thw[2] = Wo / patch;
    return rc;
}
