// tools/quantizer/host_quantize.cpp -- FIXTURE TOOLING, not part of libmllm_hip.so and not declared in include/mllm_hip.h.
//
// Built into tools/libmllm_quant.so (tools/quantlib.py) and used only to write the synthetic `*-q4_k.mllm` files the tests and bench.py
// run on.  The K-quant scale fit below has to produce the reference tool's bytes, so fit_group / q4k_block follow make_qkx2_quants /
// quantize_row_q4_K_reference (QuantizeQ4.cpp:97-293) operation by operation; that closeness is why the file lives with the tooling.
//
// Host-side weight quantisers of the .mllm tool-chain: fp32 -> Q4_K / Q4_0 / Q8_0 blocks, byte-compatible with the
// reference's `quantize` tool (tools/quantizer/QuantWriter.cpp:288-300 dispatch;
// mllm/backends/cpu/third_party/ggml/QuantizeQ4.cpp:31-64 q4_0, :99-175 make_qkx2_quants, :187-293 q4_K).
// They exist so that synthetic-weight .mllm files can be produced where the reference's tool is not available
// (the GPU box): SURVEY §8 row N1 / component 18.  Every fused multiply-add the reference binary (g++ -O2 -mfma, default
// contraction) performs in the K-quant scale fit is written here as an explicit fmaf() and this file is compiled with
// -ffp-contract=off, so the search rounds identically whatever the compiler; the byte-for-byte
// agreement with the reference tool is checked by tests/test_host.py against committed digests (tests/golden/*_digests.json).
//
// Rows are independent, so quantisation is parallelised over blocks with OpenMP (the reference tool is serial).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <immintrin.h>


// DataType values of mllm/Types.hpp:63-97
enum { QT_F32 = 0, QT_F16 = 1, QT_Q4_0 = 2, QT_Q8_0 = 8, QT_Q4_K = 12, QT_Q8_K = 15 };

namespace {

#pragma pack(push, 1)
struct BlockQ40 { uint16_t d; uint8_t qs[16]; };
struct BlockQ80 { uint16_t d; int8_t qs[32]; };
struct BlockQ4K { uint16_t d; uint16_t dmin; uint8_t scales[12]; uint8_t qs[128]; };
#pragma pack(pop)

inline uint16_t to_f16(float f) { return _cvtss_sh(f, 0); }
inline float from_f16(uint16_t h) { return _cvtsh_ss(h); }

inline int round_magic(float fval) {  // nearest_int, Quantize.hpp:174-180
    float val = fval + 12582912.F;
    int i;
    std::memcpy(&i, &val, sizeof(int));
    return (i & 0x007fffff) - 0x00400000;
}

void q40_block(const float *x, BlockQ40 *y) {
    float amax = 0.0F, max = 0.0F;
    for (int j = 0; j < 32; j++) {
        const float v = x[j];
        if (amax < fabsf(v)) { amax = fabsf(v); max = v; }
    }
    const float d = max / -8;
    const float id = d ? 1.0F / d : 0.0F;
    y->d = to_f16(d);
    for (int j = 0; j < 16; ++j) {
        // `x*id + 8.5F` is one fused multiply-add in the reference build (g++ -O2 -mfma contraction)
        int8_t a = (int8_t)fmaf(x[j], id, 8.5F), b = (int8_t)fmaf(x[16 + j], id, 8.5F);
        const uint8_t xi0 = a < 15 ? a : 15;
        const uint8_t xi1 = b < 15 ? b : 15;
        y->qs[j] = xi0;
        y->qs[j] |= xi1 << 4;
    }
}

void q80_block(const float *x, BlockQ80 *y) {  // QuantizeQ8.cpp:32-55
    float amax = 0.0f;
    for (int j = 0; j < 32; j++) { const float v = fabsf(x[j]); amax = amax > v ? amax : v; }
    const float d = amax / ((1 << 7) - 1);
    const float id = d ? 1.0f / d : 0.0f;
    y->d = to_f16(d);
    for (int j = 0; j < 32; ++j) y->qs[j] = roundf(x[j] * id);
}

// weighted min/scale fit of one 32-group to q in [0,nmax]: x ~ scale*q - the_min  (make_qkx2_quants)
float fit_group(int n, int nmax, const float *x, const float *weights, uint8_t *L, float *the_min, uint8_t *Laux,
                float rmin, float rdelta, int nstep) {
    float min = x[0];
    float max = x[0];
    float sum_w = weights[0];
    float sum_x = sum_w * x[0];
    for (int i = 1; i < n; ++i) {
        if (x[i] < min) min = x[i];
        if (x[i] > max) max = x[i];
        float w = weights[i];
        sum_w += w;
        sum_x = fmaf(w, x[i], sum_x);
    }
    if (min > 0) min = 0;
    if (max == min) {
        for (int i = 0; i < n; ++i) L[i] = 0;
        *the_min = -min;
        return 0.F;
    }
    float iscale = nmax / (max - min);
    float scale = 1 / iscale;
    float best_mad = 0;
    for (int i = 0; i < n; ++i) {
        int l = round_magic(iscale * (x[i] - min));
        L[i] = l < 0 ? 0 : (l > nmax ? nmax : l);
        float diff = fmaf(scale, (float)L[i], min) - x[i];
        diff = diff * diff;
        float w = weights[i];
        best_mad = fmaf(w, diff, best_mad);
    }
    if (nstep < 1) {
        *the_min = -min;
        return scale;
    }
    for (int is = 0; is <= nstep; ++is) {
        iscale = (fmaf(rdelta, (float)is, rmin) + nmax) / (max - min);
        float sum_l = 0;
        float sum_l2 = 0;
        float sum_xl = 0;
        for (int i = 0; i < n; ++i) {
            int l = round_magic(iscale * (x[i] - min));
            l = l < 0 ? 0 : (l > nmax ? nmax : l);
            Laux[i] = l;
            float w = weights[i];
            const float wl = w * l;
            sum_l += wl;
            sum_l2 = fmaf(wl, (float)l, sum_l2);
            sum_xl = fmaf(wl, x[i], sum_xl);
        }
        float D = fmaf(sum_w, sum_l2, -(sum_l * sum_l));
        if (D > 0) {
            // contraction as emitted for the reference build: min numerator fuses its first product, scale numerator its second
            float this_min = fmaf(sum_l2, sum_x, -(sum_l * sum_xl)) / D;
            float this_scale = fmaf(-sum_l, sum_x, sum_w * sum_xl) / D;
            if (this_min > 0) {
                this_min = 0;
                this_scale = sum_xl / sum_l2;
            }
            float mad = 0;
            for (int i = 0; i < n; ++i) {
                float diff = fmaf((float)Laux[i], this_scale, this_min) - x[i];
                diff = diff * diff;
                float w = weights[i];
                mad = fmaf(w, diff, mad);
            }
            if (mad < best_mad) {
                for (int i = 0; i < n; ++i) L[i] = Laux[i];
                best_mad = mad;
                scale = this_scale;
                min = this_min;
            }
        }
    }
    *the_min = -min;
    return scale;
}

inline void unpack_scale_min(int j, const uint8_t *q, uint8_t *d, uint8_t *m) {  // get_scale_min_k4
    if (j < 4) { *d = q[j] & 63; *m = q[j + 4] & 63; }
    else { *d = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); *m = (q[j + 4] >> 4) | ((q[j - 0] >> 6) << 4); }
}

void q4k_block(const float *x, BlockQ4K *y) {
    uint8_t L[256];
    uint8_t Laux[32];
    float weights[32];
    float mins[8];
    float scales[8];
    float max_scale = 0;
    float max_min = 0;
    for (int j = 0; j < 8; ++j) {
        float sum_x2 = 0;
        for (int l = 0; l < 32; ++l) sum_x2 = fmaf(x[32 * j + l], x[32 * j + l], sum_x2);
        float av_x = sqrtf(sum_x2 / 32);
        for (int l = 0; l < 32; ++l) weights[l] = av_x + fabsf(x[32 * j + l]);
        scales[j] = fit_group(32, 15, x + 32 * j, weights, L + 32 * j, &mins[j], Laux, -1.F, 0.1F, 20);
        float scale = scales[j];
        if (scale > max_scale) max_scale = scale;
        float min = mins[j];
        if (min > max_min) max_min = min;
    }
    float inv_scale = max_scale > 0 ? 63.F / max_scale : 0.F;
    float inv_min = max_min > 0 ? 63.F / max_min : 0.F;
    std::memset(y->scales, 0, sizeof(y->scales));
    for (int j = 0; j < 8; ++j) {
        uint8_t ls = round_magic(inv_scale * scales[j]);
        uint8_t lm = round_magic(inv_min * mins[j]);
        ls = ls < 63 ? ls : 63;
        lm = lm < 63 ? lm : 63;
        if (j < 4) {
            y->scales[j] = ls;
            y->scales[j + 4] = lm;
        } else {
            y->scales[j + 4] = (ls & 0xF) | ((lm & 0xF) << 4);
            y->scales[j - 4] |= ((ls >> 4) << 6);
            y->scales[j - 0] |= ((lm >> 4) << 6);
        }
    }
    y->d = to_f16(max_scale / 63.F);
    y->dmin = to_f16(max_min / 63.F);
    uint8_t sc, m;
    for (int j = 0; j < 8; ++j) {
        unpack_scale_min(j, y->scales, &sc, &m);
        const float d = from_f16(y->d) * sc;
        if (d == 0.0F) continue;
        const float dm = from_f16(y->dmin) * m;
        for (int ii = 0; ii < 32; ++ii) {
            int l = round_magic((x[32 * j + ii] + dm) / d);
            l = l < 0 ? 0 : (l > 15 ? 15 : l);
            L[32 * j + ii] = l;
        }
    }
    uint8_t *q = y->qs;
    for (int j = 0; j < 256; j += 64) {
        for (int l = 0; l < 32; ++l) q[l] = L[j + l] | (L[j + l + 32] << 4);
        q += 32;
    }
}

}  // namespace

extern "C" int64_t mllm_quant_nbytes(int dtype, int64_t n) {
    switch (dtype) {
    case QT_F32: return n * 4;
    case QT_F16: return n * 2;
    case QT_Q4_0: return n % 32 ? -1 : n / 32 * 18;
    case QT_Q8_0: return n % 32 ? -1 : n / 32 * 34;
    case QT_Q4_K: return n % 256 ? -1 : n / 256 * 144;
    case QT_Q8_K: return n % 256 ? -1 : n / 256 * 292;
    default: return -1;
    }
}

extern "C" int mllm_quant_rows(int dtype, const float *x, void *y, int64_t n) {
    if (mllm_quant_nbytes(dtype, n) < 0) return -2;
    if (dtype == QT_Q4_K) {
        const int64_t nb = n / 256;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < nb; ++i) q4k_block(x + i * 256, (BlockQ4K *)y + i);
    } else if (dtype == QT_Q4_0) {
        const int64_t nb = n / 32;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < nb; ++i) q40_block(x + i * 32, (BlockQ40 *)y + i);
    } else if (dtype == QT_Q8_0) {
        const int64_t nb = n / 32;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < nb; ++i) q80_block(x + i * 32, (BlockQ80 *)y + i);
    } else if (dtype == QT_F32) {
        std::memcpy(y, x, n * 4);
    } else {
        return -3;
    }
    return 0;
}
