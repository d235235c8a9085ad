/*
 * oracle/restate.c -- TEST INFRASTRUCTURE (the oracle), NOT product code.
 *
 * A plain-C CPU restatement of the reference's (yirongjie/mllm) x86 CPU numerics for the Op/Layer hot path.
 * It exists only so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check / time the HIP
 * path; nothing under mllm_amd/ may link, import or call it.  Every function cites the reference file:line it
 * restates.  It is pinned against outputs of the compiled reference itself (oracle/_ref, built by
 * oracle/Makefile.ref; vectors under tests/golden/ produced by oracle/make_golden.py).
 *
 * Build: gcc -O2 -mavx2 -mf16c -mfma -ffp-contract=off -fopenmp -shared -fPIC oracle/restate.c -o oracle/liboracle.so -lm
 * (-ffp-contract=off: every fused multiply-add below is an explicit fmaf() where the reference's AVX2 code uses an
 *  FMA intrinsic; plain a*b+c stays two roundings.)
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <immintrin.h>

#define QK_K 256
#define QK4_0 32
#define QK8_0 32

/* block layouts == on-disk layouts: mllm/DataType.hpp:75-78 (q4_0), :93-98 (q4_K), :137-140 (q8_0), :159-163 (q8_K) */
#pragma pack(push, 1)
typedef struct { uint16_t d; uint8_t qs[QK4_0 / 2]; } block_q4_0;                      /* 18 B */
typedef struct { uint16_t d; int8_t qs[QK8_0]; } block_q8_0;                           /* 34 B */
typedef struct { uint16_t d; uint16_t dmin; uint8_t scales[12]; uint8_t qs[QK_K / 2]; } block_q4_K; /* 144 B */
typedef struct { float d; int8_t qs[QK_K]; int16_t bsums[QK_K / 16]; } block_q8_K;     /* 292 B */
#pragma pack(pop)

/* fp16 <-> fp32: the reference uses F16C on x86 (_cvtss_sh/_cvtsh_ss, third_party/ggml/ComputeUtils.hpp:223) */
float orc_f16_to_f32(uint16_t h) { return _cvtsh_ss(h); }
uint16_t orc_f32_to_f16(float f) { return _cvtss_sh(f, 0); }

/* third_party/ggml/Quantize.hpp:174-180 (magic-add round-to-nearest-even) */
static inline int nearest_int(float fval) {
    float val = fval + 12582912.F;
    int i;
    memcpy(&i, &val, sizeof(int));
    return (i & 0x007fffff) - 0x00400000;
}

/* ------------------------------------------------------------------------------------------------------------
 * A4: activation quantisers.  third_party/ggml/QuantizeQ8.cpp:216-251 (Q8_K) and :32-55 (Q8_0)
 * ---------------------------------------------------------------------------------------------------------- */
void orc_quantize_row_q8_K(const float *x, void *vy, int k) {
    block_q8_K *y = (block_q8_K *)vy;
    const int nb = k / QK_K;
    for (int i = 0; i < nb; i++) {
        float max = 0, amax = 0;
        for (int j = 0; j < QK_K; ++j) {
            float ax = fabsf(x[j]);
            if (ax > amax) { amax = ax; max = x[j]; }
        }
        if (amax == 0.0F) {
            y[i].d = 0;
            memset(y[i].qs, 0, QK_K);
            /* NB: the reference leaves bsums untouched here (stale memory); we zero them (the only consistent value). */
            memset(y[i].bsums, 0, sizeof(y[i].bsums));
            x += QK_K;
            continue;
        }
        const float iscale = -128.F / max;
        for (int j = 0; j < QK_K; ++j) {
            int v = nearest_int(iscale * x[j]);
            y[i].qs[j] = v < 127 ? v : 127;
        }
        for (int j = 0; j < QK_K / 16; ++j) {
            int sum = 0;
            for (int ii = 0; ii < 16; ++ii) sum += y[i].qs[j * 16 + ii];
            y[i].bsums[j] = sum;
        }
        y[i].d = 1 / iscale;
        x += QK_K;
    }
}

void orc_quantize_row_q8_0(const float *x, void *vy, int k) {
    block_q8_0 *y = (block_q8_0 *)vy;
    const int nb = k / QK8_0;
    for (int i = 0; i < nb; i++) {
        float amax = 0.0f;
        for (int j = 0; j < QK8_0; j++) { float v = fabsf(x[i * QK8_0 + j]); amax = amax > v ? amax : v; }
        const float d = amax / ((1 << 7) - 1);
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d = orc_f32_to_f16(d);
        for (int j = 0; j < QK8_0; ++j) y[i].qs[j] = roundf(x[i * QK8_0 + j] * id);
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * A6/A8: dequantisers.  QuantizeQ4.cpp:74-93 (q4_0), :295-333 (q4_K); 6-bit scale unpack :177-184
 * ---------------------------------------------------------------------------------------------------------- */
static inline void get_scale_min_k4(int j, const uint8_t *q, uint8_t *d, uint8_t *m) {
    if (j < 4) { *d = q[j] & 63; *m = q[j + 4] & 63; }
    else { *d = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); *m = (q[j + 4] >> 4) | ((q[j - 0] >> 6) << 4); }
}

void orc_dequantize_row_q4_0(const void *vx, float *y, int k) {
    const block_q4_0 *x = (const block_q4_0 *)vx;
    const int nb = k / QK4_0;
    for (int i = 0; i < nb; i++) {
        const float d = orc_f16_to_f32(x[i].d);
        for (int j = 0; j < QK4_0 / 2; ++j) {
            const int x0 = (x[i].qs[j] & 0x0F) - 8;
            const int x1 = (x[i].qs[j] >> 4) - 8;
            y[i * QK4_0 + j + 0] = x0 * d;
            y[i * QK4_0 + j + QK4_0 / 2] = x1 * d;
        }
    }
}

void orc_dequantize_row_q4_K(const void *vx, float *y, int k) {
    const block_q4_K *x = (const block_q4_K *)vx;
    const int nb = k / QK_K;
    for (int i = 0; i < nb; i++) {
        const uint8_t *q = x[i].qs;
        const float d = orc_f16_to_f32(x[i].d), min = orc_f16_to_f32(x[i].dmin);
        int is = 0;
        uint8_t sc, m;
        for (int j = 0; j < QK_K; j += 64) {
            get_scale_min_k4(is + 0, x[i].scales, &sc, &m);
            const float d1 = d * sc, m1 = min * m;
            get_scale_min_k4(is + 1, x[i].scales, &sc, &m);
            const float d2 = d * sc, m2 = min * m;
            for (int l = 0; l < 32; ++l) *y++ = d1 * (q[l] & 0xF) - m1;
            for (int l = 0; l < 32; ++l) *y++ = d2 * (q[l] >> 4) - m2;
            q += 32; is += 2;
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * A5: vec_dot_q4_K_q8_K, restating the AVX2 path (VecDotQ4.cpp:220-283) lane by lane so the fp32 accumulation
 * order is the reference's: 8 fp32 lanes `acc` (one per 4-byte column group t of each 32-byte chunk), 4 lanes
 * `acc_m` for the mins, one fmaf per super-block per lane, then hsum_float_8 + the acc_m shuffle-add.
 * ---------------------------------------------------------------------------------------------------------- */
static inline float hsum_float_8(const float a[8]) {
    /* ComputeUtils.hpp hsum_float_8: (lo128+hi128) -> movehl add -> movehdup add_ss */
    float r0 = a[0] + a[4], r1 = a[1] + a[5], r2 = a[2] + a[6], r3 = a[3] + a[7];
    float s0 = r0 + r2, s1 = r1 + r3;
    return s0 + s1;
}

float orc_vec_dot_q4_K_q8_K(int n, const void *vx, const void *vy) {
    const block_q4_K *x = (const block_q4_K *)vx;
    const block_q8_K *y = (const block_q8_K *)vy;
    const int nb = n / QK_K;
    float acc[8] = {0}, acc_m[4] = {0};
    for (int i = 0; i < nb; ++i) {
        const float d = y[i].d * orc_f16_to_f32(x[i].d);
        const float dmin = -y[i].d * orc_f16_to_f32(x[i].dmin);
        uint8_t sc[8], mn[8];
        for (int j = 0; j < 8; ++j) get_scale_min_k4(j, x[i].scales, &sc[j], &mn[j]);
        /* q8s[k] = bsums[2k]+bsums[2k+1] (hadd_epi16), prod[u] = mins[2u]*q8s[2u] + mins[2u+1]*q8s[2u+1] (madd_epi16) */
        for (int u = 0; u < 4; ++u) {
            int q0 = (int16_t)(y[i].bsums[4 * u] + y[i].bsums[4 * u + 1]);
            int q1 = (int16_t)(y[i].bsums[4 * u + 2] + y[i].bsums[4 * u + 3]);
            int prod = mn[2 * u] * q0 + mn[2 * u + 1] * q1;
            acc_m[u] = fmaf(dmin, (float)prod, acc_m[u]);
        }
        int32_t sumi[8] = {0};
        const uint8_t *q4 = x[i].qs;
        const int8_t *q8 = y[i].qs;
        for (int j = 0; j < QK_K / 64; ++j) {
            for (int t = 0; t < 8; ++t) {
                int pl = 0, ph = 0;
                for (int b = 0; b < 4; ++b) {
                    pl += (q4[4 * t + b] & 0xF) * q8[4 * t + b];
                    ph += (q4[4 * t + b] >> 4) * q8[32 + 4 * t + b];
                }
                sumi[t] += sc[2 * j] * pl + sc[2 * j + 1] * ph;
            }
            q4 += 32; q8 += 64;
        }
        for (int t = 0; t < 8; ++t) acc[t] = fmaf(d, (float)sumi[t], acc[t]);
    }
    float m01 = acc_m[0] + acc_m[2], m11 = acc_m[1] + acc_m[3];
    return hsum_float_8(acc) + (m01 + m11);
}

/* A6: vec_dot_q4_0_q8_0 AVX path (VecDotQ4.cpp:514-545): per block q = mul_sum_i8_pairs_float -> 8 int32 lanes
 * (lane t = bytes 4t..4t+3 of the 32 unpacked nibbles-8), acc = fma(d, q, acc). bytes_from_nibbles_32 puts the 16 low
 * nibbles in bytes 0..15 and the 16 high nibbles in bytes 16..31. */
float orc_vec_dot_q4_0_q8_0(int n, const void *vx, const void *vy) {
    const block_q4_0 *x = (const block_q4_0 *)vx;
    const block_q8_0 *y = (const block_q8_0 *)vy;
    const int nb = n / QK8_0;
    float acc[8] = {0};
    for (int i = 0; i < nb; ++i) {
        const float d = orc_f16_to_f32(x[i].d) * orc_f16_to_f32(y[i].d);
        int8_t bx[32];
        for (int j = 0; j < 16; ++j) { bx[j] = (x[i].qs[j] & 0xF) - 8; bx[16 + j] = (x[i].qs[j] >> 4) - 8; }
        for (int t = 0; t < 8; ++t) {
            int s = 0;
            for (int b = 0; b < 4; ++b) s += bx[4 * t + b] * y[i].qs[4 * t + b];
            acc[t] = fmaf(d, (float)s, acc[t]);
        }
    }
    return hsum_float_8(acc);
}

/* vec_dot_fp32 AVX2 (VecDotFP32.cpp:31-58): 4 accumulators of 8 lanes over steps of 32, reduce, scalar leftovers */
float orc_vec_dot_f32(int n, const float *x, const float *y) {
    float sum[4][8];
    memset(sum, 0, sizeof(sum));
    const int np = n & ~31;
    for (int i = 0; i < np; i += 32)
        for (int j = 0; j < 4; ++j)
            for (int l = 0; l < 8; ++l) sum[j][l] = fmaf(x[i + 8 * j + l], y[i + 8 * j + l], sum[j][l]);
    /* MLLM_F32_VEC_REDUCE (ggml GGML_F32x8_REDUCE): sum0+=sum2 ; sum1+=sum3 ; sum0+=sum1 ; then 128-bit hadd,hadd */
    float a[8];
    for (int l = 0; l < 8; ++l) { float s0 = sum[0][l] + sum[2][l], s1 = sum[1][l] + sum[3][l]; a[l] = s0 + s1; }
    float t[4] = {a[0] + a[4], a[1] + a[5], a[2] + a[6], a[3] + a[7]};
    float h0 = t[0] + t[1], h1 = t[2] + t[3];
    float sumf = h0 + h1;
    for (int i = np; i < n; ++i) sumf = fmaf(x[i], y[i], sumf);   /* contracted by the reference build (g++ -O2 -mfma) */
    return sumf;
}

/* ------------------------------------------------------------------------------------------------------------
 * A1/A2: Linear = mat_mul(x, W^T) (+bias).  compute/Matmul.cpp:77-120 (quantise x rows to the weight's vec_dot_type),
 * :219-276 (vec_dot per (m,n), bias add, fp32 or fp16 store).  wdtype: 0 F32, 2 Q4_0, 12 Q4_K (mllm/Types.hpp:63-97)
 * out_f16 != 0 restates the store branch :262-268 (output aliases the fp16 KV slab).
 * ---------------------------------------------------------------------------------------------------------- */
void orc_linear(const float *x, int M, int K, const void *W, int wdtype, int N, const float *bias, void *yout, int out_f16) {
    size_t xrow = 0, wrow = 0;
    uint8_t *xq = NULL;
    if (wdtype == 12) { xrow = (size_t)K / QK_K * sizeof(block_q8_K); wrow = (size_t)K / QK_K * sizeof(block_q4_K); }
    else if (wdtype == 2) { xrow = (size_t)K / QK8_0 * sizeof(block_q8_0); wrow = (size_t)K / QK4_0 * sizeof(block_q4_0); }
    else { wrow = (size_t)K * 4; }
    if (xrow) {
        xq = (uint8_t *)malloc(xrow * M);
#pragma omp parallel for
        for (int m = 0; m < M; ++m) {
            if (wdtype == 12) orc_quantize_row_q8_K(x + (size_t)m * K, xq + xrow * m, K);
            else orc_quantize_row_q8_0(x + (size_t)m * K, xq + xrow * m, K);
        }
    }
#pragma omp parallel for collapse(2) schedule(static)
    for (int m = 0; m < M; ++m) {
        for (int n = 0; n < N; ++n) {
            const uint8_t *w = (const uint8_t *)W + wrow * n;
            float tmp;
            if (wdtype == 12) tmp = orc_vec_dot_q4_K_q8_K(K, w, xq + xrow * m);
            else if (wdtype == 2) tmp = orc_vec_dot_q4_0_q8_0(K, w, xq + xrow * m);
            else tmp = orc_vec_dot_f32(K, (const float *)w, x + (size_t)m * K);
            if (out_f16) {
                ((uint16_t *)yout)[(size_t)m * N + n] = orc_f32_to_f16(bias ? tmp + bias[n] : tmp);
            } else {
                float *y = (float *)yout;
                y[(size_t)m * N + n] = tmp;
                if (bias) y[(size_t)m * N + n] += bias[n];
            }
        }
    }
    free(xq);
}

/* A8: CPUEmbedding (op/CPUEmbedding.cpp:38-80): row gather, Q4_0 rows dequantised; ids travel as floats */
void orc_embedding(const float *ids, int S, const void *W, int wdtype, int hidden, float *out) {
    for (int s = 0; s < S; ++s) {
        int id = (int)ids[s];
        if (wdtype == 2) orc_dequantize_row_q4_0((const uint8_t *)W + (size_t)id * (hidden / QK4_0) * sizeof(block_q4_0), out + (size_t)s * hidden, hidden);
        else if (wdtype == 12) orc_dequantize_row_q4_K((const uint8_t *)W + (size_t)id * (hidden / QK_K) * sizeof(block_q4_K), out + (size_t)s * hidden, hidden);
        else memcpy(out + (size_t)s * hidden, (const float *)W + (size_t)id * hidden, hidden * 4);
    }
}

/* A9: CPURMSNorm::execute (op/CPURMSNorm.cpp:31-136): double sum of squares, rms = 1/sqrtf(mean+eps),
 * y = (x*rms) then * w (two separate roundings: vec_scale_f32 then vec_mul_fp32) */
void orc_rmsnorm(const float *x, const float *w, float *y, int M, int dim, float eps, int add_unit_offset) {
    for (int m = 0; m < M; ++m) {
        const float *xr = x + (size_t)m * dim;
        float *yr = y + (size_t)m * dim;
        double ss = 0.0;
        for (int d = 0; d < dim; ++d) ss += (double)xr[d] * xr[d];
        const float mean = ss / dim;
        const float rms = 1.0f / sqrtf(mean + eps);
        for (int d = 0; d < dim; ++d) {
            float v = xr[d] * rms;
            yr[d] = v * (add_unit_offset ? 1.0f + w[d] : w[d]);
        }
    }
}

/* A18: CPULayerNorm::execute (op/CPULayerNorm.cpp:49-88): fp32 sequential mean, sum (x-mean)^2, rms = sqrt(var+eps),
 * y = w*(x-mean)/rms + b */
void orc_layernorm(const float *x, const float *w, const float *b, float *y, int M, int dim, float eps) {
    for (int m = 0; m < M; ++m) {
        const float *xr = x + (size_t)m * dim;
        float *yr = y + (size_t)m * dim;
        float sum = 0.0F, ssq = 0.0F;
        for (int d = 0; d < dim; ++d) sum += xr[d];
        float mean = sum / dim;
        for (int d = 0; d < dim; ++d) { float c = xr[d] - mean; ssq = fmaf(c, c, ssq); yr[d] = c; }   /* contracted by the reference build */
        float rms = sqrtf(ssq / dim + eps);
        for (int d = 0; d < dim; ++d) yr[d] = b ? w[d] * yr[d] / rms + b[d] : w[d] * yr[d] / rms;
    }
}

/* A14: SiLU via the AVX2 polynomial expf (compute/ActivationFunction.hpp:96-134, mllm_v_silu :137-146), one lane */
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
float orc_v_expf(float x) {
    const float r = 0x1.8p23f;
    const float z = fmaf(x, 0x1.715476p+0f, r);
    const float n = z - r;
    const float b = fmaf(-n, 0x1.7f7d1cp-20f, fmaf(-n, 0x1.62e4p-1f, x));
    const uint32_t e = f2u(z) << 23;
    const float k = u2f(e + f2u(1.0f));
    const int c = fabsf(n) > 126.0f;
    const float u = b * b;
    const float j = fmaf(fmaf(fmaf(0x1.0e4020p-7f, b, 0x1.573e2ep-5f), u, fmaf(0x1.555e66p-3f, b, 0x1.fffdb6p-2f)), u,
                         0x1.ffffecp-1f * b);
    if (!c) return fmaf(j, k, k);
    const uint32_t g = (n <= 0.0f) ? 0x82000000u : 0;
    const float s1 = u2f(g + 0x7f000000u);
    const float s2 = u2f(e - g);
    if (fabsf(n) > 192.0f) return s1 * s1;
    return fmaf(s2, j, s2) * s1;
}
void orc_silu(const float *x, float *y, int n) {
    int i = 0;
    for (; i + 7 < n; i += 8)
        for (int l = 0; l < 8; ++l) { float v = x[i + l]; y[i + l] = v / (1.0f + orc_v_expf(0.0f - v)); }
    for (; i < n; ++i) y[i] = x[i] / (1.0f + expf(-x[i]));     /* scalar tail: mllm_silu_f32, Quantize.hpp:108 */
}

/* A18: GELU / QuickGELU through fp16 lookup tables (third_party/ggml/Quantize.hpp:74-131) */
static uint16_t tab_gelu[1 << 16], tab_qgelu[1 << 16];
static int tab_init = 0;
static void init_tabs(void) {
    if (tab_init) return;
    for (int i = 0; i < (1 << 16); ++i) {
        float f = orc_f16_to_f32((uint16_t)i);
        tab_gelu[i] = orc_f32_to_f16(0.5f * f * (1.0f + tanhf(0.79788456080286535587989211986876f * f * (1.0f + 0.044715f * f * f))));
        tab_qgelu[i] = orc_f32_to_f16(f * (1.0f / (1.0f + expf(-1.702f * f))));
    }
    tab_init = 1;
}
void orc_gelu(const float *x, float *y, int n) { init_tabs(); for (int i = 0; i < n; ++i) y[i] = orc_f16_to_f32(tab_gelu[orc_f32_to_f16(x[i])]); }
void orc_quickgelu(const float *x, float *y, int n) { init_tabs(); for (int i = 0; i < n; ++i) y[i] = orc_f16_to_f32(tab_qgelu[orc_f32_to_f16(x[i])]); }
void orc_gelu_tables(uint16_t *gelu, uint16_t *qgelu) { init_tabs(); memcpy(gelu, tab_gelu, sizeof(tab_gelu)); memcpy(qgelu, tab_qgelu, sizeof(tab_qgelu)); }

/* A15: CPUSoftMax::execute (op/CPUSoftMax.cpp:28-65) -> mllm_vec_soft_max_f32 (ActivationFunction.cpp:29-80):
 * row max, y = v_expf(x - max) (vector lanes) / expf tail, sum of 8-lane hsums, scale by 1/sum.
 * `valid` = number of un-masked leading columns (causal truncation folds the mask in). */
void orc_softmax_row(const float *x, float *y, int n, int valid) {
    float mx = -INFINITY;
    for (int i = 0; i < valid; ++i) mx = x[i] > mx ? x[i] : mx;
    float sum = 0;
    int i = 0;
    for (; i + 7 < valid; i += 8) {
        float v[8];
        for (int l = 0; l < 8; ++l) { v[l] = orc_v_expf(x[i + l] - mx); y[i + l] = v[l]; }
        float t0 = v[4] + v[0], t1 = v[5] + v[1], t2 = v[6] + v[2], t3 = v[7] + v[3];
        float s0 = t0 + t2, s1 = t1 + t3;
        sum += s0 + s1;
    }
    for (; i < valid; ++i) { float v = expf(x[i] - mx); y[i] = v; sum += v; }
    const float inv = 1.0f / sum;
    for (i = 0; i < valid; ++i) y[i] *= inv;
    for (i = valid; i < n; ++i) y[i] = 0;
}

/* ------------------------------------------------------------------------------------------------------------
 * A10/A11/A19: rotary tables (host-side libm sinf/cosf/pow exactly like the reference) and the HF half-split rotate
 * ---------------------------------------------------------------------------------------------------------- */
/* CPURoPE.cpp:22-31,100-128 (HF): theta_i = 1/pow(base, 2i/dim) (double -> float), table[s][i] = sinf/cosf(s*theta_i),
 * duplicated in both halves. sin/cos are [n_pos][dim]. */
void orc_rope_table_hf(float base, int dim, int n_pos, float *sin_t, float *cos_t) {
    int half = dim / 2;
    for (int i = 0; i < half; ++i) {
        float theta = (float)(1.0 / pow((double)base, 2.0 * i / dim));
        for (int s = 0; s < n_pos; ++s) {
            float v = (float)s * theta;
            sin_t[(size_t)s * dim + i] = sin_t[(size_t)s * dim + i + half] = sinf(v);
            cos_t[(size_t)s * dim + i] = cos_t[(size_t)s * dim + i + half] = cosf(v);
        }
    }
}
/* The same table with llama3 frequency scaling, _compute_llama3_theta (CPURoPE.cpp:33-71): theta_i as above, wavelen = 2 pi / theta_i (double division, stored as float);
 * wavelen > original/low: theta /= factor; original/high <= wavelen <= original/low: theta = (1 - s) * (theta / factor) + s * theta with s = (original / wavelen - low) /
 * (high - low), all in float -- the reference binary (g++ -O2 -mfma, default contraction) evaluates the blend as fma(1 - s, theta / factor, s * theta); shorter wavelengths keep
 * theta.  Pinned by tests/golden/rope3.npz. */
void orc_rope_table_hf_llama3(float base, int dim, int n_pos, float factor, float low_freq_factor, float high_freq_factor, float original_max_pos, float *sin_t, float *cos_t) {
    int half = dim / 2;
    float low_freq_wavelen = original_max_pos / low_freq_factor, high_freq_wavelen = original_max_pos / high_freq_factor;
    for (int i = 0; i < half; ++i) {
        float theta = (float)(1.0 / pow((double)base, 2.0 * i / dim));
        float wavelen = (float)(2 * M_PI / theta);
        if (wavelen > low_freq_wavelen) {
            theta /= factor;
        } else if (wavelen >= high_freq_wavelen && wavelen <= low_freq_wavelen) {
            float smooth = (original_max_pos / wavelen - low_freq_factor) / (high_freq_factor - low_freq_factor);
            theta = fmaf(1 - smooth, theta / factor, smooth * theta);
        }
        for (int s = 0; s < n_pos; ++s) {
            float v = (float)s * theta;
            sin_t[(size_t)s * dim + i] = sin_t[(size_t)s * dim + i + half] = sinf(v);
            cos_t[(size_t)s * dim + i] = cos_t[(size_t)s * dim + i + half] = cosf(v);
        }
    }
}
/* NTKRoPE of the MiniCPM3 / Phi-3 family (CPUNTKRoPE.cpp:27-80, get_sin_cos_emb_hf): the table is built for n_pos = max_position_embeddings positions;
 * the per-frequency divisors are the long factors when n_pos > original_max_pos, else the short ones; inv_freq[i] = 1 / powf(theta, i / dim) (the reference divides by
 * dim, not dim / 2 -- kept); angle = (s * (1 / ext[i])) * inv_freq[i] in float; sin / cos scaled by (float)sqrt(1 + logf(n_pos / orig) / log((double)orig)).
 * Pinned by tests/golden/n4_ops.npz (ntk_*). */
void orc_rope_table_ntk(float theta, int dim, int n_pos, int original_max_pos, const float *long_factor, const float *short_factor, float *sin_t, float *cos_t) {
    int half = dim / 2;
    float scale = (float)n_pos / (float)original_max_pos;
    float scaling = (float)sqrt(1 + logf(scale) / log((double)original_max_pos));
    const float *ext = n_pos > original_max_pos ? long_factor : short_factor;
    for (int i = 0; i < half; ++i) {
        float inv = 1.f / powf(theta, (float)i / (float)dim);
        float rcp = 1.0f / ext[i];
        for (int s = 0; s < n_pos; ++s) {
            float f = ((float)s * rcp) * inv;
            sin_t[(size_t)s * dim + i] = sin_t[(size_t)s * dim + i + half] = sinf(f) * scaling;
            cos_t[(size_t)s * dim + i] = cos_t[(size_t)s * dim + i + half] = cosf(f) * scaling;
        }
    }
}
/* CPUMultimodalRoPE.cpp:26-36 (theta), :84-118 (per-axis tables), :37-82 (mrope_section stitch). pos is [3][S] (t,h,w rows);
 * sin/cos out are [S][dim/2]: column c takes axis j where c falls in section j. */
void orc_mrope_table(float base, int dim, const float *pos, int S, const int *section, int n_section, float *sin_t, float *cos_t) {
    int half = dim / 2, c0 = 0;
    for (int j = 0; j < n_section; ++j) {
        int axis = j % 3;
        for (int c = c0; c < c0 + section[j]; ++c) {
            float theta = (float)(1.0 / pow((double)base, 2.0 * c / dim));
            for (int s = 0; s < S; ++s) {
                float v = theta * pos[(size_t)axis * S + s];
                sin_t[(size_t)s * half + c] = sinf(v);
                cos_t[(size_t)s * half + c] = cosf(v);
            }
        }
        c0 += section[j];
    }
}
/* CPUVisionRoPE.cpp:19-28 (inv_freq, float pow), :56-103 (merge-block ordered (h,w) ids), :29-55 + :131-147 (gather):
 * angle[p][0:q] = h(p)*inv_freq, angle[p][q:2q] = w(p)*inv_freq with q = rot_dim/2; out is [t*h*w][rot_dim] of ANGLES */
void orc_vision_rope_angles(int t, int h, int w, int merge, int rot_dim, float *angle) {
    int q = rot_dim / 2;
    float *inv = (float *)malloc(q * sizeof(float));
    for (int i = 0; i < q; ++i) inv[i] = 1.0f / powf(10000.0f, (2.0f * i) / (float)rot_dim);
    int nhb = h / merge, nwb = w / merge, p = 0;
    for (int ti = 0; ti < t; ++ti)
        for (int bh = 0; bh < nhb; ++bh)
            for (int bw = 0; bw < nwb; ++bw)
                for (int jh = 0; jh < merge; ++jh)
                    for (int jw = 0; jw < merge; ++jw, ++p) {
                        int ph = bh * merge + jh, pw = bw * merge + jw;
                        for (int i = 0; i < q; ++i) { angle[(size_t)p * rot_dim + i] = (float)ph * inv[i]; angle[(size_t)p * rot_dim + q + i] = (float)pw * inv[i]; }
                    }
    free(inv);
}
#ifndef ROTV
#define ROTV 1
#endif
#if ROTV == 0
#define ROT1(a,b,s,c) ((a)*(c)-(b)*(s))
#define ROT2(a,b,s,c) ((a)*(s)+(b)*(c))
#elif ROTV == 1
#define ROT1(a,b,s,c) fmaf((a),(c),-((b)*(s)))
#define ROT2(a,b,s,c) fmaf((a),(s),(b)*(c))
#else
#define ROT1(a,b,s,c) fmaf(-(b),(s),(a)*(c))
#define ROT2(a,b,s,c) fmaf((b),(c),(a)*(s))
#endif
/* rope_hf rotate (CPUMultimodalRoPE.cpp:153-221 / CPURoPE.cpp:261-...): x is [S][H][D] (BSHD); sin/cos [S][ld] use cols < D/2.
 * out fp32 or fp16 (K written straight into the fp16 cache). */
void orc_rope_apply(const float *x, int S, int H, int D, const float *sin_t, const float *cos_t, int ld, void *out, int out_f16) {
    int half = D / 2;
    for (int s = 0; s < S; ++s)
        for (int h = 0; h < H; ++h)
            for (int d = 0; d < half; ++d) {
                size_t o = ((size_t)s * H + h) * D + d;
                float a = x[o], b = x[o + half];
                float sv = sin_t[(size_t)s * ld + d], cv = cos_t[(size_t)s * ld + d];
                float v1 = ROT1(a, b, sv, cv), v2 = ROT2(a, b, sv, cv);
                if (out_f16) { ((uint16_t *)out)[o] = orc_f32_to_f16(v1); ((uint16_t *)out)[o + half] = orc_f32_to_f16(v2); }
                else { ((float *)out)[o] = v1; ((float *)out)[o + half] = v2; }
            }
}
/* CPUVisionRoPEFunc.hpp:21-60: same rotate with sin/cos evaluated from the angle table on the fly */
void orc_vision_rope_apply(const float *x, int S, int H, int D, const float *angle, float *out) {
    int half = D / 2;
    for (int s = 0; s < S; ++s)
        for (int h = 0; h < H; ++h)
            for (int d = 0; d < half; ++d) {
                size_t o = ((size_t)s * H + h) * D + d;
                float a = x[o], b = x[o + half];
                float sv = sinf(angle[(size_t)s * half + d]), cv = cosf(angle[(size_t)s * half + d]);
                out[o] = ROT1(a, b, sv, cv);
                out[o + half] = ROT2(a, b, sv, cv);
            }
}

/* expf as the reference's attention calls it (FlashAttention2.hpp:451,457): glibc 2.35 libm, x86-64 FMA variant (the ifunc
 * picks it on every AVX2+FMA host; the HIP kernels carry the same restatement, so it is checked here against the libm of
 * the machine the goldens were made on).  Published algorithm (sysdeps/ieee754/flt-32/e_expf.c, N = 32 table, cubic in
 * double), with the contractions of the -mfma build: kd = fma(InvLn2N, x, Shift); r = fma(InvLn2N, x, -(kd - Shift)). */
static const uint64_t ORC_EXP2F_T[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
float orc_expf(float x) {
    uint32_t ix;
    memcpy(&ix, &x, 4);
    const uint32_t abstop = (ix >> 20) & 0x7ff;
    if (abstop >= 0x42b) { /* |x| >= 88 or nan */
        if (ix == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8) return x + x;
        if (x > 0x1.62e42ep6f) return INFINITY;
        if (x < -0x1.9fe368p6f) return 0.0f;
        if (x < -0x1.9d1d9ep6f) return 0x1p-149f; /* __math_may_uflowf: 0x1.4p-75f squared, rounded */
    }
    const double xd = (double)x, InvLn2N = 0x1.71547652b82fep+5, Shift = 0x1.8p+52;
    double kd = fma(InvLn2N, xd, Shift);
    uint64_t ki;
    memcpy(&ki, &kd, 8);
    kd = kd - Shift;
    const double r = fma(InvLn2N, xd, -kd);
    uint64_t t = ORC_EXP2F_T[ki & 31] + (ki << 47);
    double sc;
    memcpy(&sc, &t, 8);
    const double z = fma(0x1.c6af84b912394p-20, r, 0x1.ebfce50fac4f3p-13);
    const double r2 = r * r;
    double y = fma(r, 0x1.62e42ff0c52d6p-6, 1.0);
    y = fma(z, r2, y);
    return (float)(y * sc);
}
/* number of inputs (out of n, spread over [lo, hi] plus every float within +-span ulps of the break points) whose
 * orc_expf differs from this machine's libm expf */
long orc_expf_mismatches(float lo, float hi, long n, uint64_t seed) {
    long bad = 0;
    uint64_t st = seed * 6364136223846793005ull + 1442695040888963407ull;
#pragma omp parallel for reduction(+ : bad)
    for (long i = 0; i < n; ++i) {
        uint64_t z = (st + (uint64_t)i * 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        const float u = (float)((double)(z >> 11) * (1.0 / 9007199254740992.0));
        const float x = lo + (hi - lo) * u;
        const float a = orc_expf(x), b = expf(x);
        if (memcmp(&a, &b, 4) != 0) bad++;
    }
    return bad;
}

/* A12/A13: F_FA2 (CPUFlashAttention2Func.hpp:52-125 -> compute/FlashAttention2.hpp), restated in the reference's own
 * evaluation order (x86 AVX2 build) so that results are bit-identical:
 *   - tiles Br = Bc = 4 when Sq >= 4, else Br = Bc = 1 (CPUFlashAttention2Func.hpp:71-72); Sq == 1 takes __fa2_decode
 *     (:225-274 / :1346-1394), which is the same recurrence with one key per tile;
 *   - mma0 (:314-357, :1432-1470): per (row, key) 8 fp32 lanes, lane l accumulating d = 8i + l by fma, then
 *     _mm256_hadd_ps (:39-46) = ((l0+l4)+(l1+l5)) + ((l2+l6)+(l3+l7));
 *   - causal (:351-357): tiles wholly right of the diagonal are skipped by every stage; entries j > i of the tile whose
 *     row end meets its column end are set to lowest(); other tiles are NOT masked (the reference's behaviour when
 *     Sk - Sq is not tile aligned -- kept);
 *   - softmax (:430-465, :1556-1599): m' = max(m, row); c = expf((m - m') * scale); p = expf((s - m') * scale);
 *     sum = ((p0 + p1) + p2) + p3; logsum = fmaf(logsum, c, sum) (GCC contracts it: vfmadd132ss in the built reference);
 *   - rescale (:577-595) acc_o *= c, mma1 (:614-636) acc_o = fmaf(p_j, v_j, acc_o) for j in tile order;
 *   - scale_and_store (:682-718): acc_o * (1.0f / logsum);
 *   - leftover columns: the fp32 class uses Sk % Bc (:152), the fp16-KV class Sk % Tc (:1277, Tc = Sk / Bc) -- both kept.
 * expf is the libm one the reference calls. */
static inline float orc_hadd8(const float l[8]) {
    return ((l[0] + l[4]) + (l[1] + l[5])) + ((l[2] + l[6]) + (l[3] + l[7]));
}
static inline float orc_kv_at(const void *P, int f16, size_t i) {
    return f16 ? orc_f16_to_f32(((const uint16_t *)P)[i]) : ((const float *)P)[i];
}
#define ORC_NEG_INF (-FLT_MAX)
/* one (row tile, column tile) step; nr x nc live entries, Bt = tile pitch of acc_s */
static void orc_fa2_tile(const float *Q, const void *K, const void *V, int kv_f16, size_t ldq, size_t ldk, int D, int r0, int nr, int c0, int nc,
                         int Bt, int delta, int causal, float scale, float *acc_s, float *acc_o, float *mx, float *logsum) {
    if (causal && (c0 - delta > (r0 + nr - 1))) return;
    for (int r = 0; r < nr; ++r)
        for (int c = 0; c < nc; ++c) {
            float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const float *q = Q + (size_t)(r0 + r) * ldq;
            size_t ko = (size_t)(c0 + c) * ldk;
            for (int i = 0; i < D; i += 8)
                for (int t = 0; t < 8; ++t) l[t] = fmaf(q[i + t], orc_kv_at(K, kv_f16, ko + i + t), l[t]);
            acc_s[r * Bt + c] = orc_hadd8(l);
        }
    if (causal && (r0 + nr == (c0 + nc) - delta))
        for (int r = 0; r < nr; ++r)
            for (int c = 0; c < nc; ++c)
                if (c > r) acc_s[r * Bt + c] = ORC_NEG_INF;
    for (int r = 0; r < nr; ++r) {
        float prev = mx[r], m = prev;
        for (int c = 0; c < nc; ++c) m = fmaxf(m, acc_s[r * Bt + c]);
        mx[r] = m;
        float cs = expf((prev - m) * scale);
        float sum = 0.0f;
        for (int c = 0; c < nc; ++c) {
            float v = expf((acc_s[r * Bt + c] - m) * scale);
            acc_s[r * Bt + c] = v;
            sum += v;
        }
        logsum[r] = fmaf(logsum[r], cs, sum);
        float *o = acc_o + (size_t)r * D;
        for (int d = 0; d < D; ++d) o[d] = o[d] * cs;
        for (int d = 0; d < D; ++d) {
            float a = o[d];
            for (int c = 0; c < nc; ++c) a = fmaf(acc_s[r * Bt + c], orc_kv_at(V, kv_f16, (size_t)(c0 + c) * ldk + d), a);
            o[d] = a;
        }
    }
}
void orc_attention(const float *Q, const void *K, const void *V, int kv_f16, float *O, int Sq, int Sk, int Hq, int Hkv, int D, int causal) {
    const float scale = 1.0f / sqrtf((float)D);
    const int grp = Hq / Hkv, delta = Sk - Sq;
    const int Bt = Sq >= 4 ? 4 : 1;
    const size_t ldq = (size_t)Hq * D, ldk = (size_t)Hkv * D;
    const int Tr = Sq / Bt, Tr_left = Sq % Bt, Tc = Sk / Bt;
    int Tc_left;
    if (Sq == 1) Tc_left = Sk % Bt;                       /* __fa2_decode, both classes (:232, :1353) */
    else Tc_left = kv_f16 ? (Tc ? Sk % Tc : 0) : Sk % Bt;  /* __fa2_prefill_append (:152 vs :1277) */
#pragma omp parallel for schedule(dynamic, 1)
    for (int h = 0; h < Hq; ++h) {
        const int kvh = h / grp;
        const float *Qh = Q + (size_t)h * D;
        const void *Kh = kv_f16 ? (const void *)((const uint16_t *)K + (size_t)kvh * D) : (const void *)((const float *)K + (size_t)kvh * D);
        const void *Vh = kv_f16 ? (const void *)((const uint16_t *)V + (size_t)kvh * D) : (const void *)((const float *)V + (size_t)kvh * D);
        float acc_s[16], mx[4], logsum[4];
        float *acc_o = (float *)malloc(sizeof(float) * 4 * D);
        for (int tr = 0; tr < Tr + (Tr_left ? 1 : 0); ++tr) {
            const int r0 = tr * Bt, nr = tr < Tr ? Bt : Tr_left;
            for (int r = 0; r < 4; ++r) { mx[r] = ORC_NEG_INF; logsum[r] = 0.0f; }
            for (int i = 0; i < 4 * D; ++i) acc_o[i] = 0.0f;
            for (int tc = 0; tc < Tc; ++tc) orc_fa2_tile(Qh, Kh, Vh, kv_f16, ldq, ldk, D, r0, nr, tc * Bt, Bt, Bt, delta, causal, scale, acc_s, acc_o, mx, logsum);
            if (Tc_left) orc_fa2_tile(Qh, Kh, Vh, kv_f16, ldq, ldk, D, r0, nr, Tc * Bt, Tc_left, Bt, delta, causal, scale, acc_s, acc_o, mx, logsum);
            for (int r = 0; r < nr; ++r) {
                const float rl = 1.0f / logsum[r];
                float *o = O + (size_t)(r0 + r) * ldq + (size_t)h * D;
                for (int d = 0; d < D; ++d) o[d] = acc_o[(size_t)r * D + d] * rl;
            }
        }
        free(acc_o);
    }
}

/* A16/A17: patch-embed convolution with kernel == stride, VALID: one output pixel = vec_dot_fp32 over the flattened
 * receptive field (compute/Convolution.cpp:35-82 conv2d, :179-235 conv3d), + bias.
 * conv3d (Qwen2-VL): patches [N][C*T*H*W] already in (c,t,h,w) order, W [OC][C*T*H*W]  -> out [N][OC]. */
void orc_patch_gemm(const float *patches, int N, int KK, const float *W, int OC, const float *bias, float *out) {
#pragma omp parallel for collapse(2)
    for (int n = 0; n < N; ++n)
        for (int oc = 0; oc < OC; ++oc) {
            float v = orc_vec_dot_f32(KK, W + (size_t)oc * KK, patches + (size_t)n * KK);
            if (bias) v += bias[oc];
            out[(size_t)n * OC + oc] = v;
        }
}
/* conv2d (ViT/CLIP): image tensor [B, H, C, W] in mllm terms (head = H, sequence = C: models/vit/processing_vit.hpp:18-26),
 * given here in logical (h, c, w) order; weight file layout [OC][C][kh][kw]. The reference flattens kernel and receptive
 * field as [c][kh][kw] (Convolution.cpp:8-33 and :45-60), so the weight needs no relayout; out is logical [H/p][OC][W/p]. */
void orc_conv2d_patch(const float *img, int H, int C, int Wd, const float *Wt, int OC, int p, const float *bias, float *out) {
    int oh = H / p, ow = Wd / p, KK = p * C * p;
#pragma omp parallel for collapse(2)
    for (int y = 0; y < oh; ++y)
        for (int x = 0; x < ow; ++x) {
            float *rf = (float *)malloc(KK * 4);
            for (int c = 0; c < C; ++c)
                for (int kh = 0; kh < p; ++kh)
                    for (int kw = 0; kw < p; ++kw) rf[(c * p + kh) * p + kw] = img[((size_t)(y * p + kh) * C + c) * Wd + x * p + kw];
            for (int oc = 0; oc < OC; ++oc) {
                float v = orc_vec_dot_f32(KK, Wt + (size_t)oc * KK, rf);
                if (bias) v += bias[oc];
                out[((size_t)y * OC + oc) * ow + x] = v;
            }
            free(rf);
        }
}

/* A20: elementwise (CPUBinaryFunc.hpp F_TTADD / F_TTMUL) */
void orc_add(const float *a, const float *b, float *y, int n) { for (int i = 0; i < n; ++i) y[i] = a[i] + b[i]; }
void orc_mul(const float *a, const float *b, float *y, int n) { for (int i = 0; i < n; ++i) y[i] = a[i] * b[i]; }

/* ------------------------------------------------------------------------------------------------------------
 * A7, eager-attention form of F_MM: gemm_fp32 / gemm_fp32_fp16 (compute/GemmFp.hpp:104-150, :233-283, x86 path) as CPUmmFunction::execute calls it per (batch, head) on
 * BHSD operands (op/CPUMatmulFunc.hpp:155-172; C is zeroed first, :158).  A [M][K], B [K][N] (fp32, or fp16 when b16 != 0), C [M][N].
 * The K axis goes in blocks of 256.  An 8 x 8 tile that is full (row block and column block both complete) runs the AVX micro-kernel: C is loaded, one fmadd per k, stored --
 * over the K blocks that is ONE fma chain over all of K starting from 0.  Elements of edge tiles take the scalar path: per K block `sum = 0; sum += a * b` (contracted to an
 * fma by the reference's build, -mfma with GCC's default -ffp-contract=fast) and `C += sum`.
 * ---------------------------------------------------------------------------------------------------------- */
void orc_gemm_fp32_bhsd(const float *a, const void *b, int b16, float *c, int heads, int M, int N, int K) {
    const int mfull = M - M % 8, nfull = N - N % 8;
    for (int h = 0; h < heads; ++h) {
        const float *A = a + (size_t)h * M * K;
        const float *B32 = (const float *)b + (size_t)h * K * N;
        const uint16_t *B16 = (const uint16_t *)b + (size_t)h * K * N;
        float *C = c + (size_t)h * M * N;
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < N; ++j) {
                float acc = 0.0f;
                if (i < mfull && j < nfull) {
                    for (int k = 0; k < K; ++k) acc = fmaf(A[(size_t)i * K + k], b16 ? orc_f16_to_f32(B16[(size_t)k * N + j]) : B32[(size_t)k * N + j], acc);
                } else {
                    for (int k0 = 0; k0 < K; k0 += 256) {
                        float sum = 0.0f;
                        const int k1 = k0 + 256 < K ? k0 + 256 : K;
                        for (int k = k0; k < k1; ++k) sum = fmaf(A[(size_t)i * K + k], b16 ? orc_f16_to_f32(B16[(size_t)k * N + j]) : B32[(size_t)k * N + j], sum);
                        acc = acc + sum;
                    }
                }
                C[(size_t)i * N + j] = acc;
            }
    }
}
