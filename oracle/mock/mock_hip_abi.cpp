// oracle/mock/mock_hip_abi.cpp -- TEST INFRASTRUCTURE, not product: a "null device" behind the device-touching entry points of include/mllm_hip.h.
//
// Purpose: run the reference-side adapter (integration/hip/) under the reference's real Module / Layer / Tensor frontend IN THIS CONTAINER (no GPU), under
// AddressSanitizer, before it is allowed near a GPU box: the risk in the adapter is plumbing (trace passes, views, reference counts, shapes, buffer sizes), and a
// faulting kernel on the pool can reset a node.  Every "device" allocation is host memory, copies are memcpy, and every launcher READS all bytes of its inputs and
// WRITES (zeros) all bytes of its outputs, with the extents the real kernels touch -- so a dangling view, a short scratch buffer or a wrong pitch is an ASan report.
// No arithmetic is mocked: results are zeros and are never compared with anything.  The host-only entry points (rotary tables, LUTs, *_bytes) are not defined
// here and resolve to the real libmllm_hip.so.  Linked only into oracle/_ref/*_mock binaries by oracle/Makefile.ref; the product never sees it.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "mllm_hip.h"

namespace {
volatile uint64_t g_sink;
void rd(const void *p, size_t n) {
#ifdef MOCK_FAST      // host-overhead timing build: the launchers cost nothing, what is left is the frontend's and the adapter's own work per Op
    (void)p; (void)n; return;
#endif
    if (!n) return;
    if (!p) { fprintf(stderr, "mock: NULL input of %zu bytes\n", n); abort(); }
    const unsigned char *b = (const unsigned char *)p;
    uint64_t s = 0;
    for (size_t i = 0; i < n; ++i) s += b[i];
    g_sink = g_sink + s;
}
void wr(void *p, size_t n) {
#ifdef MOCK_FAST
    (void)p; (void)n; return;
#endif
    if (!n) return;
    if (!p) { fprintf(stderr, "mock: NULL output of %zu bytes\n", n); abort(); }
    memset(p, 0, n);
}
// pitched windows: rows of `cols` elements of `es` bytes, row stride ld elements
void rd2(const void *p, int64_t ld, int64_t rows, int64_t cols, size_t es) { for (int64_t r = 0; r < rows; ++r) rd((const char *)p + r * ld * es, cols * es); }
void wr2(void *p, int64_t ld, int64_t rows, int64_t cols, size_t es) { for (int64_t r = 0; r < rows; ++r) wr((char *)p + r * ld * es, cols * es); }
size_t esz(int dt) { return dt == MLLM_HIP_F16 ? 2 : 4; }
}  // namespace

extern "C" {
int mllm_hip_init(int) { return 0; }
int mllm_hip_alloc(void **p, size_t n) { *p = malloc(n ? n : 16); return 0; }
int mllm_hip_free(void *p) { free(p); return 0; }
int mllm_hip_pool_alloc(void **p, size_t n, void *) { *p = malloc(n ? n : 16); return 0; }
int mllm_hip_pool_free(void *p, void *) { free(p); return 0; }
int mllm_hip_stream_create(void **s) { *s = (void *)0x1; return 0; }
int mllm_hip_stream_destroy(void *) { return 0; }
int mllm_hip_h2d(void *d, const void *s, size_t n, void *) { memcpy(d, s, n); return 0; }
int mllm_hip_upload(void *d, const void *s, size_t n, void *) { memcpy(d, s, n); return 0; }
int mllm_hip_host_register(void *, size_t) { return 0; }
int mllm_hip_host_unregister(void *) { return 0; }
int mllm_hip_d2h(void *d, const void *s, size_t n, void *) { memcpy(d, s, n); return 0; }
int mllm_hip_sync(void *) { return 0; }

int mllm_hip_quantize_q80(const float *x, int8_t *qs, uint16_t *d, int M, int K, void *) { rd(x, (size_t)M * K * 4); wr(qs, (size_t)M * K); wr(d, (size_t)M * (K / 32) * 2); return 0; }
int mllm_hip_quantize_q8k_packed(const float *x, void *xpack, int M, int K, void *) { rd(x, (size_t)M * K * 4); wr(xpack, mllm_hip_q4k_prepack_bytes(M, K)); return 0; }
int mllm_hip_q4k_prepack(const void *W, int N, int K, void *out, void *) { rd(W, (size_t)N * (K / 256) * 144); wr(out, mllm_hip_q4k_wpack_bytes(N, K)); return 0; }
int mllm_hip_repack_q40(const void *raw, uint8_t *qs, uint16_t *d, int64_t nb, void *) { rd(raw, (size_t)nb * 18); wr(qs, (size_t)nb * 16); wr(d, (size_t)nb * 2); return 0; }
int mllm_hip_linear(const void *W, int wdt, const float *bias, const float *x, void *y, int ydt, int64_t ldy, int M, int N, int K, void *ws, void *) {
    rd(W, wdt == MLLM_HIP_Q4_K ? (size_t)N * (K / 256) * 144 : wdt == MLLM_HIP_Q4_0 ? (size_t)N * (K / 32) * 18 : (size_t)N * K * 4);
    if (bias) rd(bias, (size_t)N * 4);
    rd(x, (size_t)M * K * 4);
    wr(ws, mllm_hip_linear_workspace_bytes(wdt, M, K));
    wr2(y, ldy, M, N, esz(ydt));
    return 0;
}
int mllm_hip_linear_q4kp_packed(const void *Wp, const float *bias, const void *xpack, void *y, int ydt, int64_t ldy, const float *res, int M, int N, int K, void *) {
    rd(Wp, mllm_hip_q4k_wpack_bytes(N, K)); if (bias) rd(bias, (size_t)N * 4); rd(xpack, mllm_hip_q4k_prepack_bytes(M, K));
    if (res) rd2(res, ldy, M, N, 4);
    wr2(y, ldy, M, N, esz(ydt));
    return 0;
}
int mllm_hip_linear_q40_q80(const uint8_t *Wqs, const uint16_t *Wd, const float *bias, const int8_t *xqs, const uint16_t *xd, float *y, int64_t ldy, int M, int N, int K, void *) {
    rd(Wqs, (size_t)N * K / 2); rd(Wd, (size_t)N * (K / 32) * 2); if (bias) rd(bias, (size_t)N * 4); rd(xqs, (size_t)M * K); rd(xd, (size_t)M * (K / 32) * 2);
    wr2(y, ldy, M, N, 4);
    return 0;
}
int mllm_hip_linear_f32(const float *W, const float *bias, const float *x, float *y, int64_t ldy, int M, int N, int K, void *) {
    rd(W, (size_t)N * K * 4); if (bias) rd(bias, (size_t)N * 4); rd(x, (size_t)M * K * 4); wr2(y, ldy, M, N, 4);
    return 0;
}
int mllm_hip_embedding_q40(const float *ids, const uint8_t *Wqs, const uint16_t *Wd, float *out, int S, int hidden, int vocab, void *) {
    rd(ids, (size_t)S * 4); rd(Wqs, (size_t)vocab * hidden / 2); rd(Wd, (size_t)vocab * (hidden / 32) * 2); wr(out, (size_t)S * hidden * 4);
    return 0;
}
int mllm_hip_rmsnorm(const float *x, const float *w, float *y, int8_t *, float *, int16_t *, int M, int dim, float, int, void *) { rd(x, (size_t)M * dim * 4); rd(w, (size_t)dim * 4); wr(y, (size_t)M * dim * 4); return 0; }
int mllm_hip_layernorm(const float *x, const float *w, const float *b, float *y, int8_t *, float *, int16_t *, int M, int dim, float, void *) {
    rd(x, (size_t)M * dim * 4); rd(w, (size_t)dim * 4); if (b) rd(b, (size_t)dim * 4); wr(y, (size_t)M * dim * 4);
    return 0;
}
int mllm_hip_rmsnorm_packed(const float *x, const float *w, float *y, void *xpack, int M, int dim, float, int, void *) {
    rd(x, (size_t)M * dim * 4); rd(w, (size_t)dim * 4); if (y) wr(y, (size_t)M * dim * 4); wr(xpack, mllm_hip_q4k_prepack_bytes(M, dim));
    return 0;
}
int mllm_hip_layernorm_packed(const float *x, const float *w, const float *b, float *y, void *xpack, int M, int dim, float, void *) {
    rd(x, (size_t)M * dim * 4); rd(w, (size_t)dim * 4); if (b) rd(b, (size_t)dim * 4); if (y) wr(y, (size_t)M * dim * 4); wr(xpack, mllm_hip_q4k_prepack_bytes(M, dim));
    return 0;
}
int mllm_hip_silu(const float *x, float *y, int64_t n, void *) { rd(x, n * 4); wr(y, n * 4); return 0; }
int mllm_hip_silu_rows(const float *x, float *y, int64_t rows, int dim, void *) { rd(x, rows * dim * 4); wr(y, rows * dim * 4); return 0; }
int mllm_hip_act_lut(const float *x, float *y, int64_t n, const uint16_t *lut, void *) { rd(x, n * 4); rd(lut, 65536 * 2); wr(y, n * 4); return 0; }
int mllm_hip_add(const float *a, const float *b, float *y, int64_t n, void *) { rd(a, n * 4); rd(b, n * 4); wr(y, n * 4); return 0; }
int mllm_hip_mul(const float *a, const float *b, float *y, int64_t n, void *) { rd(a, n * 4); rd(b, n * 4); wr(y, n * 4); return 0; }
int mllm_hip_softmax(const float *x, float *y, int rows, int n, const int *valid, void *) { rd(x, (size_t)rows * n * 4); if (valid) rd(valid, (size_t)rows * 4); wr(y, (size_t)rows * n * 4); return 0; }
int mllm_hip_index_put_rows_fidx(float *dst, int n_dst, const float *value, const float *idx, int n_rows, int dim, void *) {
    rd(idx, (size_t)n_rows * 4); rd(value, (size_t)n_rows * dim * 4);
    for (int r = 0; r < n_rows; ++r) { const int d = (int)idx[r]; if (d >= 0 && d < n_dst) wr(dst + (size_t)d * dim, (size_t)dim * 4); }
    return 0;
}
int mllm_hip_copy_2d_f32(const float *src, int64_t lds, float *dst, int64_t ldd, int rows, int cols, void *) { rd2(src, lds, rows, cols, 4); wr2(dst, ldd, rows, cols, 4); return 0; }
int mllm_hip_transpose_f32(const float *x, float *y, int rows, int cols, void *) { rd(x, (size_t)rows * cols * 4); wr(y, (size_t)rows * cols * 4); return 0; }
int mllm_hip_rope_apply(const float *x, int64_t ldx, const float *s, const float *c, int ld_tab, void *out, int odt, int64_t ldo, int S, int H, int D, void *) {
    rd2(x, ldx, S, (int64_t)H * D, 4); rd2(s, ld_tab, S, D / 2, 4); rd2(c, ld_tab, S, D / 2, 4); wr2(out, ldo, S, (int64_t)H * D, esz(odt));
    return 0;
}
// the lazy window's fused launches: every Op output of the run is written, every input read, with the extents of the Ops the launch replaces
int mllm_hip_row_fused_launch(const mllm_hip_row_fused *a, void *) {
    if (!mllm_hip_row_fused_supported(a)) { fprintf(stderr, "mock: row_fused launched with a shape the library refuses\n"); abort(); }
    const size_t K = (size_t)a->K;
    rd(a->xa, K * 4);
    if (a->xb) rd(a->xb, K * 4);
    if (a->sum_out) wr(a->sum_out, K * 4);
    if (a->norm_w) rd(a->norm_w, K * 4);
    if (a->norm_out) wr(a->norm_out, K * 4);
    for (int i = 0; i < a->nseg; ++i) {
        const mllm_hip_row_seg &s = a->seg[i];
        rd(s.W, (size_t)s.N * (K / 256) * 144);
        if (s.bias) rd(s.bias, (size_t)s.N * 4);
        wr(s.y, (size_t)s.N * 4);
        if (s.post_out) { rd(s.post_add, (size_t)s.N * 4); wr(s.post_out, (size_t)s.N * 4); }
    }
    if (a->mode == 1) { if (a->silu_out) wr(a->silu_out, (size_t)a->seg[0].N * 4); wr(a->mul_out, (size_t)a->seg[0].N * 4); }
    return 0;
}
int mllm_hip_rope2_store2(const float *q, const float *sq, const float *cq, int ldq, float *qo, int Hq, const float *k, const float *sk, const float *ck, int ldk, float *ko, uint16_t *k16,
                          const float *v, uint16_t *v16, int Hkv, int S, int D, void *) {
    rd(q, (size_t)S * Hq * D * 4); rd2(sq, ldq, S, D / 2, 4); rd2(cq, ldq, S, D / 2, 4); wr(qo, (size_t)S * Hq * D * 4);
    rd(k, (size_t)S * Hkv * D * 4); rd2(sk, ldk, S, D / 2, 4); rd2(ck, ldk, S, D / 2, 4); wr(ko, (size_t)S * Hkv * D * 4); wr(k16, (size_t)S * Hkv * D * 2);
    rd(v, (size_t)S * Hkv * D * 4); wr(v16, (size_t)S * Hkv * D * 2);
    return 0;
}
int mllm_hip_fa2_decode_step(const float *q, const float *sq, const float *cq, float *qo, const float *k, const float *sk, const float *ck, float *ko, const float *v, uint16_t *ks,
                             uint16_t *vs, int T, float *O, int Hq, int Hkv, int D, void *) {
    if (!mllm_hip_fa2_decode_step_supported(T, Hq, Hkv, D)) { fprintf(stderr, "mock: fa2_decode_step launched with extents the library refuses\n"); abort(); }
    rd(q, (size_t)Hq * D * 4); rd(sq, D / 2 * 4); rd(cq, D / 2 * 4); rd(k, (size_t)Hkv * D * 4); rd(sk, D / 2 * 4); rd(ck, D / 2 * 4); rd(v, (size_t)Hkv * D * 4);
    rd(ks, (size_t)T * Hkv * D * 2); rd(vs, (size_t)T * Hkv * D * 2);
    wr(qo, (size_t)Hq * D * 4); wr(ko, (size_t)Hkv * D * 4); wr(ks + (size_t)T * Hkv * D, (size_t)Hkv * D * 2); wr(vs + (size_t)T * Hkv * D, (size_t)Hkv * D * 2);
    wr(O, (size_t)Hq * D * 4);
    return 0;
}
int mllm_hip_store_f16(const float *x, int64_t ldx, uint16_t *out, int64_t ldo, int S, int n, void *) { rd2(x, ldx, S, n, 4); wr2(out, ldo, S, n, 2); return 0; }
int mllm_hip_fa2(const float *Q, int64_t ldq, const void *K, int64_t ldk, const void *V, int64_t ldv, int kvdt, float *O, int64_t ldo, int Sq, int Sk, int Hq, int Hkv, int D, int, const int *, void *, void *) {
    rd2(Q, ldq, Sq, (int64_t)Hq * D, 4); rd2(K, ldk, Sk, (int64_t)Hkv * D, esz(kvdt)); rd2(V, ldv, Sk, (int64_t)Hkv * D, esz(kvdt)); wr2(O, ldo, Sq, (int64_t)Hq * D, 4);
    return 0;
}
int mllm_hip_patch_gemm_f32(const float *p, const float *W, const float *bias, float *out, int N, int KK, int OC, void *) {
    rd(p, (size_t)N * KK * 4); rd(W, (size_t)OC * KK * 4); if (bias) rd(bias, (size_t)OC * 4); wr(out, (size_t)N * OC * 4);
    return 0;
}
int mllm_hip_im2patch_hcw(const float *img, float *patches, int H, int C, int W, int p, void *) { rd(img, (size_t)H * C * W * 4); wr(patches, (size_t)(H / p) * (W / p) * C * p * p * 4); return 0; }
int mllm_hip_im2patch_chw(const float *img, float *patches, int H, int C, int W, int p, void *) { rd(img, (size_t)H * C * W * 4); wr(patches, (size_t)(H / p) * (W / p) * C * p * p * 4); return 0; }
int mllm_hip_sliding_window_mask(const float *x, float *y, int S, int H, int keys, int, void *) { rd(x, (size_t)S * H * keys * 4); wr(y, (size_t)S * H * keys * 4); return 0; }
int mllm_hip_topk_rows(const float *x, int64_t ldx, float *v, float *i, int rows, int n, int k, void *) { rd2(x, ldx, rows, n, 4); wr(v, (size_t)rows * k * 4); wr(i, (size_t)rows * k * 4); return 0; }
int mllm_hip_gather_rows(const float *src, int64_t lds, int n_src_rows, const float *idx, float *out, int64_t ldo, int R, int D, int, void *) {
    rd(idx, (size_t)R * 4);
    for (int r = 0; r < R; ++r) { const int i = (int)idx[r]; if (i >= 0 && i < n_src_rows) rd(src + (size_t)i * lds, (size_t)D * 4); }
    wr2(out, ldo, R, D, 4);
    return 0;
}
int mllm_hip_scatter_add_rows(float *dst, int64_t, int, const float *src, int64_t lds, const float *idx, int R, int D, void *) { (void)dst; rd2(src, lds, R, D, 4); rd(idx, (size_t)R * 4); return 0; }
}
