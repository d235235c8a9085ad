// oracle/ref_drivers/ref_hip_window.cpp -- TEST INFRASTRUCTURE (oracle side), not product.
//
// Unit driver of the adapter's lazy window (integration/hip/HIPBackend.cpp: lazy / flush_lazy / emit_group): hands the backend hand-made runs of LazyOps -- the kinds, pointer
// relations and ALIASING patterns the reference's frontend produces when it re-uses released blocks -- and prints, per scenario, how many C-ABI calls reached the worker and how
// many of them were fused launches standing for how many Ops.  Built by oracle/Makefile.ref as oracle/_ref/mock_hip_window against the null device (tests/test_integration_build.py
// checks the counts in the container, under AddressSanitizer: the mock reads / writes every extent a launch names).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "backends/cpu/CPUBackend.hpp"
#include "Module.hpp"
#include "HIPBackend.hpp"

using namespace mllm;
using LO = HIPBackend::LazyOp;

static std::vector<void *> g_bufs;
static float *fbuf(size_t n) { void *p = calloc(n, 4); g_bufs.push_back(p); return (float *)p; }
static void *raw(size_t n) { void *p = calloc(n, 1); g_bufs.push_back(p); return p; }

static LO norm(const float *a, float *out, const float *w, int dim) { LO o; o.kind = LO::NORM; o.a = a; o.out = out; o.w = w; o.n = dim; o.eps = 1e-6f; return o; }
static LO lin(const float *x, float *y, const void *W, int K, int N, void *ws) { LO o; o.kind = LO::LINEAR; o.a = x; o.out = y; o.W = W; o.K = K; o.n = N; o.ws = ws; return o; }
static LO un(LO::Kind k, const float *a, float *out, int n) { LO o; o.kind = k; o.a = a; o.out = out; o.n = n; return o; }
static LO bin(LO::Kind k, const float *a, const float *b, float *out, int n) { LO o; o.kind = k; o.a = a; o.b = b; o.out = out; o.n = n; return o; }
static LO rope(const float *a, float *out, const float *s, const float *c, int H, int D) { LO o; o.kind = LO::ROPE; o.a = a; o.out = out; o.sin = s; o.cos = c; o.ld_tab = D / 2; o.S = 1; o.H = H; o.D = D; return o; }
static LO kvs(const float *a, uint16_t *dst, int n) { LO o; o.kind = LO::KVSTORE; o.a = a; o.dst16 = dst; o.n = n; o.S = 1; return o; }
static LO fa2(const float *q, const void *k, const void *v, float *out, int Hq, int Hkv, int D, int Sk) {
    LO o; o.kind = LO::FA2; o.a = q; o.kp = k; o.vp = v; o.out = out; o.H = Hq; o.Hkv = Hkv; o.D = D; o.Sk = Sk; o.causal = 1; o.kvdt = MLLM_HIP_F16; o.S = 1; return o;
}

int main() {
    Module::initBackend(MLLM_CPU);
    HIPBackend *hb = installHIPBackend(0);
    const int K = 512, N = 512, I = 1024, nb = K / 256;
    float *w = fbuf(K);
    void *Wq = raw((size_t)N * nb * 144), *Wk = raw((size_t)128 * nb * 144), *Wv = raw((size_t)128 * nb * 144), *Wg = raw((size_t)I * nb * 144), *Wu = raw((size_t)I * nb * 144);
    void *Wd = raw((size_t)N * (I / 256) * 144), *Wbig = raw((size_t)64 * 44 * 144);
    void *ws = raw(1 << 20);
    bool first = true;
    printf("[");
    auto scenario = [&](const char *name, const std::vector<LO> &ops) {
        hb->drain();
        const long c0 = hb->deferred_calls(), f0 = hb->fused_launches(), o0 = hb->fused_ops();
        for (const LO &o : ops) hb->lazy(o);
        hb->drain();
        printf("%s{\"name\": \"%s\", \"ops\": %zu, \"calls\": %ld, \"fused_launches\": %ld, \"fused_ops\": %ld}", first ? "" : ", ", name, ops.size(), hb->deferred_calls() - c0,
               hb->fused_launches() - f0, hb->fused_ops() - o0);
        first = false;
    };
    {   // input_layernorm -> q, k, v
        float *x = fbuf(K), *n1 = fbuf(K), *q = fbuf(N), *k = fbuf(128), *v = fbuf(128);
        scenario("NLLL", {norm(x, n1, w, K), lin(n1, q, Wq, K, N, ws), lin(n1, k, Wk, K, 128, ws), lin(n1, v, Wv, K, 128, ws)});
        // the q projection's output is the block the norm reads (cannot happen while x is alive; the rule must still catch it): no fusion
        scenario("NLLL_output_over_input", {norm(x, n1, w, K), lin(n1, x, Wq, K, N, ws), lin(n1, k, Wk, K, 128, ws), lin(n1, v, Wv, K, 128, ws)});
    }
    {   // o_proj + residual; the same with the sum written over the residual operand (in-place add)
        float *a = fbuf(K), *o = fbuf(N), *res = fbuf(N), *sum = fbuf(N);
        scenario("LA", {lin(a, o, Wq, K, N, ws), bin(LO::ADD, o, res, sum, N)});
        scenario("LA_other_order", {lin(a, o, Wq, K, N, ws), bin(LO::ADD, res, o, sum, N)});
        scenario("LA_sum_over_input_row", {lin(a, o, Wq, K, N, ws), bin(LO::ADD, o, res, a, N)});
    }
    {   // the MLP: up's output in the block gate's output had (released after silu)
        float *x = fbuf(K), *n2 = fbuf(K), *g = fbuf(I), *sg = fbuf(I), *u = fbuf(I), *act = fbuf(I);
        scenario("NLSLM", {norm(x, n2, w, K), lin(n2, g, Wg, K, I, ws), un(LO::SILU, g, sg, I), lin(n2, u, Wu, K, I, ws), bin(LO::MUL, sg, u, act, I)});
        scenario("NLSLM_up_reuses_gate_block", {norm(x, n2, w, K), lin(n2, g, Wg, K, I, ws), un(LO::SILU, g, sg, I), lin(n2, g, Wu, K, I, ws), bin(LO::MUL, sg, g, act, I)});
        // not the MLP's shape: the product takes something else than silu's output
        scenario("NLSLM_broken_chain", {norm(x, n2, w, K), lin(n2, g, Wg, K, I, ws), un(LO::SILU, g, sg, I), lin(n2, u, Wu, K, I, ws), bin(LO::MUL, g, u, act, I)});
    }
    {   // a Linear without a fused form (K / 256 = 44) then the add, the norm and q / k / v: the add travels as the prologue -- unless the norm's output took the add's operand's block
        float *big = fbuf(11264), *d = fbuf(K), *tmp = fbuf(K), *xn = fbuf(K), *n1 = fbuf(K), *q = fbuf(N), *k = fbuf(128), *v = fbuf(128);
        scenario("L_unfusable_then_A", {lin(big, d, Wbig, 11264, 64, ws), bin(LO::ADD, d, tmp, xn, 64)});
        scenario("ANLLL", {bin(LO::ADD, d, tmp, xn, K), norm(xn, n1, w, K), lin(n1, q, Wq, K, N, ws), lin(n1, k, Wk, K, 128, ws), lin(n1, v, Wv, K, 128, ws)});
        scenario("ANLLL_norm_output_over_add_operand", {bin(LO::ADD, d, tmp, xn, K), norm(xn, d, w, K), lin(d, q, Wq, K, N, ws), lin(d, k, Wk, K, 128, ws), lin(d, v, Wv, K, 128, ws)});
    }
    {   // the attention block of one position: 4 query heads, 2 / 4 key-value heads, D = 64, 9 keys in the cache
        const int D = 64, T = 9;
        float *s = fbuf(D / 2), *c = fbuf(D / 2);
        for (int Hkv : {2, 4}) {
            const int Hq = 4, qn = Hq * D, kn = Hkv * D;
            float *q = fbuf(qn), *k = fbuf(kn), *v = fbuf(kn), *qo = fbuf(qn), *ko = fbuf(kn), *O = fbuf(qn);
            uint16_t *ks = (uint16_t *)raw((size_t)(T + 1) * kn * 2), *vs = (uint16_t *)raw((size_t)(T + 1) * kn * 2);
            const std::string tag = Hkv == 2 ? "_gqa" : "_mha";
            scenario(("RRKKF" + tag).c_str(), {rope(q, qo, s, c, Hq, D), rope(k, ko, s, c, Hkv, D), kvs(ko, ks + (size_t)T * kn, kn), kvs(v, vs + (size_t)T * kn, kn), fa2(qo, ks, vs, O, Hq, Hkv, D, T + 1)});
            // the attention output in q's released block: every workgroup owns its head's slice on both sides
            scenario(("RRKKF_out_over_q" + tag).c_str(), {rope(q, qo, s, c, Hq, D), rope(k, ko, s, c, Hkv, D), kvs(ko, ks + (size_t)T * kn, kn), kvs(v, vs + (size_t)T * kn, kn), fa2(qo, ks, vs, q, Hq, Hkv, D, T + 1)});
            // the rotated k in (the front of) q's released block: fine with one head per group, a race between workgroups otherwise
            scenario(("RRKKF_krot_over_q" + tag).c_str(), {rope(q, qo, s, c, Hq, D), rope(k, q, s, c, Hkv, D), kvs(q, ks + (size_t)T * kn, kn), kvs(v, vs + (size_t)T * kn, kn), fa2(qo, ks, vs, O, Hq, Hkv, D, T + 1)});
            // a cache append that is not the slab's next row: no fused step (the four Ops in front still are one launch)
            scenario(("RRKKF_wrong_row" + tag).c_str(), {rope(q, qo, s, c, Hq, D), rope(k, ko, s, c, Hkv, D), kvs(ko, ks + (size_t)(T - 1) * kn, kn), kvs(v, vs + (size_t)(T - 1) * kn, kn), fa2(qo, ks, vs, O, Hq, Hkv, D, T + 1)});
        }
    }
    {   // program order: an Op that is not a LazyOp goes behind what the window holds
        float *x = fbuf(K), *n1 = fbuf(K), *y = fbuf(K);
        hb->drain();
        const long c0 = hb->deferred_calls();
        hb->lazy(norm(x, n1, w, K));
        hb->defer("mllm_hip_silu", mllm_hip_silu, (const float *)n1, y, (int64_t)K, hb->stream());
        hb->drain();
        printf(", {\"name\": \"flush_before_other\", \"ops\": 2, \"calls\": %ld, \"fused_launches\": 0, \"fused_ops\": 0}", hb->deferred_calls() - c0);
    }
    printf("]\n");
    hb->drain();
    for (void *p : g_bufs) free(p);
    return 0;
}
