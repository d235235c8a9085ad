// mllm_amd/csrc/engine.hip -- host side of the hot path: .mllm loader + the reference's Qwen2-VL model graph on the launchers.
//
// What it mirrors (all host logic, the arithmetic lives in the kernels_*.hip launchers):
//   ParamLoader            mllm/ParamLoader.cpp:157-286 (index parse, mmap), :88-141 (load)          -> MllmFile
//   Qwen2VLModel::Forward  mllm/models/qwen2_vl/modeling_qwen2_vl.hpp:381-404                         -> forward_llm()
//   QWen2Decoder/Attention/MLP  :193-335                                                              -> layer loop
//   Qwen2VisionModel       :21-191 (patch embed, VisionBlock x32, PatchMerger)                        -> forward_vision()
//   get_rope_index / get_position_ids  :413-595                                                       -> rope_index()
//   demo loop + argmax     examples/demo_qwen2_vl.cpp:53-63, processing_qwen2_vl.hpp:284-289,438-452  -> prefill/decode/generate
//   KVCache                backends/cpu/op/CPUKVCache.cpp:10-131,253-275 (fp16 slab, zero-copy append) -> kv slabs + cache_len
// Layout in HBM: one allocation per weight tensor, Q4_K rows in their native 144-B blocks; q/k/v (and gate/up) rows are
// concatenated at load so one GEMV/GEMM serves the three (two) projections; embed_tokens (Q4_0, tied lm_head) is split
// into a nibble plane and an fp16 scale plane; activations live in a handful of reusable fp32 / q8k-plane buffers sized
// for cache_limit tokens; K/V slabs are fp16 [layers][cache_limit][Hkv*D].
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "decode_launch.h"

using namespace mllm_hip;

namespace {

struct Entry { uint64_t off, len; int dtype; };

struct MllmFile {
    int fd = -1;
    uint8_t *base = nullptr;
    size_t size = 0;
    std::map<std::string, Entry> idx;
    bool open(const char *path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        size = st.st_size;
        base = (uint8_t *)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (base == MAP_FAILED) { base = nullptr; return false; }
        int32_t magic;
        memcpy(&magic, base, 4);
        if (magic != 20012) return false;  // _MAGIC_NUMBER, mllm/ParamLoader.hpp:48
        uint64_t ilen;
        memcpy(&ilen, base + 4, 8);
        const uint8_t *p = base + 12, *end = base + 12 + ilen;
        while (p < end) {
            int32_t nl;
            memcpy(&nl, p, 4); p += 4;
            std::string name((const char *)p, nl); p += nl;
            Entry e;
            memcpy(&e.len, p, 8); memcpy(&e.off, p + 8, 8);
            int32_t dt; memcpy(&dt, p + 16, 4); e.dtype = dt;
            p += 20;
            idx[name] = e;
        }
        return true;
    }
    const Entry *find(const std::string &n) const { auto it = idx.find(n); return it == idx.end() ? nullptr : &it->second; }
    ~MllmFile() { if (base) munmap(base, size); if (fd >= 0) ::close(fd); }
};

struct LinearW {          // one (possibly row-concatenated) Linear
    void *w = nullptr;    // Q4_K blocks [N][K/256] or fp32 [N][K]
    void *wp = nullptr;   // the same rows packed for the M >= 16 GEMM (mllm_hip_q4k_prepack): prefill / vision read these
    void *wd = nullptr;   // LLM decode Linears only: the rows in decode order (decode_order_q4k) for the fused decode kernels
    float *bias = nullptr;
    int N = 0, K = 0, dtype = MLLM_HIP_Q4_K;
};

struct Q8Planes { int8_t *qs = nullptr; float *d = nullptr; int16_t *bs = nullptr; };

#define EH(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)
#define HH(expr) MH_CHECK(expr)

}  // namespace

struct mllm_hip_qwen2vl {
    mllm_hip_qwen2vl_config c;
    hipStream_t st = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<void *> allocs;
    int D = 0, HD = 0, KVD = 0, QKV = 0;
    // LLM weights
    struct Layer { float *in_norm, *post_norm; LinearW qkv, o, gu, down; };
    std::vector<Layer> layers;
    float *final_norm = nullptr;
    uint8_t *emb_qs = nullptr; uint16_t *emb_d = nullptr;
    // vision weights
    struct VBlock { float *n1w, *n1b, *n2w, *n2b; LinearW qkv, proj, fc1, fc2; };
    std::vector<VBlock> vblocks;
    float *patch_w = nullptr, *lnq_w = nullptr, *lnq_b = nullptr;
    LinearW m0, m2;
    uint16_t *lut_gelu = nullptr, *lut_qgelu = nullptr;
    bool has_vision = false;
    // activations
    int max_tok = 0;
    float *h0 = nullptr, *h1 = nullptr, *qkv = nullptr, *attn = nullptr, *gu = nullptr, *act = nullptr, *logits = nullptr, *normed = nullptr;
    float *ids_f = nullptr; int *idx_i = nullptr; int *tok_dev = nullptr;
    Q8Planes xq, xq2;            // K = hidden (also v_dim / merger width), K = inter (also v_mlp)
    int8_t *x80_qs = nullptr; uint16_t *x80_d = nullptr;
    float *rope_sin = nullptr, *rope_cos = nullptr;
    uint16_t *kslab = nullptr, *vslab = nullptr;
    int vt_ld = 0;
    void *fa_ws = nullptr;
    void *xpack = nullptr;
    size_t xpack_bytes = 0;
    // vision activations
    int max_patch = 0;
    float *vx = nullptr, *vr = nullptr, *vqkv = nullptr, *vattn = nullptr, *vfc = nullptr, *vact = nullptr, *vpix = nullptr, *vsin = nullptr, *vcos = nullptr,
          *vemb = nullptr, *vm0 = nullptr;
    // fused decode path (kernels_decode.hip): device-side step state, per-step rotary rows, captured graph
    DecodeState *d_state = nullptr;
    float *dec_sin = nullptr, *dec_cos = nullptr, *part_val = nullptr, *cur_sin = nullptr, *cur_cos = nullptr;
    int *part_idx = nullptr, *history = nullptr;
    int nsplit = 0, max_parts = 4096, dec_rows = 0;
    DecodeCtx dctx;
    std::vector<DecodeLayer> dlayers;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    bool use_graph = true;
    // state
    int cache_len = 0;
    float last_pos = -1.0f;
    int64_t decode_weight_bytes = 0;

    template <typename T> int dalloc(T **p, size_t n) {
        void *q = nullptr;
        MH_CHECK(hipMalloc(&q, n ? n : 16));
        allocs.push_back(q);
        *p = (T *)q;
        return 0;
    }
    int upload(void *dst, const void *src, size_t n) { MH_CHECK(hipMemcpy(dst, src, n, hipMemcpyHostToDevice)); return 0; }
};

typedef mllm_hip_qwen2vl M;

static int need(const MllmFile &f, const std::string &n, const Entry **e, int dtype, uint64_t len) {
    *e = f.find(n);
    if (!*e || (*e)->dtype != dtype || (*e)->len != len) {
        fprintf(stderr, "mllm_hip: tensor %s missing or wrong dtype/size in .mllm (want dtype %d len %llu)\n", n.c_str(), dtype, (unsigned long long)len);
        return MLLM_HIP_ERR_IO;
    }
    return 0;
}

static int load_f32(M *m, const MllmFile &f, const std::string &n, size_t count, float **out) {
    const Entry *e;
    EH(need(f, n, &e, MLLM_HIP_F32, count * 4));
    EH(m->dalloc(out, count * 4));
    return m->upload(*out, f.base + e->off, count * 4);
}

// rows of several Q4_K Linear weights (same K) concatenated; biases concatenated (zeros where a part has none)
static int load_linear_q4k(M *m, const MllmFile &f, const std::vector<std::string> &names, const std::vector<int> &Ns, int K, bool bias, LinearW *lw) {
    int N = 0;
    for (int n : Ns) N += n;
    const size_t row = (size_t)K / 256 * 144;
    uint8_t *w;
    EH(m->dalloc(&w, row * N));
    size_t ro = 0;
    for (size_t i = 0; i < names.size(); ++i) {
        const Entry *e;
        EH(need(f, names[i] + ".weight", &e, MLLM_HIP_Q4_K, row * Ns[i]));
        EH(m->upload(w + ro * row, f.base + e->off, row * Ns[i]));
        ro += Ns[i];
    }
    lw->w = w; lw->N = N; lw->K = K; lw->dtype = MLLM_HIP_Q4_K;
    {
        uint8_t *wp;
        EH(m->dalloc(&wp, mllm_hip_q4k_prepack_bytes(N, K)));
        EH(mllm_hip_q4k_prepack(w, N, K, wp, m->st));
        lw->wp = wp;
    }
    if (bias) {
        EH(m->dalloc(&lw->bias, (size_t)N * 4));
        size_t bo = 0;
        for (size_t i = 0; i < names.size(); ++i) {
            const Entry *e;
            EH(need(f, names[i] + ".bias", &e, MLLM_HIP_F32, (uint64_t)Ns[i] * 4));
            EH(m->upload(lw->bias + bo, f.base + e->off, (size_t)Ns[i] * 4));
            bo += Ns[i];
        }
    }
    return 0;
}

static int alloc_q8(M *m, Q8Planes *p, int M_, int K) {
    EH(m->dalloc(&p->qs, (size_t)M_ * K));
    EH(m->dalloc(&p->d, (size_t)M_ * (K / 256) * 4));
    EH(m->dalloc(&p->bs, (size_t)M_ * (K / 16) * 2));
    return 0;
}

extern "C" int mllm_hip_qwen2vl_create(const mllm_hip_qwen2vl_config *cfg, const char *path, mllm_hip_qwen2vl **out) {
    if (!cfg || !path || !out) return MLLM_HIP_ERR_ARG;
    MllmFile f;
    if (!f.open(path)) { fprintf(stderr, "mllm_hip: cannot open/parse %s\n", path); return MLLM_HIP_ERR_IO; }
    M *m = new M();
    m->c = *cfg;
    const auto &c = m->c;
    if (c.hidden % c.heads || c.hidden % 256 || c.inter % 256 || c.heads % c.kv_heads) { delete m; return MLLM_HIP_ERR_SHAPE; }
    m->D = c.hidden / c.heads;
    m->HD = c.heads * m->D;
    m->KVD = c.kv_heads * m->D;
    m->QKV = m->HD + 2 * m->KVD;
    if (c.mrope_section[0] + c.mrope_section[1] + c.mrope_section[2] != m->D / 2) { delete m; return MLLM_HIP_ERR_SHAPE; }
    int rc = 0;
#define CK(expr) do { rc = (expr); if (rc) { mllm_hip_qwen2vl_destroy(m); return rc; } } while (0)
    if (hipStreamCreate(&m->st) != hipSuccess || hipEventCreate(&m->ev0) != hipSuccess || hipEventCreate(&m->ev1) != hipSuccess) {
        mllm_hip_qwen2vl_destroy(m); return MLLM_HIP_ERR_HIP;
    }
    const int H = c.hidden, I = c.inter;
    // ---- LLM ----
    {
        const Entry *e;
        const uint64_t nblk = (uint64_t)c.vocab * (H / 32);
        CK(need(f, "model.embed_tokens.weight", &e, MLLM_HIP_Q4_0, nblk * 18));
        uint8_t *raw;
        CK(m->dalloc(&raw, nblk * 18));
        CK(m->upload(raw, f.base + e->off, nblk * 18));
        CK(m->dalloc(&m->emb_qs, nblk * 16));
        CK(m->dalloc(&m->emb_d, nblk * 2));
        CK(mllm_hip_repack_q40(raw, m->emb_qs, m->emb_d, (int64_t)nblk, m->st));
        CK(hipStreamSynchronize(m->st) == hipSuccess ? 0 : MLLM_HIP_ERR_HIP);
        m->allocs.erase(std::find(m->allocs.begin(), m->allocs.end(), (void *)raw));
        CK(hipFree(raw) == hipSuccess ? 0 : MLLM_HIP_ERR_HIP);
        if (!c.tie_embedding) { fprintf(stderr, "mllm_hip: untied lm_head not implemented in this engine\n"); mllm_hip_qwen2vl_destroy(m); return MLLM_HIP_ERR_ARG; }
    }
    m->layers.resize(c.layers);
    for (int i = 0; i < c.layers; ++i) {
        auto &L = m->layers[i];
        const std::string p = "model.layers." + std::to_string(i) + ".";
        CK(load_f32(m, f, p + "input_layernorm.weight", H, &L.in_norm));
        CK(load_f32(m, f, p + "post_attention_layernorm.weight", H, &L.post_norm));
        CK(load_linear_q4k(m, f, {p + "self_attn.q_proj", p + "self_attn.k_proj", p + "self_attn.v_proj"}, {m->HD, m->KVD, m->KVD}, H, true, &L.qkv));
        CK(load_linear_q4k(m, f, {p + "self_attn.o_proj"}, {H}, m->HD, false, &L.o));
        CK(load_linear_q4k(m, f, {p + "mlp.gate_proj", p + "mlp.up_proj"}, {I, I}, H, false, &L.gu));
        CK(load_linear_q4k(m, f, {p + "mlp.down_proj"}, {H}, I, false, &L.down));
    }
    CK(load_f32(m, f, "model.norm.weight", H, &m->final_norm));
    m->decode_weight_bytes = (int64_t)c.layers * ((int64_t)(m->QKV + H) * (H / 256) * 144 + (int64_t)2 * I * (H / 256) * 144 + (int64_t)H * (I / 256) * 144)
                             + (int64_t)c.vocab * (H / 32) * 18;
    // ---- vision ----
    m->has_vision = f.find("visual.patch_embed.proj.weight") != nullptr && c.v_dim > 0;
    if (m->has_vision) {
        const int V = c.v_dim, VM = 4 * V, PE = 3 * 2 * c.v_patch * c.v_patch, MM = V * c.v_merge * c.v_merge;
        if (V % 256 || V % c.v_heads) { mllm_hip_qwen2vl_destroy(m); return MLLM_HIP_ERR_SHAPE; }
        CK(load_f32(m, f, "visual.patch_embed.proj.weight", (size_t)V * PE, &m->patch_w));
        m->vblocks.resize(c.v_blocks);
        for (int i = 0; i < c.v_blocks; ++i) {
            auto &B = m->vblocks[i];
            const std::string p = "visual.blocks." + std::to_string(i) + ".";
            CK(load_f32(m, f, p + "norm1.weight", V, &B.n1w)); CK(load_f32(m, f, p + "norm1.bias", V, &B.n1b));
            CK(load_f32(m, f, p + "norm2.weight", V, &B.n2w)); CK(load_f32(m, f, p + "norm2.bias", V, &B.n2b));
            CK(load_linear_q4k(m, f, {p + "attn.qkv"}, {3 * V}, V, true, &B.qkv));
            CK(load_linear_q4k(m, f, {p + "attn.proj"}, {V}, V, true, &B.proj));
            CK(load_linear_q4k(m, f, {p + "mlp.fc1"}, {VM}, V, true, &B.fc1));
            CK(load_linear_q4k(m, f, {p + "mlp.fc2"}, {V}, VM, true, &B.fc2));
        }
        CK(load_f32(m, f, "visual.merger.ln_q.weight", V, &m->lnq_w)); CK(load_f32(m, f, "visual.merger.ln_q.bias", V, &m->lnq_b));
        CK(load_linear_q4k(m, f, {"visual.merger.mlp.0"}, {MM}, MM, true, &m->m0));
        CK(load_linear_q4k(m, f, {"visual.merger.mlp.2"}, {H}, MM, true, &m->m2));
        std::vector<uint16_t> g(65536), q(65536);
        mllm_hip_build_act_luts(g.data(), q.data());
        CK(m->dalloc(&m->lut_gelu, 65536 * 2)); CK(m->dalloc(&m->lut_qgelu, 65536 * 2));
        CK(m->upload(m->lut_gelu, g.data(), 65536 * 2)); CK(m->upload(m->lut_qgelu, q.data(), 65536 * 2));
    }
    // ---- activations ----
    const int T = c.cache_limit;
    m->max_tok = T;
    CK(m->dalloc(&m->h0, (size_t)T * H * 4)); CK(m->dalloc(&m->h1, (size_t)T * H * 4));
    CK(m->dalloc(&m->qkv, (size_t)T * m->QKV * 4)); CK(m->dalloc(&m->attn, (size_t)T * m->HD * 4));
    CK(m->dalloc(&m->gu, (size_t)T * 2 * I * 4)); CK(m->dalloc(&m->act, (size_t)T * I * 4));
    CK(m->dalloc(&m->logits, (size_t)c.vocab * 4)); CK(m->dalloc(&m->normed, (size_t)H * 4));
    CK(m->dalloc(&m->ids_f, (size_t)T * 4)); CK(m->dalloc(&m->idx_i, (size_t)T * 4)); CK(m->dalloc(&m->tok_dev, 16));
    CK(alloc_q8(m, &m->xq, T, H > m->HD ? H : m->HD)); CK(alloc_q8(m, &m->xq2, T, I));
    CK(m->dalloc(&m->x80_qs, (size_t)H)); CK(m->dalloc(&m->x80_d, (size_t)(H / 32) * 2));
    CK(m->dalloc(&m->rope_sin, (size_t)T * (m->D / 2) * 4)); CK(m->dalloc(&m->rope_cos, (size_t)T * (m->D / 2) * 4));
    // + 64 rows: the decode attention reads whole 64-key splits speculatively
    // K slab [layers][T][KVD] (the reference's BSHD cache rows); V slab transposed [layers][KVD][vt_ld], vt_ld = T rounded up + 128 keys
    // of zero padding (the decode walk over-reads whole 16-byte vectors; prefill reads 4 keys at a time)
    m->vt_ld = ((T + 63) & ~63) + 128;
    CK(m->dalloc(&m->kslab, ((size_t)c.layers * T + 64) * m->KVD * 2)); CK(m->dalloc(&m->vslab, (size_t)c.layers * m->KVD * m->vt_ld * 2));
    CK(hipMemset(m->kslab, 0, ((size_t)c.layers * T + 64) * m->KVD * 2) == hipSuccess ? 0 : MLLM_HIP_ERR_HIP);
    CK(hipMemset(m->vslab, 0, (size_t)c.layers * m->KVD * m->vt_ld * 2) == hipSuccess ? 0 : MLLM_HIP_ERR_HIP);
    m->nsplit = (T + 63) / 64;
    {
        size_t wsb = mllm_hip_fa2_workspace_bytes(1, c.heads, m->D, T), wsd = (size_t)c.heads * m->nsplit * 136 * 4;
        CK(m->dalloc((uint8_t **)&m->fa_ws, wsb > wsd ? wsb : wsd));
    }
    CK(m->dalloc(&m->d_state, sizeof(DecodeState)));
    CK(m->dalloc(&m->dec_sin, (size_t)T * (m->D / 2) * 4)); CK(m->dalloc(&m->dec_cos, (size_t)T * (m->D / 2) * 4));
    CK(m->dalloc(&m->cur_sin, (size_t)m->D * 4)); CK(m->dalloc(&m->cur_cos, (size_t)m->D * 4));
    CK(m->dalloc(&m->part_val, (size_t)m->max_parts * 4)); CK(m->dalloc(&m->part_idx, (size_t)m->max_parts * 4));
    CK(m->dalloc(&m->history, (size_t)T * 4));
    {
        DecodeCtx &d = m->dctx;
        d.state = m->d_state; d.H = H; d.I = I; d.heads = c.heads; d.kv_heads = c.kv_heads; d.D = m->D; d.vocab = c.vocab; d.cache_limit = T;
        d.nsplit = m->nsplit; d.max_parts = m->max_parts; d.eps = c.rms_eps; d.emb_qs = m->emb_qs; d.emb_d = m->emb_d; d.final_norm = m->final_norm;
        d.x0 = m->h0; d.x1 = m->h1; d.qkv = m->qkv; d.act = m->act; d.logits = m->logits; d.fa_ws = (float *)m->fa_ws; d.part_val = m->part_val;
        d.part_idx = m->part_idx; d.tok_dev = m->tok_dev; d.history = m->history; d.rope_sin = m->dec_sin; d.rope_cos = m->dec_cos; d.cur_sin = m->cur_sin; d.cur_cos = m->cur_cos;
        d.kslab = m->kslab; d.vslab = m->vslab; d.vt_ld = m->vt_ld; d.normed = m->normed; d.x80_qs = m->x80_qs; d.x80_d = m->x80_d;
        for (auto &L : m->layers) {
            for (LinearW *lw : {&L.qkv, &L.o, &L.gu, &L.down}) {
                const int64_t nblk = (int64_t)lw->N * (lw->K / 256);
                uint8_t *wd = nullptr;
                CK(m->dalloc(&wd, (size_t)nblk * 144));
                CK(decode_order_q4k(lw->w, wd, nblk, m->st));
                lw->wd = wd;
            }
            DecodeLayer dl;
            dl.in_norm = L.in_norm; dl.post_norm = L.post_norm; dl.Wqkv = (const uint8_t *)L.qkv.wd; dl.bqkv = L.qkv.bias; dl.qkv_N = L.qkv.N;
            dl.Wo = (const uint8_t *)L.o.wd; dl.Wgu = (const uint8_t *)L.gu.wd; dl.Wdown = (const uint8_t *)L.down.wd;
            dl.Wgu_raw = (const uint8_t *)L.gu.w; dl.Wdown_raw = (const uint8_t *)L.down.w; dl.Wo_raw = (const uint8_t *)L.o.w;
            m->dlayers.push_back(dl);
        }
        m->use_graph = getenv("MLLM_HIP_NO_GRAPH") == nullptr;
    }
    *out = m;
    return MLLM_HIP_OK;
#undef CK
}

extern "C" void mllm_hip_qwen2vl_destroy(mllm_hip_qwen2vl *m) {
    if (!m) return;
    // teardown: nothing useful can be done with a failing release, the codes are dropped on purpose
    if (m->graph_exec) (void)hipGraphExecDestroy(m->graph_exec);
    if (m->graph) (void)hipGraphDestroy(m->graph);
    for (void *p : m->allocs) (void)hipFree(p);
    if (m->xpack) (void)hipFree(m->xpack);
    if (m->ev0) (void)hipEventDestroy(m->ev0);
    if (m->ev1) (void)hipEventDestroy(m->ev1);
    if (m->st) (void)hipStreamDestroy(m->st);
    delete m;
}
extern "C" int mllm_hip_qwen2vl_clear_kvcache(mllm_hip_qwen2vl *m) { m->cache_len = 0; m->last_pos = -1.0f; return MLLM_HIP_OK; }
extern "C" int64_t mllm_hip_qwen2vl_decode_weight_bytes(const mllm_hip_qwen2vl *m) { return m->decode_weight_bytes; }
extern "C" void *mllm_hip_qwen2vl_stream(mllm_hip_qwen2vl *m) { return (void *)m->st; }
// bring-up aid (not part of include/mllm_hip.h): device pointers of the prefill activations, 0 h0, 1 h1, 2 qkv, 3 attn, 4 gate|up, 5 act
extern "C" void *mllm_hip_qwen2vl_debug_ptr(mllm_hip_qwen2vl *m, int which) {
    switch (which) { case 0: return m->h0; case 1: return m->h1; case 2: return m->qkv; case 3: return m->attn; case 4: return m->gu; case 5: return m->act; case 6: return m->kslab; case 7: return m->vslab; case 8: return m->logits; }
    return nullptr;
}

// get_rope_index (modeling_qwen2_vl.hpp:436-595), batch 1, images only. pos = [3][S].
static void rope_index(const M *m, const int32_t *ids, int S, const int32_t *grid, bool has_img, std::vector<float> &pos) {
    pos.assign((size_t)3 * S, 0.0f);
    if (!has_img) {
        for (int a = 0; a < 3; ++a) for (int j = 0; j < S; ++j) pos[(size_t)a * S + j] = (float)j;
        return;
    }
    std::vector<int64_t> lp[3];
    size_t st = 0;
    int64_t cur = 0;
    int n_img = 0;
    for (int j = 0; j + 1 < S; ++j) if (ids[j] == m->c.vision_start_token_id && ids[j + 1] == m->c.image_token_id) n_img++;
    int remain = n_img;
    // count of vision starts (image or video) as the reference does
    int n_starts = 0;
    for (int j = 0; j + 1 < S; ++j) if (ids[j] == m->c.vision_start_token_id) n_starts++;
    for (int vs = 0; vs < n_starts; ++vs) {
        size_t ed = S;
        if (remain > 0) for (size_t j = st; j < (size_t)S; ++j) if (ids[j] == m->c.image_token_id) { ed = j; break; }
        if (ed == (size_t)S) break;
        const int64_t t = grid[0], h = grid[1], w = grid[2];  // one grid shared by all images of the call
        remain--;
        const int64_t gt = t, gh = h / m->c.v_merge, gw = w / m->c.v_merge;
        const size_t text_len = ed - st;
        if (text_len > 0) {
            const int64_t s0 = cur;
            for (size_t k = 0; k < text_len; ++k) for (int a = 0; a < 3; ++a) lp[a].push_back(s0 + (int64_t)k);
            cur += text_len;
        }
        for (int64_t ti = 0; ti < gt; ++ti)
            for (int64_t hi = 0; hi < gh; ++hi)
                for (int64_t wi = 0; wi < gw; ++wi) { lp[0].push_back(cur + ti); lp[1].push_back(cur + hi); lp[2].push_back(cur + wi); }
        cur = std::max(lp[0].back(), std::max(lp[1].back(), lp[2].back()));
        st = ed + gt * gh * gw;
    }
    if (st < (size_t)S) {
        const size_t text_len = S - st;
        const int64_t s0 = cur + 1;
        for (size_t k = 0; k < text_len; ++k) for (int a = 0; a < 3; ++a) lp[a].push_back(s0 + (int64_t)k);
    }
    for (int a = 0; a < 3; ++a) for (int j = 0; j < S && j < (int)lp[a].size(); ++j) pos[(size_t)a * S + j] = (float)lp[a][j];
}

// ---- quantised activations feeding a Linear.  Rows >= 16 meet the GEMM, whose activation operand is a packed layout: the
// producers (norms with fused quantisation, the quantiser) then write that layout directly into m->xpack (one scratch: every
// quantised buffer is consumed by the very next Linear); fewer rows take the GEMV on the three Q8_K planes.
static int ensure_xpack(M *m, int rows, int K) {
    const size_t need_b = mllm_hip_q4k_prepack_bytes(rows, K);
    if (need_b > m->xpack_bytes) {   // grown on demand (prefill only, never inside a captured graph)
        HH(hipStreamSynchronize(m->st));
        if (m->xpack) HH(hipFree(m->xpack));
        m->xpack = nullptr; m->xpack_bytes = 0;
        HH(hipMalloc(&m->xpack, need_b));
        m->xpack_bytes = need_b;
    }
    return 0;
}
static int q_quant(M *m, const float *x, const Q8Planes &p, int rows, int K) {
    if (rows >= 16) { EH(ensure_xpack(m, rows, K)); return mllm_hip_quantize_q8k_packed(x, m->xpack, rows, K, m->st); }
    return mllm_hip_quantize_q8k(x, p.qs, p.d, p.bs, rows, K, m->st);
}
// activation + quantiser in one launch for the GEMM path; the separate launches otherwise
static int q_act_quant(M *m, const float *x, float *scratch, const uint16_t *lut, const Q8Planes &p, int rows, int K) {
    if (rows >= 16) { EH(ensure_xpack(m, rows, K)); return mllm_hip_quantize_q8k_packed_act(x, lut, m->xpack, rows, K, m->st); }
    EH(mllm_hip_act_lut(x, scratch, (int64_t)rows * K, lut, m->st));
    return q_quant(m, scratch, p, rows, K);
}
static int q_silu_mul_quant(M *m, const float *gu, float *scratch, const Q8Planes &p, int rows, int I) {
    if (rows >= 16) { EH(ensure_xpack(m, rows, I)); return mllm_hip_quantize_q8k_packed_silu_mul(gu, m->xpack, rows, I, m->st); }
    EH(mllm_hip_silu_mul(gu, scratch, rows, I, m->st));
    return q_quant(m, scratch, p, rows, I);
}
static int q_rmsnorm(M *m, const float *x, const float *w, const Q8Planes &p, int rows, int dim, float eps) {
    if (rows >= 16) { EH(ensure_xpack(m, rows, dim)); return mllm_hip_rmsnorm_packed(x, w, nullptr, m->xpack, rows, dim, eps, 0, m->st); }
    return mllm_hip_rmsnorm(x, w, nullptr, p.qs, p.d, p.bs, rows, dim, eps, 0, m->st);
}
static int q_layernorm(M *m, const float *x, const float *w, const float *b, const Q8Planes &p, int rows, int dim, float eps) {
    if (rows >= 16) { EH(ensure_xpack(m, rows, dim)); return mllm_hip_layernorm_packed(x, w, b, nullptr, m->xpack, rows, dim, eps, m->st); }
    return mllm_hip_layernorm(x, w, b, nullptr, p.qs, p.d, p.bs, rows, dim, eps, m->st);
}
// Linear on the activations the last q_* call produced
static int lin(M *m, const LinearW &w, const Q8Planes &x, void *y, int ydt, int64_t ldy, const float *res, int Mrows) {
    if (Mrows >= 16) return mllm_hip_linear_q4kp_packed(w.wp, w.bias, m->xpack, y, ydt, ldy, res, Mrows, w.N, w.K, m->st);
    return mllm_hip_linear_q4k_q8k(w.w, w.bias, x.qs, x.d, x.bs, y, ydt, ldy, res, Mrows, w.N, w.K, m->st);
}

// Qwen2VisionModel::Forward for one image already resident in m->vpix ([N][PE]); writes [N/merge^2][hidden] to `out`
static int forward_vision(M *m, const int32_t *grid, float *out) {
    const auto &c = m->c;
    const int N = grid[0] * grid[1] * grid[2], V = c.v_dim, VM = 4 * V, PE = 3 * 2 * c.v_patch * c.v_patch;
    const int VD = V / c.v_heads, MM = V * c.v_merge * c.v_merge, NT = N / (c.v_merge * c.v_merge);
    hipStream_t st = m->st;
    {   // rotary tables (CPUVisionRoPE): rot_dim = head_dim/2
        std::vector<float> s((size_t)N * (VD / 2)), co((size_t)N * (VD / 2));
        EH(mllm_hip_vision_rope_table(grid[0], grid[1], grid[2], c.v_merge, VD / 2, s.data(), co.data()));
        HH(hipMemcpyAsync(m->vsin, s.data(), s.size() * 4, hipMemcpyHostToDevice, st));
        HH(hipMemcpyAsync(m->vcos, co.data(), co.size() * 4, hipMemcpyHostToDevice, st));
        HH(hipStreamSynchronize(st));
    }
    EH(mllm_hip_patch_gemm_f32(m->vpix, m->patch_w, nullptr, m->vx, N, PE, V, st));
    float *x = m->vx, *r = m->vr;
    for (auto &B : m->vblocks) {
        EH(q_layernorm(m, x, B.n1w, B.n1b, m->xq, N, V, 1e-6f));
        EH(lin(m, B.qkv, m->xq, m->vqkv, MLLM_HIP_F32, 3 * V, nullptr, N));
        // q and k are columns [0, 2V) of the same rows: one launch over 2 * heads heads rotates both in place
        EH(mllm_hip_rope_apply(m->vqkv, 3 * V, m->vsin, m->vcos, VD / 2, m->vqkv, MLLM_HIP_F32, 3 * V, N, 2 * c.v_heads, VD, st));
        EH(mllm_hip_fa2(m->vqkv, 3 * V, m->vqkv + V, 3 * V, m->vqkv + 2 * V, 3 * V, MLLM_HIP_F32, m->vattn, V, N, N, c.v_heads, c.v_heads, VD, 0,
                        nullptr, nullptr, st));
        EH(q_quant(m, m->vattn, m->xq, N, V));
        EH(lin(m, B.proj, m->xq, r, MLLM_HIP_F32, V, x, N));                       // residual = proj(attn) + x
        EH(q_layernorm(m, r, B.n2w, B.n2b, m->xq, N, V, 1e-6f));
        EH(lin(m, B.fc1, m->xq, m->vfc, MLLM_HIP_F32, VM, nullptr, N));
        EH(q_act_quant(m, m->vfc, m->vact, m->lut_qgelu, m->xq2, N, VM));
        EH(lin(m, B.fc2, m->xq2, x, MLLM_HIP_F32, V, r, N));                        // x = fc2(act) + residual
    }
    // PatchMerger: ln_q -> view [NT][MM] -> mlp.0 -> GELU -> mlp.2
    EH(mllm_hip_layernorm(x, m->lnq_w, m->lnq_b, r, nullptr, nullptr, nullptr, N, V, 1e-6f, st));
    EH(q_quant(m, r, m->xq2, NT, MM));
    EH(lin(m, m->m0, m->xq2, m->vm0, MLLM_HIP_F32, MM, nullptr, NT));
    EH(q_act_quant(m, m->vm0, m->vfc, m->lut_gelu, m->xq2, NT, MM));
    EH(lin(m, m->m2, m->xq2, out, MLLM_HIP_F32, c.hidden, nullptr, NT));
    return 0;
}

static int ensure_vision_buffers(M *m, int N) {
    if (N <= m->max_patch) return 0;
    const auto &c = m->c;
    const int V = c.v_dim, VM = 4 * V, PE = 3 * 2 * c.v_patch * c.v_patch, MM = V * c.v_merge * c.v_merge, NT = N / (c.v_merge * c.v_merge);
    const int VD = V / c.v_heads;
    EH(m->dalloc(&m->vx, (size_t)N * V * 4)); EH(m->dalloc(&m->vr, (size_t)N * V * 4));
    EH(m->dalloc(&m->vqkv, (size_t)N * 3 * V * 4)); EH(m->dalloc(&m->vattn, (size_t)N * V * 4));
    EH(m->dalloc(&m->vfc, (size_t)N * VM * 4)); EH(m->dalloc(&m->vact, (size_t)N * VM * 4));
    EH(m->dalloc(&m->vpix, (size_t)N * PE * 4));
    EH(m->dalloc(&m->vsin, (size_t)N * (VD / 2) * 4)); EH(m->dalloc(&m->vcos, (size_t)N * (VD / 2) * 4));
    EH(m->dalloc(&m->vemb, (size_t)NT * c.hidden * 4)); EH(m->dalloc(&m->vm0, (size_t)NT * MM * 4));
    // the shared q8k planes must hold N rows of V (xq) and of VM / NT rows of MM (xq2)
    Q8Planes a, b;
    EH(alloc_q8(m, &a, N, V)); EH(alloc_q8(m, &b, N, VM > MM ? VM : MM));
    if ((size_t)N * V > (size_t)m->max_tok * std::max(c.hidden, m->HD)) m->xq = a;
    if ((size_t)N * VM > (size_t)m->max_tok * c.inter) m->xq2 = b;
    m->max_patch = N;
    return 0;
}

// One LLM forward over S new tokens whose embeddings are in m->h0 ([S][H]); logits of the last token -> m->logits
static int forward_llm(M *m, int S, const float *pos3) {
    const auto &c = m->c;
    const int H = c.hidden, I = c.inter, D = m->D, T0 = m->cache_len;
    hipStream_t st = m->st;
    if (T0 + S > c.cache_limit) { fprintf(stderr, "mllm_hip: KV cache overflow (%d + %d > %d)\n", T0, S, c.cache_limit); return MLLM_HIP_ERR_SHAPE; }
    {   // M-RoPE tables for these S positions
        std::vector<float> s((size_t)S * (D / 2)), co((size_t)S * (D / 2));
        EH(mllm_hip_mrope_table(c.rope_theta, D, pos3, S, c.mrope_section, 3, s.data(), co.data()));
        HH(hipMemcpyAsync(m->rope_sin, s.data(), s.size() * 4, hipMemcpyHostToDevice, st));
        HH(hipMemcpyAsync(m->rope_cos, co.data(), co.size() * 4, hipMemcpyHostToDevice, st));
        HH(hipStreamSynchronize(st));  // host vectors go out of scope
    }
    float *h = m->h0, *h2 = m->h1;
    const int n_layers = getenv("MLLM_HIP_MAX_LAYERS") ? std::min(c.layers, atoi(getenv("MLLM_HIP_MAX_LAYERS"))) : c.layers;   // bring-up aid
    for (int li = 0; li < n_layers; ++li) {
        auto &L = m->layers[li];
        uint16_t *kl = m->kslab + (size_t)li * c.cache_limit * m->KVD, *vl = m->vslab + (size_t)li * m->KVD * m->vt_ld;
        EH(q_rmsnorm(m, h, L.in_norm, m->xq, S, H, c.rms_eps));
        EH(lin(m, L.qkv, m->xq, m->qkv, MLLM_HIP_F32, m->QKV, nullptr, S));
        // q_rope in place; k_rope -> fp16 slab rows [T0, T0+S); v -> fp16 slab (KVCache zero-copy append)
        EH(mllm_hip_rope_apply(m->qkv, m->QKV, m->rope_sin, m->rope_cos, D / 2, m->qkv, MLLM_HIP_F32, m->QKV, S, c.heads, D, st));
        EH(mllm_hip_rope_apply(m->qkv + m->HD, m->QKV, m->rope_sin, m->rope_cos, D / 2, kl + (size_t)T0 * m->KVD, MLLM_HIP_F16, m->KVD, S, c.kv_heads, D, st));
        EH(mllm_hip_store_f16_t(m->qkv + m->HD + m->KVD, m->QKV, vl + T0, m->vt_ld, S, m->KVD, st));
        EH(mllm_hip_fa2_vt(m->qkv, m->QKV, kl, m->KVD, vl, m->vt_ld, m->attn, m->HD, S, T0 + S, c.heads, c.kv_heads, D, 1, st));
        EH(q_quant(m, m->attn, m->xq, S, m->HD));
        EH(lin(m, L.o, m->xq, h2, MLLM_HIP_F32, H, h, S));                          // tmp = o_proj(attn) + x
        EH(q_rmsnorm(m, h2, L.post_norm, m->xq, S, H, c.rms_eps));
        EH(lin(m, L.gu, m->xq, m->gu, MLLM_HIP_F32, 2 * I, nullptr, S));
        EH(q_silu_mul_quant(m, m->gu, m->act, m->xq2, S, I));
        EH(lin(m, L.down, m->xq2, h, MLLM_HIP_F32, H, h2, S));                      // x = down(...) + tmp
    }
    // final norm on the last token only (norm then clip({-1}) == clip then norm), tied lm_head through Q8_0 activations
    EH(mllm_hip_rmsnorm(h + (size_t)(S - 1) * H, m->final_norm, m->normed, nullptr, nullptr, nullptr, 1, H, 1e-6f, 0, st));
    EH(mllm_hip_quantize_q80(m->normed, m->x80_qs, m->x80_d, 1, H, st));
    EH(mllm_hip_linear_q40_q80(m->emb_qs, m->emb_d, nullptr, m->x80_qs, m->x80_d, m->logits, c.vocab, 1, c.vocab, H, st));
    EH(mllm_hip_argmax(m->logits, c.vocab, m->tok_dev, st));
    m->cache_len = T0 + S;
    return 0;
}

static int arm_decode(M *m);

static int finish(M *m, float *logits_host, int32_t *next_token, float *elapsed_ms) {
    HH(hipEventRecord(m->ev1, m->st));
    HH(hipEventSynchronize(m->ev1));
    if (elapsed_ms) HH(hipEventElapsedTime(elapsed_ms, m->ev0, m->ev1));
    if (logits_host) HH(hipMemcpy(logits_host, m->logits, (size_t)m->c.vocab * 4, hipMemcpyDeviceToHost));
    if (next_token) HH(hipMemcpy(next_token, m->tok_dev, 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int mllm_hip_qwen2vl_prefill(mllm_hip_qwen2vl *m, const int32_t *ids, int n_ids, const float *pixel_values, const int32_t *grid_thw,
                                        float *logits_host, int32_t *next_token, float *elapsed_ms) {
    if (!m || !ids || n_ids <= 0 || n_ids > m->c.cache_limit) return MLLM_HIP_ERR_ARG;
    const auto &c = m->c;
    const bool has_img = pixel_values != nullptr;
    if (has_img && (!m->has_vision || !grid_thw)) return MLLM_HIP_ERR_ARG;
    std::vector<float> idf(n_ids);
    std::vector<int> where;
    for (int i = 0; i < n_ids; ++i) { idf[i] = (float)ids[i]; if (ids[i] == c.image_token_id) where.push_back(i); }
    int N = 0;
    if (has_img) {
        N = grid_thw[0] * grid_thw[1] * grid_thw[2];
        if ((int)where.size() != N / (c.v_merge * c.v_merge)) { fprintf(stderr, "mllm_hip: %zu image tokens but %d visual tokens\n", where.size(), N / (c.v_merge * c.v_merge)); return MLLM_HIP_ERR_SHAPE; }
        EH(ensure_vision_buffers(m, N));
        HH(hipMemcpy(m->vpix, pixel_values, (size_t)N * 3 * 2 * c.v_patch * c.v_patch * 4, hipMemcpyHostToDevice));
        HH(hipMemcpy(m->idx_i, where.data(), where.size() * 4, hipMemcpyHostToDevice));
    }
    HH(hipMemcpy(m->ids_f, idf.data(), (size_t)n_ids * 4, hipMemcpyHostToDevice));
    std::vector<float> pos;
    rope_index(m, ids, n_ids, grid_thw, has_img, pos);
    // ---- timed region: inputs resident in HBM ----
    HH(hipEventRecord(m->ev0, m->st));
    EH(mllm_hip_embedding_q40(m->ids_f, m->emb_qs, m->emb_d, m->h0, n_ids, c.hidden, c.vocab, m->st));
    if (has_img) {
        EH(forward_vision(m, grid_thw, m->vemb));
        EH(mllm_hip_index_put_rows(m->h0, m->vemb, m->idx_i, (int)where.size(), c.hidden, m->st));
    }
    EH(forward_llm(m, n_ids, pos.data()));
    m->last_pos = pos[(size_t)n_ids - 1];  // row 0 (t axis), last column: get_position_ids' decode branch
    EH(finish(m, logits_host, next_token, elapsed_ms));
    return arm_decode(m);
}

// After a prefill: device step state {T, step = 0, token}, and the M-RoPE rows of every position the decode loop can still
// reach (get_position_ids' decode branch: all three axes = last_pos + 1 + step, modeling_qwen2_vl.hpp:423-432).
static int arm_decode(M *m) {
    const auto &c = m->c;
    const int rows = c.cache_limit - m->cache_len;
    m->dec_rows = rows;
    if (rows > 0) {
        std::vector<float> pos((size_t)3 * rows), s((size_t)rows * (m->D / 2)), co((size_t)rows * (m->D / 2));
        for (int a = 0; a < 3; ++a) for (int r = 0; r < rows; ++r) pos[(size_t)a * rows + r] = m->last_pos + 1.0f + (float)r;
        EH(mllm_hip_mrope_table(c.rope_theta, m->D, pos.data(), rows, c.mrope_section, 3, s.data(), co.data()));
        HH(hipMemcpy(m->dec_sin, s.data(), s.size() * 4, hipMemcpyHostToDevice));
        HH(hipMemcpy(m->dec_cos, co.data(), co.size() * 4, hipMemcpyHostToDevice));
        HH(hipMemcpy(m->cur_sin, s.data(), (size_t)(m->D / 2) * 4, hipMemcpyHostToDevice));
        HH(hipMemcpy(m->cur_cos, co.data(), (size_t)(m->D / 2) * 4, hipMemcpyHostToDevice));
    }
    int tok = 0;
    HH(hipMemcpy(&tok, m->tok_dev, 4, hipMemcpyDeviceToHost));
    DecodeState st0 = {m->cache_len, 0, tok, 0};
    HH(hipMemcpy(m->d_state, &st0, sizeof(st0), hipMemcpyHostToDevice));
    return 0;
}

static int launch_step(M *m) {
    if (m->use_graph) {
        if (!m->graph_exec) {
            HH(hipStreamBeginCapture(m->st, hipStreamCaptureModeThreadLocal));
            int rc = decode_step_launch(m->dctx, m->dlayers.data(), (int)m->dlayers.size(), m->st);
            hipError_t e = hipStreamEndCapture(m->st, &m->graph);
            if (rc) return rc;
            HH(e);
            HH(hipGraphInstantiate(&m->graph_exec, m->graph, nullptr, nullptr, 0));
        }
        HH(hipGraphLaunch(m->graph_exec, m->st));
        return 0;
    }
    return decode_step_launch(m->dctx, m->dlayers.data(), (int)m->dlayers.size(), m->st);
}

extern "C" int mllm_hip_qwen2vl_decode(mllm_hip_qwen2vl *m, int32_t token, float *logits_host, int32_t *next_token, float *elapsed_ms) {
    if (!m || m->cache_len <= 0) return MLLM_HIP_ERR_ARG;
    if (m->cache_len + 1 > m->c.cache_limit) { fprintf(stderr, "mllm_hip: KV cache overflow (%d + 1 > %d)\n", m->cache_len, m->c.cache_limit); return MLLM_HIP_ERR_SHAPE; }
    HH(hipMemcpyAsync(&m->d_state->token, &token, 4, hipMemcpyHostToDevice, m->st));
    HH(hipStreamSynchronize(m->st));
    HH(hipEventRecord(m->ev0, m->st));
    EH(launch_step(m));
    m->cache_len += 1;
    m->last_pos += 1.0f;
    return finish(m, logits_host, next_token, elapsed_ms);
}

extern "C" int mllm_hip_qwen2vl_generate(mllm_hip_qwen2vl *m, int32_t first_token, int steps, int32_t *tokens_host, float *elapsed_ms) {
    if (!m || m->cache_len <= 0 || steps <= 0) return MLLM_HIP_ERR_ARG;
    if (m->cache_len + steps > m->c.cache_limit) { fprintf(stderr, "mllm_hip: KV cache overflow (%d + %d > %d)\n", m->cache_len, steps, m->c.cache_limit); return MLLM_HIP_ERR_SHAPE; }
    DecodeState st0;
    HH(hipMemcpy(&st0, m->d_state, sizeof(st0), hipMemcpyDeviceToHost));
    const int step0 = st0.step;
    HH(hipMemcpy(&m->d_state->token, &first_token, 4, hipMemcpyHostToDevice));
    HH(hipEventRecord(m->ev0, m->st));
    for (int s = 0; s < steps; ++s) EH(launch_step(m));
    m->cache_len += steps;
    m->last_pos += (float)steps;
    EH(finish(m, nullptr, nullptr, elapsed_ms));
    if (tokens_host) HH(hipMemcpy(tokens_host, m->history + step0, (size_t)steps * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int mllm_hip_qwen2vl_vision(mllm_hip_qwen2vl *m, const float *pixel_values_host, const int32_t *grid_thw, int n_img, float *embeds_dev,
                                       float *elapsed_ms) {
    if (!m || !m->has_vision || !grid_thw || n_img <= 0) return MLLM_HIP_ERR_ARG;
    const auto &c = m->c;
    const int N = grid_thw[0] * grid_thw[1] * grid_thw[2], PE = 3 * 2 * c.v_patch * c.v_patch, NT = N / (c.v_merge * c.v_merge);
    EH(ensure_vision_buffers(m, N));
    float total = 0.0f;
    for (int i = 0; i < n_img; ++i) {
        HH(hipMemcpy(m->vpix, pixel_values_host + (size_t)i * N * PE, (size_t)N * PE * 4, hipMemcpyHostToDevice));
        HH(hipEventRecord(m->ev0, m->st));
        EH(forward_vision(m, grid_thw, embeds_dev + (size_t)i * NT * c.hidden));
        HH(hipEventRecord(m->ev1, m->st));
        HH(hipEventSynchronize(m->ev1));
        float ms;
        HH(hipEventElapsedTime(&ms, m->ev0, m->ev1));
        total += ms;
    }
    if (elapsed_ms) *elapsed_ms = total;
    return 0;
}

extern "C" int mllm_hip_qwen2vl_time_gemv(mllm_hip_qwen2vl *m, int which, int iters, float *ms_per_launch, int64_t *bytes_per_launch) {
    // which 0..3: stand-alone Q4_K GEMV launcher on gate|up, down, qkv, o; which 10..14: fused decode kernels qkv, attn, o-proj,
    // gate|up, down.  Launch i uses layer i % layers, so consecutive launches stream different weights (28 x 15.5 MB does not
    // fit the 256 MiB Infinity Cache): the time is that of a cold HBM stream, like inside the decode step.
    if (!m || iters <= 0) return MLLM_HIP_ERR_ARG;
    int nl = (int)m->layers.size();
    if (const char *e = getenv("MLLM_HIP_TIME_LAYERS")) nl = std::max(1, std::min(nl, atoi(e)));   // fewer layers: an Infinity-Cache-warm stream
    auto launch = [&](int i) -> int {
        auto &L = m->layers[i % nl];
        if (which >= 10) return decode_kernel_launch(m->dctx, m->dlayers.data(), i % nl == 0 && which == 10 ? 1 % nl : i % nl, which - 10, m->st);
        const LinearW &w = which == 0 ? L.gu : (which == 1 ? L.down : (which == 2 ? L.qkv : L.o));
        const Q8Planes &x = which == 1 ? m->xq2 : m->xq;
        float *y = which == 0 ? m->gu : m->h1;
        return lin(m, w, x, y, MLLM_HIP_F32, w.N, nullptr, 1);
    };
    for (int i = 0; i < nl; ++i) EH(launch(i));
    HH(hipEventRecord(m->ev0, m->st));
    for (int i = 0; i < iters; ++i) EH(launch(i));
    HH(hipEventRecord(m->ev1, m->st));
    HH(hipEventSynchronize(m->ev1));
    float ms;
    HH(hipEventElapsedTime(&ms, m->ev0, m->ev1));
    if (ms_per_launch) *ms_per_launch = ms / iters;
    auto &L0 = m->layers[0];
    const int k = which >= 10 ? which - 10 : -1;
    const LinearW *w = which == 0 || k == 3 ? &L0.gu : (which == 1 || k == 4 ? &L0.down : (which == 2 || k == 0 ? &L0.qkv : &L0.o));
    if (bytes_per_launch) *bytes_per_launch = k == 1 ? (int64_t)2 * m->cache_len * m->KVD * 2 : (int64_t)w->N * (w->K / 256) * 144;
    return 0;
}
