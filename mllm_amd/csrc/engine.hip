// mllm_amd/csrc/engine.hip -- host side of the hot path: the .mllm loader and the reference's model graphs on the launchers, resident on the device.
//
// What it mirrors (all host logic, the arithmetic lives in the kernels_*.hip launchers):
//   ParamLoader            mllm/ParamLoader.cpp:157-286 (index parse, mmap), :88-141 (load)           -> MllmFile + Loader (pinned double buffer)
//   Qwen2VLModel::Forward  mllm/models/qwen2_vl/modeling_qwen2_vl.hpp:381-404                          -> prefill / forward_llm
//   QWen2Decoder/Attention/MLP :193-335; QWenDecoder modeling_qwen.hpp:60-101; TinyLLaMABlock modeling_tinyllama.hpp:15-42;
//   LLaMABlock modeling_llama.hpp:40-80 (all: RMSNorm -> MultiHeadAttention (modeling_transformer.hpp:35-219) -> +x -> RMSNorm -> SiLU MLP -> +x)
//                                                                                                      -> the layer loop of forward_llm
//   Qwen2VisionModel       modeling_qwen2_vl.hpp:21-191 (patch embed, VisionBlock x32, PatchMerger)   -> forward_vision (kind QWEN2VL)
//   LLaVAVisionModel       modeling_llava.hpp:39-98 (CLIP: conv patch embed, cls row, position rows, pre_layrnorm, ViTBlock xN, projector)
//   ViTModel               modeling_vit.hpp:63-111 (conv patch embed + bias, cls, positions, ViTBlock xN, LayerNorm(cls), classifier) -> forward_vision
//   get_rope_index / get_position_ids  modeling_qwen2_vl.hpp:413-595                                   -> rope_index()
//   demo loops + argmax    examples/demo_qwen2_vl.cpp:53-63, demo_qwen.cpp, demo_llava.cpp:39-57       -> prefill / decode / generate
//   Module::generate       mllm/Module.cpp:63-100 + Generate.cpp:17-142 (greedy / top-k / top-p)       -> generate_sampled
//   KVCache                backends/cpu/op/CPUKVCache.cpp:10-131,253-275 (fp16 slab, zero-copy append) -> kv slabs + cache_len
// Layout in HBM: one allocation per weight tensor, Q4_K rows in their native 144-B blocks; q/k/v (and gate/up) rows are
// concatenated at load so one GEMV/GEMM serves the three (two) projections; embed_tokens (Q4_0) is split into a nibble plane and
// an fp16 scale plane; activations live in a handful of reusable fp32 / q8k-plane buffers sized for cache_limit tokens; K slab fp16
// [layers][cache_limit][Hkv*D], V slab fp16 transposed [layers][Hkv*D][vt_ld].
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
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

// .mllm index: int32 magic 20012; uint64 index_len; repeat { int32 name_len; name; uint64 data_len; uint64 file_offset; int32 dtype } (ParamLoader.cpp:157-286).
// Nothing in the header is trusted: every length is checked against the mapped size before it is used.
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
        if (size < 12) return false;
        base = (uint8_t *)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (base == MAP_FAILED) { base = nullptr; return false; }
        int32_t magic;
        memcpy(&magic, base, 4);
        if (magic != 20012) return false;  // _MAGIC_NUMBER, mllm/ParamLoader.hpp:48
        uint64_t ilen;
        memcpy(&ilen, base + 4, 8);
        if (ilen > size - 12) return false;
        const uint8_t *p = base + 12, *end = base + 12 + ilen;
        while (p < end) {
            if ((size_t)(end - p) < 4) return false;
            int32_t nl;
            memcpy(&nl, p, 4); p += 4;
            if (nl < 0 || (size_t)(end - p) < (size_t)nl + 20) return false;
            std::string name((const char *)p, nl); p += nl;
            Entry e;
            memcpy(&e.len, p, 8); memcpy(&e.off, p + 8, 8);
            int32_t dt; memcpy(&dt, p + 16, 4); e.dtype = dt;
            p += 20;
            if (e.off > size || e.len > size - e.off) return false;      // data must lie inside the file
            idx[name] = e;
        }
        return true;
    }
    const Entry *find(const std::string &n) const { auto it = idx.find(n); return it == idx.end() ? nullptr : &it->second; }
    ~MllmFile() { if (base) munmap(base, size); if (fd >= 0) ::close(fd); }
};

// mmap -> pinned staging (two buffers) -> hipMemcpyAsync on a copy stream.  The CPU fills buffer b+1 from the page cache while the DMA engine
// drains buffer b; repack kernels of tensors that have landed run meanwhile on the compute stream behind fence() (SURVEY N1;
// precedent for a device upload at load: mllm/backends/opencl/OpenCLBackend.cpp:928-980).
struct Loader {
    static constexpr size_t CH = (size_t)32 << 20;
    hipStream_t copy = nullptr;
    uint8_t *pin[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr}, first = nullptr, last = nullptr, fence_ev = nullptr;
    bool used[2] = {false, false}, started = false;
    int cur = 0;
    int64_t bytes = 0;
    int init() {
        MH_CHECK(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            MH_CHECK(hipHostMalloc((void **)&pin[b], CH, hipHostMallocDefault));
            MH_CHECK(hipEventCreateWithFlags(&done[b], hipEventDisableTiming));
        }
        MH_CHECK(hipEventCreate(&first)); MH_CHECK(hipEventCreate(&last));
        MH_CHECK(hipEventCreateWithFlags(&fence_ev, hipEventDisableTiming));
        return 0;
    }
    int put(void *dst, const uint8_t *src, size_t n) {
        if (!started) { MH_CHECK(hipEventRecord(first, copy)); started = true; }
        for (size_t off = 0; off < n; off += CH) {
            const size_t len = std::min(CH, n - off);
            const int b = cur;
            if (used[b]) MH_CHECK(hipEventSynchronize(done[b]));
            memcpy(pin[b], src + off, len);
            MH_CHECK(hipMemcpyAsync((uint8_t *)dst + off, pin[b], len, hipMemcpyHostToDevice, copy));
            MH_CHECK(hipEventRecord(done[b], copy));
            used[b] = true;
            cur ^= 1;
        }
        bytes += (int64_t)n;
        return 0;
    }
    // everything put() so far is visible to work enqueued on `compute` after this call
    int fence(hipStream_t compute) {
        MH_CHECK(hipEventRecord(fence_ev, copy));
        MH_CHECK(hipStreamWaitEvent(compute, fence_ev, 0));
        return 0;
    }
    int finish(float *h2d_ms) {
        if (started) {
            MH_CHECK(hipEventRecord(last, copy));
            MH_CHECK(hipEventSynchronize(last));
            if (h2d_ms) MH_CHECK(hipEventElapsedTime(h2d_ms, first, last));
        } else if (h2d_ms) *h2d_ms = 0.0f;
        return 0;
    }
    void destroy() {
        for (int b = 0; b < 2; ++b) { if (pin[b]) (void)hipHostFree(pin[b]); if (done[b]) (void)hipEventDestroy(done[b]); pin[b] = nullptr; done[b] = nullptr; }
        if (first) (void)hipEventDestroy(first);
        if (last) (void)hipEventDestroy(last);
        if (fence_ev) (void)hipEventDestroy(fence_ev);
        if (copy) (void)hipStreamDestroy(copy);
        first = last = fence_ev = nullptr; copy = nullptr;
    }
};

struct LinearW {          // one (possibly row-concatenated) Linear
    void *w = nullptr;    // Q4_K blocks [N][K/256]
    void *wp = nullptr;   // the same rows packed for the M >= 16 GEMM (mllm_hip_q4k_prepack): prefill / vision read these
    void *wd = nullptr;   // LLM decode Linears only: the rows in decode order (decode_order_q4k) for the fused decode kernels
    float *bias = nullptr;
    int N = 0, K = 0;
};

struct Q8Planes { int8_t *qs = nullptr; float *d = nullptr; int16_t *bs = nullptr; };

#define EH(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)
#define HH(expr) MH_CHECK(expr)

enum VKind { V_NONE = 0, V_QWEN2VL = 1, V_CLIP = 2, V_VIT = 3 };

}  // namespace

struct mllm_hip_model {
    mllm_hip_model_config c;
    hipStream_t st = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int32_t *pin_tok = nullptr;      // two page-locked words: the token handed to a decode step, the token it produced (no pageable staging, no extra synchronisation)
    std::vector<void *> allocs, temps, vis_allocs;
    Loader ld;
    int D = 0, HD = 0, KVD = 0, QKV = 0;
    bool has_llm = false, mrope = false;
    std::string prefix;     // "language_model." for LLaVA's text side
    // LLM weights
    struct Layer { float *in_norm, *post_norm; LinearW qkv, o, gu, down; };
    std::vector<Layer> layers;
    float *final_norm = nullptr;
    uint8_t *emb_qs = nullptr; uint16_t *emb_d = nullptr;
    LinearW head;           // Linear lm_head when not tied
    // vision weights
    int vkind = V_NONE;
    struct VBlock { float *n1w, *n1b, *n2w, *n2b; LinearW qkv, proj, fc1, fc2; };
    std::vector<VBlock> vblocks;
    float *patch_w = nullptr, *patch_b = nullptr, *cls_tok = nullptr, *pos_emb = nullptr, *pre_w = nullptr, *pre_b = nullptr;
    float *lnq_w = nullptr, *lnq_b = nullptr;      // QWEN2VL merger.ln_q; VIT: final layernorm
    LinearW m0, m2;                                // QWEN2VL merger mlp.0 / mlp.2; LLAVA projector linear_1 / linear_2; VIT: m0 = classifier
    uint16_t *lut_gelu = nullptr, *lut_qgelu = nullptr;
    // activations
    int max_tok = 0;
    float *h0 = nullptr, *h1 = nullptr, *qkv = nullptr, *attn = nullptr, *gu = nullptr, *act = nullptr, *logits = nullptr, *normed = nullptr;
    float *ids_f = nullptr; int *idx_i = nullptr; int *tok_dev = nullptr;
    Q8Planes xq, xq2;            // K = hidden, K = inter
    Q8Planes vxq, vxq2;          // vision: K = v_dim, K = max(v_ffn, merger width)
    int8_t *x80_qs = nullptr; uint16_t *x80_d = nullptr;
    float *rope_sin = nullptr, *rope_cos = nullptr;
    std::vector<float> hf_sin, hf_cos;   // CPURoPE's static table [cache_limit][D/2] (HF rotary archs)
    uint16_t *kslab = nullptr, *vslab = nullptr;
    int vt_ld = 0;
    void *fa_ws = nullptr;
    void *xpack = nullptr;
    size_t xpack_bytes = 0;
    // vision activations (sized for max_patch tokens)
    int max_patch = 0, vis_batch = 0;     // tokens per image and images per pass the vision buffers are sized for
    float *vx = nullptr, *vr = nullptr, *vqkv = nullptr, *vattn = nullptr, *vfc = nullptr, *vact = nullptr, *vpix[2] = {nullptr, nullptr}, *vpatch = nullptr, *vsin = nullptr,
          *vcos = nullptr, *vemb = nullptr, *vm0 = nullptr;
    hipEvent_t vup[2] = {nullptr, nullptr}, vfree[2] = {nullptr, nullptr};
    float *pin_img = nullptr; size_t pin_img_bytes = 0;
    // fused decode path (kernels_decode.hip): device-side step state, per-step rotary rows, captured graph
    std::vector<float> mrope_host_s, mrope_host_c;      // forward_llm's M-RoPE tables on the host (kept alive across the asynchronous copies)
    hipEvent_t mrope_host_free = nullptr;
    int vrope_grid[3] = {0, 0, 0};                      // the grid m->vsin / m->vcos were made for
    DecodeState *d_state = nullptr;
    unsigned long long *qkv_pairs = nullptr, *x_pairs = nullptr;
    unsigned long long *attn_pairs = nullptr;      // merged attention + o-projection launch: the attention's output rows as {value, epoch} pairs, [layers][heads * D]
    int *poll_err = nullptr;
    int *pin_err = nullptr;                        // page-locked word the flag is copied to with the step's other results
    float *dec_sin = nullptr, *dec_cos = nullptr, *part_val = nullptr, *cur_sin = nullptr, *cur_cos = nullptr;
    int *part_idx = nullptr, *history = nullptr;
    int nsplit = 0, max_parts = 4096, dec_rows = 0;
    DecodeCtx dctx{};
    std::vector<DecodeLayer> dlayers;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    bool use_graph = true;
    // sampling scratch (generate_sampled)
    float *samp_val = nullptr; int *samp_idx = nullptr; float *samp_prob = nullptr; void *sort_ws = nullptr; size_t sort_ws_bytes = 0;
    // state
    int cache_len = 0;
    float last_pos = -1.0f;
    // batched decode (mllm_hip_model_batch_*): B independent sequences, each with its own KV slabs and counters; sequence 0 is the model's own.  m->kslab / vslab /
    // cache_len / last_pos always describe the SELECTED sequence (cur_seq); the others are parked in seqs[]
    struct Seq { uint16_t *kslab = nullptr, *vslab = nullptr; int cache_len = 0; float last_pos = -1.0f; };
    std::vector<Seq> seqs;
    int cur_seq = 0;
    int gemm_min_rows = 16;      // rows from which a Linear takes the packed MFMA GEMM (its 32-row tile costs the same for 1 .. 32 rows); a batched step of B >= 4 lowers it to B
    SeqKV *seqkv_dev = nullptr;
    bool needs_arm = false;      // a batched step moved the selected sequence on: the fused decode step's device state is re-armed before its next use
    float *blogits = nullptr, *bnormed = nullptr; int8_t *bx80_qs = nullptr; uint16_t *bx80_d = nullptr; int *btok = nullptr; int batch_cap = 0;
    int64_t decode_weight_bytes = 0, resident_bytes = 0, released_bytes = 0;      // device bytes held after the load / raw rows and packs not kept (load_linear_q4k)
    float load_total_ms = 0, load_h2d_ms = 0, load_tail_ms = 0;

    template <typename T> int dalloc(T **p, size_t n, std::vector<void *> *list = nullptr) {
        void *q = nullptr;
        MH_CHECK(hipMalloc(&q, n ? n : 16));
        (list ? *list : allocs).push_back(q);
        if (list != &temps) resident_bytes += (int64_t)n;
        *p = (T *)q;
        return 0;
    }
};

typedef mllm_hip_model M;

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
    return m->ld.put(*out, f.base + e->off, count * 4);
}

// rows of several Q4_K Linear weights (same K) concatenated; biases concatenated.  `forms` says which device copies the Linear's callers read:
//   LW_RAW     the rows as stored on disk: the GEMV of passes with fewer than 16 rows (lin()), dec_gateup_blk / dec_proj_blk, the Linear lm_head
//   LW_PACK    the GEMM order of mllm_hip_q4k_prepack: passes of 16 rows and more (prefill, the vision tower)
//   LW_DECODE  the class-per-lane decode order (decode_order_q4k) of the fused decode kernels
// A form nobody reads is not kept: the lm_head only ever meets one row (no LW_PACK), a tower block never fewer than 16 (its raw rows are released once packed).
enum { LW_RAW = 1, LW_PACK = 2, LW_DECODE = 4 };
static int load_linear_q4k(M *m, const MllmFile &f, const std::vector<std::string> &names, const std::vector<int> &Ns, int K, bool bias, int forms, LinearW *lw) {
    int N = 0;
    for (int n : Ns) N += n;
    const size_t row = (size_t)K / 256 * 144;
    uint8_t *w;
    const bool keep_raw = (forms & LW_RAW) != 0;
    EH(m->dalloc(&w, row * N, keep_raw ? nullptr : &m->temps));      // temps: freed when the load-time repacks have run
    if (!keep_raw) m->released_bytes += (int64_t)(row * N);
    size_t ro = 0;
    for (size_t i = 0; i < names.size(); ++i) {
        const Entry *e;
        EH(need(f, names[i] + ".weight", &e, MLLM_HIP_Q4_K, row * Ns[i]));
        EH(m->ld.put(w + ro * row, f.base + e->off, row * Ns[i]));
        ro += Ns[i];
    }
    lw->w = keep_raw ? w : nullptr; lw->N = N; lw->K = K;
    if (bias) {
        EH(m->dalloc(&lw->bias, (size_t)N * 4));
        size_t bo = 0;
        for (size_t i = 0; i < names.size(); ++i) {
            const Entry *e;
            EH(need(f, names[i] + ".bias", &e, MLLM_HIP_F32, (uint64_t)Ns[i] * 4));
            EH(m->ld.put(lw->bias + bo, f.base + e->off, (size_t)Ns[i] * 4));
            bo += Ns[i];
        }
    }
    // load-time repacks on the compute stream, behind the copies of this tensor; the next tensor's copies run meanwhile
    EH(m->ld.fence(m->st));
    if (forms & LW_PACK) {
        uint8_t *wp;
        EH(m->dalloc(&wp, mllm_hip_q4k_wpack_bytes(N, K)));
        EH(mllm_hip_q4k_prepack(w, N, K, wp, m->st));
        lw->wp = wp;
    } else {
        m->released_bytes += (int64_t)mllm_hip_q4k_wpack_bytes(N, K);      // the GEMM-order copy that is not made
    }
    if (forms & LW_DECODE) {
        const int64_t nblk = (int64_t)N * (K / 256);
        uint8_t *wd = nullptr;
        EH(m->dalloc(&wd, (size_t)nblk * 144));
        EH(decode_order_q4k(w, wd, nblk, m->st));
        lw->wd = wd;
    }
    return 0;
}

static int alloc_q8(M *m, Q8Planes *p, int M_, int K, std::vector<void *> *list = nullptr) {
    EH(m->dalloc(&p->qs, (size_t)M_ * K, list));
    EH(m->dalloc(&p->d, (size_t)M_ * (K / 256) * 4, list));
    EH(m->dalloc(&p->bs, (size_t)M_ * (K / 16) * 2, list));
    return 0;
}

static int load_vblock(M *m, const MllmFile &f, M::VBlock &B, const std::string &n1, const std::string &n2, const std::vector<std::string> &qkv,
                       const std::string &proj, const std::string &fc1, const std::string &fc2, int V, int F) {
    EH(load_f32(m, f, n1 + ".weight", V, &B.n1w)); EH(load_f32(m, f, n1 + ".bias", V, &B.n1b));
    EH(load_f32(m, f, n2 + ".weight", V, &B.n2w)); EH(load_f32(m, f, n2 + ".bias", V, &B.n2b));
    if (qkv.size() == 1) EH(load_linear_q4k(m, f, qkv, {3 * V}, V, true, LW_PACK, &B.qkv));
    else EH(load_linear_q4k(m, f, qkv, {V, V, V}, V, true, LW_PACK, &B.qkv));
    EH(load_linear_q4k(m, f, {proj}, {V}, V, true, LW_PACK, &B.proj));
    EH(load_linear_q4k(m, f, {fc1}, {F}, V, true, LW_PACK, &B.fc1));
    EH(load_linear_q4k(m, f, {fc2}, {V}, F, true, LW_PACK, &B.fc2));
    return 0;
}

static int create_impl(M *m, const MllmFile &f) {
    const auto &c = m->c;
    const int H = c.hidden, I = c.inter;
    // ---- LLM ----
    if (m->has_llm) {
        const std::string &P = m->prefix;
        {
            const Entry *e;
            const uint64_t nblk = (uint64_t)c.vocab * (H / 32);
            EH(need(f, P + "model.embed_tokens.weight", &e, MLLM_HIP_Q4_0, nblk * 18));
            uint8_t *raw;
            EH(m->dalloc(&raw, nblk * 18, &m->temps));
            EH(m->ld.put(raw, f.base + e->off, nblk * 18));
            EH(m->dalloc(&m->emb_qs, nblk * 16));
            EH(m->dalloc(&m->emb_d, nblk * 2));
            EH(m->ld.fence(m->st));
            EH(mllm_hip_repack_q40(raw, m->emb_qs, m->emb_d, (int64_t)nblk, m->st));
        }
        m->layers.resize(c.layers);
        for (int i = 0; i < c.layers; ++i) {
            auto &L = m->layers[i];
            const std::string p = P + "model.layers." + std::to_string(i) + ".";
            EH(load_f32(m, f, p + "input_layernorm.weight", H, &L.in_norm));
            EH(load_f32(m, f, p + "post_attention_layernorm.weight", H, &L.post_norm));
            EH(load_linear_q4k(m, f, {p + "self_attn.q_proj", p + "self_attn.k_proj", p + "self_attn.v_proj"}, {m->HD, m->KVD, m->KVD}, H, c.qkv_bias != 0, LW_RAW | LW_PACK | LW_DECODE, &L.qkv));
            EH(load_linear_q4k(m, f, {p + "self_attn.o_proj"}, {H}, m->HD, false, LW_RAW | LW_PACK | LW_DECODE, &L.o));
            EH(load_linear_q4k(m, f, {p + "mlp.gate_proj", p + "mlp.up_proj"}, {I, I}, H, false, LW_RAW | LW_PACK | LW_DECODE, &L.gu));
            EH(load_linear_q4k(m, f, {p + "mlp.down_proj"}, {H}, I, false, LW_RAW | LW_PACK | LW_DECODE, &L.down));
        }
        EH(load_f32(m, f, P + "model.norm.weight", H, &m->final_norm));
        m->decode_weight_bytes = (int64_t)c.layers * ((int64_t)(m->QKV + H) * (H / 256) * 144 + (int64_t)2 * I * (H / 256) * 144 + (int64_t)H * (I / 256) * 144);
        if (c.tie_embedding) {
            m->decode_weight_bytes += (int64_t)c.vocab * (H / 32) * 18;
        } else {
            EH(load_linear_q4k(m, f, {P + "lm_head"}, {c.vocab}, H, false, LW_RAW | LW_DECODE, &m->head));
            m->decode_weight_bytes += (int64_t)c.vocab * (H / 256) * 144;
        }
    }
    // ---- vision ----
    if (m->vkind != V_NONE) {
        const int V = c.v_dim;
        if (m->vkind == V_QWEN2VL) {
            const int VM = 4 * V, PE = 3 * 2 * c.v_patch * c.v_patch, MM = V * c.v_merge * c.v_merge;
            EH(load_f32(m, f, "visual.patch_embed.proj.weight", (size_t)V * PE, &m->patch_w));
            m->vblocks.resize(c.v_blocks);
            for (int i = 0; i < c.v_blocks; ++i) {
                const std::string p = "visual.blocks." + std::to_string(i) + ".";
                EH(load_vblock(m, f, m->vblocks[i], p + "norm1", p + "norm2", {p + "attn.qkv"}, p + "attn.proj", p + "mlp.fc1", p + "mlp.fc2", V, VM));
            }
            EH(load_f32(m, f, "visual.merger.ln_q.weight", V, &m->lnq_w)); EH(load_f32(m, f, "visual.merger.ln_q.bias", V, &m->lnq_b));
            EH(load_linear_q4k(m, f, {"visual.merger.mlp.0"}, {MM}, MM, true, LW_RAW | LW_PACK, &m->m0));
            EH(load_linear_q4k(m, f, {"visual.merger.mlp.2"}, {H}, MM, true, LW_RAW | LW_PACK, &m->m2));
        } else {
            const bool clip = m->vkind == V_CLIP;
            const int F = c.v_ffn, KK = 3 * c.v_patch * c.v_patch, NP = (c.v_img / c.v_patch) * (c.v_img / c.v_patch) + 1;
            // names: ViTNameConfig "clip" / "vit" (models/vit/configuration_vit.hpp:27-64)
            const std::string base = clip ? "vision_tower.vision_model." : "vit.", e = base + "embeddings.";
            EH(load_f32(m, f, e + (clip ? "patch_embedding.weight" : "patch_embeddings.projection.weight"), (size_t)V * KK, &m->patch_w));
            if (!clip) EH(load_f32(m, f, e + "patch_embeddings.projection.bias", V, &m->patch_b));
            EH(load_f32(m, f, e + (clip ? "class_embedding" : "cls_token"), V, &m->cls_tok));
            EH(load_f32(m, f, e + (clip ? "position_embedding.weight" : "position_embeddings"), (size_t)NP * V, &m->pos_emb));
            if (clip) { EH(load_f32(m, f, base + "pre_layrnorm.weight", V, &m->pre_w)); EH(load_f32(m, f, base + "pre_layrnorm.bias", V, &m->pre_b)); }
            m->vblocks.resize(c.v_blocks);
            for (int i = 0; i < c.v_blocks; ++i) {
                if (clip) {
                    const std::string p = base + "encoder.layers." + std::to_string(i) + ".";
                    EH(load_vblock(m, f, m->vblocks[i], p + "layer_norm1", p + "layer_norm2", {p + "self_attn.q_proj", p + "self_attn.k_proj", p + "self_attn.v_proj"},
                                   p + "self_attn.out_proj", p + "mlp.fc1", p + "mlp.fc2", V, F));
                } else {
                    const std::string p = base + "encoder.layer." + std::to_string(i) + ".";
                    EH(load_vblock(m, f, m->vblocks[i], p + "layernorm_before", p + "layernorm_after",
                                   {p + "attention.attention.query", p + "attention.attention.key", p + "attention.attention.value"}, p + "attention.output.dense",
                                   p + "intermediate.dense", p + "output.dense", V, F));
                }
            }
            if (clip) {
                EH(load_linear_q4k(m, f, {"multi_modal_projector.linear_1"}, {F}, V, true, LW_RAW | LW_PACK, &m->m0));
                EH(load_linear_q4k(m, f, {"multi_modal_projector.linear_2"}, {F}, F, true, LW_RAW | LW_PACK, &m->m2));
            } else {
                EH(load_f32(m, f, "vit.layernorm.weight", V, &m->lnq_w)); EH(load_f32(m, f, "vit.layernorm.bias", V, &m->lnq_b));
                EH(load_linear_q4k(m, f, {"classifier"}, {c.v_classes}, V, false, LW_RAW | LW_PACK, &m->m0));
            }
        }
        std::vector<uint16_t> g(65536), q(65536);
        mllm_hip_build_act_luts(g.data(), q.data());
        EH(m->dalloc(&m->lut_gelu, 65536 * 2)); EH(m->dalloc(&m->lut_qgelu, 65536 * 2));
        HH(hipMemcpy(m->lut_gelu, g.data(), 65536 * 2, hipMemcpyHostToDevice)); HH(hipMemcpy(m->lut_qgelu, q.data(), 65536 * 2, hipMemcpyHostToDevice));
        for (int b = 0; b < 2; ++b) { HH(hipEventCreateWithFlags(&m->vup[b], hipEventDisableTiming)); HH(hipEventCreateWithFlags(&m->vfree[b], hipEventDisableTiming)); }
    }
    if (!m->has_llm) return 0;
    // ---- LLM activations ----
    const int T = c.cache_limit;
    m->max_tok = T;
    EH(m->dalloc(&m->h0, (size_t)T * H * 4)); EH(m->dalloc(&m->h1, (size_t)T * H * 4));
    EH(m->dalloc(&m->qkv, (size_t)T * m->QKV * 4)); EH(m->dalloc(&m->attn, (size_t)T * m->HD * 4));
    EH(m->dalloc(&m->gu, (size_t)T * 2 * I * 4)); EH(m->dalloc(&m->act, (size_t)T * I * 4));
    EH(m->dalloc(&m->logits, (size_t)c.vocab * 4)); EH(m->dalloc(&m->normed, (size_t)H * 4));
    EH(m->dalloc(&m->ids_f, (size_t)T * 4)); EH(m->dalloc(&m->idx_i, (size_t)T * 4)); EH(m->dalloc(&m->tok_dev, 16));
    EH(alloc_q8(m, &m->xq, T, H > m->HD ? H : m->HD)); EH(alloc_q8(m, &m->xq2, T, I));
    EH(m->dalloc(&m->x80_qs, (size_t)H)); EH(m->dalloc(&m->x80_d, (size_t)(H / 32) * 2));
    EH(m->dalloc(&m->rope_sin, (size_t)T * (m->D / 2) * 4)); EH(m->dalloc(&m->rope_cos, (size_t)T * (m->D / 2) * 4));
    if (!m->mrope) {
        // CPURoPE's static table (CPURoPE.cpp:100-128) for every position the cache can hold; the two halves of a row are equal, half is kept
        std::vector<float> s((size_t)T * m->D), co((size_t)T * m->D);
        EH(mllm_hip_rope_table_hf(c.rope_theta, m->D, T, s.data(), co.data()));
        const int half = m->D / 2;
        m->hf_sin.resize((size_t)T * half); m->hf_cos.resize((size_t)T * half);
        for (int p = 0; p < T; ++p)
            for (int i = 0; i < half; ++i) { m->hf_sin[(size_t)p * half + i] = s[(size_t)p * m->D + i]; m->hf_cos[(size_t)p * half + i] = co[(size_t)p * m->D + i]; }
    }
    // K slab [layers][T][KVD] (the reference's BSHD cache rows; + 64 rows: the decode attention reads whole key splits speculatively); V slab
    // transposed [layers][KVD][vt_ld], vt_ld = T rounded up + 128 keys of zero padding (the decode walk over-reads whole 16-byte vectors)
    m->vt_ld = ((T + 63) & ~63) + 128;
    EH(m->dalloc(&m->kslab, ((size_t)c.layers * T + 64) * m->KVD * 2)); EH(m->dalloc(&m->vslab, (size_t)c.layers * m->KVD * m->vt_ld * 2));
    HH(hipMemsetAsync(m->kslab, 0, ((size_t)c.layers * T + 64) * m->KVD * 2, m->st));
    HH(hipMemsetAsync(m->vslab, 0, (size_t)c.layers * m->KVD * m->vt_ld * 2, m->st));
    m->nsplit = (T + 63) / 64;
    {
        size_t wsb = mllm_hip_fa2_workspace_bytes(1, c.heads, m->D, T), wsd = (size_t)c.heads * m->nsplit * 136 * 4;
        wsb = std::max(std::max(wsb, wsd), (size_t)m->HD * 4);
        EH(m->dalloc((uint8_t **)&m->fa_ws, wsb));
    }
    EH(m->dalloc(&m->d_state, sizeof(DecodeState)));
    EH(m->dalloc(&m->attn_pairs, (size_t)c.layers * m->HD * 8)); EH(m->dalloc(&m->poll_err, 4)); EH(m->dalloc(&m->qkv_pairs, (size_t)c.layers * m->QKV * 8)); EH(m->dalloc(&m->x_pairs, (size_t)c.layers * H * 8));
    HH(hipMemsetAsync(m->qkv_pairs, 0xFF, (size_t)c.layers * m->QKV * 8, m->st)); HH(hipMemsetAsync(m->x_pairs, 0xFF, (size_t)c.layers * H * 8, m->st));
    HH(hipMemsetAsync(m->attn_pairs, 0xFF, (size_t)c.layers * m->HD * 8, m->st)); HH(hipMemsetAsync(m->poll_err, 0, 4, m->st));
    HH(hipHostMalloc((void **)&m->pin_err, 4)); *m->pin_err = 0;
    EH(m->dalloc(&m->dec_sin, (size_t)T * (m->D / 2) * 4)); EH(m->dalloc(&m->dec_cos, (size_t)T * (m->D / 2) * 4));
    EH(m->dalloc(&m->cur_sin, (size_t)m->D * 4)); EH(m->dalloc(&m->cur_cos, (size_t)m->D * 4));
    EH(m->dalloc(&m->part_val, (size_t)m->max_parts * 4)); EH(m->dalloc(&m->part_idx, (size_t)m->max_parts * 4));
    EH(m->dalloc(&m->history, (size_t)T * 4));
    {
        DecodeCtx &d = m->dctx;
        d.state = m->d_state; d.H = H; d.I = I; d.heads = c.heads; d.kv_heads = c.kv_heads; d.D = m->D; d.vocab = c.vocab; d.cache_limit = T;
        d.nsplit = m->nsplit; d.max_parts = m->max_parts; d.eps = c.rms_eps; d.final_eps = c.final_eps; d.emb_qs = m->emb_qs; d.emb_d = m->emb_d; d.final_norm = m->final_norm;
        d.Whead = (const uint8_t *)m->head.wd;
        d.x0 = m->h0; d.x1 = m->h1; d.qkv = m->qkv; d.act = m->act; d.logits = m->logits; d.fa_ws = (float *)m->fa_ws; d.part_val = m->part_val;
        d.part_idx = m->part_idx; d.tok_dev = m->tok_dev; d.history = m->history; d.rope_sin = m->dec_sin; d.rope_cos = m->dec_cos; d.cur_sin = m->cur_sin; d.cur_cos = m->cur_cos;
        d.attn_pairs = m->attn_pairs; d.qkv_pairs = m->qkv_pairs; d.x_pairs = m->x_pairs; d.poll_err = m->poll_err; d.merge_o = option(OPT_MERGE_O) < 0 || option(OPT_MERGE_O) > 4 ? 4 : option(OPT_MERGE_O);      // 4 (default): a layer's down projection + the next layer's q|k|v + attention + o-projection in one launch; 3: q|k|v + attention + o-projection; 2: attention + o-projection; 1: that with two rows per wave; 0: five launches per layer      // 2 (default): one row per wave of the projection role; 1: two (the stand-alone kernel's split)      // default on; option "merge_o" = 0 keeps the two launches      // as it stood when the model was created (the captured graph holds the choice)
        d.kslab = m->kslab; d.vslab = m->vslab; d.vt_ld = m->vt_ld; d.n_layers = c.layers; d.normed = m->normed; d.x80_qs = m->x80_qs; d.x80_d = m->x80_d;
        for (auto &L : m->layers) {
            DecodeLayer dl;
            dl.in_norm = L.in_norm; dl.post_norm = L.post_norm; dl.Wqkv = (const uint8_t *)L.qkv.wd; dl.bqkv = L.qkv.bias; dl.qkv_N = L.qkv.N;
            dl.Wo = (const uint8_t *)L.o.wd; dl.Wgu = (const uint8_t *)L.gu.wd; dl.Wdown = (const uint8_t *)L.down.wd;
            dl.Wgu_raw = (const uint8_t *)L.gu.w; dl.Wdown_raw = (const uint8_t *)L.down.w; dl.Wo_raw = (const uint8_t *)L.o.w;
            m->dlayers.push_back(dl);
        }
        {      // the attention launch's weight-warming regions, one entry per layer (read when the model is created: the captured graph holds the pointer)
            std::vector<WeightWarm> tab(m->dlayers.size());
            const int fl = decode_attn_flags();
            d.attn_flags = fl;
            d.warm_tab = nullptr;
            // (the o-projection's rows need no warming when its workgroups ride in the attention's launch and fetch them there)
            if ((fl & 1) && !(fl & 4) && decode_warm_table(d, m->dlayers.data(), (int)m->dlayers.size(), decode_merges_o(d) ? fl & ~64 : fl, tab.data()) > 0) {
                WeightWarm *dev = nullptr;
                EH(m->dalloc(&dev, tab.size() * sizeof(WeightWarm)));
                HH(hipMemcpy(dev, tab.data(), tab.size() * sizeof(WeightWarm), hipMemcpyHostToDevice));
                d.warm_tab = dev;
            }
        }
        m->use_graph = getenv("MLLM_HIP_NO_GRAPH") == nullptr;
    }
    return 0;
}

extern "C" int mllm_hip_model_create(const mllm_hip_model_config *cfg, const char *path, mllm_hip_model **out) {
    if (!cfg || !path || !out) return MLLM_HIP_ERR_ARG;
    const auto &c0 = *cfg;
    if (c0.arch < MLLM_HIP_ARCH_QWEN2VL || c0.arch > MLLM_HIP_ARCH_VIT) return MLLM_HIP_ERR_ARG;
    const bool has_llm = c0.arch != MLLM_HIP_ARCH_VIT;
    const int vkind = c0.v_dim <= 0 ? V_NONE : (c0.arch == MLLM_HIP_ARCH_QWEN2VL ? V_QWEN2VL : (c0.arch == MLLM_HIP_ARCH_LLAVA ? V_CLIP : (c0.arch == MLLM_HIP_ARCH_VIT ? V_VIT : V_NONE)));
    // everything the kernels divide by or index with is validated here, before any file or device work (ERR_ARG: nonsense, ERR_SHAPE: unsupported)
    if (has_llm) {
        if (c0.hidden <= 0 || c0.inter <= 0 || c0.layers <= 0 || c0.heads <= 0 || c0.kv_heads <= 0 || c0.vocab <= 0 || c0.cache_limit <= 0) return MLLM_HIP_ERR_ARG;
        if (c0.hidden % c0.heads || c0.hidden % 256 || c0.inter % 256 || c0.heads % c0.kv_heads) return MLLM_HIP_ERR_SHAPE;
        const int D = c0.hidden / c0.heads;
        if (D != 64 && D != 128) return MLLM_HIP_ERR_SHAPE;                     // the decode attention is built for these head sizes
        if (c0.hidden > 6 * 2048 || c0.inter > 6 * 2048) return MLLM_HIP_ERR_SHAPE;   // GEMV row forms: at most 48 super-blocks per row
        if (c0.arch == MLLM_HIP_ARCH_QWEN2VL && c0.mrope_section[0] + c0.mrope_section[1] + c0.mrope_section[2] != D / 2) return MLLM_HIP_ERR_SHAPE;
    } else if (vkind == V_NONE) return MLLM_HIP_ERR_ARG;
    if (vkind != V_NONE) {
        if (c0.v_heads <= 0 || c0.v_blocks <= 0 || c0.v_patch <= 0) return MLLM_HIP_ERR_ARG;
        if (c0.v_dim % 256 || c0.v_dim % c0.v_heads) return MLLM_HIP_ERR_SHAPE;
        const int VD = c0.v_dim / c0.v_heads;
        if (VD != 16 && VD != 64 && VD != 80 && VD != 128) return MLLM_HIP_ERR_SHAPE;     // head sizes mllm_hip_fa2 is instantiated for
        if (vkind == V_QWEN2VL) { if (c0.v_merge <= 0) return MLLM_HIP_ERR_ARG; }
        else {
            if (c0.v_ffn <= 0 || c0.v_img <= 0 || c0.v_img % c0.v_patch || c0.v_ffn % 256) return MLLM_HIP_ERR_SHAPE;
            if (vkind == V_VIT && c0.v_classes <= 0) return MLLM_HIP_ERR_ARG;
            if (vkind == V_CLIP && c0.v_ffn != c0.hidden) return MLLM_HIP_ERR_SHAPE;   // linear_2 is v_ffn x v_ffn and feeds the text embedding rows
        }
    }
    MllmFile f;
    if (!f.open(path)) { fprintf(stderr, "mllm_hip: cannot open/parse %s (missing, truncated or corrupt index)\n", path); return MLLM_HIP_ERR_IO; }
    const auto t0 = std::chrono::steady_clock::now();
    M *m = new M();
    m->c = *cfg;
    m->has_llm = has_llm;
    // a text-only Qwen2-VL file (no visual.* tensors) loads as the language model alone, as the reference's loader would leave the tower unset
    m->vkind = (vkind == V_QWEN2VL && !f.find("visual.patch_embed.proj.weight")) ? V_NONE : vkind;
    m->mrope = c0.arch == MLLM_HIP_ARCH_QWEN2VL;
    m->prefix = c0.arch == MLLM_HIP_ARCH_LLAVA ? "language_model." : "";
    if (has_llm) {
        m->D = c0.hidden / c0.heads;
        m->HD = c0.heads * m->D;
        m->KVD = c0.kv_heads * m->D;
        m->QKV = m->HD + 2 * m->KVD;
    }
    if (hipStreamCreate(&m->st) != hipSuccess || hipEventCreate(&m->ev0) != hipSuccess || hipEventCreate(&m->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&m->mrope_host_free, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&m->pin_tok), 2 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess || m->ld.init() != 0) {
        mllm_hip_model_destroy(m); return MLLM_HIP_ERR_HIP;
    }
    int rc = create_impl(m, f);
    if (!rc) rc = m->ld.finish(&m->load_h2d_ms);
    const auto t1 = std::chrono::steady_clock::now();
    if (!rc && hipStreamSynchronize(m->st) != hipSuccess) rc = MLLM_HIP_ERR_HIP;
    const auto t2 = std::chrono::steady_clock::now();
    for (void *p : m->temps) (void)hipFree(p);
    m->temps.clear();
    m->ld.destroy();                      // the pinned staging buffers are only needed during the load
    if (rc) { mllm_hip_model_destroy(m); return rc; }
    m->load_total_ms = std::chrono::duration<float, std::milli>(t2 - t0).count();
    m->load_tail_ms = std::chrono::duration<float, std::milli>(t2 - t1).count();
    *out = m;
    return MLLM_HIP_OK;
}

extern "C" int mllm_hip_model_load_stats(const mllm_hip_model *m, float *total_ms, int64_t *file_bytes, float *h2d_ms, float *repack_ms) {
    if (!m) return MLLM_HIP_ERR_ARG;
    if (total_ms) *total_ms = m->load_total_ms;
    if (file_bytes) *file_bytes = m->ld.bytes;
    if (h2d_ms) *h2d_ms = m->load_h2d_ms;
    if (repack_ms) *repack_ms = m->load_tail_ms;     // repack work left on the compute stream after the last copy had landed (the part not overlapped)
    return MLLM_HIP_OK;
}

extern "C" int mllm_hip_model_memory_stats(const mllm_hip_model *m, int64_t *resident_bytes, int64_t *released_bytes) {
    if (!m) return MLLM_HIP_ERR_ARG;
    if (resident_bytes) *resident_bytes = m->resident_bytes;
    if (released_bytes) *released_bytes = m->released_bytes;
    return MLLM_HIP_OK;
}

extern "C" void mllm_hip_model_destroy(mllm_hip_model *m) {
    if (!m) return;
    // teardown: nothing useful can be done with a failing release, the codes are dropped on purpose
    if (m->st) (void)hipStreamSynchronize(m->st);
    if (m->graph_exec) (void)hipGraphExecDestroy(m->graph_exec);
    if (m->graph) (void)hipGraphDestroy(m->graph);
    for (void *p : m->allocs) (void)hipFree(p);
    for (void *p : m->vis_allocs) (void)hipFree(p);
    for (void *p : m->temps) (void)hipFree(p);
    if (m->xpack) (void)hipFree(m->xpack);
    if (m->sort_ws) (void)hipFree(m->sort_ws);
    if (m->pin_img) (void)hipHostFree(m->pin_img);
    m->ld.destroy();
    for (int b = 0; b < 2; ++b) { if (m->vup[b]) (void)hipEventDestroy(m->vup[b]); if (m->vfree[b]) (void)hipEventDestroy(m->vfree[b]); }
    if (m->pin_tok) (void)hipHostFree(m->pin_tok);
    if (m->pin_err) (void)hipHostFree(m->pin_err);
    if (m->ev0) (void)hipEventDestroy(m->ev0);
    if (m->ev1) (void)hipEventDestroy(m->ev1);
    if (m->mrope_host_free) (void)hipEventDestroy(m->mrope_host_free);
    if (m->st) (void)hipStreamDestroy(m->st);
    delete m;
}
extern "C" int mllm_hip_model_clear_kvcache(mllm_hip_model *m) { if (!m) return MLLM_HIP_ERR_ARG; m->cache_len = 0; m->last_pos = -1.0f; return MLLM_HIP_OK; }
extern "C" int mllm_hip_model_cache_len(const mllm_hip_model *m) { return m ? m->cache_len : -1; }
extern "C" int64_t mllm_hip_model_decode_weight_bytes(const mllm_hip_model *m) { return m ? m->decode_weight_bytes : 0; }
extern "C" void *mllm_hip_model_stream(mllm_hip_model *m) { return m ? (void *)m->st : nullptr; }
// bring-up aid (not part of include/mllm_hip.h): device pointers of the prefill activations, 0 h0, 1 h1, 2 qkv, 3 attn, 4 gate|up, 5 act
extern "C" void *mllm_hip_qwen2vl_debug_ptr(mllm_hip_model *m, int which) {
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
    if (rows >= m->gemm_min_rows) { EH(ensure_xpack(m, rows, K)); return mllm_hip_quantize_q8k_packed(x, m->xpack, rows, K, m->st); }
    return mllm_hip_quantize_q8k(x, p.qs, p.d, p.bs, rows, K, m->st);
}
// activation + quantiser in one launch for the GEMM path; the separate launches otherwise
static int q_act_quant(M *m, const float *x, float *scratch, const uint16_t *lut, const Q8Planes &p, int rows, int K) {
    if (rows >= m->gemm_min_rows) { EH(ensure_xpack(m, rows, K)); return mllm_hip_quantize_q8k_packed_act(x, lut, m->xpack, rows, K, m->st); }
    EH(mllm_hip_act_lut(x, scratch, (int64_t)rows * K, lut, m->st));
    return q_quant(m, scratch, p, rows, K);
}
static int q_silu_mul_quant(M *m, const float *gu, float *scratch, const Q8Planes &p, int rows, int I) {
    if (rows >= m->gemm_min_rows) { EH(ensure_xpack(m, rows, I)); return mllm_hip_quantize_q8k_packed_silu_mul(gu, m->xpack, rows, I, m->st); }
    EH(mllm_hip_silu_mul(gu, scratch, rows, I, m->st));
    return q_quant(m, scratch, p, rows, I);
}
static int q_rmsnorm(M *m, const float *x, const float *w, const Q8Planes &p, int rows, int dim, float eps) {
    if (rows >= m->gemm_min_rows) { EH(ensure_xpack(m, rows, dim)); return mllm_hip_rmsnorm_packed(x, w, nullptr, m->xpack, rows, dim, eps, 0, m->st); }
    return mllm_hip_rmsnorm(x, w, nullptr, p.qs, p.d, p.bs, rows, dim, eps, 0, m->st);
}
static int q_layernorm(M *m, const float *x, const float *w, const float *b, const Q8Planes &p, int rows, int dim, float eps) {
    if (rows >= m->gemm_min_rows) { EH(ensure_xpack(m, rows, dim)); return mllm_hip_layernorm_packed(x, w, b, nullptr, m->xpack, rows, dim, eps, m->st); }
    return mllm_hip_layernorm(x, w, b, nullptr, p.qs, p.d, p.bs, rows, dim, eps, m->st);
}
// Linear on the activations the last q_* call produced
static int lin(M *m, const LinearW &w, const Q8Planes &x, void *y, int ydt, int64_t ldy, const float *res, int Mrows) {
    if (Mrows >= m->gemm_min_rows ? !w.wp : !w.w) { fprintf(stderr, "mllm_hip: a Linear met %d rows, a form it was not loaded for\n", Mrows); return MLLM_HIP_ERR_SHAPE; }
    if (Mrows >= m->gemm_min_rows) return mllm_hip_linear_q4kp_packed(w.wp, w.bias, m->xpack, y, ydt, ldy, res, Mrows, w.N, w.K, m->st);
    return mllm_hip_linear_q4k_q8k(w.w, w.bias, x.qs, x.d, x.bs, y, ydt, ldy, res, Mrows, w.N, w.K, m->st);
}

// tokens / output rows of one image's tower for this config
static void vision_dims(const M *m, const int32_t *meta, int *n_tok, int *out_rows, int *out_cols, size_t *img_elems) {
    const auto &c = m->c;
    if (m->vkind == V_QWEN2VL) {
        const int N = meta[0] * meta[1] * meta[2];
        *n_tok = N; *out_rows = N / (c.v_merge * c.v_merge); *out_cols = c.hidden; *img_elems = (size_t)N * 3 * 2 * c.v_patch * c.v_patch;
    } else {
        const int g = c.v_img / c.v_patch;
        *n_tok = g * g + 1; *img_elems = (size_t)c.v_img * 3 * c.v_img;
        if (m->vkind == V_CLIP) { *out_rows = g * g; *out_cols = c.v_ffn; } else { *out_rows = 1; *out_cols = c.v_classes; }
    }
}
extern "C" int mllm_hip_model_vision_shape(const mllm_hip_model *m, const int32_t *image_meta, int *rows, int *cols) {
    if (!m || m->vkind == V_NONE || (m->vkind == V_QWEN2VL && !image_meta)) return MLLM_HIP_ERR_ARG;
    int nt, r, cc; size_t ie;
    vision_dims(m, image_meta, &nt, &r, &cc, &ie);
    if (rows) *rows = r;
    if (cols) *cols = cc;
    return MLLM_HIP_OK;
}

// VisionBlock (modeling_qwen2_vl.hpp:116-140) / ViTBlock (modeling_vit.hpp:31-61): LN -> qkv -> [2-D rotary] -> FlashAttention2 (non-causal, fp32 K/V)
// -> proj + x -> LN -> fc1 -> act (fp16 LUT) -> fc2 + residual.  x in m->vx (N rows), result back in m->vx.
// NB images of N tokens each sit row after row in m->vx: everything row-wise (LayerNorm, quantiser, Linear, activation) runs ONCE over the NB * N rows -- a row's
// result does not depend on how many rows share its launch, and the GEMMs of a small image (197 rows for ViT-B/16) otherwise leave most CUs idle -- while the
// rotary and the attention, which see one image at a time, run per image.
static int vision_blocks(M *m, int N, int NB, float ln_eps, const uint16_t *lut, bool rope) {
    const auto &c = m->c;
    const int V = c.v_dim, VD = V / c.v_heads, R = N * NB;
    hipStream_t st = m->st;
    float *x = m->vx, *r = m->vr;
    for (auto &B : m->vblocks) {
        const int F = B.fc1.N;
        EH(q_layernorm(m, x, B.n1w, B.n1b, m->vxq, R, V, ln_eps));
        EH(lin(m, B.qkv, m->vxq, m->vqkv, MLLM_HIP_F32, 3 * V, nullptr, R));
        // q and k are columns [0, 2V) of the same rows: one launch over 2 * heads heads rotates both in place (per image: the table rows are its token positions)
        if (rope) EH(rope_apply_periodic(m->vqkv, 3 * V, m->vsin, m->vcos, VD / 2, R, N, 2 * c.v_heads, VD, st));      // every image of the pass: row r takes table row r % N
        if (N >= 4) {
            EH(mllm_hip_fa2_batch(m->vqkv, 3 * V, m->vqkv + V, 3 * V, m->vqkv + 2 * V, 3 * V, MLLM_HIP_F32, m->vattn, V, N, N, c.v_heads, c.v_heads, VD, 0, NB,
                                  (int64_t)N * 3 * V, (int64_t)N * 3 * V, (int64_t)N * 3 * V, (int64_t)N * V, st));
        } else {
            for (int b = 0; b < NB; ++b) {
                float *qkv = m->vqkv + (size_t)b * N * 3 * V;
                EH(mllm_hip_fa2(qkv, 3 * V, qkv + V, 3 * V, qkv + 2 * V, 3 * V, MLLM_HIP_F32, m->vattn + (size_t)b * N * V, V, N, N, c.v_heads, c.v_heads, VD, 0, nullptr,
                                nullptr, st));
            }
        }
        EH(q_quant(m, m->vattn, m->vxq, R, V));
        EH(lin(m, B.proj, m->vxq, r, MLLM_HIP_F32, V, x, R));                       // residual = proj(attn) + x
        EH(q_layernorm(m, r, B.n2w, B.n2b, m->vxq, R, V, ln_eps));
        EH(lin(m, B.fc1, m->vxq, m->vfc, MLLM_HIP_F32, F, nullptr, R));
        EH(q_act_quant(m, m->vfc, m->vact, lut, m->vxq2, R, F));
        EH(lin(m, B.fc2, m->vxq2, x, MLLM_HIP_F32, V, r, R));                        // x = fc2(act) + residual
    }
    return 0;
}

// NB images of the same geometry already resident one after the other in `pix` (QWEN2VL: [N][PE] patches each; CLIP / VIT: [H][C][W] each); writes the tower's output
// rows, image after image, to `out` (device)
static int forward_vision(M *m, const float *pix, const int32_t *meta, float *out, int NB = 1) {
    const auto &c = m->c;
    const int V = c.v_dim;
    hipStream_t st = m->st;
    if (m->vkind == V_QWEN2VL) {
        const int N = meta[0] * meta[1] * meta[2], PE = 3 * 2 * c.v_patch * c.v_patch;
        const int VD = V / c.v_heads, MM = V * c.v_merge * c.v_merge, NT = NB * (N / (c.v_merge * c.v_merge)), R = NB * N;
        // the tower's Linears are resident in the packed (M >= 16) form only: a pass needs 16 patch rows.  The reference's default min_pixels (56 x 56 = a 4 x 4 grid,
        // processing_qwen2_vl.hpp:84-109) never makes fewer; the C ABI takes any grid, so say it here instead of failing inside the first block
        if (meta[0] <= 0 || meta[1] <= 0 || meta[2] <= 0 || meta[1] % c.v_merge || meta[2] % c.v_merge || R < 16) {
            set_error_msg("mllm_hip vision tower: grid_thw = (%d, %d, %d) x %d image(s): needs positive multiples of the merge size %d and at least 16 patches per pass", meta[0], meta[1],
                          meta[2], NB, c.v_merge);
            return MLLM_HIP_ERR_SHAPE;
        }
        {   // rotary tables (CPUVisionRoPE): rot_dim = head_dim/2
            // a grid's tables are a pure function of (t, h, w): kept on the device for the grid they were made for (a serving loop sees the same few grids again and again)
            if (meta[0] != m->vrope_grid[0] || meta[1] != m->vrope_grid[1] || meta[2] != m->vrope_grid[2]) {
                std::vector<float> s((size_t)N * (VD / 2)), co((size_t)N * (VD / 2));
                EH(mllm_hip_vision_rope_table(meta[0], meta[1], meta[2], c.v_merge, VD / 2, s.data(), co.data()));
                HH(hipMemcpyAsync(m->vsin, s.data(), s.size() * 4, hipMemcpyHostToDevice, st));
                HH(hipMemcpyAsync(m->vcos, co.data(), co.size() * 4, hipMemcpyHostToDevice, st));
                HH(hipStreamSynchronize(st));
                for (int i = 0; i < 3; ++i) m->vrope_grid[i] = meta[i];
            }
        }
        EH(mllm_hip_patch_gemm_f32(pix, m->patch_w, nullptr, m->vx, R, PE, V, st));
        EH(vision_blocks(m, N, NB, 1e-6f, m->lut_qgelu, true));
        // PatchMerger: ln_q -> view [NT][MM] -> mlp.0 -> GELU -> mlp.2
        EH(mllm_hip_layernorm(m->vx, m->lnq_w, m->lnq_b, m->vr, nullptr, nullptr, nullptr, R, V, 1e-6f, st));
        EH(q_quant(m, m->vr, m->vxq2, NT, MM));
        EH(lin(m, m->m0, m->vxq2, m->vm0, MLLM_HIP_F32, MM, nullptr, NT));
        EH(q_act_quant(m, m->vm0, m->vfc, m->lut_gelu, m->vxq2, NT, MM));
        EH(lin(m, m->m2, m->vxq2, out, MLLM_HIP_F32, c.hidden, nullptr, NT));
        return 0;
    }
    // LLaVAVisionEmbedding (modeling_llava.hpp:39-60) / ViTEmbedding (modeling_vit.hpp:63-83): Conv2D patch embedding as im2patch + GEMM over the
    // receptive fields; row 0 = the class token; + position rows (an Embedding over 0..N-1 / a Parameter: both a plain row-wise add)
    const int g = c.v_img / c.v_patch, NP = g * g, N = NP + 1, KK = 3 * c.v_patch * c.v_patch, F = c.v_ffn, R = NB * N;
    const size_t ie = (size_t)c.v_img * 3 * c.v_img;
    for (int b = 0; b < NB; ++b) {
        float *vp = m->vpatch + (size_t)b * NP * KK, *rows = m->vr + (size_t)b * N * V;
        EH(mllm_hip_im2patch_hcw(pix + b * ie, vp, c.v_img, 3, c.v_img, c.v_patch, st));
        EH(mllm_hip_patch_gemm_f32(vp, m->patch_w, m->patch_b, rows + V, NP, KK, V, st));
        HH(hipMemcpyAsync(rows, m->cls_tok, (size_t)V * 4, hipMemcpyDeviceToDevice, st));
        EH(mllm_hip_add(m->pos_emb, rows, m->vx + (size_t)b * N * V, (int64_t)N * V, st));
    }
    if (m->vkind == V_CLIP) {
        EH(mllm_hip_layernorm(m->vx, m->pre_w, m->pre_b, m->vr, nullptr, nullptr, nullptr, R, V, 1e-6f, st));       // pre_layrnorm (:69)
        HH(hipMemcpyAsync(m->vx, m->vr, (size_t)R * V * 4, hipMemcpyDeviceToDevice, st));
        EH(vision_blocks(m, N, NB, 1e-5f, m->lut_qgelu, false));
        // clip off the class row (:90), multi_modal_projector: linear_1 -> GELU -> linear_2
        const float *patches = m->vx + V;
        if (NB > 1) {     // the patch rows of the images, packed together
            for (int b = 0; b < NB; ++b)
                HH(hipMemcpyAsync(m->vattn + (size_t)b * NP * V, m->vx + ((size_t)b * N + 1) * V, (size_t)NP * V * 4, hipMemcpyDeviceToDevice, st));
            patches = m->vattn;
        }
        EH(q_quant(m, patches, m->vxq, NB * NP, V));
        EH(lin(m, m->m0, m->vxq, m->vfc, MLLM_HIP_F32, F, nullptr, NB * NP));
        EH(q_act_quant(m, m->vfc, m->vact, m->lut_gelu, m->vxq2, NB * NP, F));
        EH(lin(m, m->m2, m->vxq2, out, MLLM_HIP_F32, F, nullptr, NB * NP));
        return 0;
    }
    // VIT: the class row alone goes on (clip({0}), modeling_vit.hpp:106): LayerNorm 1e-6, classifier without bias
    EH(vision_blocks(m, N, NB, 1e-5f, m->lut_gelu, false));
    const float *cls = m->vx;
    if (NB > 1) {
        for (int b = 0; b < NB; ++b) HH(hipMemcpyAsync(m->vattn + (size_t)b * V, m->vx + (size_t)b * N * V, (size_t)V * 4, hipMemcpyDeviceToDevice, st));
        cls = m->vattn;
    }
    EH(q_layernorm(m, cls, m->lnq_w, m->lnq_b, m->vxq, NB, V, 1e-6f));
    EH(lin(m, m->m0, m->vxq, out, MLLM_HIP_F32, c.v_classes, nullptr, NB));
    return 0;
}

static int ensure_vision_buffers(M *m, const int32_t *meta, int NB = 1) {
    const auto &c = m->c;
    int N1, orows1, ocols; size_t img_elems1;
    vision_dims(m, meta, &N1, &orows1, &ocols, &img_elems1);
    if (N1 <= m->max_patch && NB <= m->vis_batch) return 0;
    NB = std::max(NB, m->vis_batch);
    const int Ntok = std::max(N1, m->max_patch);                               // tokens of the largest image so far
    const int N = Ntok * NB;                                                   // rows of one pass
    // (only the Qwen2-VL tower sees images of different sizes: its output rows and patch elements scale with the tokens)
    const int orows = (m->vkind == V_QWEN2VL ? Ntok / (c.v_merge * c.v_merge) : orows1) * NB;
    const size_t img_elems = (m->vkind == V_QWEN2VL ? (size_t)Ntok * 3 * 2 * c.v_patch * c.v_patch : img_elems1) * NB;
    // a larger image than any before: release the previous set (nothing of it is in flight after the sync) and size a new one
    HH(hipStreamSynchronize(m->st));
    for (void *p : m->vis_allocs) HH(hipFree(p));
    m->vis_allocs.clear();
    m->max_patch = 0; m->vis_batch = 0;
    auto *L = &m->vis_allocs;
    const int V = c.v_dim;
    const int F = m->vkind == V_QWEN2VL ? 4 * V : c.v_ffn, MM = m->vkind == V_QWEN2VL ? V * c.v_merge * c.v_merge : 0;
    const int VD = V / c.v_heads;
    const int Fw = std::max(F, MM);
    EH(m->dalloc(&m->vx, (size_t)N * V * 4, L)); EH(m->dalloc(&m->vr, (size_t)N * V * 4, L));
    EH(m->dalloc(&m->vqkv, (size_t)N * 3 * V * 4, L)); EH(m->dalloc(&m->vattn, (size_t)N * V * 4, L));
    EH(m->dalloc(&m->vfc, (size_t)N * Fw * 4, L)); EH(m->dalloc(&m->vact, (size_t)N * Fw * 4, L));
    EH(m->dalloc(&m->vpix[0], img_elems * 4, L)); EH(m->dalloc(&m->vpix[1], img_elems * 4, L));
    EH(m->dalloc(&m->vemb, (size_t)orows * ocols * 4, L));
    if (m->vkind == V_QWEN2VL) {
        EH(m->dalloc(&m->vsin, (size_t)N * (VD / 2) * 4, L)); EH(m->dalloc(&m->vcos, (size_t)N * (VD / 2) * 4, L));
        m->vrope_grid[0] = m->vrope_grid[1] = m->vrope_grid[2] = 0;      // new blocks: the tables have to be made again
        EH(m->dalloc(&m->vm0, (size_t)orows * MM * 4, L));
    } else {
        EH(m->dalloc(&m->vpatch, (size_t)N * 3 * c.v_patch * c.v_patch * 4, L));
    }
    // Q8_K planes for fewer than 16 rows only (larger row counts go through the packed GEMM operand m->xpack); sized generously for N rows anyway
    EH(alloc_q8(m, &m->vxq, N, V, L)); EH(alloc_q8(m, &m->vxq2, N, Fw, L));
    if (img_elems * 4 > m->pin_img_bytes) {
        if (m->pin_img) HH(hipHostFree(m->pin_img));
        m->pin_img = nullptr; m->pin_img_bytes = 0;
        HH(hipHostMalloc((void **)&m->pin_img, 2 * img_elems * 4, hipHostMallocDefault));
        m->pin_img_bytes = img_elems * 4;
    }
    m->max_patch = Ntok; m->vis_batch = NB;
    return 0;
}

// One LLM forward over S new tokens whose embeddings are in m->h0 ([S][H]); logits of the last token -> m->logits.
// pos3: QWEN2VL position ids [3][S]; the HF-rotary archs take positions cache_len .. cache_len + S - 1 (CPURoPE's h_cnt_, CPURoPE.cpp:510-513)
static int forward_llm(M *m, int S, const float *pos3) {
    const auto &c = m->c;
    const int H = c.hidden, I = c.inter, D = m->D, T0 = m->cache_len, half = D / 2;
    hipStream_t st = m->st;
    if (T0 + S > c.cache_limit) { fprintf(stderr, "mllm_hip: KV cache overflow (%d + %d > %d)\n", T0, S, c.cache_limit); return MLLM_HIP_ERR_SHAPE; }
    if (m->mrope) {   // M-RoPE tables for these S positions
        // the host tables live in the model (they must outlive the copies): no synchronisation here -- this point comes right behind the vision tower's launches in an
        // image prefill, and waiting for the stream would leave the device idle while the LLM's launches are only being issued
        std::vector<float> &s = m->mrope_host_s, &co = m->mrope_host_c;
        if (!s.empty()) HH(hipEventSynchronize(m->mrope_host_free));      // the previous forward's copies out of these vectors have been made (long ago)
        s.resize((size_t)S * half); co.resize((size_t)S * half);
        EH(mllm_hip_mrope_table(c.rope_theta, D, pos3, S, c.mrope_section, 3, s.data(), co.data()));
        HH(hipMemcpyAsync(m->rope_sin, s.data(), s.size() * 4, hipMemcpyHostToDevice, st));
        HH(hipMemcpyAsync(m->rope_cos, co.data(), co.size() * 4, hipMemcpyHostToDevice, st));
        HH(hipEventRecord(m->mrope_host_free, st));
    } else {
        HH(hipMemcpyAsync(m->rope_sin, m->hf_sin.data() + (size_t)T0 * half, (size_t)S * half * 4, hipMemcpyHostToDevice, st));
        HH(hipMemcpyAsync(m->rope_cos, m->hf_cos.data() + (size_t)T0 * half, (size_t)S * half * 4, hipMemcpyHostToDevice, st));
    }
    float *h = m->h0, *h2 = m->h1;
    for (int li = 0; li < c.layers; ++li) {
        auto &L = m->layers[li];
        uint16_t *kl = m->kslab + (size_t)li * c.cache_limit * m->KVD, *vl = m->vslab + (size_t)li * m->KVD * m->vt_ld;
        EH(q_rmsnorm(m, h, L.in_norm, m->xq, S, H, c.rms_eps));
        EH(lin(m, L.qkv, m->xq, m->qkv, MLLM_HIP_F32, m->QKV, nullptr, S));
        // q_rope in place; k_rope -> fp16 slab rows [T0, T0+S); v -> fp16 slab (KVCache zero-copy append)
        EH(mllm_hip_qkv_rope_append(m->qkv, m->QKV, m->rope_sin, m->rope_cos, half, kl + (size_t)T0 * m->KVD, m->KVD, vl + T0, m->vt_ld, S, c.heads, c.kv_heads, D, st));
        EH(mllm_hip_fa2_vt(m->qkv, m->QKV, kl, m->KVD, vl, m->vt_ld, m->attn, m->HD, S, T0 + S, c.heads, c.kv_heads, D, 1, st));
        EH(q_quant(m, m->attn, m->xq, S, m->HD));
        EH(lin(m, L.o, m->xq, h2, MLLM_HIP_F32, H, h, S));                          // tmp = o_proj(attn) + x
        EH(q_rmsnorm(m, h2, L.post_norm, m->xq, S, H, c.rms_eps));
        EH(lin(m, L.gu, m->xq, m->gu, MLLM_HIP_F32, 2 * I, nullptr, S));
        EH(q_silu_mul_quant(m, m->gu, m->act, m->xq2, S, I));
        EH(lin(m, L.down, m->xq2, h, MLLM_HIP_F32, H, h2, S));                      // x = down(...) + tmp
    }
    // final norm on the last token only (norm then clip({-1}) == clip then norm; TinyLLaMAModel does not clip (modeling_tinyllama.hpp:67-75) but the demo
    // reads the last row), then the head: tied = Tensor::mm with embed_tokens^T through Q8_0 activations, else the Linear lm_head through Q8_K
    if (c.tie_embedding) {
        EH(mllm_hip_rmsnorm(h + (size_t)(S - 1) * H, m->final_norm, m->normed, nullptr, nullptr, nullptr, 1, H, c.final_eps, 0, st));
        EH(mllm_hip_quantize_q80(m->normed, m->x80_qs, m->x80_d, 1, H, st));
        EH(mllm_hip_linear_q40_q80(m->emb_qs, m->emb_d, nullptr, m->x80_qs, m->x80_d, m->logits, c.vocab, 1, c.vocab, H, st));
    } else {
        EH(mllm_hip_rmsnorm(h + (size_t)(S - 1) * H, m->final_norm, nullptr, m->xq.qs, m->xq.d, m->xq.bs, 1, H, c.final_eps, 0, st));
        EH(mllm_hip_linear_q4k_q8k(m->head.w, nullptr, m->xq.qs, m->xq.d, m->xq.bs, m->logits, MLLM_HIP_F32, c.vocab, nullptr, 1, c.vocab, H, st));
    }
    EH(argmax_row_launch(m->dctx, m->logits, c.vocab, m->tok_dev, st));
    m->cache_len = T0 + S;
    return 0;
}

static int arm_decode(M *m);

static int finish(M *m, float *logits_host, int32_t *next_token, float *elapsed_ms) {
    HH(hipEventRecord(m->ev1, m->st));
    // the copies ride the stream behind the step and ONE synchronisation covers the step and both of them
    if (logits_host) HH(hipMemcpyAsync(logits_host, m->logits, (size_t)m->c.vocab * 4, hipMemcpyDeviceToHost, m->st));
    if (next_token) HH(hipMemcpyAsync(m->pin_tok + 1, m->tok_dev, 4, hipMemcpyDeviceToHost, m->st));
    if (m->dctx.merge_o) HH(hipMemcpyAsync(m->pin_err, m->poll_err, 4, hipMemcpyDeviceToHost, m->st));
    HH(hipStreamSynchronize(m->st));
    if (m->dctx.merge_o && *m->pin_err) {      // a polled hand-off inside a merged launch gave up: the results of the step(s) are not valid
        const int site = *m->pin_err;
        *m->pin_err = 0;
        HH(hipMemset(m->poll_err, 0, 4));
        static const char *const who[] = {"?", "q|k|v role waiting for the layer input row", "o-projection role waiting for the attention's row", "attention role waiting for q|k|v"};
        set_error_msg("a merged decode launch timed out waiting for its producer workgroups (option merge_o): %s", who[site >= 1 && site <= 3 ? site : 0]);
        return MLLM_HIP_ERR_ARG;
    }
    if (next_token) *next_token = m->pin_tok[1];
    if (elapsed_ms) HH(hipEventElapsedTime(elapsed_ms, m->ev0, m->ev1));
    return 0;
}

extern "C" int mllm_hip_model_prefill(mllm_hip_model *m, const int32_t *ids, int n_ids, const float *image, const int32_t *image_meta, const float *visual_dev,
                                      int n_visual_rows, float *logits_host, int32_t *next_token, float *elapsed_ms) {
    if (!m || !m->has_llm || !ids || n_ids <= 0 || n_ids > m->c.cache_limit) return MLLM_HIP_ERR_ARG;
    const auto &c = m->c;
    const bool has_img = image != nullptr || visual_dev != nullptr;
    if (has_img && m->vkind == V_NONE) return MLLM_HIP_ERR_ARG;
    if (has_img && m->vkind == V_QWEN2VL && !image_meta) return MLLM_HIP_ERR_ARG;
    std::vector<float> idf(n_ids);
    std::vector<int> where;
    for (int i = 0; i < n_ids; ++i) { idf[i] = (float)ids[i]; if (ids[i] == c.image_token_id) where.push_back(i); }
    int vrows = 0, vcols = 0, S = n_ids;
    const float *vis = visual_dev;
    if (has_img) {
        int nt; size_t ie;
        vision_dims(m, image_meta, &nt, &vrows, &vcols, &ie);
        if (visual_dev && n_visual_rows != vrows) return MLLM_HIP_ERR_SHAPE;
        if (m->vkind == V_QWEN2VL) {
            if ((int)where.size() != vrows) { fprintf(stderr, "mllm_hip: %zu image tokens but %d visual tokens\n", where.size(), vrows); return MLLM_HIP_ERR_SHAPE; }
        } else {
            if (where.size() != 1) { fprintf(stderr, "mllm_hip: LLaVA takes exactly one <image> token per prompt, got %zu\n", where.size()); return MLLM_HIP_ERR_SHAPE; }
            S = n_ids - 1 + vrows;
            if (S > c.cache_limit) return MLLM_HIP_ERR_SHAPE;
        }
        if (!visual_dev) {
            EH(ensure_vision_buffers(m, image_meta));
            HH(hipMemcpy(m->vpix[0], image, ie * 4, hipMemcpyDefault));      // host or device pixels (mllm_hip_qwen2vl_preprocess leaves them on the device)
            vis = m->vemb;
        }
        if (m->vkind == V_QWEN2VL) HH(hipMemcpy(m->idx_i, where.data(), where.size() * 4, hipMemcpyHostToDevice));
    }
    HH(hipMemcpy(m->ids_f, idf.data(), (size_t)n_ids * 4, hipMemcpyHostToDevice));
    std::vector<float> pos;
    if (m->mrope) rope_index(m, ids, n_ids, image_meta, has_img, pos);
    // ---- timed region: inputs resident in HBM ----
    HH(hipEventRecord(m->ev0, m->st));
    if (has_img && !visual_dev) EH(forward_vision(m, m->vpix[0], image_meta, m->vemb));
    if (has_img && m->vkind == V_CLIP) {
        // text rows before the <image> row, the visual rows in its place, the text rows after it (Tensor::where + index_put, modeling_llava.hpp:128-132)
        const int at = where[0];
        EH(mllm_hip_embedding_q40(m->ids_f, m->emb_qs, m->emb_d, m->h0, at, c.hidden, c.vocab, m->st));
        HH(hipMemcpyAsync(m->h0 + (size_t)at * c.hidden, vis, (size_t)vrows * c.hidden * 4, hipMemcpyDeviceToDevice, m->st));
        EH(mllm_hip_embedding_q40(m->ids_f + at + 1, m->emb_qs, m->emb_d, m->h0 + (size_t)(at + vrows) * c.hidden, n_ids - at - 1, c.hidden, c.vocab, m->st));
    } else {
        EH(mllm_hip_embedding_q40(m->ids_f, m->emb_qs, m->emb_d, m->h0, n_ids, c.hidden, c.vocab, m->st));
        if (has_img) EH(mllm_hip_index_put_rows(m->h0, vis, m->idx_i, (int)where.size(), c.hidden, m->st));
    }
    EH(forward_llm(m, S, pos.data()));
    // the position the decode loop continues from: M-RoPE row 0 (t axis), last column (get_position_ids' decode branch); HF rotary: the token count
    m->last_pos = m->mrope ? pos[(size_t)n_ids - 1] : (float)(m->cache_len - 1);
    EH(finish(m, logits_host, next_token, elapsed_ms));
    m->needs_arm = false;
    return arm_decode(m);
}

// After a prefill: device step state {T, step = 0, token}, and the rotary rows of every position the decode loop can still reach
// (QWEN2VL: all three axes = last_pos + 1 + step, modeling_qwen2_vl.hpp:423-432; HF rotary: position = tokens in the cache).
static int arm_decode(M *m) {
    const auto &c = m->c;
    const int rows = c.cache_limit - m->cache_len, half = m->D / 2;
    m->dec_rows = rows;
    if (rows > 0) {
        if (m->mrope) {
            std::vector<float> pos((size_t)3 * rows), s((size_t)rows * half), co((size_t)rows * half);
            for (int a = 0; a < 3; ++a) for (int r = 0; r < rows; ++r) pos[(size_t)a * rows + r] = m->last_pos + 1.0f + (float)r;
            EH(mllm_hip_mrope_table(c.rope_theta, m->D, pos.data(), rows, c.mrope_section, 3, s.data(), co.data()));
            HH(hipMemcpy(m->dec_sin, s.data(), s.size() * 4, hipMemcpyHostToDevice));
            HH(hipMemcpy(m->dec_cos, co.data(), co.size() * 4, hipMemcpyHostToDevice));
        } else {
            HH(hipMemcpy(m->dec_sin, m->hf_sin.data() + (size_t)m->cache_len * half, (size_t)rows * half * 4, hipMemcpyHostToDevice));
            HH(hipMemcpy(m->dec_cos, m->hf_cos.data() + (size_t)m->cache_len * half, (size_t)rows * half * 4, hipMemcpyHostToDevice));
        }
        HH(hipMemcpy(m->cur_sin, m->dec_sin, (size_t)half * 4, hipMemcpyDeviceToDevice));
        HH(hipMemcpy(m->cur_cos, m->dec_cos, (size_t)half * 4, hipMemcpyDeviceToDevice));
    }
    int tok = 0;
    HH(hipMemcpy(&tok, m->tok_dev, 4, hipMemcpyDeviceToHost));
    DecodeState st0 = {m->cache_len, 0, tok, 0};
    HH(hipMemcpy(m->d_state, &st0, sizeof(st0), hipMemcpyHostToDevice));
    if (m->dctx.merge_o) { HH(hipMemset(m->attn_pairs, 0xFF, (size_t)m->c.layers * m->HD * 8)); HH(hipMemset(m->qkv_pairs, 0xFF, (size_t)m->c.layers * m->QKV * 8)); HH(hipMemset(m->x_pairs, 0xFF, (size_t)m->c.layers * m->c.hidden * 8)); }      // no pair of an earlier run may carry an epoch this run will count up to      // no pair of an earlier run may carry an epoch this run will count up to
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

// After a failure inside a decode loop the host's counters follow the device's DecodeState (which the steps that did run have advanced), so a caller that carries on
// does so from the state the cache is really in.  Best effort: if the device cannot be read either, the counters stay as they were.
static int resync_after_error(mllm_hip_model *m, int rc) {
    DecodeState st;
    if (hipStreamSynchronize(m->st) == hipSuccess && hipMemcpy(&st, m->d_state, sizeof(st), hipMemcpyDeviceToHost) == hipSuccess && st.T >= 0 && st.T <= m->c.cache_limit) {
        m->last_pos += (float)(st.T - m->cache_len);
        m->cache_len = st.T;
    }
    return rc;
}

extern "C" int mllm_hip_model_decode(mllm_hip_model *m, int32_t token, float *logits_host, int32_t *next_token, float *elapsed_ms) {
    if (!m || !m->has_llm || m->cache_len <= 0) return MLLM_HIP_ERR_ARG;
    if (m->cache_len + 1 > m->c.cache_limit) { fprintf(stderr, "mllm_hip: KV cache overflow (%d + 1 > %d)\n", m->cache_len, m->c.cache_limit); return MLLM_HIP_ERR_SHAPE; }
    if (m->needs_arm) { EH(arm_decode(m)); m->needs_arm = false; }
    m->pin_tok[0] = token;      // (page-locked: the copy is stream-ordered and the word is not written again before finish() has synchronised)
    HH(hipMemcpyAsync(&m->d_state->token, m->pin_tok, 4, hipMemcpyHostToDevice, m->st));
    HH(hipEventRecord(m->ev0, m->st));
    EH(launch_step(m));
    m->cache_len += 1;
    m->last_pos += 1.0f;
    return finish(m, logits_host, next_token, elapsed_ms);
}

extern "C" int mllm_hip_model_generate(mllm_hip_model *m, int32_t first_token, int steps, int32_t *tokens_host, float *elapsed_ms) {
    if (!m || !m->has_llm || m->cache_len <= 0 || steps <= 0) return MLLM_HIP_ERR_ARG;
    if (m->cache_len + steps > m->c.cache_limit) { fprintf(stderr, "mllm_hip: KV cache overflow (%d + %d > %d)\n", m->cache_len, steps, m->c.cache_limit); return MLLM_HIP_ERR_SHAPE; }
    if (m->needs_arm) { EH(arm_decode(m)); m->needs_arm = false; }
    DecodeState st0;
    HH(hipMemcpy(&st0, m->d_state, sizeof(st0), hipMemcpyDeviceToHost));
    const int step0 = st0.step;
    HH(hipMemcpy(&m->d_state->token, &first_token, 4, hipMemcpyHostToDevice));
    HH(hipEventRecord(m->ev0, m->st));
    for (int s = 0; s < steps; ++s)
        if (int rc = launch_step(m)) return resync_after_error(m, rc);
    m->cache_len += steps;
    m->last_pos += (float)steps;
    EH(finish(m, nullptr, nullptr, elapsed_ms));
    if (tokens_host) HH(hipMemcpy(tokens_host, m->history + step0, (size_t)steps * 4, hipMemcpyDeviceToHost));
    return 0;
}

// The decode step once more, launch by launch: `steps` greedy steps run eagerly (no graph) with a HIP event either side of every launch on the engine's stream, the
// event-to-event times summed per kind of launch (decode_launch.h StepMarks).  The cache advances exactly as under mllm_hip_model_generate (same tokens).  An interval holds
// the kernel and what the command processor spends between the marker and the kernel, so a kind's figure sits a little above the kernel trace's duration of the same kernel
// inside the captured graph; bench.py reports both the step's dominant launch from here and the whole token from the graph replay.
namespace {
struct StepTimer {
    hipStream_t st;
    std::vector<hipEvent_t> ev;
    std::vector<int> tag;
    size_t n = 0;
    unsigned flags = hipEventDisableSystemFence;
    ~StepTimer() { for (auto e : ev) (void)hipEventDestroy(e); }
};
int step_mark(void *user, int kind, int after) {
    auto *t = static_cast<StepTimer *>(user);
    if (t->n == t->ev.size()) {
        hipEvent_t e;
        // (no system-scope fence: a default event's marker writes the L2s back for the host between two kernels, 1.5-2.5 us per launch on top of the kernel; measured
        //  on the 2 B step: sum of the intervals 944 us with default events, 825 with these, against 792-815 for the captured graph -- scratch/time_step_flags.py)
        if (hipEventCreateWithFlags(&e, t->flags) != hipSuccess && hipEventCreate(&e) != hipSuccess) return MLLM_HIP_ERR_HIP;
        t->ev.push_back(e);
        t->tag.push_back(0);
    }
    t->tag[t->n] = kind * 2 + after;
    if (hipEventRecord(t->ev[t->n], t->st) != hipSuccess) return MLLM_HIP_ERR_HIP;
    ++t->n;
    return 0;
}
}  // namespace
extern "C" int mllm_hip_model_time_step(mllm_hip_model *m, int32_t first_token, int steps, float *us_by_kind, int32_t *launches_by_kind, int32_t *last_token) {
    if (!m || !m->has_llm || m->cache_len <= 0 || steps <= 0 || !us_by_kind || !launches_by_kind) return MLLM_HIP_ERR_ARG;
    if (m->cache_len + steps > m->c.cache_limit) { fprintf(stderr, "mllm_hip: KV cache overflow (%d + %d > %d)\n", m->cache_len, steps, m->c.cache_limit); return MLLM_HIP_ERR_SHAPE; }
    if (m->needs_arm) { EH(arm_decode(m)); m->needs_arm = false; }
    HH(hipMemcpy(&m->d_state->token, &first_token, 4, hipMemcpyHostToDevice));
    StepTimer t;
    t.st = m->st;
    StepMarks marks = {step_mark, &t};
    double us[STEP_KINDS] = {0};
    int cnt[STEP_KINDS] = {0};
    std::vector<size_t> first(steps + 1, 0);
    HH(hipEventRecord(m->ev0, m->st));
    for (int s = 0; s < steps; ++s) {      // all steps are queued before the one synchronisation: after the first step the host runs ahead of the device
        first[s] = t.n;
        if (int rc = decode_step_launch(m->dctx, m->dlayers.data(), (int)m->dlayers.size(), m->st, &marks)) return resync_after_error(m, rc);
    }
    first[steps] = t.n;
    m->cache_len += steps;
    m->last_pos += (float)steps;
    EH(finish(m, nullptr, last_token, nullptr));
    const int counted = steps > 1 ? steps - 1 : 1;      // the first step's launches wait for the host; it is left out when there is another
    for (size_t i = first[steps > 1 ? 1 : 0]; i + 1 < t.n; i += 2) {
        const int kind = t.tag[i] >> 1;
        if ((t.tag[i] & 1) || t.tag[i + 1] != kind * 2 + 1) return MLLM_HIP_ERR_ARG;      // marks come in (before, after) pairs of one kind
        float ms = 0;
        HH(hipEventElapsedTime(&ms, t.ev[i], t.ev[i + 1]));
        us[kind] += (double)ms * 1e3;
        cnt[kind] += 1;
    }
    for (int k = 0; k < STEP_KINDS; ++k) { us_by_kind[k] = cnt[k] ? (float)(us[k] / cnt[k]) : 0.0f; launches_by_kind[k] = cnt[k] / counted; }
    return 0;
}

// ---- batched decode: B independent sequences share ONE pass over the weights per step ---------------------------------------------------------------------------
// The reference's hook for this is KVCache_batch (mllm/Types.hpp:26-33: the KVCache Op's slab gets a batch dimension); its models then run [B, 1, S, H] activations
// through the same Ops.  A decode token of this engine is launch- and latency-bound, not byte-bound, so B sequences stepped together cost little more
// than one.  Rows never mix: every Op of the graph is row-wise (norms, Q8_K quantisation, the GEMV's per-row dot products, SiLU, residual adds) except attention, which
// runs per sequence on that sequence's own cache -- so row b of a batched step is bit for bit what sequence b would have produced stepping alone (tests/test_batched_decode.py).
// The step is composed from the per-Op launchers like the prefill (the M < 16 GEMV form reads each weight row once for all B rows); the fused single-sequence decode
// kernels and their captured graph stay the headline path (bench.py's `value`); `batched_decode` is reported beside it.
static void seq_park(M *m) {
    if (m->seqs.empty()) m->seqs.resize(1);
    auto &q = m->seqs[m->cur_seq];
    q.kslab = m->kslab; q.vslab = m->vslab; q.cache_len = m->cache_len; q.last_pos = m->last_pos;
}
static int seq_select(M *m, int s) {
    seq_park(m);
    if (s == m->cur_seq) return 0;
    const auto &q = m->seqs[s];
    m->kslab = q.kslab; m->vslab = q.vslab; m->cache_len = q.cache_len; m->last_pos = q.last_pos;
    m->dctx.kslab = q.kslab; m->dctx.vslab = q.vslab;
    m->cur_seq = s;
    m->needs_arm = m->cache_len > 0;
    if (m->graph_exec) {      // the captured step holds the previous sequence's slab pointers
        HH(hipStreamSynchronize(m->st));
        HH(hipGraphExecDestroy(m->graph_exec)); m->graph_exec = nullptr;
        if (m->graph) { HH(hipGraphDestroy(m->graph)); m->graph = nullptr; }
    }
    return 0;
}
extern "C" int mllm_hip_model_batch_begin(mllm_hip_model *m, int B) {
    if (!m || !m->has_llm || B < 1 || B > 15) return MLLM_HIP_ERR_ARG;      // 15: the rows of a step go through the M < 16 GEMV form
    const auto &c = m->c;
    seq_park(m);
    const size_t kbytes = ((size_t)c.layers * c.cache_limit + 64) * m->KVD * 2, vbytes = (size_t)c.layers * m->KVD * m->vt_ld * 2;
    while ((int)m->seqs.size() < B) {
        M::Seq q;
        EH(m->dalloc(&q.kslab, kbytes)); EH(m->dalloc(&q.vslab, vbytes));
        HH(hipMemsetAsync(q.kslab, 0, kbytes, m->st));
        HH(hipMemsetAsync(q.vslab, 0, vbytes, m->st));
        m->seqs.push_back(q);
    }
    if (B > m->batch_cap) {
        EH(m->dalloc(&m->blogits, (size_t)B * c.vocab * 4)); EH(m->dalloc(&m->bnormed, (size_t)B * c.hidden * 4));
        EH(m->dalloc(&m->bx80_qs, (size_t)B * c.hidden)); EH(m->dalloc(&m->bx80_d, (size_t)B * (c.hidden / 32) * 2 + 64)); EH(m->dalloc(&m->btok, (size_t)B * 4));
        EH(m->dalloc(&m->seqkv_dev, (size_t)B * sizeof(SeqKV)));
        m->batch_cap = B;
    }
    HH(hipStreamSynchronize(m->st));
    return MLLM_HIP_OK;
}
extern "C" int mllm_hip_model_batch_select(mllm_hip_model *m, int seq) {
    if (!m || !m->has_llm) return MLLM_HIP_ERR_ARG;
    seq_park(m);
    if (seq < 0 || seq >= (int)m->seqs.size()) return MLLM_HIP_ERR_ARG;
    return seq_select(m, seq);
}
extern "C" int mllm_hip_model_batch_decode(mllm_hip_model *m, int B, const int32_t *tokens, float *logits_host, int32_t *next_tokens, float *elapsed_ms) {
    if (!m || !m->has_llm || !tokens || B < 1 || B > m->batch_cap) return MLLM_HIP_ERR_ARG;
    const auto &c = m->c;
    seq_park(m);
    if ((int)m->seqs.size() < B) return MLLM_HIP_ERR_ARG;
    for (int b = 0; b < B; ++b) {
        if (m->seqs[b].cache_len <= 0) { set_error_msg("mllm_hip_model_batch_decode: sequence %d has no prefill", b); return MLLM_HIP_ERR_ARG; }
        if (m->seqs[b].cache_len + 1 > c.cache_limit) { fprintf(stderr, "mllm_hip: KV cache overflow (%d + 1 > %d)\n", m->seqs[b].cache_len, c.cache_limit); return MLLM_HIP_ERR_SHAPE; }
    }
    const int H = c.hidden, I = c.inter, D = m->D, half = D / 2;
    hipStream_t st = m->st;
    // the rotary row of every sequence's next position (QWEN2VL: all three axes = last_pos + 1, modeling_qwen2_vl.hpp:423-432; HF rotary: position = tokens in its cache)
    std::vector<float> idf(B), s((size_t)B * half), co((size_t)B * half);
    for (int b = 0; b < B; ++b) idf[b] = (float)tokens[b];
    if (m->mrope) {
        std::vector<float> pos((size_t)3 * B);
        for (int a = 0; a < 3; ++a) for (int b = 0; b < B; ++b) pos[(size_t)a * B + b] = m->seqs[b].last_pos + 1.0f;
        EH(mllm_hip_mrope_table(c.rope_theta, D, pos.data(), B, c.mrope_section, 3, s.data(), co.data()));
    } else {
        for (int b = 0; b < B; ++b) {
            memcpy(s.data() + (size_t)b * half, m->hf_sin.data() + (size_t)m->seqs[b].cache_len * half, (size_t)half * 4);
            memcpy(co.data() + (size_t)b * half, m->hf_cos.data() + (size_t)m->seqs[b].cache_len * half, (size_t)half * 4);
        }
    }
    HH(hipMemcpyAsync(m->rope_sin, s.data(), s.size() * 4, hipMemcpyHostToDevice, st));
    HH(hipMemcpyAsync(m->rope_cos, co.data(), co.size() * 4, hipMemcpyHostToDevice, st));
    HH(hipMemcpyAsync(m->ids_f, idf.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    std::vector<SeqKV> desc(B);
    for (int b = 0; b < B; ++b) desc[b] = SeqKV{m->seqs[b].kslab, m->seqs[b].vslab, m->seqs[b].cache_len, 0};
    HH(hipMemcpyAsync(m->seqkv_dev, desc.data(), (size_t)B * sizeof(SeqKV), hipMemcpyHostToDevice, st));
    HH(hipStreamSynchronize(st));      // the host vectors go out of scope; inputs resident when the clock starts
    // from four rows on the Linears take the packed MFMA GEMM: its 32-row tile costs the same for 1 .. 32 rows, the M < 16 GEMV form pays its chain tables per row
    struct MinRows { M *m; int keep; ~MinRows() { m->gemm_min_rows = keep; } } restore{m, m->gemm_min_rows};
    if (B >= 4) m->gemm_min_rows = B;
    HH(hipEventRecord(m->ev0, st));
    EH(mllm_hip_embedding_q40(m->ids_f, m->emb_qs, m->emb_d, m->h0, B, H, c.vocab, st));
    float *h = m->h0, *h2 = m->h1;
    for (int li = 0; li < c.layers; ++li) {
        auto &L = m->layers[li];
        EH(q_rmsnorm(m, h, L.in_norm, m->xq, B, H, c.rms_eps));
        EH(lin(m, L.qkv, m->xq, m->qkv, MLLM_HIP_F32, m->QKV, nullptr, B));
        // attention is the one Op that is not row-wise: sequence b's new key / value go to ITS slabs at ITS position, its query walks ITS cache -- all B in one launch each
        const int64_t koff = (int64_t)li * c.cache_limit * m->KVD, voff = (int64_t)li * m->KVD * m->vt_ld;
        EH(seqs_rope_append_launch(m->qkv, m->QKV, m->rope_sin, m->rope_cos, half, m->seqkv_dev, koff, voff, m->KVD, m->vt_ld, B, c.heads, c.kv_heads, D, st));
        EH(seqs_fa2_decode_launch(m->qkv, m->QKV, m->seqkv_dev, koff, voff, m->KVD, m->vt_ld, m->attn, m->HD, B, c.heads, c.kv_heads, D, c.cache_limit, st));
        EH(q_quant(m, m->attn, m->xq, B, m->HD));
        EH(lin(m, L.o, m->xq, h2, MLLM_HIP_F32, H, h, B));
        EH(q_rmsnorm(m, h2, L.post_norm, m->xq, B, H, c.rms_eps));
        EH(lin(m, L.gu, m->xq, m->gu, MLLM_HIP_F32, 2 * I, nullptr, B));
        EH(q_silu_mul_quant(m, m->gu, m->act, m->xq2, B, I));
        EH(lin(m, L.down, m->xq2, h, MLLM_HIP_F32, H, h2, B));
    }
    if (c.tie_embedding) {
        EH(mllm_hip_rmsnorm(h, m->final_norm, m->bnormed, nullptr, nullptr, nullptr, B, H, c.final_eps, 0, st));
        EH(mllm_hip_quantize_q80(m->bnormed, m->bx80_qs, m->bx80_d, B, H, st));
        EH(mllm_hip_linear_q40_q80(m->emb_qs, m->emb_d, nullptr, m->bx80_qs, m->bx80_d, m->blogits, c.vocab, B, c.vocab, H, st));
    } else {
        EH(mllm_hip_rmsnorm(h, m->final_norm, nullptr, m->xq.qs, m->xq.d, m->xq.bs, B, H, c.final_eps, 0, st));
        EH(mllm_hip_linear_q4k_q8k(m->head.w, nullptr, m->xq.qs, m->xq.d, m->xq.bs, m->blogits, MLLM_HIP_F32, c.vocab, nullptr, B, c.vocab, H, st));
    }
    for (int b = 0; b < B; ++b) EH(argmax_row_launch(m->dctx, m->blogits + (size_t)b * c.vocab, c.vocab, m->btok + b, st));
    HH(hipEventRecord(m->ev1, st));
    if (logits_host) HH(hipMemcpyAsync(logits_host, m->blogits, (size_t)B * c.vocab * 4, hipMemcpyDeviceToHost, st));
    if (next_tokens) HH(hipMemcpyAsync(next_tokens, m->btok, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    HH(hipStreamSynchronize(st));
    if (elapsed_ms) HH(hipEventElapsedTime(elapsed_ms, m->ev0, m->ev1));
    for (int b = 0; b < B; ++b) { m->seqs[b].cache_len += 1; m->seqs[b].last_pos += 1.0f; }
    if (m->cur_seq < B) { m->cache_len = m->seqs[m->cur_seq].cache_len; m->last_pos = m->seqs[m->cur_seq].last_pos; m->needs_arm = true; }
    return MLLM_HIP_OK;
}

// Module::generate's loop (mllm/Module.cpp:63-100) with the method switch of :76-88: greedy / top-k / top-p.  The forward and the candidate selection
// run on the device; what crosses PCIe per step is the k candidates (top-k) or the sorted prefix that reaches the nucleus mass (top-p, in chunks), and
// the chosen id going back.  The temperature softmax over the candidates (Generate.cpp:69-87 / :120-136: float exp results, double sum, float
// renormalisation) and the draw are host arithmetic as in the reference; the draw consumes u01[step] by inverse CDF over the float probabilities where
// the reference seeds a std::discrete_distribution from std::random_device (Generate.hpp:38-44), which no caller can reproduce.
extern "C" int mllm_hip_model_generate_sampled(mllm_hip_model *m, int32_t first_token, int steps, int method, int top_k, float top_p, float temperature,
                                               const float *u01, int32_t eos, int32_t *tokens_host, int *n_out, float *elapsed_ms) {
    if (!m || !m->has_llm || m->cache_len <= 0 || steps <= 0 || method < 0 || method > 2) return MLLM_HIP_ERR_ARG;
    if (method != 0 && (!u01 || !(temperature > 0.0f))) return MLLM_HIP_ERR_ARG;
    if (method == 1 && (top_k < 0 || top_k > 64 || top_k > m->c.vocab)) return MLLM_HIP_ERR_SHAPE;
    if (method == 2 && !(top_p > 0.0f)) return MLLM_HIP_ERR_ARG;      // p <= 0 or NaN keeps no candidate at all (the reference then indexes an empty vector, Generate.cpp:116-118)
    if (m->cache_len + steps > m->c.cache_limit) { fprintf(stderr, "mllm_hip: KV cache overflow (%d + %d > %d)\n", m->cache_len, steps, m->c.cache_limit); return MLLM_HIP_ERR_SHAPE; }
    if (m->needs_arm) { EH(arm_decode(m)); m->needs_arm = false; }
    const int V = m->c.vocab;
    if (method != 0 && !m->samp_val) {
        EH(m->dalloc(&m->samp_val, (size_t)V * 4)); EH(m->dalloc(&m->samp_idx, (size_t)V * 4)); EH(m->dalloc(&m->samp_prob, (size_t)V * 4));
        m->sort_ws_bytes = mllm_hip_sort_desc_workspace_bytes(V);
        HH(hipMalloc(&m->sort_ws, m->sort_ws_bytes));
    }
    const auto t0 = std::chrono::steady_clock::now();
    int32_t tok = first_token;
    int made = 0;
    std::vector<float> val, prob;
    std::vector<int> idx;
    // one step; the host counters advance once the step's launches are enqueued, and every later failure re-reads the device state (resync_after_error)
    auto one_step = [&](int s, int32_t *out) -> int {
        HH(hipMemcpyAsync(&m->d_state->token, &tok, 4, hipMemcpyHostToDevice, m->st));
        EH(launch_step(m));
        m->cache_len += 1;
        m->last_pos += 1.0f;
        int32_t next = 0;
        if (method == 0 || (method == 1 && top_k <= 1)) {
            // greedy: the device argmax of the step (std::max_element, first maximum); top-k with k in {0, 1} is the same (Generate.cpp:50-54)
            HH(hipMemcpyAsync(&next, m->tok_dev, 4, hipMemcpyDeviceToHost, m->st));
            HH(hipStreamSynchronize(m->st));
        } else if (method == 1) {
            EH(mllm_hip_topk(m->logits, V, top_k, m->samp_val, m->samp_idx, m->st));
            val.resize(top_k); idx.resize(top_k); prob.resize(top_k);
            HH(hipMemcpyAsync(val.data(), m->samp_val, (size_t)top_k * 4, hipMemcpyDeviceToHost, m->st));
            HH(hipMemcpyAsync(idx.data(), m->samp_idx, (size_t)top_k * 4, hipMemcpyDeviceToHost, m->st));
            HH(hipStreamSynchronize(m->st));
            EH(mllm_hip_topk_probs_host(val.data(), top_k, temperature, prob.data()));
            next = idx[mllm_hip_sample_index_host(prob.data(), top_k, u01[s])];
        } else {
            // top-p expects probabilities (it throws when the largest score exceeds 1, Generate.cpp:104-106): the caller's graph ends in a softmax over the
            // vocabulary row (CPUSoftMax), then all scores are sorted descending and the prefix whose running float sum reaches p is kept
            EH(mllm_hip_softmax(m->logits, m->samp_prob, 1, V, nullptr, m->st));
            EH(mllm_hip_sort_desc(m->samp_prob, V, m->samp_val, m->samp_idx, m->sort_ws, m->sort_ws_bytes, m->st));
            val.clear(); idx.clear();
            float p = 0.0f;
            int have = 0;
            bool done = false;
            while (!done && have < V) {
                const int chunk = std::min(V - have, have == 0 ? 256 : 4096);
                val.resize(have + chunk); idx.resize(have + chunk);
                HH(hipMemcpyAsync(val.data() + have, m->samp_val + have, (size_t)chunk * 4, hipMemcpyDeviceToHost, m->st));
                HH(hipMemcpyAsync(idx.data() + have, m->samp_idx + have, (size_t)chunk * 4, hipMemcpyDeviceToHost, m->st));
                HH(hipStreamSynchronize(m->st));
                int n = have;
                while (n < have + chunk && p < top_p) { p += val[n]; ++n; }       // `while (p < m_p)` of Generate.cpp:108-115, float accumulation
                if (p >= top_p || n < have + chunk) { val.resize(n); idx.resize(n); done = true; }
                have = n;
            }
            const int k = (int)val.size();
            if (k <= 1) next = idx.empty() ? 0 : idx[0];
            else {
                prob.resize(k);
                EH(mllm_hip_topk_probs_host(val.data(), k, temperature, prob.data()));
                next = idx[mllm_hip_sample_index_host(prob.data(), k, u01[s])];
            }
        }
        *out = next;
        return 0;
    };
    for (int s = 0; s < steps; ++s) {
        int32_t next = 0;
        if (int rc = one_step(s, &next)) { if (n_out) *n_out = made; return resync_after_error(m, rc); }
        if (tokens_host) tokens_host[made] = next;
        ++made;
        tok = next;
        if (eos >= 0 && next == eos) break;
    }
    HH(hipStreamSynchronize(m->st));
    if (n_out) *n_out = made;
    if (elapsed_ms) *elapsed_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

// n_img images through the tower, NB at a time (one pass over NB * tokens rows); the upload of group g+1 (pinned staging, its own copy stream) runs while the tower works on group g
extern "C" int mllm_hip_model_vision(mllm_hip_model *m, const float *images_host, const int32_t *image_meta, int n_img, float *out_dev, float *elapsed_ms) {
    if (!m || m->vkind == V_NONE || !images_host || !out_dev || n_img <= 0) return MLLM_HIP_ERR_ARG;
    if (m->vkind == V_QWEN2VL && !image_meta) return MLLM_HIP_ERR_ARG;
    int nt, orows, ocols; size_t ie;
    vision_dims(m, image_meta, &nt, &orows, &ocols, &ie);
    // images per pass: up to about 6400 token rows (the row-wise kernels then fill the chip even for 197-token images), at most 32, the images spread evenly over
    // the passes (8 images at 7 per pass = 4 + 4, not 7 + 1)
    int NB = std::max(1, std::min(std::min(32, 6400 / std::max(nt, 1)), n_img));
    if (option(OPT_VISION_BATCH) > 0) NB = std::max(1, std::min(option(OPT_VISION_BATCH), n_img));
    NB = (n_img + (n_img + NB - 1) / NB - 1) / ((n_img + NB - 1) / NB);
    EH(ensure_vision_buffers(m, image_meta, NB));
    const int ngroups = (n_img + NB - 1) / NB;
    const size_t gstride = ie * (size_t)m->vis_batch;                  // the staging buffers hold vis_batch images each
    hipStream_t cp;
    HH(hipStreamCreateWithFlags(&cp, hipStreamNonBlocking));
    auto upload = [&](int gi) -> int {
        const int b = gi & 1, cnt = std::min(NB, n_img - gi * NB);
        float *pin = m->pin_img + (size_t)b * gstride;
        if (gi >= 2) HH(hipEventSynchronize(m->vfree[b]));          // the tower has finished with group gi-2, which sat in this device buffer
        memcpy(pin, images_host + (size_t)gi * NB * ie, ie * 4 * cnt);   // (its H2D copy was waited for by the compute stream before that)
        HH(hipMemcpyAsync(m->vpix[b], pin, ie * 4 * cnt, hipMemcpyHostToDevice, cp));
        HH(hipEventRecord(m->vup[b], cp));
        return 0;
    };
    int rc = upload(0);
    if (!rc) { hipError_t e = hipEventRecord(m->ev0, m->st); if (e != hipSuccess) rc = MLLM_HIP_ERR_HIP; }
    for (int gi = 0; gi < ngroups && !rc; ++gi) {
        const int b = gi & 1, cnt = std::min(NB, n_img - gi * NB);
        if (hipStreamWaitEvent(m->st, m->vup[b], 0) != hipSuccess) { rc = MLLM_HIP_ERR_HIP; break; }
        if (gi + 1 < ngroups) rc = upload(gi + 1);
        if (!rc) rc = forward_vision(m, m->vpix[b], image_meta, out_dev + (size_t)gi * NB * orows * ocols, cnt);
        if (!rc && hipEventRecord(m->vfree[b], m->st) != hipSuccess) rc = MLLM_HIP_ERR_HIP;
    }
    if (!rc) {
        if (hipEventRecord(m->ev1, m->st) != hipSuccess || hipEventSynchronize(m->ev1) != hipSuccess) rc = MLLM_HIP_ERR_HIP;
        else if (elapsed_ms && hipEventElapsedTime(elapsed_ms, m->ev0, m->ev1) != hipSuccess) rc = MLLM_HIP_ERR_HIP;
    }
    (void)hipStreamSynchronize(cp);
    (void)hipStreamDestroy(cp);
    return rc;
}

extern "C" int mllm_hip_model_time_kernel(mllm_hip_model *m, int which, int iters, float *ms_per_launch, int64_t *bytes_per_launch) {
    // which 0..3: stand-alone Q4_K GEMV launcher on gate|up, down, qkv, o; which 10..14: fused decode kernels qkv, attn, o-proj,
    // gate|up, down.  Launch i uses layer i % layers, so consecutive launches stream different weights (28 x 15.5 MB does not
    // fit the 256 MiB Infinity Cache): the time is that of a cold HBM stream, like inside the decode step.
    if (!m || !m->has_llm || iters <= 0) return MLLM_HIP_ERR_ARG;
    int nl = (int)m->layers.size();
    if (option(OPT_TIME_LAYERS) > 0) nl = std::max(1, std::min(nl, option(OPT_TIME_LAYERS)));   // fewer layers: an Infinity-Cache-warm stream
    DecodeCtx alone = m->dctx;      // the kernels one at a time: the step's merged launch (attention + o-projection) is taken apart
    alone.merge_o = 0;
    auto launch = [&](int i) -> int {
        auto &L = m->layers[i % nl];
        if (which >= 10) return decode_kernel_launch(alone, m->dlayers.data(), i % nl == 0 && which == 10 ? 1 % nl : i % nl, which - 10, m->st);
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

// ---- round-1 entry points of the Qwen2-VL engine: forwards onto the generic engine ----------------------------------------------------------
extern "C" int mllm_hip_qwen2vl_create(const mllm_hip_qwen2vl_config *q, const char *path, mllm_hip_qwen2vl **out) {
    if (!q || !path || !out) return MLLM_HIP_ERR_ARG;
    mllm_hip_model_config c;
    memset(&c, 0, sizeof(c));
    c.arch = MLLM_HIP_ARCH_QWEN2VL;
    c.hidden = q->hidden; c.inter = q->inter; c.layers = q->layers; c.heads = q->heads; c.kv_heads = q->kv_heads; c.vocab = q->vocab;
    c.rms_eps = q->rms_eps; c.final_eps = 1e-6f;      // Qwen2VLModel's model.norm: RMSNorm(hidden_dim, 1e-6, ...) (modeling_qwen2_vl.hpp:374)
    c.rope_theta = q->rope_theta;
    for (int i = 0; i < 3; ++i) c.mrope_section[i] = q->mrope_section[i];
    c.cache_limit = q->cache_limit; c.tie_embedding = q->tie_embedding; c.qkv_bias = 1;
    c.v_dim = q->v_dim; c.v_heads = q->v_heads; c.v_blocks = q->v_blocks; c.v_patch = q->v_patch; c.v_merge = q->v_merge;
    c.image_token_id = q->image_token_id; c.vision_start_token_id = q->vision_start_token_id; c.vision_end_token_id = q->vision_end_token_id;
    c.video_token_id = q->video_token_id;
    return mllm_hip_model_create(&c, path, out);
}
extern "C" void mllm_hip_qwen2vl_destroy(mllm_hip_qwen2vl *m) { mllm_hip_model_destroy(m); }
extern "C" int mllm_hip_qwen2vl_clear_kvcache(mllm_hip_qwen2vl *m) { return mllm_hip_model_clear_kvcache(m); }
extern "C" int mllm_hip_qwen2vl_prefill(mllm_hip_qwen2vl *m, const int32_t *ids, int n_ids, const float *pixel_values, const int32_t *grid_thw,
                                        float *logits_host, int32_t *next_token, float *elapsed_ms) {
    if (pixel_values && !grid_thw) return MLLM_HIP_ERR_ARG;
    return mllm_hip_model_prefill(m, ids, n_ids, pixel_values, grid_thw, nullptr, 0, logits_host, next_token, elapsed_ms);
}
extern "C" int mllm_hip_qwen2vl_decode(mllm_hip_qwen2vl *m, int32_t token, float *logits_host, int32_t *next_token, float *elapsed_ms) {
    return mllm_hip_model_decode(m, token, logits_host, next_token, elapsed_ms);
}
extern "C" int mllm_hip_qwen2vl_generate(mllm_hip_qwen2vl *m, int32_t first_token, int steps, int32_t *tokens_host, float *elapsed_ms) {
    return mllm_hip_model_generate(m, first_token, steps, tokens_host, elapsed_ms);
}
extern "C" int mllm_hip_qwen2vl_vision(mllm_hip_qwen2vl *m, const float *pixel_values_host, const int32_t *grid_thw, int n_img, float *embeds_dev, float *elapsed_ms) {
    if (!grid_thw) return MLLM_HIP_ERR_ARG;
    return mllm_hip_model_vision(m, pixel_values_host, grid_thw, n_img, embeds_dev, elapsed_ms);
}
extern "C" int64_t mllm_hip_qwen2vl_decode_weight_bytes(const mllm_hip_qwen2vl *m) { return mllm_hip_model_decode_weight_bytes(m); }
extern "C" void *mllm_hip_qwen2vl_stream(mllm_hip_qwen2vl *m) { return mllm_hip_model_stream(m); }
extern "C" int mllm_hip_qwen2vl_time_gemv(mllm_hip_qwen2vl *m, int which, int iters, float *ms_per_launch, int64_t *bytes_per_launch) {
    return mllm_hip_model_time_kernel(m, which, iters, ms_per_launch, bytes_per_launch);
}
