// integration/hip/HIPOps.cpp -- the HIP*Op classes: mllm::Op subclasses whose execute() is one or two calls into the C ABI (include/mllm_hip.h).
// Each class names the CPU Op it stands in for (mllm/backends/cpu/op/...) and the Layer / Tensor function that creates it with its OpParam keys (mllm/Layer.hpp,
// mllm/Tensor.cpp).  Shapes a launcher does not cover are refused at opCreate time (nullptr => CPU fallback, mllm/Layer.hpp:214-218) or, for what is only known
// with the tensors in hand, thrown from reshape() before anything is launched -- never at execute.
// Device layout: every activation is the reference's BSHD tensor in memory order [batch][sequence][head][dimension], contiguous, fp32 (fp16 for the KV slabs);
// Tensor views (F_VIEW, F_CLIP, F_TRANPOSE, KVCACHE, PARAMETER outputs) are reference-counted aliases of their producer's block (HIPBackend::view_of).
// Compiled against the reference's headers by oracle/Makefile.ref (test infrastructure; see HIPBackend.hpp).
#include <cmath>
#include <cstring>

#include "HIPBackend.hpp"
#include "ParamLoader.hpp"

namespace mllm {

namespace {

using TensorList = const vector<shared_ptr<Tensor>> &;      // HIPOp::reshape_ / setUp_ / execute_: by reference (HIPBackend.hpp)
using TensorListV = vector<shared_ptr<Tensor>>;             // Op::free keeps the reference's by-value signature

inline int geti(const OpParam &p, const char *k, int def = 0) { auto it = p.find(k); return it == p.end() ? def : (int)it->second; }
inline float getf(const OpParam &p, const char *k, float def = 0.f) { auto it = p.find(k); return it == p.end() ? def : it->second; }
inline int rows_of(const shared_ptr<Tensor> &t) { return t->batch() * t->sequence() * t->head(); }       // BSHD: rows of `dimension()` values
inline size_t elem_bytes(DataType dt) { return dt == MLLM_TYPE_F16 ? 2 : 4; }
inline void need(bool ok, const char *what) { if (!ok) throw std::runtime_error(std::string("mllm_hip adapter: ") + what); }
#define HIPCHK(call) HIPBackend::check((call), #call)      // direct call (load time: the caller drains the deferred queue first)
// an execute()-time launch: the arguments are evaluated HERE (device pointers, extents), the call itself is issued by the backend's worker thread in program order
#define HIPQ(fn, ...) hb()->defer(#fn, fn, __VA_ARGS__)

// loads `<op>.weight` / `<op>.bias` onto the device in the file's storage dtype (ParamLoader::load -> Backend::load_from_file fast path)
void load_tensor(Tensor &t, Backend *bn, AbstructLoader &loader, const string &name, int rows, int cols) {
    const DataType dt = loader.getDataType(name);
    if (dt == MLLM_TYPE_COUNT) throw std::runtime_error("mllm_hip adapter: tensor '" + name + "' is not in the weight file");
    t.setName(name);
    t.setBackend(bn);
    t.reshape(1, 1, rows, cols);
    t.setDtype(dt);
    t.alloc();
    if (!loader.load(&t)) throw std::runtime_error("mllm_hip adapter: loading '" + name + "' failed");
}
// output = fp32 BSHD tensor of the given shape, allocated from the pool (what Op::setUp does, mllm/Op.hpp:63-70, spelled out so that 5-D inputs do not leak their ctype)
void alloc_f32(const shared_ptr<Tensor> &out) {
    out->setDtype(MLLM_TYPE_F32);
    out->alloc();
}

// ---- LINEAR: CPULinear (op/CPULinear.cpp:23-234); params in_features, out_features, bias (Layer.hpp:230-234) -----------------------------------------------
class HIPLinearOp final : public HIPOp {
public:
    HIPLinearOp(Backend *bn, const string &name, int in, int out, bool bias) : HIPOp(bn, name), in_(in), out_(out), has_bias_(bias) {}
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        need(inputs[0]->dimension() == in_, "LINEAR: input width differs from in_features");
        outputs[0]->reshape(inputs[0]->batch(), inputs[0]->head(), inputs[0]->sequence(), out_);
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList outputs) override { alloc_f32(outputs[0]); return MLLM_NO_ERROR; }
    ErrorCode load(AbstructLoader &loader) override {
        load_tensor(weight_, backend_, loader, name() + ".weight", out_, in_);
        const DataType dt = weight_.dtype();
        if (dt != MLLM_TYPE_Q4_K && dt != MLLM_TYPE_Q4_0 && dt != MLLM_TYPE_F32) throw std::runtime_error("HIPLinearOp: weight dtype not on the hot path: " + name());      // precedent OpenCLLinearOp.cpp:26-52
        if (dt == MLLM_TYPE_Q4_K) {      // resident Linears are packed once for the M >= 16 GEMM (mllm_hip_q4k_prepack)
            packed_ = hb()->dev_alloc(mllm_hip_q4k_wpack_bytes(out_, in_));
            hb()->drain();
            HIPCHK(mllm_hip_q4k_prepack(weight_.device_memory().handle, out_, in_, packed_, hb()->stream()));
        } else if (dt == MLLM_TYPE_Q4_0) {  // 18-byte blocks -> nibble plane + fp16 scale plane (mllm_hip_repack_q40)
            const int64_t nblk = (int64_t)out_ * (in_ / 32);
            q40_qs_ = hb()->dev_alloc((size_t)nblk * 16);
            q40_d_ = hb()->dev_alloc((size_t)nblk * 2);
            hb()->drain();
            HIPCHK(mllm_hip_repack_q40(weight_.device_memory().handle, (uint8_t *)q40_qs_, (uint16_t *)q40_d_, nblk, hb()->stream()));
        }
        if (has_bias_) load_tensor(bias_, backend_, loader, name() + ".bias", 1, out_);
        return MLLM_NO_ERROR;
    }
    ErrorCode execute_(TensorList inputs, TensorList outputs) override {
        auto *b = hb();
        const int M = rows_of(inputs[0]);
        if (M == 0) return MLLM_NO_ERROR;
        const float *bias = has_bias_ ? (const float *)bias_.device_memory().handle : nullptr;
        const float *x = (const float *)dptr(inputs[0]);
        void *y = dptr(outputs[0]);
        switch (weight_.dtype()) {
        case MLLM_TYPE_Q4_K: {
            // activations to Q8_K (quantize_row_q8_K_reference), then vec_dot_q4_K_q8_K per (row, output): GEMV below 16 rows, packed GEMM from 16 on
            if (M < 16) {
                uint8_t *ws = (uint8_t *)b->scratch(0, mllm_hip_linear_workspace_bytes(MLLM_HIP_Q4_K, M, in_));
                if (M == 1) {      // one row: a candidate for the lazy window's fused launches (HIPBackend::lazy)
                    HIPBackend::LazyOp o;
                    o.kind = HIPBackend::LazyOp::LINEAR; o.a = x; o.out = (float *)y; o.n = out_; o.w = bias; o.W = weight_.device_memory().handle; o.K = in_; o.ws = ws;
                    b->lazy(o);
                } else
                HIPQ(mllm_hip_linear, weight_.device_memory().handle, MLLM_HIP_Q4_K, bias, x, y, MLLM_HIP_F32, out_, M, out_, in_, ws, b->stream());
            } else {
                void *xpack = b->scratch(1, mllm_hip_q4k_prepack_bytes(M, in_));
                HIPBackend::LazyOp o;      // the window lets the Linears on one set of rows share one packed operand, written by the norm in front of them when there is one
                o.kind = HIPBackend::LazyOp::LINEAR_M; o.a = x; o.out = (float *)y; o.n = out_; o.w = bias; o.Wpacked = packed_; o.K = in_; o.M = M; o.ws = xpack;
                b->lazy(o);
            }
            break;
        }
        case MLLM_TYPE_Q4_0: {
            uint8_t *ws = (uint8_t *)b->scratch(0, mllm_hip_linear_workspace_bytes(MLLM_HIP_Q4_0, M, in_));
            int8_t *qs = (int8_t *)ws;
            uint16_t *d = (uint16_t *)(ws + (((size_t)M * in_ + 255) & ~(size_t)255));
            HIPQ(mllm_hip_quantize_q80, x, qs, d, M, in_, b->stream());
            HIPQ(mllm_hip_linear_q40_q80, (const uint8_t *)q40_qs_, (const uint16_t *)q40_d_, bias, qs, d, (float *)y, out_, M, out_, in_, b->stream());
            break;
        }
        default:
            HIPQ(mllm_hip_linear_f32, (const float *)weight_.device_memory().handle, bias, x, (float *)y, out_, M, out_, in_, b->stream());
        }
        return MLLM_NO_ERROR;
    }
    ErrorCode free(TensorListV, TensorListV) override {
        weight_.free();
        if (has_bias_) bias_.free();
        for (void **p : {&packed_, &q40_qs_, &q40_d_}) if (*p) { hb()->dev_release(*p); *p = nullptr; }
        return MLLM_NO_ERROR;
    }

private:
    int in_, out_;
    bool has_bias_;
    Tensor weight_, bias_;
    void *packed_ = nullptr, *q40_qs_ = nullptr, *q40_d_ = nullptr;
};

// ---- EMBEDDING: CPUEmbedding (op/CPUEmbedding.cpp:38-80); hidden_size, vocab_size (Layer.hpp:434-435); ids are fp32 ------------------------------------------
class HIPEmbeddingOp final : public HIPOp {
public:
    HIPEmbeddingOp(Backend *bn, const string &name, int hidden, int vocab) : HIPOp(bn, name), hidden_(hidden), vocab_(vocab) {}
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        outputs[0]->reshape(inputs[0]->batch(), 1, inputs[0]->sequence(), hidden_);
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList outputs) override { alloc_f32(outputs[0]); return MLLM_NO_ERROR; }
    ErrorCode load(AbstructLoader &loader) override {
        const DataType dt = loader.getDataType(name() + ".weight");
        if (dt == MLLM_TYPE_Q4_0) table_ = hb()->q40_table(loader, name() + ".weight", vocab_, hidden_);      // shared with the tied lm_head's PARAMETER
        else if (dt == MLLM_TYPE_F32) load_tensor(f32_, backend_, loader, name() + ".weight", vocab_, hidden_);      // CLIP's position_embedding (an "embeddings" name: fp32 in the file)
        else throw std::runtime_error("HIPEmbeddingOp: Q4_0 or fp32 tables (what *-q4_k.mllm files hold): " + name());
        return MLLM_NO_ERROR;
    }
    ErrorCode execute_(TensorList inputs, TensorList outputs) override {
        const int S = inputs[0]->batch() * inputs[0]->sequence();
        if (S == 0) return MLLM_NO_ERROR;
        if (table_) HIPQ(mllm_hip_embedding_q40, (const float *)dptr(inputs[0]), (const uint8_t *)table_->qs, (const uint16_t *)table_->d, (float *)dptr(outputs[0]), S, hidden_, vocab_, hb()->stream());
        else HIPQ(mllm_hip_gather_rows, (const float *)f32_.device_memory().handle, hidden_, vocab_, (const float *)dptr(inputs[0]), (float *)dptr(outputs[0]), hidden_, S, hidden_, 0, hb()->stream());      // CPUEmbedding.cpp:46-60: a row copy
        return MLLM_NO_ERROR;
    }
    ErrorCode free(TensorListV, TensorListV) override { if (!table_) f32_.free(); return MLLM_NO_ERROR; }

private:
    int hidden_, vocab_;
    std::shared_ptr<HIPQ40Table> table_;
    Tensor f32_;
};

// ---- PARAMETER: CPUParameter (op/CPUParameter.cpp; batch, seq, head, dim, Layer.hpp:904-918): hands out its weight.  The one on the hot path is the tied lm_head,
// `lm_head()` = the Q4_0 embedding table (modeling_qwen2_vl.hpp:376,399; modeling_qwen.hpp:148-159): a view of the shared table's raw blocks -----------------------
class HIPParameterOp final : public HIPOp {
public:
    HIPParameterOp(Backend *bn, const string &name, int b, int h, int s, int d) : HIPOp(bn, name), b_(b), h_(h), s_(s), d_(d) {}
    bool keeps_shadow() const override { return true; }
    ErrorCode reshape_(TensorList, TensorList outputs) override {
        outputs[0]->reshape(b_, h_, s_, d_);
        return MLLM_NO_ERROR;
    }
    ErrorCode load(AbstructLoader &loader) override {
        const DataType dt = loader.getDataType(name());
        if (dt == MLLM_TYPE_Q4_0 && b_ == 1 && h_ == 1) { table_ = hb()->q40_table(loader, name(), s_, d_); return MLLM_NO_ERROR; }
        if (dt != MLLM_TYPE_F32) throw std::runtime_error("HIPParameterOp: Q4_0 tables and fp32 parameters only: " + name());
        weight_.setName(name());
        weight_.setBackend(backend_);
        weight_.reshape(b_, h_, s_, d_);
        weight_.setDtype(dt);
        weight_.alloc();
        loader.load(&weight_);
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList outputs) override {
        Tensor &w = table_ ? table_->raw : weight_;
        outputs[0]->setDtype(w.dtype());
        hb()->view_of(outputs[0], w.device_memory().handle, w.cntSize());
        return MLLM_NO_ERROR;
    }

private:
    int b_, h_, s_, d_;
    std::shared_ptr<HIPQ40Table> table_;
    Tensor weight_;
};

// ---- RMSNORM / LAYERNORM: CPURMSNorm (op/CPURMSNorm.cpp:31-136; norm_size, epsilon, add_unit_offset), CPULayerNorm (op/CPULayerNorm.cpp:49-88; norm_size, epsilon, bias) ----
class HIPNormOp final : public HIPOp {
public:
    HIPNormOp(Backend *bn, const string &name, bool layer, int dim, float eps, bool bias, bool unit_offset) :
        HIPOp(bn, name), layer_(layer), dim_(dim), eps_(eps), has_bias_(bias), unit_offset_(unit_offset) {}
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        need(inputs[0]->dimension() == dim_, "norm: input width differs from norm_size");
        outputs[0]->reshape(inputs[0]->batch(), inputs[0]->head(), inputs[0]->sequence(), inputs[0]->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList outputs) override { alloc_f32(outputs[0]); return MLLM_NO_ERROR; }
    ErrorCode load(AbstructLoader &loader) override {
        load_tensor(weight_, backend_, loader, name() + ".weight", 1, dim_);
        if (layer_ && has_bias_) load_tensor(bias_, backend_, loader, name() + ".bias", 1, dim_);
        return MLLM_NO_ERROR;
    }
    ErrorCode execute_(TensorList inputs, TensorList outputs) override {
        const int M = rows_of(inputs[0]);
        if (M == 0) return MLLM_NO_ERROR;
        const float *w = (const float *)weight_.device_memory().handle;
        if (M >= 16 && dim_ % 256 == 0) {      // prefill rows: the window may let the norm write the packed operand of the Linears behind it
            HIPBackend::LazyOp o;
            o.kind = HIPBackend::LazyOp::NORM_M; o.a = (const float *)dptr(inputs[0]); o.out = (float *)dptr(outputs[0]); o.n = dim_; o.w = w; o.eps = eps_; o.M = M;
            o.layer_norm = layer_ ? 1 : 0; o.unit_offset = unit_offset_ ? 1 : 0; o.b = layer_ && has_bias_ ? (const float *)bias_.device_memory().handle : nullptr;
            hb()->lazy(o);
        } else if (layer_) HIPQ(mllm_hip_layernorm, (const float *)dptr(inputs[0]), w, has_bias_ ? (const float *)bias_.device_memory().handle : nullptr, (float *)dptr(outputs[0]), nullptr,
                                              nullptr, nullptr, M, dim_, eps_, hb()->stream());
        else if (M == 1 && !unit_offset_) {
            HIPBackend::LazyOp o;
            o.kind = HIPBackend::LazyOp::NORM; o.a = (const float *)dptr(inputs[0]); o.out = (float *)dptr(outputs[0]); o.n = dim_; o.w = w; o.eps = eps_;
            hb()->lazy(o);
        } else HIPQ(mllm_hip_rmsnorm, (const float *)dptr(inputs[0]), w, (float *)dptr(outputs[0]), nullptr, nullptr, nullptr, M, dim_, eps_, unit_offset_ ? 1 : 0, hb()->stream());
        return MLLM_NO_ERROR;
    }

private:
    bool layer_;
    int dim_;
    float eps_;
    bool has_bias_, unit_offset_;
    Tensor weight_, bias_;
};

// ---- ROPE (HF half-split table, CPURoPE.cpp:100-128,200-232; pose_type, rope_theta, max_position_embeddings) and MULTIMODALROPE (CPUMultimodalRoPE.cpp:84-264) ----
// input / output [B, H, S, D] in BSHD memory order = rows of H*D per position.  The position counter h_cnt_ lives in the op (CPURoPE.cpp:510-513) and clearCache()
// resets it; the multimodal form takes its positions from the second input, [3,1,1,S] (the host shadow of its upload, HIPBackend fact 3): the sin / cos tables of a
// forward are built once on the host (libm sinf / cosf, as the reference) and shared by all the MULTIMODALROPE Ops of the model (HIPBackend::mrope_tables).
class HIPRoPEOp final : public HIPOp {
public:
    HIPRoPEOp(Backend *bn, const string &name, bool multimodal, float theta, int max_pos, std::vector<int> section) :
        HIPOp(bn, name), multimodal_(multimodal), theta_(theta), max_pos_(max_pos), section_(std::move(section)) {}
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        if (!multimodal_ && h_cnt_ + inputs[0]->sequence() > max_pos_) throw std::runtime_error("HIPRoPEOp: position beyond max_position_embeddings");
        need(inputs[0]->batch() <= 1, "ROPE: batch 1");
        outputs[0]->reshape(inputs[0]->batch(), inputs[0]->head(), inputs[0]->sequence(), inputs[0]->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList outputs) override { alloc_f32(outputs[0]); return MLLM_NO_ERROR; }
    ErrorCode execute_(TensorList inputs, TensorList outputs) override {
        auto *b = hb();
        const int S = inputs[0]->sequence(), H = inputs[0]->head(), D = inputs[0]->dimension(), half = D / 2;
        if (S == 0) return MLLM_NO_ERROR;
        const float *ds, *dc;
        int ld_tab = half;
        if (multimodal_) {
            const HIPBackend::RopeTables t = b->mrope_tables(inputs[1], theta_, D, section_);
            need(t.S == S, "MULTIMODALROPE: position_ids hold a different number of positions than the input");
            ds = t.sin; dc = t.cos;
        } else {
            if (table_dim_ != D) {      // CPURoPE's static table [max_pos][D]: one per (theta, D, max_pos) in the backend, shared by every ROPE Op of the model
                tab_ = b->rope_hf_tables(theta_, D, max_pos_);
                table_dim_ = D;
            }
            ds = tab_ + (size_t)h_cnt_ * D;
            dc = tab_ + (size_t)max_pos_ * D + (size_t)h_cnt_ * D;
            ld_tab = D;
        }
        {
            HIPBackend::LazyOp o;
            o.kind = HIPBackend::LazyOp::ROPE; o.a = (const float *)dptr(inputs[0]); o.out = (float *)dptr(outputs[0]); o.sin = ds; o.cos = dc; o.ld_tab = ld_tab; o.S = S; o.H = H; o.D = D;
            b->lazy(o);
        }
        if (!multimodal_) h_cnt_ += S;
        return MLLM_NO_ERROR;
    }
    void clearCache() override { h_cnt_ = 0; }

private:
    bool multimodal_;
    float theta_;
    int max_pos_;
    std::vector<int> section_;
    int h_cnt_ = 0, table_dim_ = 0;
    const float *tab_ = nullptr;      // the backend's (HIPBackend::rope_hf_tables)
};

// ---- VISIONROPE: CPUVisionRoPE (op/CPUVisionRoPE.cpp:12-147; dim, spatial_merge_size): grid (t, h, w) -> the angle table [1,1,N,dim].  The grid is a host-side
// scalar tensor the model reads again right behind this layer (modeling_qwen2_vl.hpp:179-182) => host_inputs().  Besides the angles (the layer's contract) the Op
// uploads their libm sin / cos, which is what F_APPLY_VISIOROPE evaluates per use (CPUVisionRoPEFunc.hpp:21-60), and files them under the output's handle ----------
class HIPVisionRoPEOp final : public HIPOp {
public:
    HIPVisionRoPEOp(Backend *bn, const string &name, int dim, int merge) : HIPOp(bn, name), dim_(dim), merge_(merge) {}
    bool host_inputs() const override { return true; }
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        for (int i = 0; i < 3; ++i) g_[i] = (int)inputs[0]->dataAt<float>(0, 0, 0, i);
        need(g_[0] > 0 && g_[1] > 0 && g_[2] > 0 && g_[1] % merge_ == 0 && g_[2] % merge_ == 0, "VISIONROPE: grid_thw must be positive multiples of the merge size");
        outputs[0]->reshape(1, 1, g_[0] * g_[1] * g_[2], 2 * (dim_ / 2));
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList outputs) override { alloc_f32(outputs[0]); return MLLM_NO_ERROR; }
    ErrorCode execute_(TensorList, TensorList outputs) override {
        auto *b = hb();
        const int N = g_[0] * g_[1] * g_[2], rd = 2 * (dim_ / 2);
        // the tables of a grid are a pure function of (t, h, w): 82 k libm sines and cosines for a 448 x 448 image, about 2 ms of the caller's thread per forward -- kept for the
        // grid they were made for (angles, sines, cosines in one device block: [3][cap_]); the Op's output is filled from the device copy
        if (g_[0] != made_[0] || g_[1] != made_[1] || g_[2] != made_[2]) {
            std::vector<float> ang((size_t)N * rd), s((size_t)N * rd), c((size_t)N * rd);
            HIPCHK(mllm_hip_vision_rope_angles(g_[0], g_[1], g_[2], merge_, rd, ang.data()));
            HIPCHK(mllm_hip_vision_rope_table(g_[0], g_[1], g_[2], merge_, rd, s.data(), c.data()));
            if ((size_t)N * rd > cap_) {
                if (tab_) b->dev_release(tab_);
                cap_ = (size_t)N * rd;
                tab_ = (float *)b->dev_alloc(3 * cap_ * 4);
            }
            b->upload(tab_, s.data(), s.size() * 4);
            b->upload(tab_ + cap_, c.data(), c.size() * 4);
            b->upload(tab_ + 2 * cap_, ang.data(), ang.size() * 4);
            for (int i = 0; i < 3; ++i) made_[i] = g_[i];
        }
        HIPQ(mllm_hip_copy_2d_f32, (const float *)(tab_ + 2 * cap_), (int64_t)rd, (float *)dptr(outputs[0]), (int64_t)rd, N, rd, b->stream());
        b->set_vision_tables(dptr(outputs[0]), tab_, tab_ + cap_, N, rd);
        return MLLM_NO_ERROR;
    }

private:
    int dim_, merge_, g_[3] = {0, 0, 0}, made_[3] = {0, 0, 0};
    float *tab_ = nullptr;
    size_t cap_ = 0;
};
// F_APPLY_VISIOROPE: CPUVisionRoPEFuncFunction (op/CPUVisionRoPEFunc.hpp): x [1,H,N,D], angles [1,1,N,D/2] -> rotate-half with sin / cos of the angles
class HIPApplyVisionRoPEOp final : public HIPOp {
public:
    HIPApplyVisionRoPEOp(Backend *bn, const string &name) : HIPOp(bn, name) {}
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        need(inputs[0]->batch() == 1 && inputs[1]->sequence() == inputs[0]->sequence() && inputs[1]->dimension() == inputs[0]->dimension() / 2 && inputs[0]->dtype() == MLLM_TYPE_F32,
             "F_APPLY_VISIOROPE: x [1,H,N,D] fp32 with angles [1,1,N,D/2]");
        outputs[0]->reshape(inputs[0]->batch(), inputs[0]->head(), inputs[0]->sequence(), inputs[0]->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList outputs) override { alloc_f32(outputs[0]); return MLLM_NO_ERROR; }
    ErrorCode execute_(TensorList inputs, TensorList outputs) override {
        auto *b = hb();
        const int S = inputs[0]->sequence(), H = inputs[0]->head(), D = inputs[0]->dimension(), half = D / 2;
        HIPBackend::RopeTables t;
        if (!b->vision_tables(dptr(inputs[1]), &t) || t.S != S || t.half != half) {      // angles that did not come from the VISIONROPE Op: sin / cos on the host, once per call
            const std::vector<float> &ang = b->host_floats(inputs[1]);
            std::vector<float> s(ang.size()), c(ang.size());
            for (size_t i = 0; i < ang.size(); ++i) { s[i] = std::sin(ang[i]); c[i] = std::cos(ang[i]); }
            float *ds = (float *)b->scratch(4, ang.size() * 8), *dc = ds + ang.size();
            b->upload(ds, s.data(), s.size() * 4);
            b->upload(dc, c.data(), c.size() * 4);
            t = HIPBackend::RopeTables{ds, dc, S, half};
        }
        HIPQ(mllm_hip_rope_apply, (const float *)dptr(inputs[0]), (int64_t)H * D, t.sin, t.cos, half, dptr(outputs[0]), MLLM_HIP_F32, (int64_t)H * D, S, H, D, b->stream());
        return MLLM_NO_ERROR;
    }
};

// ---- KVCACHE: CPUKVCache (op/CPUKVCache.cpp:10-131,253-275; head, hidden, n_rep, cache_max, fa2).  FlashAttention2 mode: n_rep is forced to 1 (:41-43) and the
// slab is fp16 (KVCache_TYPE = 16, Types.hpp:26) -----------------------------------------------------------------------------------------------------------------
// The slab [cache_max][H*D] fp16 is the op's; execute() appends the S new rows at cache_seq_len_ (fp32 -> fp16, the rounding of the reference's fp16 stores into the
// slab: RoPE's MLLM_FP32_TO_FP16 and mat_mul's fp16 branch, Matmul.cpp:262-268) and the output is a view of rows [0, T + S) -- never a copy of the cache.
class HIPKVCacheOp final : public HIPOp {
public:
    HIPKVCacheOp(Backend *bn, const string &name, int cache_max) : HIPOp(bn, name), cache_max_(cache_max) {}
    bool keeps_shadow() const override { return true; }
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        need(inputs[0]->batch() == 1 && inputs[0]->dtype() == MLLM_TYPE_F32, "KVCACHE: batch 1, fp32 producer");
        if (cache_seq_len_ + inputs[0]->sequence() > cache_max_) {      // CPUKVCache.cpp:121-126
            fprintf(stderr, "\n[ERROR]: Current tokens exceed cache limit: %d>%d;\n         Please set args `--limits` >%d\n", cache_seq_len_ + inputs[0]->sequence(), cache_max_, cache_max_);
            exit(1);
        }
        outputs[0]->reshape(inputs[0]->batch(), inputs[0]->head(), cache_seq_len_ + inputs[0]->sequence(), inputs[0]->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList inputs, TensorList outputs) override {
        const size_t row = (size_t)inputs[0]->head() * inputs[0]->dimension();
        if (!slab_ || row != row_) {
            need(cache_seq_len_ == 0, "KVCACHE: the row width changed with tokens in the cache");
            if (slab_) hb()->dev_release(slab_);
            slab_ = hb()->dev_alloc((size_t)cache_max_ * row * 2);
            row_ = row;
        }
        outputs[0]->setDtype(MLLM_TYPE_F16);
        hb()->view_of(outputs[0], slab_, (size_t)outputs[0]->sequence() * row * 2);
        return MLLM_NO_ERROR;
    }
    ErrorCode execute_(TensorList inputs, TensorList) override {
        const int S = inputs[0]->sequence(), n = (int)row_;
        if (S) {
            HIPBackend::LazyOp o;
            o.kind = HIPBackend::LazyOp::KVSTORE; o.a = (const float *)dptr(inputs[0]); o.dst16 = (uint16_t *)slab_ + (size_t)cache_seq_len_ * n; o.n = n; o.S = S;
            hb()->lazy(o);
        }
        cache_seq_len_ += S;
        return MLLM_NO_ERROR;
    }
    int getCacheSeqLen() override { return cache_seq_len_; }
    void clearCache() override { cache_seq_len_ = 0; }

private:
    int cache_max_, cache_seq_len_ = 0;
    void *slab_ = nullptr;
    size_t row_ = 0;
};

// ---- F_FA2: CPUFlashAttention2Func (op/CPUFlashAttention2Func.hpp:29-125; causal_mask) -> flash_attention_2_forward (compute/FlashAttention2.hpp:2236-2284) -------
class HIPFlashAttention2Op final : public HIPOp {
public:
    HIPFlashAttention2Op(Backend *bn, const string &name, bool causal) : HIPOp(bn, name), causal_(causal) {}
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        auto &q = inputs[0], &k = inputs[1], &v = inputs[2];
        need(q->batch() == 1 && q->dtype() == MLLM_TYPE_F32 && k->dtype() == v->dtype() && (k->dtype() == MLLM_TYPE_F16 || k->dtype() == MLLM_TYPE_F32), "F_FA2: batch 1, fp32 q, fp16 or fp32 k / v");
        need(k->head() == v->head() && k->head() > 0 && q->head() % k->head() == 0 && k->sequence() == v->sequence() && k->dimension() == q->dimension() && v->dimension() == q->dimension(), "F_FA2: head / length mismatch");
        outputs[0]->reshape(q->batch(), q->head(), q->sequence(), q->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList outputs) override { alloc_f32(outputs[0]); return MLLM_NO_ERROR; }
    ErrorCode execute_(TensorList inputs, TensorList outputs) override {
        auto &q = inputs[0], &k = inputs[1], &v = inputs[2];
        const int Hq = q->head(), Hkv = k->head(), D = q->dimension();
        if (q->sequence() == 0) return MLLM_NO_ERROR;
        const int kvdt = k->dtype() == MLLM_TYPE_F16 ? MLLM_HIP_F16 : MLLM_HIP_F32;
        if (q->sequence() == 1) {      // one position: the window may fold the rotary Ops and the cache appends in front of it into the launch (HIPBackend::lazy)
            HIPBackend::LazyOp o;
            o.kind = HIPBackend::LazyOp::FA2; o.a = (const float *)dptr(q); o.kp = dptr(k); o.vp = dptr(v); o.out = (float *)dptr(outputs[0]); o.H = Hq; o.Hkv = Hkv; o.D = D;
            o.Sk = k->sequence(); o.causal = causal_ ? 1 : 0; o.kvdt = kvdt; o.S = 1;
            hb()->lazy(o);
            return MLLM_NO_ERROR;
        }
        HIPQ(mllm_hip_fa2, (const float *)dptr(q), (int64_t)Hq * D, dptr(k), (int64_t)Hkv * D, dptr(v), (int64_t)Hkv * D, kvdt, (float *)dptr(outputs[0]), (int64_t)Hq * D, q->sequence(),
                            k->sequence(), Hq, Hkv, D, causal_ ? 1 : 0, nullptr, nullptr, hb()->stream());
        return MLLM_NO_ERROR;
    }

private:
    bool causal_;
};

// ---- elementwise: CPUSiLU (op/CPUSiLU.cpp:24-52), CPUGELU / CPUQuickGELU through the fp16 LUTs (op/CPUGELU.cpp:24-46, op/CPUQuickGELU.cpp:22-43), F_TTADD / F_TTMUL (op/CPUBinaryFunc.hpp) ----
class HIPUnaryOp final : public HIPOp {
public:
    enum Kind { SILU_K, GELU_K, QUICKGELU_K };
    HIPUnaryOp(Backend *bn, const string &name, Kind k) : HIPOp(bn, name), kind_(k) {}
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        outputs[0]->reshape(inputs[0]->batch(), inputs[0]->head(), inputs[0]->sequence(), inputs[0]->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList outputs) override { alloc_f32(outputs[0]); return MLLM_NO_ERROR; }
    ErrorCode execute_(TensorList inputs, TensorList outputs) override {
        auto *b = hb();
        const int64_t n = inputs[0]->count();
        if (n == 0) return MLLM_NO_ERROR;
        if (kind_ == SILU_K && inputs[0]->dimension() % 8 != 0)      // rows with a libm tail (CPUSiLU.cpp:35-47 applies the vector form per row)
            HIPQ(mllm_hip_silu_rows, (const float *)dptr(inputs[0]), (float *)dptr(outputs[0]), n / inputs[0]->dimension(), inputs[0]->dimension(), b->stream());
        else if (kind_ == SILU_K) {
            HIPBackend::LazyOp o;
            o.kind = HIPBackend::LazyOp::SILU; o.a = (const float *)dptr(inputs[0]); o.out = (float *)dptr(outputs[0]); o.n = n;
            b->lazy(o);
        }
        else HIPQ(mllm_hip_act_lut, (const float *)dptr(inputs[0]), (float *)dptr(outputs[0]), n, kind_ == GELU_K ? b->gelu_lut() : b->quickgelu_lut(), b->stream());
        return MLLM_NO_ERROR;
    }

private:
    Kind kind_;
};
class HIPBinaryOp final : public HIPOp {
public:
    HIPBinaryOp(Backend *bn, const string &name, bool mul) : HIPOp(bn, name), mul_(mul) {}
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        need(inputs[0]->count() == inputs[1]->count(), "F_TTADD / F_TTMUL: operands of one shape (no broadcast on this path)");
        outputs[0]->reshape(inputs[0]->batch(), inputs[0]->head(), inputs[0]->sequence(), inputs[0]->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList outputs) override { alloc_f32(outputs[0]); return MLLM_NO_ERROR; }
    ErrorCode execute_(TensorList inputs, TensorList outputs) override {
        if (inputs[0]->count()) {
            HIPBackend::LazyOp o;
            o.kind = mul_ ? HIPBackend::LazyOp::MUL : HIPBackend::LazyOp::ADD;
            o.a = (const float *)dptr(inputs[0]); o.b = (const float *)dptr(inputs[1]); o.out = (float *)dptr(outputs[0]); o.n = (int64_t)inputs[0]->count();
            hb()->lazy(o);
        }
        return MLLM_NO_ERROR;
    }

private:
    bool mul_;
};

// ---- ops of the other model families -----------------------------------------------------------------------------------------------------------------
// SLIDINGWINDOWMASK (op/CPUSlidingWindowMask.cpp:30-58) on BSHD scores [1][heads][S][keys]; batch 1 like the rest of the adapter
class HIPSlidingWindowMaskOp final : public HIPOp {
public:
    HIPSlidingWindowMaskOp(Backend *bn, const string &name, int window) : HIPOp(bn, name), window_(window) {}
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        need(inputs[0]->batch() == 1, "SLIDINGWINDOWMASK: batch 1");
        outputs[0]->reshape(inputs[0]->batch(), inputs[0]->head(), inputs[0]->sequence(), inputs[0]->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList outputs) override { alloc_f32(outputs[0]); return MLLM_NO_ERROR; }
    ErrorCode execute_(TensorList inputs, TensorList outputs) override {
        HIPQ(mllm_hip_sliding_window_mask, (const float *)dptr(inputs[0]), (float *)dptr(outputs[0]), inputs[0]->sequence(), inputs[0]->head(), inputs[0]->dimension(), window_,
                                            hb()->stream());
        return MLLM_NO_ERROR;
    }

private:
    int window_;
};
// F_TOPK (op/CPUTopkFunc.hpp:27-92): outputs[0] = values, outputs[1] = indices (floats).  DIMENSION: [b][h][s][D] -> [b][h][s][k]; HEAD (input [1][H][S][1]) ->
// [1][k][S][1] -- in BSHD memory the rows [S][H] -> [S][k], the same launch with n = H
class HIPTopkOp final : public HIPOp {
public:
    HIPTopkOp(Backend *bn, const string &name, int k, bool head_axis) : HIPOp(bn, name), k_(k), head_(head_axis) {}
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        if (head_ && (inputs[0]->dimension() != 1 || inputs[0]->batch() != 1)) throw std::runtime_error("HIPTopkOp: the HEAD axis form takes [1][H][S][1]");
        for (int o = 0; o < 2; ++o) {
            if (head_) outputs[o]->reshape(inputs[0]->batch(), k_, inputs[0]->sequence(), 1);
            else outputs[o]->reshape(inputs[0]->batch(), inputs[0]->head(), inputs[0]->sequence(), k_);
        }
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList outputs) override { alloc_f32(outputs[0]); alloc_f32(outputs[1]); return MLLM_NO_ERROR; }
    ErrorCode execute_(TensorList inputs, TensorList outputs) override {
        const int rows = head_ ? inputs[0]->sequence() : inputs[0]->batch() * inputs[0]->head() * inputs[0]->sequence();
        const int n = head_ ? inputs[0]->head() : inputs[0]->dimension();
        HIPQ(mllm_hip_topk_rows, (const float *)dptr(inputs[0]), n, (float *)dptr(outputs[0]), (float *)dptr(outputs[1]), rows, n, k_, hb()->stream());
        return MLLM_NO_ERROR;
    }

private:
    int k_;
    bool head_;
};
// F_SCATTERADD on SEQUENCE (op/CPUScatterAddFunc.hpp:27-60): inputs = (dest [1][1][S][D], src [1][1][R][D], indices [1][1][1][R]); dest is updated in place, no outputs
class HIPScatterAddOp final : public HIPOp {
public:
    HIPScatterAddOp(Backend *bn, const string &name) : HIPOp(bn, name) {}
    ErrorCode reshape_(TensorList, TensorList) override { return MLLM_NO_ERROR; }
    ErrorCode setUp_(TensorList, TensorList) override { return MLLM_NO_ERROR; }
    ErrorCode execute_(TensorList inputs, TensorList) override {
        if (inputs[1]->batch() == 0) return MLLM_NO_ERROR;
        const int D = inputs[0]->dimension();
        HIPQ(mllm_hip_scatter_add_rows, (float *)dptr(inputs[0]), D, inputs[0]->sequence(), (const float *)dptr(inputs[1]), D, (const float *)dptr(inputs[2]), inputs[2]->dimension(), D, hb()->stream());
        return MLLM_NO_ERROR;
    }
};

// ---- SOFTMAX: CPUSoftMax (op/CPUSoftMax.cpp:28-65; axis, do_causal_mask); DIMENSION axis.  With do_causal_mask and more than one query row, row s keeps its first
// s + 1 + old_dim columns (old_dim = columns - rows, or the cache length passed as a second input minus the rows, :35-46) and the rest of the output row is zero -------
class HIPSoftMaxOp final : public HIPOp {
public:
    HIPSoftMaxOp(Backend *bn, const string &name, bool causal) : HIPOp(bn, name), causal_(causal) {}
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        need(inputs[0]->ctype() == BSHD, "SOFTMAX: BSHD scores");
        outputs[0]->reshape(inputs[0]->batch(), inputs[0]->head(), inputs[0]->sequence(), inputs[0]->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList outputs) override { alloc_f32(outputs[0]); return MLLM_NO_ERROR; }
    ErrorCode execute_(TensorList inputs, TensorList outputs) override {
        auto *b = hb();
        const int rows = rows_of(inputs[0]), n = inputs[0]->dimension(), S = inputs[0]->sequence(), H = inputs[0]->head();
        if (rows == 0) return MLLM_NO_ERROR;
        int classes = n, old_dim = n - S;
        if (inputs.size() > 1) { classes = (int)b->host_floats(inputs[1])[0]; old_dim = classes - S; }
        const int *valid = nullptr;
        if ((causal_ && S > 1) || classes != n) {
            std::vector<int> v((size_t)rows);
            for (int r = 0; r < rows; ++r) v[r] = causal_ && S > 1 ? std::min(n, (r / H) % S + 1 + old_dim) : std::min(n, classes);      // BSHD rows: r = (b*S + s)*H + h
            int *dv = (int *)b->scratch(5, v.size() * 4);
            b->upload(dv, v.data(), v.size() * 4);
            valid = dv;
        }
        HIPQ(mllm_hip_softmax, (const float *)dptr(inputs[0]), (float *)dptr(outputs[0]), rows, n, valid, b->stream());
        return MLLM_NO_ERROR;
    }

private:
    bool causal_;
};

// ---- CONVOLUTION3D (Qwen2-VL patch embed, kernel == stride, VALID, no bias: op/CPUConvolution3D.cpp:56-100) and CONVOLUTION2D (ViT / CLIP patch embed,
// kernel == stride, VALID: op/CPUConvolution2D.cpp:29-149) as GEMMs over the flattened receptive fields (compute/Convolution.cpp:35-82,179-235) ---------------------
class HIPPatchConvOp final : public HIPOp {
public:
    HIPPatchConvOp(Backend *bn, const string &name, bool is3d, int in_ch, int out_ch, int kt, int kh, int kw, bool bias) :
        HIPOp(bn, name), is3d_(is3d), in_ch_(in_ch), out_ch_(out_ch), kt_(kt), kh_(kh), kw_(kw), has_bias_(bias) {}
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        if (is3d_) {
            need(inputs[0]->ctype() == BCTHW && inputs[0]->channel() == in_ch_ && inputs[0]->time() == kt_ && inputs[0]->height() == kh_ && inputs[0]->width() == kw_,
                 "CONVOLUTION3D: one receptive field per batch entry, [N, C, kt, kh, kw]");
            outputs[0]->reshape(inputs[0]->batch(), out_ch_, 1, 1, 1);            // [N, OC, 1, 1, 1], viewed [1,1,N,OC] by the model (modeling_qwen2_vl.hpp:31-35)
        } else {
            need(inputs[0]->batch() == 1 && inputs[0]->sequence() == in_ch_ && inputs[0]->head() % kh_ == 0 && inputs[0]->dimension() % kw_ == 0, "CONVOLUTION2D: image [1, H, C, W], H and W multiples of the patch");
            outputs[0]->reshape(inputs[0]->batch(), inputs[0]->head() / kh_, out_ch_, inputs[0]->dimension() / kw_);      // image [B, H, C, W] -> [B, H/p, OC, W/p]
        }
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList outputs) override { alloc_f32(outputs[0]); return MLLM_NO_ERROR; }
    ErrorCode load(AbstructLoader &loader) override {
        load_tensor(weight_, backend_, loader, name() + ".weight", out_ch_, in_ch_ * kt_ * kh_ * kw_);
        need(weight_.dtype() == MLLM_TYPE_F32, "patch-embedding convolutions are fp32 in *-q4_k.mllm files");
        if (has_bias_) load_tensor(bias_, backend_, loader, name() + ".bias", 1, out_ch_);
        return MLLM_NO_ERROR;
    }
    ErrorCode execute_(TensorList inputs, TensorList outputs) override {
        auto *b = hb();
        const float *w = (const float *)weight_.device_memory().handle, *bias = has_bias_ ? (const float *)bias_.device_memory().handle : nullptr;
        const int KK = in_ch_ * kt_ * kh_ * kw_;
        if (is3d_) {
            if (inputs[0]->batch()) HIPQ(mllm_hip_patch_gemm_f32, (const float *)dptr(inputs[0]), w, bias, (float *)dptr(outputs[0]), inputs[0]->batch(), KK, out_ch_, b->stream());
        } else {
            const int H = inputs[0]->head(), W = inputs[0]->dimension(), N = (H / kh_) * (W / kw_);
            float *patches = (float *)b->scratch(3, (size_t)N * KK * 4), *rows = (float *)b->scratch(2, (size_t)N * out_ch_ * 4);
            HIPQ(mllm_hip_im2patch_chw, (const float *)dptr(inputs[0]), patches, H, in_ch_, W, kh_, b->stream());      // the image Tensor [1, H, C, W] is BSHD: [C][H][W] in memory
            // rows [oh*ow][OC] -> the reference's output [B, H/p, OC, W/p], which in BSHD memory order is [OC][oh][ow]
            HIPQ(mllm_hip_patch_gemm_f32, patches, w, bias, rows, N, KK, out_ch_, b->stream());
            HIPQ(mllm_hip_transpose_f32, rows, (float *)dptr(outputs[0]), N, out_ch_, b->stream());
        }
        return MLLM_NO_ERROR;
    }

private:
    bool is3d_;
    int in_ch_, out_ch_, kt_, kh_, kw_;
    bool has_bias_;
    Tensor weight_, bias_;
};

// ---- metadata functions: no kernel, the output aliases the input's block ------------------------------------------------------------------------
// F_VIEW (Tensor::view, op/CPUViewFunc.hpp:30-139): in each accepted pattern -1 means "this axis keeps its extent" and the two named axes regroup; every pattern is a
// reinterpretation of the contiguous BSHD (or BCTHW) buffer.  Patterns the reference rejects with exit(-2) are rejected here with an exception.
class HIPViewOp final : public HIPOp {
public:
    HIPViewOp(Backend *bn, const string &name, int b, int h, int s, int d) : HIPOp(bn, name), b_(b), h_(h), s_(s), d_(d) {}
    bool keeps_shadow() const override { return true; }
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        auto &in = inputs[0];
        const bool five = in->ctype() == BCTHW;
        const int64_t n = in->count();
        int B = in->batch(), H = five ? 1 : in->head(), S = five ? 1 : in->sequence(), D = five ? 1 : in->dimension();
        auto pick = [&](int want, int other, int64_t prod) -> int {      // one of two regrouped axes: given, or what the other leaves (ANYDIM)
            return want != ANYDIM ? want : (int)(prod / other);
        };
        if (b_ == -1 && h_ == 1 && s_ == 1 && d_ == -1 && !five) { D = S * H * D; S = 1; H = 1; }
        else if (b_ == -1 && h_ == -1 && s_ == 1 && d_ == 1 && !five) { S = S * H * D; H = 1; D = 1; }
        else if (b_ == 1 && h_ == 1 && s_ == -1 && d_ > 0) { B = 1; H = 1; D = d_; S = (int)(n / d_); }      // everything but the row width folds into the sequence (also the BCTHW patch rows)
        else if (b_ == -1 && h_ != -1 && s_ == -1 && d_ != -1 && !five) {      // head & dimension
            const int64_t hd = (int64_t)H * D;
            need(h_ != ANYDIM || d_ != ANYDIM, "F_VIEW: head and dimension both open");
            H = pick(h_, d_, hd); D = pick(d_, h_, hd);
            need((int64_t)H * D == hd, "F_VIEW: head * dimension must be kept");
        } else if (b_ == -1 && h_ != -1 && s_ != -1 && d_ == -1 && !five) {    // head & sequence
            const int64_t hs = (int64_t)H * S;
            need(h_ != ANYDIM || s_ != ANYDIM, "F_VIEW: head and sequence both open");
            H = pick(h_, s_, hs); S = pick(s_, h_, hs);
            need((int64_t)H * S == hs, "F_VIEW: head * sequence must be kept");
        } else if (b_ != -1 && h_ == -1 && s_ != -1 && d_ == -1 && !five) {    // batch & sequence
            const int64_t bs = (int64_t)B * S;
            need(b_ != ANYDIM || s_ != ANYDIM, "F_VIEW: batch and sequence both open");
            B = pick(b_, s_, bs); S = pick(s_, b_, bs);
            need((int64_t)B * S == bs, "F_VIEW: batch * sequence must be kept");
        } else {
            throw std::runtime_error("F_VIEW [" + std::to_string(b_) + ", " + std::to_string(h_) + ", " + std::to_string(s_) + ", " + std::to_string(d_) + "] is not one of the reference's patterns");
        }
        need((int64_t)B * H * S * D == n, "F_VIEW: element count must be kept");
        outputs[0]->reshape(B, H, S, D);      // a fresh shell: BSHD
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList inputs, TensorList outputs) override {
        outputs[0]->setDtype(inputs[0]->dtype());
        hb()->view_of(outputs[0], dptr(inputs[0]), inputs[0]->device_memory().size_in_bytes);
        return MLLM_NO_ERROR;
    }

private:
    int b_, h_, s_, d_;
};

// F_CLIP (op/CPUClipFunc.hpp) on the SEQUENCE axis of a batch-1 BSHD tensor: clip({}, {}, {-1}, {}) = the last position (every causal LM ends with it,
// modeling_qwen2_vl.hpp:395-397), clip({}, {}, {a, b}, {}) = positions [a, b) (LLaVA drops the class row, modeling_llava.hpp:90).  Rows of a position are contiguous,
// so the result is a pointer offset into the input.
class HIPClipSeqOp final : public HIPOp {
public:
    HIPClipSeqOp(Backend *bn, const string &name, int a, int b, bool single) : HIPOp(bn, name), a_(a), b_(b), single_(single) {}
    bool keeps_shadow() const override { return true; }
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        const int S = inputs[0]->sequence();
        lo_ = a_ < 0 ? S + a_ : a_;
        hi_ = single_ ? lo_ + 1 : (b_ < 0 ? S + b_ : b_);
        if (inputs[0]->batch() != 1 || lo_ < 0 || hi_ > S || hi_ <= lo_) throw std::runtime_error("HIPClipSeqOp: range outside the sequence");
        outputs[0]->reshape(1, inputs[0]->head(), hi_ - lo_, inputs[0]->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList inputs, TensorList outputs) override {
        outputs[0]->setDtype(inputs[0]->dtype());
        const size_t row = (size_t)inputs[0]->head() * inputs[0]->dimension() * elem_bytes(inputs[0]->dtype());
        hb()->view_of(outputs[0], (char *)dptr(inputs[0]) + (size_t)lo_ * row, (size_t)(hi_ - lo_) * row);
        return MLLM_NO_ERROR;
    }

private:
    int a_, b_, lo_ = 0, hi_ = 0;
    bool single_;
};

// F_TRANPOSE (Tensor::transpose, op/CPUTransposeFunc.hpp): the (SEQUENCE, DIMENSION) swap is a flip of the axis map on shared memory in the reference too
// (:31-84 -- the producer is re-pointed, nothing moves); it is how a weight becomes the right operand of Tensor::mm (`lm_head().transpose(SEQUENCE, DIMENSION)`).
class HIPTransposeOp final : public HIPOp {
public:
    HIPTransposeOp(Backend *bn, const string &name, Chl a, Chl b) : HIPOp(bn, name), a_(a), b_(b) {}
    bool keeps_shadow() const override { return true; }
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        need(inputs[0]->ctype() == BSHD, "F_TRANPOSE: BSHD input");
        outputs[0]->transCopyShape(inputs[0]->shape());
        outputs[0]->chls() = inputs[0]->chls();
        std::swap(outputs[0]->chls()[a_], outputs[0]->chls()[b_]);
        outputs[0]->changeCtype((int)inputs[0]->shape().size());
        outputs[0]->undiffusion() = true;
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList inputs, TensorList outputs) override {
        outputs[0]->setDtype(inputs[0]->dtype());
        hb()->view_of(outputs[0], dptr(inputs[0]), inputs[0]->device_memory().size_in_bytes);
        return MLLM_NO_ERROR;
    }

private:
    Chl a_, b_;
};

// F_TRANPOSE with the pair list {(SEQUENCE, DIMENSION), (HEAD, SEQUENCE)}: how ViTEmbedding / LLaVAVisionEmbedding turn the patch convolution's output [1, oh, OC, ow]
// (BSHD memory [OC][oh][ow]) into patch rows (models/vit/modeling_vit.hpp:78, models/llava/modeling_llava.hpp:54).  On the CPU it is a flip of the axis map the
// producer then writes through (op/CPUTransposeFunc.hpp:53-86,106-119: ctype BDSH); the device keeps every activation contiguous BSHD, so here the data moves:
// out (head = ow, sequence = oh, dimension = OC), memory [oh][ow][OC] = the transpose of [OC][oh * ow].
class HIPPatchRowsTransposeOp final : public HIPOp {
public:
    HIPPatchRowsTransposeOp(Backend *bn, const string &name) : HIPOp(bn, name) {}
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        need(inputs[0]->ctype() == BSHD && inputs[0]->batch() == 1 && inputs[0]->dtype() == MLLM_TYPE_F32, "F_TRANPOSE {(S,D),(H,S)}: one fp32 BSHD image's patch grid");
        outputs[0]->reshape(1, inputs[0]->dimension(), inputs[0]->head(), inputs[0]->sequence());
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList outputs) override { alloc_f32(outputs[0]); return MLLM_NO_ERROR; }
    ErrorCode execute_(TensorList inputs, TensorList outputs) override {
        const int OC = inputs[0]->sequence(), N = inputs[0]->head() * inputs[0]->dimension();
        if (OC && N) HIPQ(mllm_hip_transpose_f32, (const float *)dptr(inputs[0]), (float *)dptr(outputs[0]), OC, N, hb()->stream());
        return MLLM_NO_ERROR;
    }
};

// F_FLATTEN (Tensor::flatten, op/CPUFlattenFunc.hpp:249-332; always called in place, Tensor.cpp:506-512): two neighbouring axes of a contiguous BSHD tensor become
// one -- (HEAD, SEQUENCE): [b, h, s, d] -> [b, 1, s*h, d] (position s*H + h: the memory order); (HEAD, DIMENSION): -> [b, 1, s, h*d]; (BATCH, SEQUENCE) with one head:
// -> [1, 1, b*s, d].  No data moves, as on the CPU (:297-300).
class HIPFlattenOp final : public HIPOp {
public:
    HIPFlattenOp(Backend *bn, const string &name, Chl a, Chl b) : HIPOp(bn, name), a_(a), b_(b) {}
    bool keeps_shadow() const override { return true; }
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        auto &in = inputs[0];
        need(in->ctype() == BSHD, "F_FLATTEN: contiguous BSHD input (the 5-D forms are not on the five configs' path)");
        int B = in->batch(), H = in->head(), S = in->sequence(), D = in->dimension();
        if (a_ == HEAD && b_ == SEQUENCE) { S *= H; H = 1; }
        else if (a_ == HEAD && b_ == DIMENSION) { D *= H; H = 1; }
        else if (a_ == BATCH && b_ == SEQUENCE && H == 1) { S *= B; B = 1; }
        else throw std::runtime_error("F_FLATTEN: axis pair not on this backend");
        outputs[0]->reshape(B, H, S, D);
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList inputs, TensorList outputs) override {
        if (outputs[0].get() != inputs[0].get()) {
            outputs[0]->setDtype(inputs[0]->dtype());
            hb()->view_of(outputs[0], dptr(inputs[0]), inputs[0]->device_memory().size_in_bytes);
        }
        return MLLM_NO_ERROR;
    }

private:
    Chl a_, b_;
};

// F_CAT on SEQUENCE (Tensor::cat, op/CPUCatFunc.hpp:611-624: the one-head branch is a memcpy per input): the class row in front of the patch rows
// (`Tensor::cat({cls_token(), embd}, SEQUENCE)`, modeling_vit.hpp:80, modeling_llava.hpp:56).  fp32 [1, 1, s_i, D] inputs; one pitched copy each.
class HIPCatSeqOp final : public HIPOp {
public:
    HIPCatSeqOp(Backend *bn, const string &name) : HIPOp(bn, name) {}
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        int S = 0;
        for (auto &in : inputs) {
            need(in->batch() == 1 && in->head() == 1 && in->dimension() == inputs[0]->dimension() && in->dtype() == MLLM_TYPE_F32 && in->dimension() % 4 == 0,
                 "F_CAT(SEQUENCE): fp32 [1, 1, s, D] inputs of one width (a multiple of 4)");
            S += in->sequence();
        }
        outputs[0]->reshape(1, 1, S, inputs[0]->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList outputs) override { alloc_f32(outputs[0]); return MLLM_NO_ERROR; }
    ErrorCode execute_(TensorList inputs, TensorList outputs) override {
        const int D = inputs[0]->dimension();
        float *dst = (float *)dptr(outputs[0]);
        for (auto &in : inputs) {
            if (in->sequence()) HIPQ(mllm_hip_copy_2d_f32, (const float *)dptr(in), D, dst, D, in->sequence(), D, hb()->stream());
            dst += (size_t)in->sequence() * D;
        }
        return MLLM_NO_ERROR;
    }
};

// F_SPLIT on DIMENSION (Tensor::split, op/CPUSplitFunc.hpp:32-172): the reference scatters the producer straight into the parts ("aggregated" children) or runs
// efficient_split; here one pitched copy per part (precedent opencl/kernel/split.cl).
class HIPSplitOp final : public HIPOp {
public:
    HIPSplitOp(Backend *bn, const string &name, std::vector<int> each) : HIPOp(bn, name), each_(std::move(each)) {}
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        int total = 0;
        for (int e : each_) total += e;
        need(outputs.size() == each_.size() && total == inputs[0]->dimension() && inputs[0]->dtype() == MLLM_TYPE_F32, "F_SPLIT: the parts must add up to the fp32 input's dimension");
        for (size_t i = 0; i < each_.size(); ++i) {
            need(each_[i] % 4 == 0, "F_SPLIT: parts of a multiple of 4 columns");
            outputs[i]->reshape(inputs[0]->batch(), inputs[0]->head(), inputs[0]->sequence(), each_[i]);
        }
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList outputs) override { for (auto &o : outputs) alloc_f32(o); return MLLM_NO_ERROR; }
    ErrorCode execute_(TensorList inputs, TensorList outputs) override {
        const int rows = rows_of(inputs[0]), ld = inputs[0]->dimension();
        int off = 0;
        for (size_t i = 0; i < each_.size(); ++i) {
            if (rows) HIPQ(mllm_hip_copy_2d_f32, (const float *)dptr(inputs[0]) + off, ld, (float *)dptr(outputs[i]), each_[i], rows, each_[i], hb()->stream());
            off += each_[i];
        }
        return MLLM_NO_ERROR;
    }

private:
    std::vector<int> each_;
};

// ---- F_MM: CPUmmFunction (op/CPUMatmulFunc.hpp:86-181).  The form on the hot path: activations [1,1,M,K] times a transposed weight view (axis map flipped by
// F_TRANPOSE, chls[SEQUENCE] == 3) = the tied lm_head `Tensor::mm(x, lm_head().transpose(SEQUENCE, DIMENSION))` (modeling_qwen2_vl.hpp:399): mat_mul with a Q4_0
// right operand quantises x to Q8_0 and runs vec_dot_q4_0_q8_0 (Matmul.cpp:77-120,219-276); an fp32 table runs vec_dot_fp32 -----------------------------------------
class HIPMatmulOp final : public HIPOp {
public:
    HIPMatmulOp(Backend *bn, const string &name) : HIPOp(bn, name) {}
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        auto &x = inputs[0], &w = inputs[1];
        need(x->ctype() == BSHD && x->dtype() == MLLM_TYPE_F32 && x->head() == 1, "F_MM: fp32 BSHD activations [b,1,s,K]");
        need(w->chls()[SEQUENCE] == 3 && w->batch() == 1 && w->head() == 1 && (w->dtype() == MLLM_TYPE_Q4_0 || w->dtype() == MLLM_TYPE_F32),
             "F_MM: the right operand must be a transposed [1,1,K,N] view of a Q4_0 or fp32 weight (the eager-attention forms are not on this backend)");
        need(x->dimension() == w->sequence(), "F_MM: inner extents differ");
        if (w->dtype() == MLLM_TYPE_Q4_0) need(hb()->q40_table_at(dptr(w)) != nullptr, "F_MM: Q4_0 operand that is not a PARAMETER's table");
        outputs[0]->reshape(x->batch(), x->head(), x->sequence(), w->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList outputs) override { alloc_f32(outputs[0]); return MLLM_NO_ERROR; }
    ErrorCode execute_(TensorList inputs, TensorList outputs) override {
        auto *b = hb();
        auto &x = inputs[0], &w = inputs[1];
        const int M = rows_of(x), K = x->dimension(), N = w->dimension();
        if (M == 0) return MLLM_NO_ERROR;
        if (w->dtype() == MLLM_TYPE_Q4_0) {
            auto t = b->q40_table_at(dptr(w));
            uint8_t *ws = (uint8_t *)b->scratch(0, mllm_hip_linear_workspace_bytes(MLLM_HIP_Q4_0, M, K));
            int8_t *qs = (int8_t *)ws;
            uint16_t *d = (uint16_t *)(ws + (((size_t)M * K + 255) & ~(size_t)255));
            HIPQ(mllm_hip_quantize_q80, (const float *)dptr(x), qs, d, M, K, b->stream());
            HIPQ(mllm_hip_linear_q40_q80, (const uint8_t *)t->qs, (const uint16_t *)t->d, nullptr, qs, d, (float *)dptr(outputs[0]), N, M, N, K, b->stream());
        } else {
            HIPQ(mllm_hip_linear_f32, (const float *)dptr(w), nullptr, (const float *)dptr(x), (float *)dptr(outputs[0]), N, M, N, K, b->stream());
        }
        return MLLM_NO_ERROR;
    }
};

// ---- F_WHERE (Tensor::where, op/CPUWhereFunc.hpp; value, axis) + F_INDEX_PUT (Tensor::index_put, op/CPUIndexPutFunc.hpp:25-92; accumulate = false): the splice of
// the visual rows into the text embeddings (modeling_qwen2_vl.hpp:386-389).  The count of matches shapes the output, so the ids are read on the host (their upload's
// shadow; no device round trip) and the index tensor goes up as floats, which is what index_put receives in the reference too --------------------------------------
class HIPWhereOp final : public HIPOp {
public:
    HIPWhereOp(Backend *bn, const string &name, float value, int axis) : HIPOp(bn, name), value_(value), axis_(axis) {}
    bool keeps_shadow() const override { return true; }
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        auto &in = inputs[0];
        const std::vector<float> &v = hb()->host_floats(in);
        for (auto &x : idx_) x.clear();
        const int B = in->batch(), S = in->sequence(), H = in->head(), D = in->dimension();
        size_t i = 0;      // BSHD memory order, which is also the reference's visiting order (b, s, h, d)
        for (int b = 0; b < B; ++b) for (int s = 0; s < S; ++s) for (int h = 0; h < H; ++h) for (int d = 0; d < D; ++d, ++i)
            if (v[i] == value_) { idx_[0].push_back((float)b); idx_[1].push_back((float)h); idx_[2].push_back((float)s); idx_[3].push_back((float)d); }
        const int num = (int)idx_[0].size();
        if (axis_ == -1) outputs[0]->reshape(1, 1, 4, num);
        else outputs[0]->reshape(1, 1, 1, num);
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList outputs) override { alloc_f32(outputs[0]); return MLLM_NO_ERROR; }
    ErrorCode execute_(TensorList, TensorList outputs) override {
        const size_t num = idx_[0].size();
        if (num == 0) return MLLM_NO_ERROR;
        std::vector<float> flat;
        if (axis_ == -1) for (auto &x : idx_) flat.insert(flat.end(), x.begin(), x.end());      // rows b, h, s, d
        else flat = idx_[axis_ == BATCH ? 0 : axis_ == HEAD ? 1 : axis_ == SEQUENCE ? 2 : 3];
        hb()->upload(dptr(outputs[0]), flat.data(), flat.size() * 4);
        hb()->remember_host(dptr(outputs[0]), flat.data(), flat.size());
        return MLLM_NO_ERROR;
    }

private:
    float value_;
    int axis_;
    std::vector<float> idx_[4];
};
class HIPIndexPutOp final : public HIPOp {
public:
    HIPIndexPutOp(Backend *bn, const string &name) : HIPOp(bn, name) {}
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        if (inputs.size() > 1 && inputs[1]->batch() == 0) return MLLM_NO_ERROR;      // no image: the destination passes through (:44-53)
        need(inputs.size() == 3 && outputs[0].get() == inputs[0].get(), "F_INDEX_PUT: (dest, value, indices), in place");
        auto &dst = inputs[0], &src = inputs[1], &idx = inputs[2];
        need(dst->batch() == 1 && dst->head() == 1 && src->head() == 1 && dst->dimension() == src->dimension() && dst->dtype() == MLLM_TYPE_F32 && src->dtype() == MLLM_TYPE_F32,
             "F_INDEX_PUT: fp32 rows of one width, batch 1, one head");
        need(idx->dimension() <= src->batch() * src->sequence(), "F_INDEX_PUT: more indices than value rows");
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList) override { return MLLM_NO_ERROR; }
    ErrorCode execute_(TensorList inputs, TensorList) override {
        if (inputs.size() > 1 && inputs[1]->batch() == 0) return MLLM_NO_ERROR;
        HIPQ(mllm_hip_index_put_rows_fidx, (float *)dptr(inputs[0]), inputs[0]->sequence(), (const float *)dptr(inputs[1]), (const float *)dptr(inputs[2]), inputs[2]->dimension(),
                                            inputs[0]->dimension(), hb()->stream());
        return MLLM_NO_ERROR;
    }
};
// F_INDEX_PUT with accumulate = true (op/CPUIndexPutFunc.hpp:61-69,93-121): the LLaVA splice -- the ONE row of `dest` that holds the <image> id is replaced by all
// rows of `value` [1, 1, R, D], so the sequence grows by R - 1 (`embd.index_put(vision, where_idx, true)`, modeling_llava.hpp:131).  A fresh output, three pitched
// copies: the rows before the marker, the visual rows, the rows behind it.  The index is read from the host shadow of F_WHERE's upload (it shapes nothing here, but
// the copies need it as a host integer).  One image per call: with more the reference's loop re-reads the first image's rows under a racing `omp parallel for`.
class HIPIndexPutGrowOp final : public HIPOp {
public:
    HIPIndexPutGrowOp(Backend *bn, const string &name) : HIPOp(bn, name) {}
    ErrorCode reshape_(TensorList inputs, TensorList outputs) override {
        need(inputs.size() == 3 && inputs[1]->batch() == 1 && inputs[2]->dimension() == 1, "F_INDEX_PUT(accumulate): (dest, one image's rows, one index)");
        auto &dst = inputs[0], &src = inputs[1];
        need(dst->batch() == 1 && dst->head() == 1 && src->head() == 1 && dst->dimension() == src->dimension() && dst->dimension() % 4 == 0 && dst->dtype() == MLLM_TYPE_F32 &&
                 src->dtype() == MLLM_TYPE_F32, "F_INDEX_PUT(accumulate): fp32 rows of one width (a multiple of 4), batch 1, one head");
        at_ = (int)hb()->host_floats(inputs[2])[0];
        need(at_ >= 0 && at_ < dst->sequence(), "F_INDEX_PUT(accumulate): index outside the destination");
        outputs[0]->reshape(1, 1, dst->sequence() - 1 + src->sequence(), dst->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp_(TensorList, TensorList outputs) override { alloc_f32(outputs[0]); return MLLM_NO_ERROR; }
    ErrorCode execute_(TensorList inputs, TensorList outputs) override {
        const int D = inputs[0]->dimension(), S = inputs[0]->sequence(), R = inputs[1]->sequence();
        const float *dst = (const float *)dptr(inputs[0]), *src = (const float *)dptr(inputs[1]);
        float *out = (float *)dptr(outputs[0]);
        void *st = hb()->stream();
        if (at_) HIPQ(mllm_hip_copy_2d_f32, dst, D, out, D, at_, D, st);
        if (R) HIPQ(mllm_hip_copy_2d_f32, src, D, out + (size_t)at_ * D, D, R, D, st);
        if (S - at_ - 1 > 0) HIPQ(mllm_hip_copy_2d_f32, dst + (size_t)(at_ + 1) * D, D, out + (size_t)(at_ + R) * D, D, S - at_ - 1, D, st);
        return MLLM_NO_ERROR;
    }

private:
    int at_ = 0;
};

}  // namespace

// ---- registry: OpType -> creator (Backend::registerOps, mllm/Backend.hpp:104-105; OpDefined.hpp:10-134).  Anything not listed, or a listed Op with parameters the
// launchers do not cover, makes opCreate return nullptr: a Layer then runs on the CPU backend with automatic tensor migration (Layer.hpp:128-135,159-163); a Tensor
// function has no such migration in the reference (Tensor.cpp:402-406 only swaps the Op), so every function the five configs call on device tensors is listed ----
void HIPBackend::registerOps() {
    creators_[LINEAR] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        const int in = geti(p, "in_features"), out = geti(p, "out_features");
        if (in <= 0 || out <= 0 || in % 256) return nullptr;      // Q4_K super-blocks; other K go to the CPU
        return new HIPLinearOp(b, n, in, out, geti(p, "bias") != 0);
    };
    creators_[EMBEDDING] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        const int h = geti(p, "hidden_size");
        return h % 32 ? nullptr : new HIPEmbeddingOp(b, n, h, geti(p, "vocab_size"));
    };
    creators_[PARAMETER] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        return new HIPParameterOp(b, n, geti(p, "batch"), geti(p, "head"), geti(p, "seq"), geti(p, "dim"));
    };
    creators_[RMSNORM] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        return new HIPNormOp(b, n, false, geti(p, "norm_size"), getf(p, "epsilon", 1e-6f), false, geti(p, "add_unit_offset") != 0);
    };
    creators_[LAYERNORM] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        return new HIPNormOp(b, n, true, geti(p, "norm_size"), getf(p, "epsilon", 1e-6f), geti(p, "bias") != 0, false);
    };
    creators_[ROPE] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        if (geti(p, "pose_type") != (int)HFHUBROPE || p.count("rope_type") || getf(p, "partial_rotary_factor", 1.f) != 1.f) return nullptr;      // the half-split table of the five configs
        return new HIPRoPEOp(b, n, false, getf(p, "rope_theta", 10000.f), geti(p, "max_position_embeddings", 16384), {});
    };
    creators_[MULTIMODALROPE] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        std::vector<int> sec;
        for (int i = 0; i < 3; ++i) { auto it = p.find("mrope_section_" + std::to_string(i)); if (it != p.end()) sec.push_back((int)it->second); }
        if (sec.empty()) sec = {16, 24, 24};
        return new HIPRoPEOp(b, n, true, getf(p, "rope_theta", 1000000.f), geti(p, "max_position_embeddings", 32768), sec);
    };
    creators_[VISIONROPE] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        const int dim = geti(p, "dim"), merge = geti(p, "spatial_merge_size");
        return dim < 2 || merge < 1 ? nullptr : new HIPVisionRoPEOp(b, n, dim, merge);
    };
    creators_[F_APPLY_VISIOROPE] = [](HIPBackend *b, const OpParam &, const std::string &n) -> Op * { return new HIPApplyVisionRoPEOp(b, n); };
    creators_[KVCACHE] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        if (!geti(p, "fa2") || geti(p, "for_xnn")) return nullptr;      // FlashAttention2 mode (the default attn_implementation): fp16 slab, no head repeat whatever n_rep says (CPUKVCache.cpp:41-43)
        return new HIPKVCacheOp(b, n, geti(p, "cache_max", 100));
    };
    creators_[F_FA2] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * { return new HIPFlashAttention2Op(b, n, geti(p, "causal_mask") != 0); };
    creators_[SILU] = [](HIPBackend *b, const OpParam &, const std::string &n) -> Op * { return new HIPUnaryOp(b, n, HIPUnaryOp::SILU_K); };
    creators_[OP_GELU] = [](HIPBackend *b, const OpParam &, const std::string &n) -> Op * { return new HIPUnaryOp(b, n, HIPUnaryOp::GELU_K); };
    creators_[QUICKGLUE] = [](HIPBackend *b, const OpParam &, const std::string &n) -> Op * { return new HIPUnaryOp(b, n, HIPUnaryOp::QUICKGELU_K); };
    creators_[F_TTADD] = [](HIPBackend *b, const OpParam &, const std::string &n) -> Op * { return new HIPBinaryOp(b, n, false); };
    creators_[F_TTMUL] = [](HIPBackend *b, const OpParam &, const std::string &n) -> Op * { return new HIPBinaryOp(b, n, true); };
    creators_[SOFTMAX] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        return geti(p, "axis") != (int)DIMENSION ? nullptr : new HIPSoftMaxOp(b, n, geti(p, "do_causal_mask") != 0);
    };
    creators_[CONVOLUTION3D] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        if (geti(p, "kernal_t") != geti(p, "stride_t") || geti(p, "kernal_h") != geti(p, "stride_h") || geti(p, "kernal_w") != geti(p, "stride_w") || geti(p, "padding") != (int)VALID) return nullptr;
        return new HIPPatchConvOp(b, n, true, geti(p, "in_channel"), geti(p, "out_channel"), geti(p, "kernal_t"), geti(p, "kernal_h"), geti(p, "kernal_w"), geti(p, "bias") != 0);
    };
    creators_[CONVOLUTION2D] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        if (geti(p, "kernal_h") != geti(p, "stride_h") || geti(p, "kernal_w") != geti(p, "stride_w") || geti(p, "kernal_h") != geti(p, "kernal_w") || geti(p, "padding") != (int)VALID) return nullptr;
        return new HIPPatchConvOp(b, n, false, geti(p, "in_channel"), geti(p, "out_channel"), 1, geti(p, "kernal_h"), geti(p, "kernal_w"), geti(p, "bias") != 0);
    };
    creators_[SLIDINGWINDOWMASK] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * { return new HIPSlidingWindowMaskOp(b, n, geti(p, "window_size")); };
    creators_[F_TOPK] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        const Chl dim = (Chl)geti(p, "dim");
        return dim == DIMENSION || dim == HEAD ? new HIPTopkOp(b, n, geti(p, "k"), dim == HEAD) : nullptr;
    };
    creators_[F_SCATTERRADD] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        return p.count("dim") && (Chl)geti(p, "dim") != SEQUENCE ? nullptr : new HIPScatterAddOp(b, n);
    };
    creators_[F_CLIP] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        // SEQUENCE-only clips; anything touching batch / head / dimension is not on this backend
        if (geti(p, "b_size") || geti(p, "h_size") || geti(p, "d_size")) return nullptr;
        const int ss = geti(p, "s_size");
        if (ss == 1) return new HIPClipSeqOp(b, n, geti(p, "s_0"), 0, true);
        if (ss == 2) return new HIPClipSeqOp(b, n, geti(p, "s_0"), geti(p, "s_1"), false);
        return nullptr;
    };
    creators_[F_VIEW] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * { return new HIPViewOp(b, n, geti(p, "b"), geti(p, "h"), geti(p, "s"), geti(p, "d")); };
    creators_[F_TRANPOSE] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        if (geti(p, "num_pairs") == 2 && (Chl)geti(p, "axis1_0") == SEQUENCE && (Chl)geti(p, "axis2_0") == DIMENSION && (Chl)geti(p, "axis1_1") == HEAD && (Chl)geti(p, "axis2_1") == SEQUENCE)
            return new HIPPatchRowsTransposeOp(b, n);      // the patch-grid -> patch-rows form of the ViT / CLIP embeddings
        if (geti(p, "num_pairs") != 1) return nullptr;
        const Chl a = (Chl)geti(p, "axis1_0"), c = (Chl)geti(p, "axis2_0");
        const bool sd = (a == SEQUENCE && c == DIMENSION) || (a == DIMENSION && c == SEQUENCE);
        return sd ? new HIPTransposeOp(b, n, a, c) : nullptr;      // the (HEAD, SEQUENCE) transposes of the eager-attention branch move data: not on this backend
    };
    creators_[F_FLATTEN] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        const Chl a = (Chl)geti(p, "axis_start"), c = (Chl)geti(p, "axis_end");
        const bool ok = (a == HEAD && (c == SEQUENCE || c == DIMENSION)) || (a == BATCH && c == SEQUENCE);
        return ok ? new HIPFlattenOp(b, n, a, c) : nullptr;
    };
    creators_[F_CAT] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * { return (Chl)geti(p, "axis") == SEQUENCE ? new HIPCatSeqOp(b, n) : nullptr; };
    creators_[F_SPLIT] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        if ((Chl)geti(p, "split_dim") != DIMENSION) return nullptr;      // HD / D_HD splits of fused in_proj layouts (Chl::HD, Types.hpp:139-140): not on the five configs' path
        std::vector<int> each;
        for (int i = 0; i < geti(p, "num_splits"); ++i) each.push_back(geti(p, ("dim_" + std::to_string(i)).c_str()));
        return each.empty() ? nullptr : new HIPSplitOp(b, n, each);
    };
    creators_[F_MM] = [](HIPBackend *b, const OpParam &, const std::string &n) -> Op * { return new HIPMatmulOp(b, n); };
    creators_[F_WHERE] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * { return new HIPWhereOp(b, n, getf(p, "value"), geti(p, "axis", -1)); };
    creators_[F_INDEX_PUT] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        if (geti(p, "accumulate")) return new HIPIndexPutGrowOp(b, n);
        return new HIPIndexPutOp(b, n);
    };
}

}  // namespace mllm
