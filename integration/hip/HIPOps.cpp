// integration/hip/HIPOps.cpp -- the HIP*Op classes: mllm::Op subclasses whose execute() is one call into the C ABI (include/mllm_hip.h).
// Each class names the CPU Op it stands in for (mllm/backends/cpu/op/...) and the Layer that creates it with its OpParam keys (mllm/Layer.hpp).
// Shapes a launcher does not cover are refused at opCreate time (nullptr => CPU fallback, mllm/Layer.hpp:214-218), never at execute.
// Compiled against the reference's headers by oracle/Makefile.ref (test infrastructure; see HIPBackend.hpp).
#include <cmath>
#include <cstring>

#include "HIPBackend.hpp"
#include "ParamLoader.hpp"

namespace mllm {

namespace {

inline HIPBackend *hb(Backend *b) { return static_cast<HIPBackend *>(b); }
inline int geti(const OpParam &p, const char *k, int def = 0) { auto it = p.find(k); return it == p.end() ? def : (int)it->second; }
inline float getf(const OpParam &p, const char *k, float def = 0.f) { auto it = p.find(k); return it == p.end() ? def : it->second; }
inline int rows_of(const shared_ptr<Tensor> &t) { return t->batch() * t->sequence() * t->head(); }       // BSHD: rows of `dimension()` values
#define HIPCHK(call) HIPBackend::check((call), #call)

// loads `<op>.weight` / `<op>.bias` onto the device in the file's storage dtype (ParamLoader::load -> Backend::load_from_file fast path)
void load_tensor(Tensor &t, Backend *bn, AbstructLoader &loader, const string &name, int rows, int cols) {
    t.setName(name);
    t.setBackend(bn);
    t.reshape(1, 1, rows, cols);
    t.setDtype(loader.getDataType(name));
    t.alloc();
    loader.load(&t);
}

// ---- LINEAR: CPULinear (op/CPULinear.cpp:23-234); params in_features, out_features, bias (Layer.hpp:230-234) -----------------------------------------------
class HIPLinearOp final : public Op {
public:
    HIPLinearOp(Backend *bn, const string &name, int in, int out, bool bias) : Op(bn, name), in_(in), out_(out), has_bias_(bias) {}
    ErrorCode reshape(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        outputs[0]->reshape(inputs[0]->batch(), inputs[0]->head(), inputs[0]->sequence(), out_);
        return MLLM_NO_ERROR;
    }
    ErrorCode load(AbstructLoader &loader) override {
        load_tensor(weight_, backend_, loader, name() + ".weight", out_, in_);
        const DataType dt = weight_.dtype();
        if (dt != MLLM_TYPE_Q4_K && dt != MLLM_TYPE_Q4_0 && dt != MLLM_TYPE_F32) throw std::runtime_error("HIPLinearOp: weight dtype not on the hot path: " + name());
        if (dt == MLLM_TYPE_Q4_K) {      // resident Linears are packed once for the M >= 16 GEMM (mllm_hip_q4k_prepack)
            const size_t pb = mllm_hip_q4k_wpack_bytes(out_, in_);
            HIPCHK(mllm_hip_alloc(&packed_, pb));
            HIPCHK(mllm_hip_q4k_prepack(weight_.device_memory().handle, out_, in_, packed_, hb(backend_)->stream()));
        } else if (dt == MLLM_TYPE_Q4_0) {  // 18-byte blocks -> nibble plane + fp16 scale plane (mllm_hip_repack_q40)
            const int64_t nblk = (int64_t)out_ * (in_ / 32);
            HIPCHK(mllm_hip_alloc(&q40_qs_, (size_t)nblk * 16));
            HIPCHK(mllm_hip_alloc(&q40_d_, (size_t)nblk * 2));
            HIPCHK(mllm_hip_repack_q40(weight_.device_memory().handle, (uint8_t *)q40_qs_, (uint16_t *)q40_d_, nblk, hb(backend_)->stream()));
        }
        if (has_bias_) load_tensor(bias_, backend_, loader, name() + ".bias", 1, out_);
        return MLLM_NO_ERROR;
    }
    ErrorCode execute(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        auto *b = hb(backend_);
        const int M = rows_of(inputs[0]);
        const float *bias = has_bias_ ? (const float *)bias_.device_memory().handle : nullptr;
        const float *x = (const float *)dptr(inputs[0]);
        void *y = dptr(outputs[0]);
        const int ydt = outputs[0]->dtype() == MLLM_TYPE_F16 ? MLLM_HIP_F16 : MLLM_HIP_F32;      // fp16 when the output aliases the KV slab (Matmul.cpp:262-268)
        switch (weight_.dtype()) {
        case MLLM_TYPE_Q4_K: {
            // activations to Q8_K planes (quantize_row_q8_K_reference), then vec_dot_q4_K_q8_K per (row, output): GEMV below 16 rows, packed GEMM from 16 on
            uint8_t *ws = (uint8_t *)b->scratch(0, mllm_hip_linear_workspace_bytes(MLLM_HIP_Q4_K, M, in_));
            if (M < 16) {
                HIPCHK(mllm_hip_linear(weight_.device_memory().handle, MLLM_HIP_Q4_K, bias, x, y, ydt, out_, M, out_, in_, ws, b->stream()));
            } else {
                void *xpack = b->scratch(1, mllm_hip_q4k_prepack_bytes(M, in_));
                HIPCHK(mllm_hip_quantize_q8k_packed(x, xpack, M, in_, b->stream()));
                HIPCHK(mllm_hip_linear_q4kp_packed(packed_, bias, xpack, y, ydt, out_, nullptr, M, out_, in_, b->stream()));
            }
            break;
        }
        case MLLM_TYPE_Q4_0: {
            uint8_t *ws = (uint8_t *)b->scratch(0, mllm_hip_linear_workspace_bytes(MLLM_HIP_Q4_0, M, in_));
            int8_t *qs = (int8_t *)ws;
            uint16_t *d = (uint16_t *)(ws + (((size_t)M * in_ + 255) & ~(size_t)255));
            HIPCHK(mllm_hip_quantize_q80(x, qs, d, M, in_, b->stream()));
            HIPCHK(mllm_hip_linear_q40_q80((const uint8_t *)q40_qs_, (const uint16_t *)q40_d_, bias, qs, d, (float *)y, out_, M, out_, in_, b->stream()));
            break;
        }
        default:
            HIPCHK(mllm_hip_linear_f32((const float *)weight_.device_memory().handle, bias, x, (float *)y, out_, M, out_, in_, b->stream()));
        }
        return MLLM_NO_ERROR;
    }
    ErrorCode free(vector<shared_ptr<Tensor>>, vector<shared_ptr<Tensor>>) override {
        weight_.free();
        if (has_bias_) bias_.free();
        for (void **p : {&packed_, &q40_qs_, &q40_d_}) if (*p) { mllm_hip_free(*p); *p = nullptr; }
        return MLLM_NO_ERROR;
    }

private:
    int in_, out_;
    bool has_bias_;
    Tensor weight_, bias_;
    void *packed_ = nullptr, *q40_qs_ = nullptr, *q40_d_ = nullptr;
};

// ---- EMBEDDING: CPUEmbedding (op/CPUEmbedding.cpp:38-80); hidden_size, vocab_size (Layer.hpp:434-435); ids are fp32 ------------------------------------------
class HIPEmbeddingOp final : public Op {
public:
    HIPEmbeddingOp(Backend *bn, const string &name, int hidden, int vocab) : Op(bn, name), hidden_(hidden), vocab_(vocab) {}
    ErrorCode reshape(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        outputs[0]->reshape(inputs[0]->batch(), 1, inputs[0]->sequence(), hidden_);
        return MLLM_NO_ERROR;
    }
    ErrorCode load(AbstructLoader &loader) override {
        load_tensor(weight_, backend_, loader, name() + ".weight", vocab_, hidden_);
        if (weight_.dtype() != MLLM_TYPE_Q4_0) throw std::runtime_error("HIPEmbeddingOp: only the Q4_0 table of *-q4_k.mllm files");
        const int64_t nblk = (int64_t)vocab_ * (hidden_ / 32);
        HIPCHK(mllm_hip_alloc(&qs_, (size_t)nblk * 16));
        HIPCHK(mllm_hip_alloc(&d_, (size_t)nblk * 2));
        HIPCHK(mllm_hip_repack_q40(weight_.device_memory().handle, (uint8_t *)qs_, (uint16_t *)d_, nblk, hb(backend_)->stream()));
        return MLLM_NO_ERROR;
    }
    ErrorCode execute(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        HIPCHK(mllm_hip_embedding_q40((const float *)dptr(inputs[0]), (const uint8_t *)qs_, (const uint16_t *)d_, (float *)dptr(outputs[0]),
                                      inputs[0]->batch() * inputs[0]->sequence(), hidden_, vocab_, hb(backend_)->stream()));
        return MLLM_NO_ERROR;
    }
    // the planes also serve the tied lm_head (Tensor::mm with the transposed table, CPUMatmulFunc.hpp:86-181)
    const void *qs() const { return qs_; }
    const void *d() const { return d_; }

private:
    int hidden_, vocab_;
    Tensor weight_;
    void *qs_ = nullptr, *d_ = nullptr;
};

// ---- RMSNORM / LAYERNORM: CPURMSNorm (op/CPURMSNorm.cpp:31-136; norm_size, epsilon, add_unit_offset), CPULayerNorm (op/CPULayerNorm.cpp:49-88; norm_size, epsilon, bias) ----
class HIPNormOp final : public Op {
public:
    HIPNormOp(Backend *bn, const string &name, bool layer, int dim, float eps, bool bias, bool unit_offset) :
        Op(bn, name), layer_(layer), dim_(dim), eps_(eps), has_bias_(bias), unit_offset_(unit_offset) {}
    ErrorCode reshape(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        outputs[0]->reshape(inputs[0]->batch(), inputs[0]->head(), inputs[0]->sequence(), inputs[0]->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode load(AbstructLoader &loader) override {
        load_tensor(weight_, backend_, loader, name() + ".weight", 1, dim_);
        if (layer_ && has_bias_) load_tensor(bias_, backend_, loader, name() + ".bias", 1, dim_);
        return MLLM_NO_ERROR;
    }
    ErrorCode execute(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        const int M = rows_of(inputs[0]);
        const float *w = (const float *)weight_.device_memory().handle;
        if (layer_) HIPCHK(mllm_hip_layernorm((const float *)dptr(inputs[0]), w, has_bias_ ? (const float *)bias_.device_memory().handle : nullptr, (float *)dptr(outputs[0]), nullptr,
                                              nullptr, nullptr, M, dim_, eps_, hb(backend_)->stream()));
        else HIPCHK(mllm_hip_rmsnorm((const float *)dptr(inputs[0]), w, (float *)dptr(outputs[0]), nullptr, nullptr, nullptr, M, dim_, eps_, unit_offset_ ? 1 : 0, hb(backend_)->stream()));
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
// input / output [B, H, S, D] in BSHD memory order = rows of H*D per position.  The position counter h_cnt_ lives in the op (CPURoPE.cpp:510-513) and
// clearCache() resets it; the multimodal form takes its positions from the second input, a host-side [3,1,1,S] tensor (such scalars stay on the host).
class HIPRoPEOp final : public Op {
public:
    HIPRoPEOp(Backend *bn, const string &name, bool multimodal, float theta, int max_pos, std::vector<int> section) :
        Op(bn, name), multimodal_(multimodal), theta_(theta), max_pos_(max_pos), section_(std::move(section)) {}
    ErrorCode reshape(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        outputs[0]->reshape(inputs[0]->batch(), inputs[0]->head(), inputs[0]->sequence(), inputs[0]->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode execute(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        auto *b = hb(backend_);
        const int S = inputs[0]->sequence(), H = inputs[0]->head(), D = inputs[0]->dimension(), half = D / 2;
        std::vector<float> s((size_t)S * half), c((size_t)S * half);
        if (multimodal_) {
            std::vector<float> pos((size_t)3 * S);
            for (int a = 0; a < 3; ++a) for (int j = 0; j < S; ++j) pos[(size_t)a * S + j] = inputs[1]->dataAt<float>(a, 0, 0, j);
            HIPCHK(mllm_hip_mrope_table(theta_, D, pos.data(), S, section_.data(), (int)section_.size(), s.data(), c.data()));
        } else {
            if (h_cnt_ + S > max_pos_) throw std::runtime_error("HIPRoPEOp: position beyond max_position_embeddings");
            if (table_dim_ != D) {      // CPURoPE's static table, built once per head size
                sin_.assign((size_t)max_pos_ * D, 0.f); cos_.assign((size_t)max_pos_ * D, 0.f);
                HIPCHK(mllm_hip_rope_table_hf(theta_, D, max_pos_, sin_.data(), cos_.data()));
                table_dim_ = D;
            }
            for (int j = 0; j < S; ++j) {
                memcpy(&s[(size_t)j * half], &sin_[(size_t)(h_cnt_ + j) * D], (size_t)half * 4);
                memcpy(&c[(size_t)j * half], &cos_[(size_t)(h_cnt_ + j) * D], (size_t)half * 4);
            }
        }
        float *ds = (float *)b->scratch(2, (size_t)2 * S * half * 4), *dc = ds + (size_t)S * half;
        HIPCHK(mllm_hip_h2d(ds, s.data(), s.size() * 4, b->stream()));
        HIPCHK(mllm_hip_h2d(dc, c.data(), c.size() * 4, b->stream()));
        b->sync();      // the host vectors go out of scope
        const int odt = outputs[0]->dtype() == MLLM_TYPE_F16 ? MLLM_HIP_F16 : MLLM_HIP_F32;      // K straight into the fp16 cache slab
        HIPCHK(mllm_hip_rope_apply((const float *)dptr(inputs[0]), (int64_t)H * D, ds, dc, half, dptr(outputs[0]), odt, (int64_t)H * D, S, H, D, b->stream()));
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
    std::vector<float> sin_, cos_;
};

// ---- KVCACHE: CPUKVCache (op/CPUKVCache.cpp:10-131,253-275; head, hidden, n_rep, cache_max, fa2).  FlashAttention2 mode: fp16 slab, n_rep = 1 -----------------
// The slab [cache_max][H*D] fp16 is the op's; execute() appends the S new rows at cache_seq_len_ (fp32 -> fp16 like the fp16 store of mat_mul) and hands out a
// non-owning view of rows [0, T + S) (TensorImpl::owns_device_memory_ = false keeps the view from freeing the slab, mllm/TensorImpl.hpp:51,137).
class HIPKVCacheOp final : public Op {
public:
    HIPKVCacheOp(Backend *bn, const string &name, int cache_max) : Op(bn, name), cache_max_(cache_max) {}
    ErrorCode reshape(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        if (cache_seq_len_ + inputs[0]->sequence() > cache_max_) { fprintf(stderr, "KVCache overflow: %d + %d > %d\n", cache_seq_len_, inputs[0]->sequence(), cache_max_); exit(1); }      // CPUKVCache.cpp:121-126
        outputs[0]->reshape(inputs[0]->batch(), inputs[0]->head(), cache_seq_len_ + inputs[0]->sequence(), inputs[0]->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        const size_t row = (size_t)inputs[0]->head() * inputs[0]->dimension();
        if (!slab_) HIPCHK(mllm_hip_alloc(&slab_, (size_t)cache_max_ * row * 2));
        outputs[0]->setDtype(MLLM_TYPE_F16);
        outputs[0]->setCtype(inputs[0]->ctype());
        DeviceMemory &m = outputs[0]->device_memory();
        m.handle = slab_; m.type = MEM_TYPE_GENERIC; m.size_in_bytes = (size_t)outputs[0]->sequence() * row * 2;
        return MLLM_NO_ERROR;
    }
    ErrorCode execute(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        const int S = inputs[0]->sequence(), n = inputs[0]->head() * inputs[0]->dimension();
        if (inputs[0]->dtype() == MLLM_TYPE_F32)
            HIPCHK(mllm_hip_store_f16((const float *)dptr(inputs[0]), n, (uint16_t *)slab_ + (size_t)cache_seq_len_ * n, n, S, n, hb(backend_)->stream()));
        // an fp16 producer (RoPE writing K rows) was pointed at the slab row by the model adapter and has appended in place already
        cache_seq_len_ += S;
        return MLLM_NO_ERROR;
    }
    int getCacheSeqLen() override { return cache_seq_len_; }
    void clearCache() override { cache_seq_len_ = 0; }

private:
    int cache_max_, cache_seq_len_ = 0;
    void *slab_ = nullptr;
};

// ---- F_FA2: CPUFlashAttention2Func (op/CPUFlashAttention2Func.hpp:29-125; causal_mask) -> flash_attention_2_forward (compute/FlashAttention2.hpp:2236-2284) -------
class HIPFlashAttention2Op final : public Op {
public:
    HIPFlashAttention2Op(Backend *bn, const string &name, bool causal) : Op(bn, name), causal_(causal) {}
    ErrorCode reshape(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        outputs[0]->reshape(inputs[0]->batch(), inputs[0]->head(), inputs[0]->sequence(), inputs[0]->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode execute(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        auto &q = inputs[0], &k = inputs[1], &v = inputs[2];
        const int Hq = q->head(), Hkv = k->head(), D = q->dimension();
        const int kvdt = k->dtype() == MLLM_TYPE_F16 ? MLLM_HIP_F16 : MLLM_HIP_F32;
        HIPCHK(mllm_hip_fa2((const float *)dptr(q), (int64_t)Hq * D, dptr(k), (int64_t)Hkv * D, dptr(v), (int64_t)Hkv * D, kvdt, (float *)dptr(outputs[0]), (int64_t)Hq * D, q->sequence(),
                            k->sequence(), Hq, Hkv, D, causal_ ? 1 : 0, nullptr, nullptr, hb(backend_)->stream()));
        return MLLM_NO_ERROR;
    }

private:
    bool causal_;
};

// ---- elementwise: CPUSiLU (op/CPUSiLU.cpp:24-52), CPUGELU / CPUQuickGELU through the fp16 LUTs (op/CPUGELU.cpp:24-46, op/CPUQuickGELU.cpp:22-43), F_TTADD / F_TTMUL (op/CPUBinaryFunc.hpp) ----
class HIPUnaryOp final : public Op {
public:
    enum Kind { SILU_K, GELU_K, QUICKGELU_K };
    HIPUnaryOp(Backend *bn, const string &name, Kind k) : Op(bn, name), kind_(k) {}
    ErrorCode reshape(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        outputs[0]->reshape(inputs[0]->batch(), inputs[0]->head(), inputs[0]->sequence(), inputs[0]->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode execute(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        auto *b = hb(backend_);
        const int64_t n = inputs[0]->count();
        if (kind_ == SILU_K) HIPCHK(mllm_hip_silu((const float *)dptr(inputs[0]), (float *)dptr(outputs[0]), n, b->stream()));
        else HIPCHK(mllm_hip_act_lut((const float *)dptr(inputs[0]), (float *)dptr(outputs[0]), n, kind_ == GELU_K ? b->gelu_lut() : b->quickgelu_lut(), b->stream()));
        return MLLM_NO_ERROR;
    }

private:
    Kind kind_;
};
class HIPBinaryOp final : public Op {
public:
    HIPBinaryOp(Backend *bn, const string &name, bool mul) : Op(bn, name), mul_(mul) {}
    ErrorCode reshape(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        outputs[0]->reshape(inputs[0]->batch(), inputs[0]->head(), inputs[0]->sequence(), inputs[0]->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode execute(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        auto fn = mul_ ? mllm_hip_mul : mllm_hip_add;
        HIPCHK(fn((const float *)dptr(inputs[0]), (const float *)dptr(inputs[1]), (float *)dptr(outputs[0]), (int64_t)inputs[0]->count(), hb(backend_)->stream()));
        return MLLM_NO_ERROR;
    }

private:
    bool mul_;
};

// ---- ops of the other model families -----------------------------------------------------------------------------------------------------------------
// SLIDINGWINDOWMASK (op/CPUSlidingWindowMask.cpp:30-58) on BSHD scores [1][heads][S][keys]; batch 1 like the rest of the adapter
class HIPSlidingWindowMaskOp final : public Op {
public:
    HIPSlidingWindowMaskOp(Backend *bn, const string &name, int window) : Op(bn, name), window_(window) {}
    ErrorCode reshape(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        outputs[0]->reshape(inputs[0]->batch(), inputs[0]->head(), inputs[0]->sequence(), inputs[0]->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode execute(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        if (inputs[0]->batch() != 1) throw std::runtime_error("HIPSlidingWindowMaskOp: batch 1 only");
        HIPCHK(mllm_hip_sliding_window_mask((const float *)dptr(inputs[0]), (float *)dptr(outputs[0]), inputs[0]->sequence(), inputs[0]->head(), inputs[0]->dimension(), window_,
                                            hb(backend_)->stream()));
        return MLLM_NO_ERROR;
    }

private:
    int window_;
};
// F_TOPK (op/CPUTopkFunc.hpp:27-92): outputs[0] = values, outputs[1] = indices (floats).  DIMENSION: [b][h][s][D] -> [b][h][s][k]; HEAD (input [1][H][S][1]) ->
// [1][k][S][1] -- in BSHD memory the rows [S][H] -> [S][k], the same launch with n = H
class HIPTopkOp final : public Op {
public:
    HIPTopkOp(Backend *bn, const string &name, int k, bool head_axis) : Op(bn, name), k_(k), head_(head_axis) {}
    ErrorCode reshape(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        if (head_ && (inputs[0]->dimension() != 1 || inputs[0]->batch() != 1)) throw std::runtime_error("HIPTopkOp: the HEAD axis form takes [1][H][S][1]");
        for (int o = 0; o < 2; ++o) {
            if (head_) outputs[o]->reshape(inputs[0]->batch(), k_, inputs[0]->sequence(), 1);
            else outputs[o]->reshape(inputs[0]->batch(), inputs[0]->head(), inputs[0]->sequence(), k_);
            outputs[o]->setDtype(inputs[0]->dtype());
        }
        return MLLM_NO_ERROR;
    }
    ErrorCode execute(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        const int rows = head_ ? inputs[0]->sequence() : inputs[0]->batch() * inputs[0]->head() * inputs[0]->sequence();
        const int n = head_ ? inputs[0]->head() : inputs[0]->dimension();
        HIPCHK(mllm_hip_topk_rows((const float *)dptr(inputs[0]), n, (float *)dptr(outputs[0]), (float *)dptr(outputs[1]), rows, n, k_, hb(backend_)->stream()));
        return MLLM_NO_ERROR;
    }

private:
    int k_;
    bool head_;
};
// F_SCATTERADD on SEQUENCE (op/CPUScatterAddFunc.hpp:27-60): inputs = (dest [1][1][S][D], src [1][1][R][D], indices [1][1][1][R]); dest is updated in place, no outputs
class HIPScatterAddOp final : public Op {
public:
    HIPScatterAddOp(Backend *bn, const string &name) : Op(bn, name) {}
    ErrorCode reshape(vector<shared_ptr<Tensor>>, vector<shared_ptr<Tensor>>) override { return MLLM_NO_ERROR; }
    ErrorCode execute(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>>) override {
        if (inputs[1]->batch() == 0) return MLLM_NO_ERROR;
        const int D = inputs[0]->dimension();
        HIPCHK(mllm_hip_scatter_add_rows((float *)dptr(inputs[0]), D, (const float *)dptr(inputs[1]), D, (const float *)dptr(inputs[2]), inputs[2]->dimension(), D, hb(backend_)->stream()));
        return MLLM_NO_ERROR;
    }
};

// ---- SOFTMAX: CPUSoftMax (op/CPUSoftMax.cpp:28-65; axis, do_causal_mask); DIMENSION axis only (the eager-attention form) -----------------------------------------
class HIPSoftMaxOp final : public Op {
public:
    HIPSoftMaxOp(Backend *bn, const string &name) : Op(bn, name) {}
    ErrorCode reshape(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        outputs[0]->reshape(inputs[0]->batch(), inputs[0]->head(), inputs[0]->sequence(), inputs[0]->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode execute(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        HIPCHK(mllm_hip_softmax((const float *)dptr(inputs[0]), (float *)dptr(outputs[0]), rows_of(inputs[0]), inputs[0]->dimension(), nullptr, hb(backend_)->stream()));
        return MLLM_NO_ERROR;
    }
};

// ---- CONVOLUTION3D (Qwen2-VL patch embed, kernel == stride, VALID, no bias: op/CPUConvolution3D.cpp:56-100) and CONVOLUTION2D (ViT / CLIP patch embed,
// kernel == stride, VALID: op/CPUConvolution2D.cpp:29-149) as GEMMs over the flattened receptive fields (compute/Convolution.cpp:35-82,179-235) ---------------------
class HIPPatchConvOp final : public Op {
public:
    HIPPatchConvOp(Backend *bn, const string &name, bool is3d, int in_ch, int out_ch, int kt, int kh, int kw, bool bias) :
        Op(bn, name), is3d_(is3d), in_ch_(in_ch), out_ch_(out_ch), kt_(kt), kh_(kh), kw_(kw), has_bias_(bias) {}
    ErrorCode reshape(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        if (is3d_) outputs[0]->reshape(inputs[0]->batch(), out_ch_, 1, 1, 1);            // [N, OC, 1, 1, 1], viewed [1,1,N,OC] by the model (modeling_qwen2_vl.hpp:31-35)
        else outputs[0]->reshape(inputs[0]->batch(), inputs[0]->head() / kh_, out_ch_, inputs[0]->dimension() / kw_);      // image [B, H, C, W] -> [B, H/p, OC, W/p]
        return MLLM_NO_ERROR;
    }
    ErrorCode load(AbstructLoader &loader) override {
        load_tensor(weight_, backend_, loader, name() + ".weight", out_ch_, in_ch_ * kt_ * kh_ * kw_);
        if (has_bias_) load_tensor(bias_, backend_, loader, name() + ".bias", 1, out_ch_);
        return MLLM_NO_ERROR;
    }
    ErrorCode execute(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        auto *b = hb(backend_);
        const float *w = (const float *)weight_.device_memory().handle, *bias = has_bias_ ? (const float *)bias_.device_memory().handle : nullptr;
        const int KK = in_ch_ * kt_ * kh_ * kw_;
        if (is3d_) {
            HIPCHK(mllm_hip_patch_gemm_f32((const float *)dptr(inputs[0]), w, bias, (float *)dptr(outputs[0]), inputs[0]->batch(), KK, out_ch_, b->stream()));
        } else {
            const int H = inputs[0]->head(), W = inputs[0]->dimension(), N = (H / kh_) * (W / kw_);
            float *patches = (float *)b->scratch(3, (size_t)N * KK * 4), *rows = (float *)b->scratch(2, (size_t)N * out_ch_ * 4);
            HIPCHK(mllm_hip_im2patch_hcw((const float *)dptr(inputs[0]), patches, H, in_ch_, W, kh_, b->stream()));
            // rows [oh*ow][OC] -> the reference's output [B, H/p, OC, W/p], which in BSHD memory order is [OC][oh][ow]
            HIPCHK(mllm_hip_patch_gemm_f32(patches, w, bias, rows, N, KK, out_ch_, b->stream()));
            HIPCHK(mllm_hip_transpose_f32(rows, (float *)dptr(outputs[0]), N, out_ch_, b->stream()));
        }
        return MLLM_NO_ERROR;
    }

private:
    bool is3d_;
    int in_ch_, out_ch_, kt_, kh_, kw_;
    bool has_bias_;
    Tensor weight_, bias_;
};

// ---- metadata functions: F_VIEW (op/CPUViewFunc.hpp:30-139) is a shape change on the contiguous BSHD buffer, F_CLIP {-1} (op/CPUClipFunc.hpp) a
// pointer offset to the last row; both hand out non-owning views ---------------------------------------------------------------------------------------------------
class HIPViewOp final : public Op {
public:
    HIPViewOp(Backend *bn, const string &name, int b, int h, int s, int d) : Op(bn, name), b_(b), h_(h), s_(s), d_(d) {}
    ErrorCode reshape(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        const int64_t n = inputs[0]->count();
        int dims[4] = {b_, h_, s_, d_};
        int64_t known = 1;
        for (int v : dims) if (v > 0) known *= v;
        for (int &v : dims) if (v <= 0) v = (int)(n / known);      // one axis may be -1 (CPUViewFunc.hpp:34-50)
        outputs[0]->reshape(dims[0], dims[1], dims[2], dims[3]);
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        outputs[0]->setDtype(inputs[0]->dtype());
        outputs[0]->setCtype(inputs[0]->ctype());
        DeviceMemory &m = outputs[0]->device_memory();
        m = inputs[0]->device_memory();
        return MLLM_NO_ERROR;
    }

private:
    int b_, h_, s_, d_;
};

// F_CLIP (op/CPUClipFunc.hpp) on the SEQUENCE axis of a batch-1 BSHD tensor: clip({}, {}, {-1}, {}) = the last position (every causal LM ends with it,
// modeling_qwen2_vl.hpp:395-397), clip({}, {}, {a, b}, {}) = positions [a, b) (LLaVA drops the class row, modeling_llava.hpp:90).  Rows of a position are contiguous,
// so the result is a pointer offset into the input: a non-owning view.
class HIPClipSeqOp final : public Op {
public:
    HIPClipSeqOp(Backend *bn, const string &name, int a, int b, bool single) : Op(bn, name), a_(a), b_(b), single_(single) {}
    ErrorCode reshape(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        const int S = inputs[0]->sequence();
        lo_ = a_ < 0 ? S + a_ : a_;
        hi_ = single_ ? lo_ + 1 : (b_ < 0 ? S + b_ : b_);
        if (inputs[0]->batch() != 1 || lo_ < 0 || hi_ > S || hi_ <= lo_) throw std::runtime_error("HIPClipSeqOp: range outside the sequence");
        outputs[0]->reshape(1, inputs[0]->head(), hi_ - lo_, inputs[0]->dimension());
        return MLLM_NO_ERROR;
    }
    ErrorCode setUp(vector<shared_ptr<Tensor>> inputs, vector<shared_ptr<Tensor>> outputs) override {
        outputs[0]->setDtype(inputs[0]->dtype());
        outputs[0]->setCtype(inputs[0]->ctype());
        const size_t row = (size_t)inputs[0]->head() * inputs[0]->dimension() * (inputs[0]->dtype() == MLLM_TYPE_F16 ? 2 : 4);
        DeviceMemory &m = outputs[0]->device_memory();
        m = inputs[0]->device_memory();
        m.handle = (char *)m.handle + (size_t)lo_ * row;
        m.size_in_bytes = (size_t)(hi_ - lo_) * row;
        return MLLM_NO_ERROR;
    }

private:
    int a_, b_, lo_ = 0, hi_ = 0;
    bool single_;
};

}  // namespace

// ---- registry: OpType -> creator (Backend::registerOps, mllm/Backend.hpp:104-105; OpDefined.hpp:10-134).  Anything not listed, or a listed Op with parameters the
// launchers do not cover, makes opCreate return nullptr and the framework run that Op on the CPU backend with automatic tensor migration (Layer.hpp:128-135,159-163) ----
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
    creators_[KVCACHE] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        if (geti(p, "n_rep", 1) != 1 || !geti(p, "fa2")) return nullptr;      // FlashAttention2 mode only (n_rep = 1, fp16): the default attn_implementation
        return new HIPKVCacheOp(b, n, geti(p, "cache_max", 100));
    };
    creators_[F_FA2] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * { return new HIPFlashAttention2Op(b, n, geti(p, "causal_mask") != 0); };
    creators_[SILU] = [](HIPBackend *b, const OpParam &, const std::string &n) -> Op * { return new HIPUnaryOp(b, n, HIPUnaryOp::SILU_K); };
    creators_[OP_GELU] = [](HIPBackend *b, const OpParam &, const std::string &n) -> Op * { return new HIPUnaryOp(b, n, HIPUnaryOp::GELU_K); };
    creators_[QUICKGLUE] = [](HIPBackend *b, const OpParam &, const std::string &n) -> Op * { return new HIPUnaryOp(b, n, HIPUnaryOp::QUICKGELU_K); };
    creators_[F_TTADD] = [](HIPBackend *b, const OpParam &, const std::string &n) -> Op * { return new HIPBinaryOp(b, n, false); };
    creators_[F_TTMUL] = [](HIPBackend *b, const OpParam &, const std::string &n) -> Op * { return new HIPBinaryOp(b, n, true); };
    creators_[SOFTMAX] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * {
        return geti(p, "axis") != (int)DIMENSION || geti(p, "do_causal_mask") ? nullptr : new HIPSoftMaxOp(b, n);
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
        // SEQUENCE-only clips; anything touching batch / head / dimension goes to the CPU
        if (geti(p, "b_size") || geti(p, "h_size") || geti(p, "d_size")) return nullptr;
        const int ss = geti(p, "s_size");
        if (ss == 1) return new HIPClipSeqOp(b, n, geti(p, "s_0"), 0, true);
        if (ss == 2) return new HIPClipSeqOp(b, n, geti(p, "s_0"), geti(p, "s_1"), false);
        return nullptr;
    };
    creators_[F_VIEW] = [](HIPBackend *b, const OpParam &p, const std::string &n) -> Op * { return new HIPViewOp(b, n, geti(p, "b"), geti(p, "h"), geti(p, "s"), geti(p, "d")); };
}

}  // namespace mllm
