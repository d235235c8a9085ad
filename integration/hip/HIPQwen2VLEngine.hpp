// integration/hip/HIPQwen2VLEngine.hpp -- the same drop-in one level up: a mllm::Module with Qwen2VLModel's calling convention whose Forward is the resident engine
// of libmllm_hip.so (mllm_hip_model_*: streamed load, one prefill call, fused decode step), for hosts that want the engine's speed without leaving the reference's
// frontend -- `model({input_ids, pixel_values, image_grid_thw})` returns the logits Tensor of the last position exactly as Qwen2VLModel::Forward does
// (models/qwen2_vl/modeling_qwen2_vl.hpp:380-404), the demo's host loop (argmax, chatPostProcessing, profiling()) stays as it is (examples/demo_qwen2_vl.cpp:53-63).
// Where the Backend / Op adapter (HIPBackend) runs the reference's own graph Op by Op -- 630 launches per token --, this Module hands the whole forward to the engine.
// The weights come straight from the .mllm file (mllm_hip_model_create): Module::load / Module::to are not used.  Includes only the reference's headers and the C ABI.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "Module.hpp"
#include "Tensor.hpp"
#include "models/qwen2_vl/configuration_qwen2_vl.hpp"

#include "mllm_hip.h"

namespace mllm {

class HIPQwen2VLEngine final : public Module {
public:
    HIPQwen2VLEngine(const Qwen2VLConfig &config, const std::string &mllm_path, int device = 0) : vocab_(config.vocab_size) {
        if (mllm_hip_init(device) != MLLM_HIP_OK) throw std::runtime_error(std::string("mllm_hip_init: ") + mllm_hip_last_error());
        mllm_hip_model_config c{};
        c.arch = MLLM_HIP_ARCH_QWEN2VL;
        c.hidden = config.hidden_size; c.inter = config.intermediate_size; c.layers = config.num_hidden_layers;
        c.heads = config.num_attention_heads; c.kv_heads = config.num_key_value_heads; c.vocab = config.vocab_size;
        c.rms_eps = (float)config.rms_norm_eps; c.final_eps = 1e-6f;      // model.norm: RMSNorm(hidden_dim, 1e-6, ...) (modeling_qwen2_vl.hpp:374)
        c.rope_theta = (float)config.rope_theta;
        for (int i = 0; i < 3; ++i) c.mrope_section[i] = config.mrope_section[i];
        c.cache_limit = config.cache_limit; c.tie_embedding = config.tie_embedding_words ? 1 : 0; c.qkv_bias = 1;      // tied embed_tokens head or a separate lm_head Linear (modeling_qwen2_vl.hpp:375-401)
        c.v_dim = config.vision_embed_dim; c.v_heads = 16; c.v_blocks = 32; c.v_patch = 14; c.v_merge = config.spatial_merge_size;      // Qwen2VisionModel(..., 16, ..., 14, 336, 32, ...) (:371)
        c.image_token_id = config.image_token_id; c.vision_start_token_id = config.vision_start_token_id;
        c.vision_end_token_id = config.vision_end_token_id; c.video_token_id = config.video_token_id;
        if (mllm_hip_model_create(&c, mllm_path.c_str(), &m_) != MLLM_HIP_OK) throw std::runtime_error(std::string("mllm_hip_model_create: ") + mllm_hip_last_error());
        Module::llm_model_ptr = this;
    }
    ~HIPQwen2VLEngine() override {
        if (pinned_) mllm_hip_host_unregister(pinned_);
        if (m_) mllm_hip_model_destroy(m_);
    }
    HIPQwen2VLEngine(const HIPQwen2VLEngine &) = delete;
    HIPQwen2VLEngine &operator=(const HIPQwen2VLEngine &) = delete;

    // Qwen2VLModel::get_position_ids builds the M-RoPE position tensor on the host (:406-470); the engine derives the same positions from the ids and the grid itself
    void get_position_ids(std::vector<Tensor> &) {}
    void clear_kvcache() { mllm_hip_model_clear_kvcache(m_); }

    // inputs: input_ids [1, 1, S, 1] (floats holding the ids), pixel_values [n_patch, 3, 2, 14, 14] or empty, image_grid_thw [1, 1, 1, 3] or empty.
    // An empty cache or S > 1: a prefill of these ids (with the image, if one is handed over); S == 1 on a filled cache: one decode step for this token.
    // Returns {logits [1, 1, 1, vocab]} on the CPU backend -- the SAME page-locked Tensor every call (a caller that keeps the logits of an earlier step copies them:
    // the reference hands out a fresh Tensor per Forward).
    std::vector<Tensor> Forward(std::vector<Tensor> inputs, std::vector<std::any>) override {
        Tensor &ids = inputs[0];
        const int S = ids.sequence();
        // the logits row lives in ONE host Tensor for the life of the Module (Tensor copies share their TensorImpl), page-locked once: a fresh 608 KB Tensor per
        // token is an mmap, 150 page faults and a staged pageable copy -- together more than a tenth of the 0.9 ms step
        if (!pinned_) {
            out_ = Tensor(1, 1, 1, vocab_, Backend::global_backends[MLLM_CPU].get(), true);
            out_.setName("lm_logits");
            if (mllm_hip_host_register(out_.hostPtr<float>(), (size_t)vocab_ * sizeof(float)) == MLLM_HIP_OK) pinned_ = out_.hostPtr<float>();
        }
        float *logits = out_.hostPtr<float>();
        int32_t next = 0;
        int rc;
        if (S > 1 || mllm_hip_model_cache_len(m_) == 0) {
            std::vector<int32_t> id(S);
            for (int i = 0; i < S; ++i) id[i] = (int32_t)ids.dataAt<float>(0, 0, i, 0);
            const float *pix = nullptr;
            int32_t meta[3] = {0, 0, 0};
            if (inputs.size() > 2 && inputs[1].count() > 0 && inputs[2].count() > 0) {
                pix = inputs[1].hostPtr<float>();      // [n_patch][3 * 2 * 14 * 14], contiguous (processing_qwen2_vl.hpp:249-252)
                for (int d = 0; d < 3; ++d) meta[d] = (int32_t)inputs[2].dataAt<float>(0, 0, 0, d);
            }
            rc = mllm_hip_model_prefill(m_, id.data(), S, pix, pix ? meta : nullptr, nullptr, 0, logits, &next, nullptr);
        } else {
            rc = mllm_hip_model_decode(m_, (int32_t)ids.dataAt<float>(0, 0, 0, 0), logits, &next, nullptr);
        }
        if (rc != MLLM_HIP_OK) throw std::runtime_error(std::string("HIPQwen2VLEngine::Forward: ") + mllm_hip_last_error());
        return {out_};
    }

private:
    mllm_hip_model *m_ = nullptr;
    int vocab_;
    Tensor out_;
    float *pinned_ = nullptr;
};

}  // namespace mllm
