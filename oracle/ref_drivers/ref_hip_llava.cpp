// oracle/ref_drivers/ref_hip_llava.cpp -- TEST INFRASTRUCTURE (oracle side), not product.
//
// BASELINE config 5 through the boundary: the LLaVA graph of mllm/models/llava/modeling_llava.hpp:39-137 composed from the reference's OWN modules -- Convolution2D /
// Parameter / Embedding / LayerNorm / ViTBlock (models/vit/modeling_vit.hpp:31-61) / Linear / GELU, Embedding + LLaMABodyModel (modeling_llava.hpp:15-37), Tensor::where +
// index_put(accumulate) (:130-131), clip({-1}) -- on the HIP backend of integration/hip/.  It is the composition of ref_llava_parts.cpp (see its header: LLaVAModel itself
// cannot be loaded at this snapshot, its Tensor::range has no inputs and the default Op::setUp dereferences inputs[0], mllm/Op.hpp:61-68, so the position ids 0..N-1
// arrive as an input tensor; position_embedding is a row gather, no arithmetic changes), moved to the device as examples/demo_qwen.cpp:43-59 does and driven like
// examples/demo_llava.cpp:40-55.  Same arguments and outputs as ref_llava_parts.cpp, plus the JSON report of hip_driver_common.hpp.
//
// usage: ref_hip_llava --model f.mllm --ids ids.i32 --img img.f32 --steps 6 --threads 4 --out dir [--dump-vision 1] [--dump-every 1]
//                      --cfg hidden,heads,ffn,layers,vocab,cache_limit,v_hidden,v_heads,v_ffn,v_blocks,patch,img_hw
#include <chrono>
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "models/llava/configuration_llava.hpp"
#include "models/llava/modeling_llava.hpp"
#include "backends/cpu/CPUBackend.hpp"

#include "hip_driver_common.hpp"

using namespace mllm;

// LLaVAVisionEmbedding (modeling_llava.hpp:39-61) with the position ids handed in (inputs[1]) instead of Tensor::range(0, range_len_)
class PartsVisionEmbedding final : public Module {
    Layer patch_embedding;
    Parameter cls_token;
    Layer position_embedding;

public:
    PartsVisionEmbedding() = default;
    PartsVisionEmbedding(int hidden_dim, int patch, int img_hw, const ViTNameConfig &names, const string &base_name) {
        patch_embedding = Convolution2D(3, hidden_dim, {patch, patch}, {patch, patch}, VALID, false, base_name + names._patch_embedding_name);
        cls_token = Parameter(1, 1, 1, hidden_dim, base_name + names._cls_token_name);
        const int range_len = std::ceil(img_hw / patch) * std::ceil(img_hw / patch) + 1;
        position_embedding = Embedding(range_len, hidden_dim, base_name + names._position_embeddings_name);
    }
    vector<Tensor> Forward(vector<Tensor> inputs, vector<std::any> args) override {
        auto embd = patch_embedding(inputs[0]);
        embd = embd.transpose({{SEQUENCE, DIMENSION}, {HEAD, SEQUENCE}});
        embd = embd.flatten(HEAD, SEQUENCE);
        embd = Tensor::cat({cls_token(), embd}, SEQUENCE);
        embd = position_embedding(inputs[1]) + embd;
        return {embd};
    }
};

// LLaVAVisionModel (modeling_llava.hpp:63-98)
class PartsVisionModel final : public Module {
    PartsVisionEmbedding embedding;
    Layer pre_layrnorm;
    vector<ViTBlock> blocks;
    Layer linear_1;
    Layer gelu;
    Layer linear_2;
    int clip_len_{};

public:
    PartsVisionModel() = default;
    PartsVisionModel(int hidden_dim, int head_size, int ffn_hidden, int patch, int img_hw, int block_num, string attn_implementation, const ViTNameConfig &names,
                     const string &base_name) {
        embedding = PartsVisionEmbedding(hidden_dim, patch, img_hw, names, base_name + names._embd_name);
        pre_layrnorm = LayerNorm(hidden_dim, true, 1e-6, base_name + names._vision_pre_layrnorm_name);
        blocks = List<ViTBlock>(block_num, hidden_dim, head_size, ffn_hidden, "QuickGELU", attn_implementation, names, base_name + names._layer_name);
        clip_len_ = std::ceil(img_hw / patch) * std::ceil(img_hw / patch) + 1;
        linear_1 = Linear(hidden_dim, ffn_hidden, true, "multi_modal_projector.linear_1");
        gelu = GELU("multi_modal_projector.act");
        linear_2 = Linear(ffn_hidden, ffn_hidden, true, "multi_modal_projector.linear_2");
    }
    vector<Tensor> Forward(vector<Tensor> inputs, vector<std::any> args) override {
        auto x = embedding(inputs)[0];
        x = pre_layrnorm(x);
        for (auto &block : blocks) x = block({x})[0];
        x = x.clip({}, {}, {1, clip_len_}, {});
        x = linear_1(x);
        x = gelu(x);
        x = linear_2(x);
        return {x};
    }
};

// LLaVAModel (modeling_llava.hpp:99-137); inputs: ids, image ([1,H,3,W], or batch 0 after the first step), position ids
class PartsLLaVAModel final : public Module {
    Layer text_embedding;
    PartsVisionModel vision_tower;
    LLaMABodyModel llama_body;

public:
    explicit PartsLLaVAModel(const LLaVAConfig &c) {
        text_embedding = Embedding(c.vocab_size, c.hidden_dim, c.names_config.token_embd_name);
        llama_body = LLaMABodyModel(c.vocab_size, c.hidden_dim, c.head_size, c.ffn_hidden, c.block_num, c.RoPE_type, c.rope_theta, c.max_position_embeddings, c.cache_limit,
                                    c.attn_implementation, c.names_config, c.names_config.blk_name);
        vision_tower = PartsVisionModel(c.vision_hidden_dim, c.vision_head_size, c.vision_ffn_hidden, c.patch, c.img_hw, c.vision_block_num, c.attn_implementation,
                                        c.vit_names_config, c.vit_names_config.vison_model_name);
    }
    vector<Tensor> Forward(vector<Tensor> inputs, vector<std::any> args) override {
        auto embd = text_embedding(inputs[0]);
        if (inputs[1].batch() > 0) {
            auto vision = vision_tower({inputs[1], inputs[2]})[0];
            auto where_idx = inputs[0].where(32000, SEQUENCE);
            embd = embd.index_put(vision, where_idx, true);
        }
        embd = llama_body({embd})[0];
        embd = embd.clip({}, {}, {-1}, {});
        return {embd};
    }
};

class PartsVisionOnly final : public Module {
public:
    PartsVisionModel vision_tower;
    explicit PartsVisionOnly(const LLaVAConfig &c) {
        vision_tower = PartsVisionModel(c.vision_hidden_dim, c.vision_head_size, c.vision_ffn_hidden, c.patch, c.img_hw, c.vision_block_num, c.attn_implementation,
                                        c.vit_names_config, c.vit_names_config.vison_model_name);
    }
    vector<Tensor> Forward(vector<Tensor> inputs, vector<std::any> args) override { return vision_tower({inputs[0], inputs[1]}); }
};

int main(int argc, char **argv) {
    std::string model_path, ids_path, img_path, out_dir = ".", cfg_s;
    int steps = 6, threads = 4, dump_vision = 0, dump_every = 1;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i], v = argv[i + 1];
        if (k == "--model") model_path = v;
        else if (k == "--ids") ids_path = v;
        else if (k == "--img") img_path = v;
        else if (k == "--steps") steps = std::stoi(v);
        else if (k == "--threads") threads = std::stoi(v);
        else if (k == "--out") out_dir = v;
        else if (k == "--cfg") cfg_s = v;
        else if (k == "--dump-vision") dump_vision = std::stoi(v);
        else if (k == "--dump-every") dump_every = std::stoi(v);
    }
    CPUBackend::cpu_threads = threads;
    Module::initBackend(MLLM_CPU);
    HIPBackend *hip = installHIPBackend(0);
    auto cv = parse_ints(cfg_s);
    if (cv.size() != 12) { fprintf(stderr, "--cfg needs 12 integers\n"); return 2; }
    LLaVAConfig config(cv[5], "7B", cv[4]);
    config.hidden_dim = cv[0];
    config.head_size = cv[1];
    config.num_key_value_heads = cv[1];
    config.ffn_hidden = cv[2];
    config.block_num = cv[3];
    config.vision_hidden_dim = cv[6];
    config.vision_head_size = cv[7];
    config.vision_ffn_hidden = cv[8];
    config.vision_block_num = cv[9];
    config.patch = cv[10];
    config.img_hw = cv[11];

    Backend *bn = Backend::global_backends[MLLM_CPU].get();
    auto ids = read_file<int32_t>(ids_path);
    const int hw = cv[11], n_pos = (hw / cv[10]) * (hw / cv[10]) + 1;
    auto img = read_file<float>(img_path);
    auto make_inputs = [&](Tensor &t_ids, Tensor &t_img, Tensor &t_pos) {
        Tensor::tensor_status = TENSOR_STATIC_INIT;
        t_ids = Tensor(1, 1, (int)ids.size(), 1, bn, true);
        t_ids.setName("input_ids");
        t_ids.setTtype(INPUT_TENSOR);
        for (size_t i = 0; i < ids.size(); ++i) t_ids.setDataAt<float>(0, 0, (int)i, 0, (float)ids[i]);
        t_img = Tensor(1, hw, 3, hw, bn, true);
        t_img.setName("input_img");
        t_img.setTtype(INPUT_TENSOR);
        for (int h = 0; h < hw; ++h)
            for (int c = 0; c < 3; ++c)
                for (int w = 0; w < hw; ++w) t_img.setDataAt<float>(0, h, c, w, img[((size_t)h * 3 + c) * hw + w]);
        t_pos = Tensor(1, 1, n_pos, 1, bn, true);
        t_pos.setName("input_pos");
        t_pos.setTtype(INPUT_TENSOR);
        for (int i = 0; i < n_pos; ++i) t_pos.setDataAt<float>(0, 0, i, 0, (float)i);
    };

    if (dump_vision) {
        auto vm = PartsVisionOnly(config);
        vm.to(MLLM_HIP_BACKEND_TYPE);
        vm.load(model_path);
        hip->sync();
        Tensor t_ids, t_img, t_pos;
        make_inputs(t_ids, t_img, t_pos);
        std::vector<double> ms;
        auto t0 = std::chrono::steady_clock::now();
        auto r = vm({t_img, t_pos});
        Tensor &v = r[0];
        if (v.backend()->type() != MLLM_CPU) v.cpu();
        ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        std::vector<float> rows((size_t)v.sequence() * v.dimension());
        for (int s = 0; s < v.sequence(); ++s)
            for (int d = 0; d < v.dimension(); ++d) rows[(size_t)s * v.dimension() + d] = v.dataAt<float>(0, 0, s, d);
        write_file<float>(out_dir + "/vision.f32", rows.data(), rows.size());
        char extra[96];
        snprintf(extra, sizeof extra, "\"vision_rows\": %d, \"vision_dim\": %d, ", v.sequence(), v.dimension());
        hip_report(hip, 0, ms, extra);
        return 0;
    }

    auto model = PartsLLaVAModel(config);
    model.to(MLLM_HIP_BACKEND_TYPE);      // examples/demo_qwen.cpp:57
    model.load(model_path);
    hip->sync();
    Tensor t_ids, t_img, t_pos;
    make_inputs(t_ids, t_img, t_pos);
    std::vector<int32_t> tokens;
    std::vector<double> ms;
    for (int step = 0; step < steps; ++step) {
        auto t0 = std::chrono::steady_clock::now();
        auto result = model({t_ids, t_img, t_pos});
        auto t1 = std::chrono::steady_clock::now();
        ms.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
        Tensor &lg = result[0];
        if (lg.backend()->type() != MLLM_CPU) lg.cpu();      // mllm/Module.cpp:93-95
        int V = lg.dimension(), s = lg.sequence() - 1;
        std::vector<float> row(V);
        for (int i = 0; i < V; ++i) row[i] = lg.dataAt<float>(0, 0, s, i);
        int best = 0;
        for (int i = 1; i < V; ++i) if (row[i] > row[best]) best = i;
        tokens.push_back(best);
        if (step == 0 || step == steps - 1 || (dump_every > 0 && step % dump_every == 0)) write_file<float>(out_dir + "/logits_" + std::to_string(step) + ".f32", row.data(), V);
        // examples/demo_llava.cpp:52 chatPostProcessing(out_token, input_ids, {&img}): next input = the sampled id, the image tensor emptied; the layers migrated both to
        // the device, so they come back first (mllm/Module.cpp:65-70)
        t_ids.cpu();
        t_ids.reshape(1, 1, 1, 1);
        t_ids.alloc();
        t_ids.setDataAt<float>(0, 0, 0, 0, (float)best);
        if (t_img.backend()->type() != MLLM_CPU) t_img.cpu();
        t_img.reshape(0, 0, 0, 0);
        t_img.alloc();
    }
    write_file<int32_t>(out_dir + "/tokens.i32", tokens.data(), tokens.size());
    hip_report(hip, ids.size() + n_pos - 2, ms);
    model.profiling();
    return 0;
}
