// oracle/ref_drivers/ref_hip_qwen2vl.cpp -- TEST INFRASTRUCTURE (oracle side), not product.
//
// The reference's own Qwen2VLModel (mllm/models/qwen2_vl/modeling_qwen2_vl.hpp:337-404) -- its Module / Layer / Tensor frontend compiled from /root/reference,
// unchanged -- moved onto the HIP backend of integration/hip/ the way examples/demo_qwen.cpp:43-58 moves a model onto a device (`model.to(device)` before
// `model.load(path)`), then driven like examples/demo_qwen2_vl.cpp:53-63 (get_position_ids -> model(input) -> host argmax -> chatPostProcessing), with the one
// device-aware step the reference's own generate loop has: tensors the host edits between forwards come back with `.cpu()` first (mllm/Module.cpp:65-70,93-95).
// Same arguments and outputs as ref_qwen2vl.cpp, plus a JSON line saying which Ops the backend refused (= ran on the CPU) and how many ran on the device.
// Built by oracle/Makefile.ref into oracle/_ref/ (the binary travels to the GPU box, the sources it was compiled from do not); run by tests/test_gpu_adapter.py.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "models/qwen2_vl/configuration_qwen2_vl.hpp"
#include "models/qwen2_vl/modeling_qwen2_vl.hpp"
#include "processor/PostProcess.hpp"
#include "backends/cpu/CPUBackend.hpp"

#include "HIPBackend.hpp"
#include "HIPQwen2VLEngine.hpp"

using namespace mllm;

template <typename T>
static std::vector<T> read_file(const std::string &p) {
    std::ifstream f(p, std::ios::binary | std::ios::ate);
    if (!f) { fprintf(stderr, "cannot open %s\n", p.c_str()); exit(2); }
    size_t n = f.tellg();
    f.seekg(0);
    std::vector<T> v(n / sizeof(T));
    f.read((char *)v.data(), n);
    return v;
}
template <typename T>
static void write_file(const std::string &p, const T *d, size_t n) {
    std::ofstream f(p, std::ios::binary);
    f.write((const char *)d, n * sizeof(T));
}
static std::vector<int> parse_ints(const std::string &s) {
    std::vector<int> v;
    size_t pos = 0;
    while (pos < s.size()) {
        size_t e = s.find(',', pos);
        if (e == std::string::npos) e = s.size();
        v.push_back(std::stoi(s.substr(pos, e - pos)));
        pos = e + 1;
    }
    return v;
}

int main(int argc, char **argv) {
    std::string model_path, ids_path, pix_path, out_dir = ".", grid_s = "1,32,32", cfg_s;
    int steps = 8, threads = 8, dump_every = 1, engine = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i], v = argv[i + 1];
        if (k == "--model") model_path = v;
        else if (k == "--ids") ids_path = v;
        else if (k == "--pix") pix_path = v;
        else if (k == "--grid") grid_s = v;
        else if (k == "--steps") steps = std::stoi(v);
        else if (k == "--threads") threads = std::stoi(v);
        else if (k == "--out") out_dir = v;
        else if (k == "--cfg") cfg_s = v;
        else if (k == "--dump-every") dump_every = std::stoi(v);
        else if (k == "--engine") engine = std::stoi(v);      // 1: the Module of HIPQwen2VLEngine.hpp (the resident engine behind the reference's frontend) instead of the Op-by-Op adapter
    }
    CPUBackend::cpu_threads = threads;
    Module::initBackend(MLLM_CPU);
    HIPBackend *hip = installHIPBackend(0);      // what `case MLLM_HIP` of Module::initBackend does upstream

    int cache_limit = 800;
    std::vector<int> cv;
    if (!cfg_s.empty()) { cv = parse_ints(cfg_s); cache_limit = cv[7]; }
    Qwen2VLConfig config(cache_limit, "1.5b");
    if (!cv.empty()) {
        config.hidden_size = cv[0];
        config.intermediate_size = cv[1];
        config.num_hidden_layers = cv[2];
        config.num_attention_heads = cv[3];
        config.num_key_value_heads = cv[4];
        config.vocab_size = cv[5];
        config.vision_embed_dim = cv[6];
        config.image_token_id = cv[8];
        config.vision_start_token_id = cv[9];
        config.vision_end_token_id = cv[10];
        config.video_token_id = cv[11];
        if (cv.size() > 12) config.tie_embedding_words = cv[12] != 0;      // optional 13th field: 0 = a separate lm_head Linear (modeling_qwen2_vl.hpp:375-401)
    }

    auto ids = read_file<int32_t>(ids_path);
    auto grid = parse_ints(grid_s);
    Backend *bn = Backend::global_backends[MLLM_CPU].get();

    Tensor t_ids(1, 1, (int)ids.size(), 1, bn, true);
    t_ids.setName("input_ids");
    Tensor::tensor_status = TENSOR_STATIC_INIT;
    t_ids.setTtype(INPUT_TENSOR);
    for (size_t i = 0; i < ids.size(); ++i) t_ids.setDataAt<float>(0, 0, (int)i, 0, (float)ids[i]);

    std::vector<Tensor> input = {t_ids};
    if (!pix_path.empty()) {
        auto pix = read_file<float>(pix_path);
        int n_patch = grid[0] * grid[1] * grid[2];
        int pe = (int)(pix.size() / n_patch);
        // same construction as Qwen2VLImageProcessor::process (processing_qwen2_vl.hpp:249-252)
        Tensor t_pix(1, n_patch, 1, pe, bn, true);
        t_pix.setName("pixel_values");
        t_pix.setTtype(INPUT_TENSOR);
        for (int p = 0; p < n_patch; ++p)
            for (int d = 0; d < pe; ++d) t_pix.setDataAt<float>(0, p, 0, d, pix[(size_t)p * pe + d]);
        t_pix.reshape(t_pix.head(), 3, 2, 14, 14);
        Tensor t_grid(1, 1, 1, 3, bn, true);
        t_grid.setName("image_grid_thw");
        t_grid.setTtype(INPUT_TENSOR);
        for (int d = 0; d < 3; ++d) t_grid.setDataAt<float>(0, 0, 0, d, (float)grid[d]);
        input.push_back(t_pix);
        input.push_back(t_grid);
    } else {
        Tensor e1(0, 0, 0, 0, MLLM_CPU, true), e2(0, 0, 0, 0, MLLM_CPU, true);
        input.push_back(e1);
        input.push_back(e2);
    }

    std::vector<int32_t> tokens;
    std::vector<double> ms;
    auto drive = [&](auto &model) {
    for (int step = 0; step < steps; ++step) {
        model.get_position_ids(input);
        auto t0 = std::chrono::steady_clock::now();
        auto result = model(input);
        auto t1 = std::chrono::steady_clock::now();
        ms.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
        Tensor &lg = result[0];
        if (lg.backend()->type() != MLLM_CPU) lg.cpu();      // mllm/Module.cpp:93-95
        int V = lg.dimension(), s = lg.sequence() - 1;
        std::vector<float> row(V);
        for (int i = 0; i < V; ++i) row[i] = lg.dataAt<float>(0, 0, s, i);
        int best = 0;  // std::max_element semantics (first max), processing_qwen2_vl.hpp:284-289
        for (int i = 1; i < V; ++i) if (row[i] > row[best]) best = i;
        tokens.push_back(best);
        if (step == 0 || step == steps - 1 || (dump_every > 0 && step % dump_every == 0))
            write_file<float>(out_dir + "/logits_" + std::to_string(step) + ".f32", row.data(), V);
        // the host edits ids and position ids between forwards: bring them back first, as Module::generate's chatPostProcessing does (mllm/Module.cpp:65-70)
        input[0].cpu();
        if (input.size() > 3) input[3].cpu();
        chatPostProcessing((unsigned)best, input[0], {&input[1], &input[2]});
    }
    };
    std::unique_ptr<Qwen2VLModel> graph_model;
    std::unique_ptr<HIPQwen2VLEngine> engine_model;
    if (engine) {
        engine_model = std::make_unique<HIPQwen2VLEngine>(config, model_path);
        drive(*engine_model);
    } else {
        graph_model = std::make_unique<Qwen2VLModel>(config);
        graph_model->to(MLLM_HIP_BACKEND_TYPE);      // examples/demo_qwen.cpp:57
        graph_model->load(model_path);               // :59 -- every Op::load goes through Backend::load_from_file
        hip->sync();
        drive(*graph_model);
    }
    write_file<int32_t>(out_dir + "/tokens.i32", tokens.data(), tokens.size());
    double dec = 0;
    for (size_t i = 1; i < ms.size(); ++i) dec += ms[i];
    std::string refused = "[";
    for (size_t i = 0; i < hip->refused().size(); ++i)
        refused += std::string(i ? ", " : "") + "[" + std::to_string(hip->refused()[i].first) + ", \"" + hip->refused()[i].second + "\"]";
    refused += "]";
    printf("{\"backend\": \"hip\", \"prefill_tokens\": %zu, \"prefill_ms\": %.3f, \"decode_steps\": %zu, \"decode_ms_mean\": %.4f, "
           "\"decode_tok_s\": %.3f, \"hip_ops_run\": %ld, \"fused_launches\": %ld, \"fused_ops\": %ld, \"cpu_fallback_ops\": %zu, \"refused\": %s, \"live_device_blocks\": %zu}\n",
           ids.size(), ms[0], ms.size() - 1, ms.size() > 1 ? dec / (ms.size() - 1) : 0.0,
           ms.size() > 1 ? 1000.0 * (ms.size() - 1) / dec : 0.0, hip->ops_run(), hip->fused_launches(), hip->fused_ops(), hip->refused().size(), refused.c_str(), hip->live_blocks());
    fflush(stdout);
    if (engine) engine_model->profiling(); else graph_model->profiling();
    return 0;
}
