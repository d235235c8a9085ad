// oracle/ref_drivers/ref_qwen2vl.cpp -- TEST INFRASTRUCTURE (oracle), not product.
//
// Drives the *reference's own* Qwen2VLModel (mllm/models/qwen2_vl/modeling_qwen2_vl.hpp:337-404) on its x86 CPU
// backend with synthetic inputs, the way examples/demo_qwen2_vl.cpp:32-66 does (get_position_ids -> model(input)
// -> host argmax -> chatPostProcessing), but with raw id / pixel files instead of tokenizer + image decode.
// Emits greedy token ids, last-row logits and per-forward wall times (Module::profiling() definition,
// mllm/Module.cpp:35-42).  Built only by oracle/Makefile.ref into oracle/_ref/ against /root/reference.
//
// usage: ref_qwen2vl --model f.mllm --ids ids.i32 --pix pix.f32 --grid 1,32,32 --steps 8 --threads 8 --out dir
//                    [--cfg hidden,inter,layers,heads,kv_heads,vocab,v_dim,cache_limit,img_tok,vstart,vend,video_tok]
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "models/qwen2_vl/configuration_qwen2_vl.hpp"
#include "models/qwen2_vl/modeling_qwen2_vl.hpp"
#include "processor/PostProcess.hpp"
#include "backends/cpu/CPUBackend.hpp"

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
    int steps = 8, threads = 8, dump_every = 1;
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
    }
    CPUBackend::cpu_threads = threads;
    Module::initBackend(MLLM_CPU);

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
    auto model = Qwen2VLModel(config);
    model.load(model_path);

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
    for (int step = 0; step < steps; ++step) {
        model.get_position_ids(input);
        auto t0 = std::chrono::steady_clock::now();
        auto result = model(input);
        auto t1 = std::chrono::steady_clock::now();
        ms.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
        Tensor &lg = result[0];
        int V = lg.dimension(), s = lg.sequence() - 1;
        std::vector<float> row(V);
        for (int i = 0; i < V; ++i) row[i] = lg.dataAt<float>(0, 0, s, i);
        int best = 0;  // std::max_element semantics (first max), processing_qwen2_vl.hpp:284-289
        for (int i = 1; i < V; ++i) if (row[i] > row[best]) best = i;
        tokens.push_back(best);
        if (step == 0 || step == steps - 1 || (dump_every > 0 && step % dump_every == 0))
            write_file<float>(out_dir + "/logits_" + std::to_string(step) + ".f32", row.data(), V);
        chatPostProcessing((unsigned)best, input[0], {&input[1], &input[2]});
    }
    write_file<int32_t>(out_dir + "/tokens.i32", tokens.data(), tokens.size());
    double dec = 0;
    for (size_t i = 1; i < ms.size(); ++i) dec += ms[i];
    printf("{\"prefill_tokens\": %zu, \"prefill_ms\": %.3f, \"decode_steps\": %zu, \"decode_ms_mean\": %.4f, "
           "\"decode_tok_s\": %.3f, \"threads\": %d}\n",
           ids.size(), ms[0], ms.size() - 1, ms.size() > 1 ? dec / (ms.size() - 1) : 0.0,
           ms.size() > 1 ? 1000.0 * (ms.size() - 1) / dec : 0.0, threads);
    model.profiling();
    return 0;
}
