// oracle/ref_drivers/ref_llava.cpp -- TEST INFRASTRUCTURE (oracle), not product.
//
// Drives the reference's own LLaVAModel (mllm/models/llava/modeling_llava.hpp:101-137, BASELINE config "demo_llava": CLIP vision
// tower -> multi-modal projector -> index_put into the text embeddings -> LLaMA body) on its x86 CPU backend, looping like
// examples/demo_llava.cpp:39-57 (forward -> argmax -> chatPostProcessing clears the image tensor) with raw ids and a raw fp32
// image [H][C][W] (CLIP img2Tensor layout, models/clip/processing_clip.hpp:28-44).
//
// STATUS at the reference snapshot of 2025-10-31: LLaVAModel cannot run -- LLaVAVisionEmbedding::Forward calls Tensor::range
// (modeling_llava.hpp:56), whose Op has no inputs, and the default Op::setUp dereferences inputs[0] (mllm/Op.hpp:61-68): the trace pass
// inside Module::load segfaults.  The driver is kept so that the golden can be produced once the reference is fixed; until then
// config 5 is pinned op by op (every Op it uses is pinned by tests/golden/ops.npz and by the configs 1-4 graphs), not as a whole graph.
//
// usage: ref_llava --model f.mllm --ids ids.i32 --img img.f32 --steps 6 --threads 4 --out dir
//                  --cfg hidden,heads,ffn,layers,vocab,cache_limit,v_hidden,v_heads,v_ffn,v_blocks,patch,img_hw
#include <chrono>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "models/llava/configuration_llava.hpp"
#include "models/llava/modeling_llava.hpp"
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
    std::string model_path, ids_path, img_path, out_dir = ".", cfg_s;
    int steps = 6, threads = 4;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i], v = argv[i + 1];
        if (k == "--model") model_path = v;
        else if (k == "--ids") ids_path = v;
        else if (k == "--img") img_path = v;
        else if (k == "--steps") steps = std::stoi(v);
        else if (k == "--threads") threads = std::stoi(v);
        else if (k == "--out") out_dir = v;
        else if (k == "--cfg") cfg_s = v;
    }
    CPUBackend::cpu_threads = threads;
    Module::initBackend(MLLM_CPU);
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
    auto model = LLaVAModel(config);
    model.load(model_path);

    Backend *bn = Backend::global_backends[MLLM_CPU].get();
    auto ids = read_file<int32_t>(ids_path);
    Tensor t_ids(1, 1, (int)ids.size(), 1, bn, true);
    t_ids.setName("input_ids");
    Tensor::tensor_status = TENSOR_STATIC_INIT;
    t_ids.setTtype(INPUT_TENSOR);
    for (size_t i = 0; i < ids.size(); ++i) t_ids.setDataAt<float>(0, 0, (int)i, 0, (float)ids[i]);
    const int hw = cv[11];
    auto img = read_file<float>(img_path);
    Tensor t_img(1, hw, 3, hw, bn, true);
    t_img.setName("input_img");
    t_img.setTtype(INPUT_TENSOR);
    for (int h = 0; h < hw; ++h)
        for (int c = 0; c < 3; ++c)
            for (int w = 0; w < hw; ++w) t_img.setDataAt<float>(0, h, c, w, img[((size_t)h * 3 + c) * hw + w]);

    std::vector<int32_t> tokens;
    for (int step = 0; step < steps; ++step) {
        auto result = model({t_ids, t_img});
        Tensor &lg = result[0];
        int V = lg.dimension(), s = lg.sequence() - 1;
        std::vector<float> row(V);
        for (int i = 0; i < V; ++i) row[i] = lg.dataAt<float>(0, 0, s, i);
        int best = 0;
        for (int i = 1; i < V; ++i) if (row[i] > row[best]) best = i;
        tokens.push_back(best);
        write_file<float>(out_dir + "/logits_" + std::to_string(step) + ".f32", row.data(), V);
        chatPostProcessing((unsigned)best, t_ids, {&t_img});
    }
    write_file<int32_t>(out_dir + "/tokens.i32", tokens.data(), tokens.size());
    return 0;
}
