// oracle/ref_drivers/ref_vit.cpp -- TEST INFRASTRUCTURE (oracle), not product.
//
// Drives the reference's own ViTModel (mllm/models/vit/modeling_vit.hpp:91-111, BASELINE config "demo_vit") on its x86 CPU
// backend, one image per forward like examples/demo_vit.cpp:33-38, with raw fp32 images [H][C][W] (the layout
// ViTProcessor::img2Tensor builds, processing_vit.hpp:18-29) instead of decoded files.  Emits the class logits of every image.
//
// usage: ref_vit --model f.mllm --img imgs.f32 --n 2 --threads 8 --out dir --cfg hidden,heads,ffn,blocks,patch,img_hw,classes
#include <chrono>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "models/vit/configuration_vit.hpp"
#include "models/vit/modeling_vit.hpp"
#include "backends/cpu/CPUBackend.hpp"

using namespace mllm;

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
    std::string model_path, img_path, out_dir = ".", cfg_s;
    int n_img = 1, threads = 8;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i], v = argv[i + 1];
        if (k == "--model") model_path = v;
        else if (k == "--img") img_path = v;
        else if (k == "--n") n_img = std::stoi(v);
        else if (k == "--threads") threads = std::stoi(v);
        else if (k == "--out") out_dir = v;
        else if (k == "--cfg") cfg_s = v;
    }
    CPUBackend::cpu_threads = threads;
    Module::initBackend(MLLM_CPU);
    auto cv = parse_ints(cfg_s);   // hidden,heads,ffn,blocks,patch,img_hw,classes
    if (cv.size() != 7) { fprintf(stderr, "--cfg needs 7 integers\n"); return 2; }
    ViTConfig config("base", cv[4], cv[5], cv[6]);
    config.hidden_dim = cv[0];
    config.head_size = cv[1];
    config.ffn_hidden = cv[2];
    config.block_num = cv[3];
    auto model = ViTModel(config);
    model.load(model_path);

    const int hw = cv[5];
    const size_t per = (size_t)hw * 3 * hw;
    std::ifstream f(img_path, std::ios::binary);
    if (!f) { fprintf(stderr, "cannot open %s\n", img_path.c_str()); return 2; }
    std::vector<float> img(per * n_img);
    f.read((char *)img.data(), img.size() * 4);
    Backend *bn = Backend::global_backends[MLLM_CPU].get();
    std::vector<float> all;
    double ms = 0;
    for (int b = 0; b < n_img; ++b) {
        Tensor t(1, hw, 3, hw, bn, true);
        t.setName("input");
        Tensor::tensor_status = TENSOR_STATIC_INIT;
        t.setTtype(INPUT_TENSOR);
        const float *p = img.data() + per * b;
        for (int h = 0; h < hw; ++h)
            for (int c = 0; c < 3; ++c)
                for (int w = 0; w < hw; ++w) t.setDataAt<float>(0, h, c, w, p[((size_t)h * 3 + c) * hw + w]);
        auto t0 = std::chrono::steady_clock::now();
        auto result = model({t});
        auto t1 = std::chrono::steady_clock::now();
        ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
        Tensor &lg = result[0];
        for (int i = 0; i < lg.dimension(); ++i) all.push_back(lg.dataAt<float>(0, 0, 0, i));
    }
    std::ofstream o(out_dir + "/vit_logits.f32", std::ios::binary);
    o.write((const char *)all.data(), all.size() * 4);
    printf("{\"images\": %d, \"ms_per_image\": %.3f, \"threads\": %d}\n", n_img, ms / n_img, threads);
    return 0;
}
