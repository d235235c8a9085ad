// oracle/ref_drivers/ref_hip_vit.cpp -- TEST INFRASTRUCTURE (oracle side), not product.
//
// BASELINE config 3 through the boundary: the reference's own ViTModel (mllm/models/vit/modeling_vit.hpp:91-111) -- frontend compiled from /root/reference, unchanged --
// moved onto the HIP backend of integration/hip/ (`model.to(device)` then `model.load(path)`, examples/demo_qwen.cpp:43-59) and fed one image per forward like
// examples/demo_vit.cpp:30-38, raw fp32 images [H][C][W] (ViTProcessor::img2Tensor's layout, processing_vit.hpp:18-29).  Same arguments and outputs as ref_vit.cpp, plus
// the JSON report of hip_driver_common.hpp.  Built by oracle/Makefile.ref into oracle/_ref/; run by tests/test_gpu_adapter.py.
//
// usage: ref_hip_vit --model f.mllm --img imgs.f32 --n 2 --threads 4 --out dir --cfg hidden,heads,ffn,blocks,patch,img_hw,classes
#include <chrono>
#include <cstdio>
#include <string>
#include <vector>

#include "models/vit/configuration_vit.hpp"
#include "models/vit/modeling_vit.hpp"
#include "backends/cpu/CPUBackend.hpp"

#include "hip_driver_common.hpp"

using namespace mllm;

int main(int argc, char **argv) {
    std::string model_path, img_path, out_dir = ".", cfg_s;
    int n_img = 1, threads = 4;
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
    HIPBackend *hip = installHIPBackend(0);
    auto cv = parse_ints(cfg_s);   // hidden,heads,ffn,blocks,patch,img_hw,classes
    if (cv.size() != 7) { fprintf(stderr, "--cfg needs 7 integers\n"); return 2; }
    ViTConfig config("base", cv[4], cv[5], cv[6]);
    config.hidden_dim = cv[0];
    config.head_size = cv[1];
    config.ffn_hidden = cv[2];
    config.block_num = cv[3];
    auto model = ViTModel(config);
    model.to(MLLM_HIP_BACKEND_TYPE);
    model.load(model_path);
    hip->sync();

    const int hw = cv[5];
    const size_t per = (size_t)hw * 3 * hw;
    auto img = read_file<float>(img_path);
    if (img.size() < per * n_img) { fprintf(stderr, "image file too short\n"); return 2; }
    Backend *bn = Backend::global_backends[MLLM_CPU].get();
    std::vector<float> all;
    std::vector<double> ms;
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
        ms.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
        Tensor &lg = result[0];
        if (lg.backend()->type() != MLLM_CPU) lg.cpu();
        for (int i = 0; i < lg.dimension(); ++i) all.push_back(lg.dataAt<float>(0, 0, 0, i));
    }
    write_file<float>(out_dir + "/vit_logits.f32", all.data(), all.size());
    double tot = 0;
    for (double m : ms) tot += m;
    char extra[128];
    snprintf(extra, sizeof extra, "\"images\": %d, \"ms_per_image\": %.3f, \"ms_last_image\": %.3f, ", n_img, tot / n_img, ms.back());
    hip_report(hip, 0, ms, extra);
    return 0;
}
