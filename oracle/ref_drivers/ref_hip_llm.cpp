// oracle/ref_drivers/ref_hip_llm.cpp -- TEST INFRASTRUCTURE (oracle side), not product.
//
// BASELINE configs 1 and 2 through the boundary: the reference's own QWenForCausalLM (mllm/models/qwen/modeling_qwen.hpp:131-179) and TinyLLaMAModel
// (mllm/models/tinyllama/modeling_tinyllama.hpp:44-84) -- Module / Layer / Tensor frontend compiled from /root/reference, unchanged -- moved onto the HIP backend of
// integration/hip/ exactly as examples/demo_qwen.cpp:43-59 moves a model onto a device (`model.to(device)`, then `model.load(path)`), and driven like the demos' loop
// (forward -> host argmax -> the next input is the sampled id), with the one device-aware step the reference's own generate loop has: tensors the host reads or edits
// between forwards come back with `.cpu()` first (mllm/Module.cpp:65-70,93-95).  Same arguments and outputs as ref_llm.cpp, plus the JSON report of hip_driver_common.hpp.
// Built by oracle/Makefile.ref into oracle/_ref/; run by tests/test_gpu_adapter.py.
//
// usage: ref_hip_llm --family qwen|tinyllama --model f.mllm --ids ids.i32 --steps 8 --threads 4 --out dir --cfg hidden,inter,layers,heads,kv_heads,vocab,cache_limit,tie
#include <chrono>
#include <cstdio>
#include <string>
#include <vector>

#include "models/qwen/configuration_qwen.hpp"
#include "models/qwen/modeling_qwen.hpp"
#include "models/tinyllama/configuration_tinyllama.hpp"
#include "models/tinyllama/modeling_tinyllama.hpp"
#include "backends/cpu/CPUBackend.hpp"

#include "hip_driver_common.hpp"

using namespace mllm;

template <typename Model>
static void run(Model &model, HIPBackend *hip, const std::vector<int32_t> &ids, int steps, int dump_every, const std::string &out_dir) {
    Backend *bn = Backend::global_backends[MLLM_CPU].get();
    Tensor t_ids(1, 1, (int)ids.size(), 1, bn, true);
    t_ids.setName("input_ids");
    Tensor::tensor_status = TENSOR_STATIC_INIT;
    t_ids.setTtype(INPUT_TENSOR);
    for (size_t i = 0; i < ids.size(); ++i) t_ids.setDataAt<float>(0, 0, (int)i, 0, (float)ids[i]);
    std::vector<int32_t> tokens;
    std::vector<double> ms;
    for (int step = 0; step < steps; ++step) {
        auto t0 = std::chrono::steady_clock::now();
        auto result = model({t_ids});
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
        t_ids.cpu();                 // the Embedding layer migrated the ids to the device (mllm/Layer.hpp:159-163); the host writes the next id
        t_ids.reshape(1, 1, 1, 1);   // the demos' chatPostProcessing: the next input is the sampled id alone
        t_ids.alloc();
        t_ids.setDataAt<float>(0, 0, 0, 0, (float)best);
    }
    write_file<int32_t>(out_dir + "/tokens.i32", tokens.data(), tokens.size());
    hip_report(hip, ids.size(), ms);
    model.profiling();
}

int main(int argc, char **argv) {
    std::string family = "qwen", model_path, ids_path, out_dir = ".", cfg_s;
    int steps = 8, threads = 4, dump_every = 1;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i], v = argv[i + 1];
        if (k == "--family") family = v;
        else if (k == "--model") model_path = v;
        else if (k == "--ids") ids_path = v;
        else if (k == "--steps") steps = std::stoi(v);
        else if (k == "--threads") threads = std::stoi(v);
        else if (k == "--out") out_dir = v;
        else if (k == "--cfg") cfg_s = v;
        else if (k == "--dump-every") dump_every = std::stoi(v);
    }
    CPUBackend::cpu_threads = threads;
    Module::initBackend(MLLM_CPU);
    HIPBackend *hip = installHIPBackend(0);      // what `case MLLM_HIP` of Module::initBackend does upstream
    auto cv = parse_ints(cfg_s);   // hidden,inter,layers,heads,kv_heads,vocab,cache_limit,tie
    if (cv.size() != 8) { fprintf(stderr, "--cfg needs 8 integers\n"); return 2; }
    auto ids = read_file<int32_t>(ids_path);
    if (family == "qwen") {
        QWenConfig config(cv[6], "0.5B", RoPEType::HFHUBROPE);
        config.hidden_size = cv[0];
        config.intermediate_size = cv[1];
        config.num_hidden_layers = cv[2];
        config.num_attention_heads = cv[3];
        config.num_key_value_heads = cv[4];
        config.vocab_size = cv[5];
        config.tie_embedding_words = cv[7] != 0;
        auto model = QWenForCausalLM(config);
        model.to(MLLM_HIP_BACKEND_TYPE);      // examples/demo_qwen.cpp:57
        model.load(model_path);               // :59 -- every Op::load goes through Backend::load_from_file
        hip->sync();
        run(model, hip, ids, steps, dump_every, out_dir);
    } else {
        TinyLLaMAConfig config(cv[6], "1.1B", HFHUBROPE, cv[5]);
        config.hidden_dim = cv[0];
        config.ffn_hidden = cv[1];
        config.block_num = cv[2];
        config.head_size = cv[3];
        config.kv_head_size = cv[4];
        auto model = TinyLLaMAModel(config);
        model.to(MLLM_HIP_BACKEND_TYPE);
        model.load(model_path);
        hip->sync();
        run(model, hip, ids, steps, dump_every, out_dir);
    }
    return 0;
}
