// oracle/ref_drivers/ref_llm.cpp -- TEST INFRASTRUCTURE (oracle), not product.
//
// Drives the reference's own text-only causal LMs on its x86 CPU backend with raw token ids: QWenForCausalLM
// (mllm/models/qwen/modeling_qwen.hpp:131-179, BASELINE config "demo_qwen") and TinyLLaMAModel
// (mllm/models/tinyllama/modeling_tinyllama.hpp:44-84, BASELINE config "demo_tinyllama"), the way examples/demo_qwen.cpp and
// examples/demo_tinyllama.cpp loop (forward -> host argmax -> next id), without the tokenizer.  Emits greedy ids, the last-row
// logits of every step and per-forward wall times.  Built only by oracle/Makefile.ref into oracle/_ref/ against /root/reference.
//
// usage: ref_llm --family qwen|tinyllama --model f.mllm --ids ids.i32 --steps 8 --threads 8 --out dir
//                --cfg hidden,inter,layers,heads,kv_heads,vocab,cache_limit,tie
#include <chrono>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "models/qwen/configuration_qwen.hpp"
#include "models/qwen/modeling_qwen.hpp"
#include "models/tinyllama/configuration_tinyllama.hpp"
#include "models/tinyllama/modeling_tinyllama.hpp"
#include "backends/cpu/CPUBackend.hpp"

using namespace mllm;

static std::vector<int32_t> read_ids(const std::string &p) {
    std::ifstream f(p, std::ios::binary | std::ios::ate);
    if (!f) { fprintf(stderr, "cannot open %s\n", p.c_str()); exit(2); }
    size_t n = f.tellg();
    f.seekg(0);
    std::vector<int32_t> v(n / 4);
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

template <typename Model>
static void run(Model &model, const std::vector<int32_t> &ids, int steps, int threads, const std::string &out_dir) {
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
        int V = lg.dimension(), s = lg.sequence() - 1;
        std::vector<float> row(V);
        for (int i = 0; i < V; ++i) row[i] = lg.dataAt<float>(0, 0, s, i);
        int best = 0;
        for (int i = 1; i < V; ++i) if (row[i] > row[best]) best = i;
        tokens.push_back(best);
        write_file<float>(out_dir + "/logits_" + std::to_string(step) + ".f32", row.data(), V);
        t_ids.reshape(1, 1, 1, 1);   // the demos' chatPostProcessing: the next input is the sampled id alone
        t_ids.alloc();
        t_ids.setDataAt<float>(0, 0, 0, 0, (float)best);
    }
    write_file<int32_t>(out_dir + "/tokens.i32", tokens.data(), tokens.size());
    double dec = 0;
    for (size_t i = 1; i < ms.size(); ++i) dec += ms[i];
    printf("{\"prefill_tokens\": %zu, \"prefill_ms\": %.3f, \"decode_steps\": %zu, \"decode_tok_s\": %.3f, \"threads\": %d}\n", ids.size(), ms[0],
           ms.size() - 1, ms.size() > 1 ? 1000.0 * (ms.size() - 1) / dec : 0.0, threads);
}

int main(int argc, char **argv) {
    std::string family = "qwen", model_path, ids_path, out_dir = ".", cfg_s;
    int steps = 8, threads = 8;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i], v = argv[i + 1];
        if (k == "--family") family = v;
        else if (k == "--model") model_path = v;
        else if (k == "--ids") ids_path = v;
        else if (k == "--steps") steps = std::stoi(v);
        else if (k == "--threads") threads = std::stoi(v);
        else if (k == "--out") out_dir = v;
        else if (k == "--cfg") cfg_s = v;
    }
    CPUBackend::cpu_threads = threads;
    Module::initBackend(MLLM_CPU);
    auto cv = parse_ints(cfg_s);   // hidden,inter,layers,heads,kv_heads,vocab,cache_limit,tie
    if (cv.size() != 8) { fprintf(stderr, "--cfg needs 8 integers\n"); return 2; }
    auto ids = read_ids(ids_path);
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
        model.load(model_path);
        run(model, ids, steps, threads, out_dir);
    } else {
        TinyLLaMAConfig config(cv[6], "1.1B", HFHUBROPE, cv[5]);
        config.hidden_dim = cv[0];
        config.ffn_hidden = cv[1];
        config.block_num = cv[2];
        config.head_size = cv[3];
        config.kv_head_size = cv[4];
        auto model = TinyLLaMAModel(config);
        model.load(model_path);
        run(model, ids, steps, threads, out_dir);
    }
    return 0;
}
