// oracle/ref_drivers/ref_sampling.cpp -- TEST INFRASTRUCTURE (oracle), not product.
//
// Pins the pre-draw state of the reference's sampling methods: _LlmTextGenerateTopkSamplingMethod::generate and
// _LlmTextGenerateToppSamplingMethod::generate (mllm/Generate.cpp:45-142) run, from the compiled reference library, on rows of scores this driver reads
// from a file, and the candidate ids + probabilities they hand to their draw are written out.
//
// The draw itself, _sample_element<unsigned int> (mllm/Generate.hpp:38-44), seeds a std::mt19937 from std::random_device, so its result cannot be compared.
// It is a weak template instantiation inside libmllm_cpu_ref.so; this executable defines the same specialisation, which the dynamic linker binds the
// library's call to (symbol interposition; the driver is linked with -rdynamic), records the two vectors the method passes in, and returns the first
// candidate.  Everything up to that call -- std::partial_sort / std::sort, the float/double temperature softmax, the renormalisation -- is the reference's
// own compiled code.
//
// usage: ref_sampling --in rows.f32 --n N --rows R --k 5 --p 0.92 --temp 0.7 --out dir
//        rows.f32 holds R rows of N floats: logits for top-k, probabilities (a softmax row) for top-p.
// out:   topk_<r>.idx / topk_<r>.prob, topp_<r>.idx / topp_<r>.prob (uint32 ids, float32 probabilities as passed to the draw)
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "Generate.hpp"
#include "backends/cpu/CPUBackend.hpp"
#include "Module.hpp"

static std::vector<unsigned int> g_elems;
static std::vector<float> g_probs;
static int g_calls = 0;

namespace mllm {
template <>
unsigned int _sample_element<unsigned int>(const std::vector<unsigned int> &elements, const std::vector<float> &probabilities) {
    g_elems = elements;
    g_probs = probabilities;
    ++g_calls;
    return elements[0];
}
}  // namespace mllm

using namespace mllm;

template <typename T>
static void write_file(const std::string &p, const T *d, size_t n) {
    std::ofstream f(p, std::ios::binary);
    f.write((const char *)d, n * sizeof(T));
}

int main(int argc, char **argv) {
    std::string in_path, out_dir = ".";
    int n = 0, rows = 1, k = 5;
    float p = 0.92f, temp = 0.7f;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string a = argv[i], v = argv[i + 1];
        if (a == "--in") in_path = v;
        else if (a == "--n") n = std::stoi(v);
        else if (a == "--rows") rows = std::stoi(v);
        else if (a == "--k") k = std::stoi(v);
        else if (a == "--p") p = std::stof(v);
        else if (a == "--temp") temp = std::stof(v);
        else if (a == "--out") out_dir = v;
    }
    Module::initBackend(MLLM_CPU);
    Backend *bn = Backend::global_backends[MLLM_CPU].get();
    std::vector<float> data((size_t)rows * n);
    {
        std::ifstream f(in_path, std::ios::binary);
        if (!f) { fprintf(stderr, "cannot open %s\n", in_path.c_str()); return 2; }
        f.read((char *)data.data(), data.size() * 4);
    }
    _LlmTextGenerateTopkSamplingMethod topk(k, temp);
    _LlmTextGenerateToppSamplingMethod topp(p, temp);
    for (int r = 0; r < rows; ++r) {
        Tensor::tensor_status = TENSOR_STATIC_INIT;
        Tensor t(1, 1, 1, n, bn, true);
        for (int i = 0; i < n; ++i) t.setDataAt<float>(0, 0, 0, i, data[(size_t)r * n + i]);
        const int before = g_calls;
        const float mx = *std::max_element(data.begin() + (size_t)r * n, data.begin() + (size_t)(r + 1) * n);
        topk.generate(t);
        if (g_calls == before) { fprintf(stderr, "the library's draw was not interposed (inlined call?)\n"); return 3; }
        write_file<unsigned int>(out_dir + "/topk_" + std::to_string(r) + ".idx", g_elems.data(), g_elems.size());
        write_file<float>(out_dir + "/topk_" + std::to_string(r) + ".prob", g_probs.data(), g_probs.size());
        if (mx <= 1.0f) {      // top-p takes probabilities only (it throws on a score above 1, Generate.cpp:104-106)
            g_elems.clear(); g_probs.clear();
            const int b2 = g_calls;
            unsigned int ret = topp.generate(t);
            if (g_calls == b2) { g_elems.assign(1, ret); g_probs.assign(1, 1.0f); }      // a one-element nucleus returns before the draw (:117-119)
            write_file<unsigned int>(out_dir + "/topp_" + std::to_string(r) + ".idx", g_elems.data(), g_elems.size());
            write_file<float>(out_dir + "/topp_" + std::to_string(r) + ".prob", g_probs.data(), g_probs.size());
        }
    }
    printf("{\"rows\": %d, \"calls\": %d}\n", rows, g_calls);
    return 0;
}
