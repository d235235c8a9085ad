// oracle/ref_drivers/hip_driver_common.hpp -- TEST INFRASTRUCTURE (oracle side), not product.  What the ref_hip_* drivers share: file helpers and the one-line JSON
// report (which Ops the HIP backend refused = ran on the CPU backend, how many ran on the device, the device blocks still alive at exit).
#ifndef HIP_DRIVER_COMMON_HPP
#define HIP_DRIVER_COMMON_HPP
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "HIPBackend.hpp"

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
// {"backend": "hip", ..., "hip_ops_run": n, "cpu_fallback_ops": n, "refused": [[optype, "name"], ...], "live_device_blocks": n}
static void hip_report(mllm::HIPBackend *hip, size_t prefill_tokens, const std::vector<double> &ms, const char *extra = "") {
    double dec = 0;
    for (size_t i = 1; i < ms.size(); ++i) dec += ms[i];
    std::string refused = "[";
    for (size_t i = 0; i < hip->refused().size(); ++i)
        refused += std::string(i ? ", " : "") + "[" + std::to_string(hip->refused()[i].first) + ", \"" + hip->refused()[i].second + "\"]";
    refused += "]";
    printf("{\"backend\": \"hip\", \"prefill_tokens\": %zu, \"prefill_ms\": %.3f, \"decode_steps\": %zu, \"decode_ms_mean\": %.4f, \"decode_tok_s\": %.3f, %s"
           "\"hip_ops_run\": %ld, \"fused_launches\": %ld, \"fused_ops\": %ld, \"cpu_fallback_ops\": %zu, \"refused\": %s, \"live_device_blocks\": %zu}\n",
           prefill_tokens, ms.empty() ? 0.0 : ms[0], ms.empty() ? (size_t)0 : ms.size() - 1, ms.size() > 1 ? dec / (ms.size() - 1) : 0.0,
           ms.size() > 1 ? 1000.0 * (ms.size() - 1) / dec : 0.0, extra, hip->ops_run(), hip->fused_launches(), hip->fused_ops(), hip->refused().size(), refused.c_str(), hip->live_blocks());
    fflush(stdout);
}
#endif
