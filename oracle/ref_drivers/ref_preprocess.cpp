// oracle/ref_drivers/ref_preprocess.cpp -- TEST INFRASTRUCTURE (oracle), not product.
//
// the reference's own image preprocessing of Qwen2-VL, Qwen2VLImageProcessor::preprocess_images (mllm/models/qwen2_vl/processing_qwen2_vl.hpp:190-235:
// stbi decode -> RescaleImage / 255 -> smart_resize + stb_image_resize2 cubic B-spline (PreProcess.cpp:84-154) -> NormalizeImages -> the frame doubled ->
// convertPatches), on an image file given as bytes.  Writes the flattened patches `[grid_t*grid_h*grid_w][3*2*14*14]` and the grid.
// Linked with the reference's mllm/processor/PreProcess.cpp compiled in place (oracle/Makefile.ref).
//
// usage: ref_preprocess --img file.bmp --out dir      -> dir/patches.f32, dir/grid.i32
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "models/qwen2_vl/processing_qwen2_vl.hpp"

using namespace mllm;

int main(int argc, char **argv) {
    std::string img, out_dir = ".";
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i], v = argv[i + 1];
        if (k == "--img") img = v;
        else if (k == "--out") out_dir = v;
    }
    std::ifstream f(img, std::ios::binary | std::ios::ate);
    if (!f) { fprintf(stderr, "cannot open %s\n", img.c_str()); return 2; }
    size_t n = f.tellg();
    f.seekg(0);
    std::vector<uint8_t> bytes(n);
    f.read((char *)bytes.data(), n);
    Qwen2VLImageProcessor proc;
    auto res = proc.preprocess_images(bytes.data(), bytes.size());
    auto &rows = res.first;
    std::vector<float> flat;
    for (auto &r : rows) flat.insert(flat.end(), r.begin(), r.end());
    std::ofstream(out_dir + "/patches.f32", std::ios::binary).write((const char *)flat.data(), flat.size() * 4);
    std::vector<int32_t> grid(res.second.begin(), res.second.end());
    std::ofstream(out_dir + "/grid.i32", std::ios::binary).write((const char *)grid.data(), grid.size() * 4);
    printf("{\"rows\": %zu, \"cols\": %zu, \"grid\": [%d, %d, %d]}\n", rows.size(), rows.empty() ? 0 : rows[0].size(), grid[0], grid[1], grid[2]);
    return 0;
}
