// oracle/ref_drivers/ref_ops.cpp -- TEST INFRASTRUCTURE (oracle), not product.
//
// Runs single reference ops / blocks on the reference's x86 CPU backend through its own frontend (Layer / Module /
// Tensor functional API) on raw fp32 inputs, and dumps the fp32 outputs.  This is how golden vectors for the A-rows
// of SURVEY §8(a) are captured (the reference's own tests hold none): same call protocol as
// test/cpu/CPUTest.hpp:18-53 but driven through Module::load + operator() so the trace/load/execute passes
// (mllm/backends/cpu/CPUBackend.cpp:314-405) are the real ones.
//
// usage: ref_ops case=<name> weights=<file.mllm> out=<dir> [p=a,b,c,...] call=<file:b,h,s,d[+file:shape...]> [call=...]
//   5-number shapes (n,c,t,h,w) build the BCTHW patch tensor the way processing_qwen2_vl.hpp:249-252 does.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "models/qwen2_vl/configuration_qwen2_vl.hpp"
#include "models/qwen2_vl/modeling_qwen2_vl.hpp"
#include "models/minicpm_moe/modeling_minicpm_moe.hpp"
#include "backends/cpu/CPUBackend.hpp"

using namespace mllm;

static std::vector<float> read_f32(const std::string &p) {
    std::ifstream f(p, std::ios::binary | std::ios::ate);
    if (!f) { fprintf(stderr, "cannot open %s\n", p.c_str()); exit(2); }
    size_t n = f.tellg();
    f.seekg(0);
    std::vector<float> v(n / 4);
    f.read((char *)v.data(), n);
    return v;
}
static std::vector<std::string> split(const std::string &s, char c) {
    std::vector<std::string> o;
    std::stringstream ss(s);
    std::string t;
    while (std::getline(ss, t, c)) o.push_back(t);
    return o;
}
static std::vector<float> P;  // numeric params of the case
static float p(int i, float dflt = 0) { return i < (int)P.size() ? P[i] : dflt; }

class OpsModule final : public Module {
public:
    std::string kind;
    Layer l0, l1;
    Parameter param;
    Softmax sm;
    explicit OpsModule(const std::string &k) : kind(k) {
        if (k == "linear") l0 = Linear((int)p(0), (int)p(1), p(2) != 0, "lin");
        else if (k == "rmsnorm") l0 = RMSNorm((int)p(0), p(1), "norm");
        else if (k == "layernorm") l0 = LayerNorm((int)p(0), true, p(1), "ln");
        else if (k == "silu") l0 = SiLU("act");
        else if (k == "gelu") l0 = GELU("act");
        else if (k == "quickgelu") l0 = QuickGELU("act");
        else if (k == "softmax") sm = Softmax(DIMENSION, p(0) != 0, "sm");
        else if (k == "embedding") l0 = Embedding((int)p(0), (int)p(1), "model.embed_tokens");
        else if (k == "mm_tied") param = Parameter(1, (int)p(0), 1, (int)p(1), "model.embed_tokens.weight");
        else if (k == "conv3d") l0 = Convolution3D(3, (int)p(0), {2, 14, 14}, {2, 14, 14}, VALID, false, "proj");
        else if (k == "conv2d") l0 = Convolution2D(3, (int)p(0), {(int)p(1), (int)p(1)}, {(int)p(1), (int)p(1)}, VALID, p(2) != 0, "proj");
        else if (k == "vrope") l0 = VisionRoPE((int)p(0), (int)p(1), "rot");
        else if (k == "mrope") l0 = MultimodalRoPE(p(0), (int)p(1), {16, 24, 24}, "rope");
        else if (k == "rope") l0 = RoPE((int)p(0), p(1), (int)p(2), "rope");
        else if (k == "rope3") {   // llama3 frequency scaling (Layer.hpp:493-531 -> CPURoPE.cpp:33-71): p = pose_type, theta, max_pos, heads, D, factor, low, high, original max_pos
            map<string, std::any> sc = {{"rope_type", std::string("llama3")}, {"factor", p(5)}, {"low_freq_factor", p(6)}, {"high_freq_factor", p(7)},
                                        {"original_max_position_embeddings", (int)p(8)}};
            RoPEConfig cfg = {{"rope_theta", p(1)}, {"max_position_embeddings", (int)p(2)}, {"rope_scaling", sc}};
            l0 = RoPE((int)p(0), cfg, "rope");
        }
        else if (k == "fa2" || k == "mm_bhsd") {}
        // ---- N4: ops of the other model families
        else if (k == "swmask") l0 = SlidingWindowMask((int)p(0), "mask");                       // p: window, heads, keys
        else if (k == "ntkrope") {                                                                // p: theta, max_pos, original max_pos, heads, D, then D/2 long and D/2 short factors
            const int half = (int)p(4) / 2;
            std::vector<float> lf, sf;
            for (int i = 0; i < half; ++i) { lf.push_back(p(5 + i)); sf.push_back(p(5 + half + i)); }
            l0 = NTKRoPE(HFHUBROPE, p(0), (int)p(1), (int)p(2), lf, sf, "rope");
        }
        else if (k == "topk" || k == "scatter_add" || k == "fuyu_gather" || k == "gather_rows" || k == "argsort" || k == "bincount") {}
        else { fprintf(stderr, "unknown case %s\n", k.c_str()); exit(2); }
    }
    vector<Tensor> Forward(vector<Tensor> in, vector<std::any> args) override {
        if (kind == "softmax") return {sm(in[0])};
        if (kind == "mm_tied") return {Tensor::mm(in[0], param().transpose(Chl::SEQUENCE, Chl::DIMENSION))};
        if (kind == "conv3d") { auto e = l0(in[0]); return {e.view(1, 1, -1, (int)p(0))}; }
        if (kind == "vrope") {
            // in[0] = q [1,1,S,H*D], in[1] = grid_thw ; rotary table then F_APPLY_VISIOROPE (modeling_qwen2_vl.hpp:70-74,179)
            auto rot = l0(in[1]);
            auto q = in[0].view(-1, (int)p(2), -1, (int)p(3));
            q = Tensor::apply_rotary_pos_emb_vision(q, rot);
            return {q.view(-1, 1, -1, (int)(p(2) * p(3)))};
        }
        if (kind == "mrope") {
            auto q = in[0].view(-1, (int)p(2), -1, (int)p(3));
            MultimodalRoPE &r = (MultimodalRoPE &)l0;
            q = r(q, in[1]);
            return {q.view(-1, 1, -1, (int)(p(2) * p(3)))};
        }
        if (kind == "rope" || kind == "rope3") {
            auto q = in[0].view(-1, (int)p(3), -1, (int)p(4));
            RoPE &r = (RoPE &)l0;
            q = r(q);
            return {q.view(-1, 1, -1, (int)(p(3) * p(4)))};
        }
        if (kind == "swmask") {      // scores [1, H, S, keys] (CPUSlidingWindowMask.cpp:30-58)
            auto x = in[0].view(-1, (int)p(1), -1, (int)p(2));
            return {l0(x).view(-1, 1, -1, (int)(p(1) * p(2)))};
        }
        if (kind == "ntkrope") {     // CPUNTKRoPE.cpp:27-80 table, :81-190 rotation
            auto q = in[0].view(-1, (int)p(3), -1, (int)p(4));
            NTKRoPE &r = (NTKRoPE &)l0;
            q = r(q);
            return {q.view(-1, 1, -1, (int)(p(3) * p(4)))};
        }
        if (kind == "topk") {        // CPUTopkFunc.hpp:48-92; p: k, which output (0 values, 1 indices), axis (0 DIMENSION, 1 HEAD: input [1, H, S, 1])
            if (p(2) != 0) {
                auto x = in[0].view(-1, (int)p(3), -1, 1);
                auto r = Tensor::topk(x, (int)p(0), HEAD);
                return {r[(int)p(1)].view(-1, 1, -1, (int)p(0))};
            }
            auto r = Tensor::topk(in[0], (int)p(0), DIMENSION);
            return {r[(int)p(1)]};
        }
        if (kind == "argsort") return {in[0].argsort()};      // CPUArgSortFunc.hpp
        if (kind == "bincount") return {in[0].bincount()};
        if (kind == "scatter_add") { // CPUScatterAddFunc.hpp:27-60: dest [1,1,S,D] += src [1,1,R,D] at rows idx [1,1,1,R], in order
            auto dst = in[0];
            dst.scatter_add(in[1], in[2]);
            return {dst};
        }
        if (kind == "gather_rows") return {in[0].clip(in[1], SEQUENCE)};       // Tensor.cpp:580 -> CPUClipTensorFunc
        if (kind == "fuyu_gather") return {Tensor::fuyu_gather_embd(in[0], in[1], in[2])};   // CPUFuyuGatherEmbdFunc.hpp:45-62
        if (kind == "mm_bhsd") {     // the eager-attention form of F_MM (CPUMatmulFunc.hpp:123-172 -> compute/GemmFp.hpp:104-150): p = transpose the right operand first (q k^T) or not (p v)
            auto a = in[0].transpose(Chl::HEAD, Chl::SEQUENCE);      // BSHD -> BHSD, a data move (CPUTransposeFunc.hpp:129-160)
            auto b = in[1].transpose(Chl::HEAD, Chl::SEQUENCE);
            if (p(0) != 0) b = b.transpose(Chl::SEQUENCE, Chl::DIMENSION);      // BHSD -> BHDS (:167-185)
            return {Tensor::mm(a, b)};
        }
        if (kind == "fa2") {
            // in: q [1,1,Sq,Hq*D], k,v [1,1,Sk,Hkv*D] all fp32 (the vision / fp32-KV path, FlashAttention2.hpp:87)
            auto q = in[0].view(-1, (int)p(0), -1, (int)p(2));
            auto k = in[1].view(-1, (int)p(1), -1, (int)p(2));
            auto v = in[2].view(-1, (int)p(1), -1, (int)p(2));
            auto o = Tensor::flash_attention2_forward(q, k, v, p(3) != 0);
            return {o.view(-1, 1, -1, (int)(p(0) * p(2)))};
        }
        return {l0(in[0])};
    }
};

// one VisionBlock (modeling_qwen2_vl.hpp:112-137) driven with its rotary table: inputs x [1,1,N,V], grid_thw
class VBlockModule final : public Module {
public:
    Layer rot;
    VisionBlock blk;
    VBlockModule(int V, const std::string &base) {
        Qwen2VLConfig cfg(64, "1.5b");
        rot = VisionRoPE((V / 16) / 2, 2, "visual.rot_pos_emb");
        blk = VisionBlock(V, 16, V * 4, "QuickGELU", cfg.attn_implementation, cfg.vision_names_config, base);
    }
    vector<Tensor> Forward(vector<Tensor> in, vector<std::any> args) override {
        auto r = rot(in[1]);
        vector<float> cu = {0.0F, 1.0F};
        auto cu_t = Tensor(cu);
        return blk({in[0], cu_t, r});
    }
};

// the vision tower re-assembled from the reference's public sub-modules, truncated after `nblk` blocks, merger optional:
// used to localise divergences (p: hidden, V, nblk, merger)
class VisionPartModule final : public Module {
public:
    Qwen2PatchEmbed pe;
    Layer rot;
    vector<VisionBlock> blocks;
    PatchMerger merger;
    bool do_merger;
    VisionPartModule(int hidden, int V, int nblk, bool mg) : do_merger(mg) {
        Qwen2VLConfig cfg(64, "1.5b");
        auto &n = cfg.vision_names_config;
        std::string base = n.vison_model_name;
        pe = Qwen2PatchEmbed(V, 14, 336, n, base + n.patch_embed_name);
        rot = VisionRoPE((V / 16) / 2, 2, base + ".rot_pos_emb");
        blocks = List<VisionBlock>(nblk, V, 16, V * 4, "QuickGELU", cfg.attn_implementation, n, base + n._layer_name);
        merger = PatchMerger(hidden, V, 2, n, base + n._merger_name);
    }
    vector<Tensor> Forward(vector<Tensor> in, vector<std::any> args) override {
        auto h = pe({in[0]})[0];
        auto r = rot(in[1]);
        vector<float> cu = {0.0F, 1.0F};
        auto cu_t = Tensor(cu);
        for (auto &b : blocks) h = b({h, cu_t, r})[0];
        if (do_merger) h = merger({h})[0];
        return {h};
    }
};

static Tensor make_input(const std::string &spec, int idx) {
    auto parts = split(spec, ':');
    auto data = read_f32(parts[0]);
    auto shp = split(parts[1], ',');
    Backend *bn = Backend::global_backends[MLLM_CPU].get();
    Tensor::tensor_status = TENSOR_STATIC_INIT;
    if (shp.size() == 5) {
        int n = std::stoi(shp[0]), pe = std::stoi(shp[1]) * std::stoi(shp[2]) * std::stoi(shp[3]) * std::stoi(shp[4]);
        Tensor t(1, n, 1, pe, bn, true);
        t.setName("input" + std::to_string(idx));
        t.setTtype(INPUT_TENSOR);
        for (int i = 0; i < n; ++i)
            for (int d = 0; d < pe; ++d) t.setDataAt<float>(0, i, 0, d, data[(size_t)i * pe + d]);
        t.reshape(n, std::stoi(shp[1]), std::stoi(shp[2]), std::stoi(shp[3]), std::stoi(shp[4]));
        return t;
    }
    int b = std::stoi(shp[0]), h = std::stoi(shp[1]), s = std::stoi(shp[2]), d = std::stoi(shp[3]);
    Tensor t(b, h, s, d, bn, true);
    t.setName("input" + std::to_string(idx));
    t.setTtype(INPUT_TENSOR);
    // file order is logical [b][h][s][d]; setDataAt honours the tensor's memory layout (BSHD)
    size_t i = 0;
    for (int bi = 0; bi < b; ++bi)
        for (int hi = 0; hi < h; ++hi)
            for (int si = 0; si < s; ++si)
                for (int di = 0; di < d; ++di) t.setDataAt<float>(bi, hi, si, di, data[i++]);
    return t;
}

static void dump(Tensor &o, const std::string &path) {
    std::vector<float> v;
    // logical [b][h][s][d] order
    for (int b = 0; b < o.batch(); ++b)
        for (int h = 0; h < o.head(); ++h)
            for (int s = 0; s < o.sequence(); ++s)
                for (int d = 0; d < o.dimension(); ++d) {
                    if (o.dtype() == MLLM_TYPE_F16) v.push_back(MLLM_FP16_TO_FP32(o.dataAt<mllm_fp16_t>(b, h, s, d)));
                    else v.push_back(o.dataAt<float>(b, h, s, d));
                }
    std::ofstream f(path, std::ios::binary);
    f.write((const char *)v.data(), v.size() * 4);
    printf("%s shape %d,%d,%d,%d dtype %d\n", path.c_str(), o.batch(), o.head(), o.sequence(), o.dimension(), (int)o.dtype());
}

int main(int argc, char **argv) {
    std::string kind, weights, out = ".";
    std::vector<std::string> calls;
    int threads = 4;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto eq = a.find('=');
        std::string k = a.substr(0, eq), v = a.substr(eq + 1);
        if (k == "case") kind = v;
        else if (k == "weights") weights = v;
        else if (k == "out") out = v;
        else if (k == "threads") threads = std::stoi(v);
        else if (k == "p") for (auto &s : split(v, ',')) P.push_back(std::stof(s));
        else if (k == "call") calls.push_back(v);
    }
    CPUBackend::cpu_threads = threads;
    Module::initBackend(MLLM_CPU);

    std::unique_ptr<Module> model;
    if (kind == "vision") {
        // p: hidden_dim(out), vision_embed_dim  (heads 16, mlp 4x, QuickGELU, patch 14, 32 blocks: modeling_qwen2_vl.hpp:371)
        Qwen2VLConfig cfg(64, "1.5b");
        model.reset(new Qwen2VisionModel((int)p(0), (int)p(1), 16, (int)p(1) * 4, "QuickGELU", 14, 336, 32, 2,
                                         cfg.attn_implementation, cfg.vision_names_config, cfg.vision_names_config.vison_model_name));
    } else if (kind == "visionpart") {
        model.reset(new VisionPartModule((int)p(0), (int)p(1), (int)p(2), p(3) != 0));
    } else if (kind == "vblock") {
        model.reset(new VBlockModule((int)p(0), "visual.blocks." + std::to_string((int)p(1)) + "."));
    } else if (kind == "attn" || kind == "decoder" || kind == "mlp") {
        // p: hidden, inter, heads, kv_heads, cache_limit
        Qwen2VLConfig cfg((int)p(4, 64), "1.5b");
        cfg.hidden_size = (int)p(0);
        cfg.intermediate_size = (int)p(1);
        cfg.num_attention_heads = (int)p(2);
        cfg.num_key_value_heads = (int)p(3);
        std::string base = cfg.names_config.blk_name + "0.";
        if (kind == "attn") model.reset(new QWen2Attention(cfg, cfg.names_config, base + cfg.names_config._attn_base_name));
        else if (kind == "mlp") model.reset(new QWen2MLP(cfg.hidden_size, cfg.intermediate_size, cfg.names_config, base + cfg.names_config._ffn_base_name));
        else model.reset(new QWen2Decoder(cfg, cfg.names_config, base));
    } else if (kind == "moe") {
        // the reference's own sparse-MoE block (models/minicpm_moe/modeling_minicpm_moe.hpp:41-115): router Linear -> softmax -> top-k -> renormalise -> argsort / bincount
        // routing -> per expert gather, gate / up / down MLP, row scale, scatter_add.  p: hidden, inter, experts, experts per token
        MiniCPMConfig cfg(64, "2B");
        cfg.hidden_size = (int)p(0);
        cfg.intermediate_size = (int)p(1);
        cfg.num_experts = (int)p(2);
        cfg.num_experts_per_tok = (int)p(3);
        model.reset(new MiniCPMMoE(cfg, cfg.names_config, cfg.names_config.blk_name + "0." + cfg.names_config._ffn_base_name));
    } else {
        model.reset(new OpsModule(kind));
    }
    model->load(weights);

    for (size_t c = 0; c < calls.size(); ++c) {
        std::vector<Tensor> in;
        int idx = 0;
        for (auto &spec : split(calls[c], '+')) in.push_back(make_input(spec, idx++));
        auto res = (*model)(in);
        dump(res[0], out + "/out" + std::to_string(c) + ".f32");
    }
    return 0;
}
