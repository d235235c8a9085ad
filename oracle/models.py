"""oracle/models.py -- TEST INFRASTRUCTURE.  The reference's Qwen2-VL graphs composed from the oracle's restated ops
(numpy + oracle/restate.c), used to localise divergences stage by stage and as the `port` CPU baseline.
Mirrors mllm/models/qwen2_vl/modeling_qwen2_vl.hpp:21-191 (vision) and :193-404 (LLM)."""
from __future__ import annotations

import numpy as np

from mllm_amd import mllmfile as mf
from . import oracle as orc


class Weights:
    def __init__(self, path):
        self.f = mf.MllmFile(path)

    def lin(self, x, name, N, bias=True, out_f16=False):
        b = self.f.f32(name + ".bias") if bias else None
        return orc.linear(x, self.f.raw(name + ".weight"), self.f.dtype(name + ".weight"), N, b, out_f16=out_f16)

    def v(self, name):
        return self.f.f32(name)


def vision_forward(w: Weights, cfg, pix, grid, stages=None):
    t, h, wd = [int(v) for v in grid]
    N, V, H, D = t * h * wd, cfg.v_dim, cfg.v_heads, cfg.v_head_dim
    x = orc.patch_gemm(pix, w.v("visual.patch_embed.proj.weight"))
    if stages is not None:
        stages["patch"] = x.copy()
    ang = orc.vision_rope_angles(t, h, wd, cfg.v_merge, D // 2)
    for i in range(cfg.v_blocks):
        p = f"visual.blocks.{i}."
        y = orc.layernorm(x, w.v(p + "norm1.weight"), w.v(p + "norm1.bias"), 1e-6)
        qkv = w.lin(y, p + "attn.qkv", 3 * V)
        q = orc.vision_rope_apply(np.ascontiguousarray(qkv[:, :V]), N, H, D, ang)
        k = orc.vision_rope_apply(np.ascontiguousarray(qkv[:, V:2 * V]), N, H, D, ang)
        o = orc.attention(q, k, np.ascontiguousarray(qkv[:, 2 * V:]), N, N, H, H, D, False)
        r = w.lin(o, p + "attn.proj", V) + x
        y = orc.layernorm(r, w.v(p + "norm2.weight"), w.v(p + "norm2.bias"), 1e-6)
        a = orc.quickgelu(w.lin(y, p + "mlp.fc1", 4 * V))
        x = w.lin(a, p + "mlp.fc2", V) + r
        if stages is not None:
            stages[f"block{i}"] = x.copy()
    y = orc.layernorm(x, w.v("visual.merger.ln_q.weight"), w.v("visual.merger.ln_q.bias"), 1e-6)
    MM = V * cfg.v_merge * cfg.v_merge
    y = y.reshape(-1, MM)
    y = orc.gelu(w.lin(y, "visual.merger.mlp.0", MM))
    return w.lin(y, "visual.merger.mlp.2", cfg.hidden)


def rope_index(cfg, ids, grid):
    """get_rope_index (modeling_qwen2_vl.hpp:436-595) for batch 1 and at most one image grid; returns pos [3][S] fp32."""
    ids = [int(v) for v in ids]
    S = len(ids)
    if grid is None:
        return np.tile(np.arange(S, dtype=np.float32), (3, 1))
    lp = [[], [], []]
    st, cur = 0, 0
    n_img = sum(1 for j in range(S - 1) if ids[j] == cfg.vision_start_token_id and ids[j + 1] == cfg.image_token_id)
    n_starts = sum(1 for j in range(S - 1) if ids[j] == cfg.vision_start_token_id)
    remain = n_img
    for _ in range(n_starts):
        ed = S
        if remain > 0:
            for j in range(st, S):
                if ids[j] == cfg.image_token_id:
                    ed = j
                    break
        if ed == S:
            break
        t, h, w = [int(v) for v in grid]
        remain -= 1
        gt, gh, gw = t, h // cfg.v_merge, w // cfg.v_merge
        tl = ed - st
        if tl > 0:
            for k in range(tl):
                for a in range(3):
                    lp[a].append(cur + k)
            cur += tl
        for ti in range(gt):
            for hi in range(gh):
                for wi in range(gw):
                    lp[0].append(cur + ti); lp[1].append(cur + hi); lp[2].append(cur + wi)
        cur = max(lp[0][-1], lp[1][-1], lp[2][-1])
        st = ed + gt * gh * gw
    if st < S:
        for k in range(S - st):
            for a in range(3):
                lp[a].append(cur + 1 + k)
    pos = np.zeros((3, S), dtype=np.float32)
    for a in range(3):
        n = min(S, len(lp[a]))
        pos[a, :n] = np.asarray(lp[a][:n], dtype=np.float32)
    return pos


class LLM:
    """Qwen2VLModel::Forward (modeling_qwen2_vl.hpp:376-403) + the demo's greedy loop (examples/demo_qwen2_vl.cpp:53-63) composed
    from oracle ops, with the fp16 K/V slabs of CPUKVCache.cpp:10-131 kept as uint16 arrays."""

    def __init__(self, w: Weights, cfg):
        self.w, self.cfg = w, cfg
        self.k = [np.zeros((0, cfg.kv_heads * cfg.head_dim), dtype=np.uint16) for _ in range(cfg.layers)]
        self.v = [np.zeros((0, cfg.kv_heads * cfg.head_dim), dtype=np.uint16) for _ in range(cfg.layers)]
        self.last_pos = -1.0

    def forward(self, ids, pos, image_embeds=None, stages=None):
        w, c = self.w, self.cfg
        H, D, heads, kvh = c.hidden, c.head_dim, c.heads, c.kv_heads
        ids = np.asarray(ids, dtype=np.int32)
        S = ids.size
        emb_name = "model.embed_tokens.weight"
        x = orc.embedding(ids, w.f.raw(emb_name), w.f.dtype(emb_name), H)
        if image_embeds is not None:
            x[ids == c.image_token_id] = image_embeds
        s, co = orc.mrope_table(c.rope_theta, D, pos)
        for i in range(c.layers):
            p = f"model.layers.{i}."
            y = orc.rmsnorm(x, w.v(p + "input_layernorm.weight"), c.rms_eps)
            q = w.lin(y, p + "self_attn.q_proj", heads * D)
            k = w.lin(y, p + "self_attn.k_proj", kvh * D)
            v = w.lin(y, p + "self_attn.v_proj", kvh * D, out_f16=True)
            q = orc.rope_apply(q, S, heads, D, s, co)
            k16 = orc.rope_apply(k, S, kvh, D, s, co, out_f16=True)
            self.k[i] = np.concatenate([self.k[i], k16.reshape(S, kvh * D)])
            self.v[i] = np.concatenate([self.v[i], v.reshape(S, kvh * D)])
            o = orc.attention(q, self.k[i], self.v[i], S, self.k[i].shape[0], heads, kvh, D, True)
            r = w.lin(o, p + "self_attn.o_proj", H, bias=False) + x
            y = orc.rmsnorm(r, w.v(p + "post_attention_layernorm.weight"), c.rms_eps)
            g = w.lin(y, p + "mlp.gate_proj", c.inter, bias=False)
            u = w.lin(y, p + "mlp.up_proj", c.inter, bias=False)
            x = w.lin(orc.silu(g) * u, p + "mlp.down_proj", H, bias=False) + r
            if stages is not None:
                stages[f"layer{i}"] = x.copy()
        x = orc.rmsnorm(x[-1:], w.v("model.norm.weight"), c.rms_eps)
        logits = orc.linear(x, w.f.raw(emb_name), w.f.dtype(emb_name), c.vocab)
        self.last_pos = float(pos.max())
        return logits[0]

    def prefill(self, ids, pix=None, grid=None, stages=None):
        emb = vision_forward(self.w, self.cfg, pix, grid) if pix is not None else None
        pos = rope_index(self.cfg, ids, grid if pix is not None else None)
        return self.forward(ids, pos, emb, stages)

    def decode(self, token):
        pos = np.full((3, 1), self.last_pos + 1.0, dtype=np.float32)
        return self.forward([token], pos)


class CausalLM:
    """QWenForCausalLM::Forward (models/qwen/modeling_qwen.hpp:151-164) and TinyLLaMAModel::Forward (models/tinyllama/
    modeling_tinyllama.hpp:67-74) composed from oracle ops: MultiHeadAttention with HF rotary at positions [T, T+S)
    (models/transformer/modeling_transformer.hpp:136-218, CPURoPE.cpp:100-128,510-513), fp16 KV slabs, FlashAttention2, SiLU MLP."""

    def __init__(self, w: Weights, cfg):
        self.w, self.cfg = w, cfg
        self.k = [np.zeros((0, cfg.kv_heads * cfg.head_dim), dtype=np.uint16) for _ in range(cfg.layers)]
        self.v = [np.zeros((0, cfg.kv_heads * cfg.head_dim), dtype=np.uint16) for _ in range(cfg.layers)]
        self.T = 0

    def forward(self, ids):
        w, c = self.w, self.cfg
        H, D, heads, kvh = c.hidden, c.head_dim, c.heads, c.kv_heads
        ids = np.asarray(ids, dtype=np.int32).ravel()
        S = ids.size
        emb = "model.embed_tokens.weight"
        return self.body_forward(orc.embedding(ids, w.f.raw(emb), w.f.dtype(emb), H))

    PREFIX = ""

    def body_forward(self, x):
        w, c, P = self.w, self.cfg, self.PREFIX
        H, D, heads, kvh = c.hidden, c.head_dim, c.heads, c.kv_heads
        S = x.shape[0]
        emb = P + "model.embed_tokens.weight"
        s, co = orc.rope_table_hf(c.rope_theta, D, self.T + S)
        s, co = np.ascontiguousarray(s[self.T:]), np.ascontiguousarray(co[self.T:])
        for i in range(c.layers):
            p = P + f"model.layers.{i}."
            y = orc.rmsnorm(x, w.v(p + "input_layernorm.weight"), c.rms_eps)
            q = w.lin(y, p + "self_attn.q_proj", heads * D, bias=c.qkv_bias)
            k = w.lin(y, p + "self_attn.k_proj", kvh * D, bias=c.qkv_bias)
            v = w.lin(y, p + "self_attn.v_proj", kvh * D, bias=c.qkv_bias, out_f16=True)
            q = orc.rope_apply(q, S, heads, D, s, co)
            k16 = orc.rope_apply(k, S, kvh, D, s, co, out_f16=True)
            self.k[i] = np.concatenate([self.k[i], k16.reshape(S, kvh * D)])
            self.v[i] = np.concatenate([self.v[i], v.reshape(S, kvh * D)])
            o = orc.attention(q, self.k[i], self.v[i], S, self.k[i].shape[0], heads, kvh, D, True)
            r = w.lin(o, p + "self_attn.o_proj", H, bias=False) + x
            y = orc.rmsnorm(r, w.v(p + "post_attention_layernorm.weight"), c.rms_eps)
            g = w.lin(y, p + "mlp.gate_proj", c.inter, bias=False)
            u = w.lin(y, p + "mlp.up_proj", c.inter, bias=False)
            x = w.lin(orc.silu(g) * u, p + "mlp.down_proj", H, bias=False) + r
        self.T += S
        x = orc.rmsnorm(x[-1:], w.v(P + "model.norm.weight"), c.rms_eps)
        if c.tie_embedding:
            return orc.linear(x, w.f.raw(emb), w.f.dtype(emb), c.vocab)[0]
        return w.lin(x, P + "lm_head", c.vocab, bias=False)[0]


def vit_forward(w: Weights, cfg, img_hcw):
    """ViTModel::Forward (models/vit/modeling_vit.hpp:21-104) on one image [H][C][W]: Conv2D patch embedding (+bias), cls token,
    position embeddings, pre-LN blocks with non-causal FlashAttention2 on fp32 K/V and a GELU MLP, final LayerNorm of the cls row, head."""
    H, heads, D, p = cfg.hidden, cfg.heads, cfg.head_dim, cfg.patch
    e = "vit.embeddings."
    pe = orc.conv2d_patch(img_hcw, cfg.img, 3, cfg.img, w.v(e + "patch_embeddings.projection.weight"), H, p, w.v(e + "patch_embeddings.projection.bias"))
    g = cfg.img // p
    tok = np.ascontiguousarray(pe.reshape(g, H, g).transpose(0, 2, 1).reshape(g * g, H))     # [oh][OC][ow] -> rows (oh, ow)
    x = np.concatenate([w.v(e + "cls_token").reshape(1, H), tok])
    x = w.v(e + "position_embeddings").reshape(-1, H) + x
    N = x.shape[0]
    for i in range(cfg.blocks):
        b = f"vit.encoder.layer.{i}."
        y = orc.layernorm(x, w.v(b + "layernorm_before.weight"), w.v(b + "layernorm_before.bias"), 1e-5)
        q = w.lin(y, b + "attention.attention.query", H)
        k = w.lin(y, b + "attention.attention.key", H)
        v = w.lin(y, b + "attention.attention.value", H)
        o = orc.attention(q, k, v, N, N, heads, heads, D, False)
        r = w.lin(o, b + "attention.output.dense", H) + x
        y = orc.layernorm(r, w.v(b + "layernorm_after.weight"), w.v(b + "layernorm_after.bias"), 1e-5)
        a = orc.gelu(w.lin(y, b + "intermediate.dense", cfg.ffn))
        x = w.lin(a, b + "output.dense", H) + r
    y = orc.layernorm(x[:1], w.v("vit.layernorm.weight"), w.v("vit.layernorm.bias"), 1e-6)
    return w.lin(y, "classifier", cfg.classes, bias=False)[0]


def llava_vision(w: Weights, cfg, img_hcw):
    """LLaVAVisionModel::Forward (models/llava/modeling_llava.hpp:39-98): CLIP embeddings (Conv2D without bias, class_embedding row, position
    Embedding over range ids), pre_layrnorm, ViTBlocks with QuickGELU, the cls row clipped away, linear_1 -> GELU -> linear_2."""
    V, heads, D, p = cfg.v_hidden, cfg.v_heads, cfg.v_head_dim, cfg.patch
    base = "vision_tower.vision_model."
    e = base + "embeddings."
    g = cfg.img // p
    pe = orc.conv2d_patch(img_hcw, cfg.img, 3, cfg.img, w.v(e + "patch_embedding.weight"), V, p)
    tok = np.ascontiguousarray(pe.reshape(g, V, g).transpose(0, 2, 1).reshape(g * g, V))
    x = np.concatenate([w.v(e + "class_embedding").reshape(1, V), tok])
    x = w.v(e + "position_embedding.weight").reshape(-1, V) + x
    x = orc.layernorm(x, w.v(base + "pre_layrnorm.weight"), w.v(base + "pre_layrnorm.bias"), 1e-6)
    N = x.shape[0]
    for i in range(cfg.v_blocks):
        b = base + f"encoder.layers.{i}."
        y = orc.layernorm(x, w.v(b + "layer_norm1.weight"), w.v(b + "layer_norm1.bias"), 1e-5)
        q, k, v = (w.lin(y, b + f"self_attn.{nm}", V) for nm in ("q_proj", "k_proj", "v_proj"))
        o = orc.attention(q, k, v, N, N, heads, heads, D, False)
        r = w.lin(o, b + "self_attn.out_proj", V) + x
        y = orc.layernorm(r, w.v(b + "layer_norm2.weight"), w.v(b + "layer_norm2.bias"), 1e-5)
        x = w.lin(orc.quickgelu(w.lin(y, b + "mlp.fc1", cfg.v_ffn)), b + "mlp.fc2", V) + r
    y = orc.gelu(w.lin(np.ascontiguousarray(x[1:]), "multi_modal_projector.linear_1", cfg.v_ffn))
    return w.lin(y, "multi_modal_projector.linear_2", cfg.v_ffn)


class LLaVA(CausalLM):
    """LLaVAModel::Forward (modeling_llava.hpp:126-136): text embedding, the <image> row expanded into the projected visual rows
    (index_put with accumulate, CPUIndexPutFunc.hpp:84-110), LLaMA body, last row."""

    def __init__(self, w: Weights, cfg):
        super().__init__(w, cfg.body())
        self.lcfg = cfg

    PREFIX = "language_model."

    def forward(self, ids, img=None):
        ids = np.asarray(ids, dtype=np.int32).ravel()
        w, c = self.w, self.lcfg
        emb = self.PREFIX + "model.embed_tokens.weight"
        x = orc.embedding(ids, w.f.raw(emb), w.f.dtype(emb), c.hidden)
        if img is not None:
            vis = llava_vision(w, c, img)
            at = int(np.nonzero(ids == c.image_token_id)[0][0])
            x = np.concatenate([x[:at], vis, x[at + 1:]])
        return self.body_forward(x)


def moe_block(w: Weights, cfg, x):
    """The reference's sparse-MoE feed-forward block, MiniCPMMoE::Forward (models/minicpm_moe/modeling_minicpm_moe.hpp:52-105), on token rows x [S, hidden]:
    scores = softmax(gate(x)) (:58-59); top-k pairs per token in the function's pair order (:60, CPUTopkFunc.hpp:48-70); weights renormalised by their sequential sum
    (:64, CPUSumFunc.hpp: `sum += x[d]` from 0, then CPUBinaryFunc's division); argsort of the flattened expert ids + bincount route the (token, slot) pairs to experts
    (:67-69: std::sort on the id alone -- the order inside an expert's run is unspecified, and immaterial: a token meets an expert at most once, rows are independent);
    per expert, ascending: gather its tokens (:83), gate / up / down MLP (:84), scale each row by its weight (:86), scatter_add into the zero-initialised output (:87)."""
    S, H = x.shape
    E, k, b = cfg.experts, cfg.per_tok, cfg.base
    scores = orc.softmax(w.lin(x, b + "gate", E, bias=False))
    tv, ti = orc.topk_rows(scores, k)
    den = np.zeros(S, dtype=np.float32)
    for d in range(k):
        den = (den + tv[:, d]).astype(np.float32)
    wn = (tv / den[:, None]).astype(np.float32)
    flat_e = ti.reshape(-1).astype(np.int64)                  # pair j = token j // k, slot j % k
    out = np.zeros((S, H), dtype=np.float32)
    for e in range(E):
        pairs = np.nonzero(flat_e == e)[0]
        if pairs.size == 0:
            continue
        tok = pairs // k
        p = f"{b}experts.{e}."
        xe = x[tok]
        g = orc.silu(w.lin(xe, p + "w1", cfg.inter, bias=False))
        u = w.lin(xe, p + "w3", cfg.inter, bias=False)
        y = w.lin((g * u).astype(np.float32), p + "w2", H, bias=False)
        y = (y * wn.reshape(-1)[pairs][:, None]).astype(np.float32)
        out = orc.scatter_add_rows(out, y, tok.astype(np.float32))
    return out
