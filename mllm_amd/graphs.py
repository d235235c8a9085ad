"""Host-side graphs of the other BASELINE configs, composed from the C-ABI launchers exactly the way the reference's model
headers compose its Ops: text-only causal LMs (QWenForCausalLM, models/qwen/modeling_qwen.hpp:131-179; TinyLLaMAModel,
models/tinyllama/modeling_tinyllama.hpp:44-84 -- both through MultiHeadAttention, models/transformer/modeling_transformer.hpp:35-219)
and ViTModel (models/vit/modeling_vit.hpp:14-111).  These are the callers of SURVEY §8(a) row A21 for configs 1-3; the Qwen2-VL
graph (config 4) is the resident C++ engine (csrc/engine.hip).  One Python call per reference Op: this layer is for parity and
for op-level timing, not the fused decode path.  torch is device memory and the current stream only; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import lib as L
from . import mllmfile as mf
from . import ops
from .lib import check, i64, vp


class DeviceWeights:
    """Tensors of a .mllm uploaded once, in the file's own storage dtypes (ParamLoader::load, mllm/ParamLoader.cpp:157-286)."""

    def __init__(self, path: str):
        ops.require_gpu()
        self.f = mf.MllmFile(path)
        self._raw, self._f32, self._q40 = {}, {}, {}

    def dtype(self, name):
        return self.f.dtype(name)

    def raw(self, name):
        if name not in self._raw:
            self._raw[name] = torch.from_numpy(np.ascontiguousarray(self.f.raw(name)).view(np.uint8).copy()).cuda()
        return self._raw[name]

    def f32(self, name):
        if name not in self._f32:
            self._f32[name] = torch.from_numpy(np.ascontiguousarray(self.f.f32(name)).copy()).cuda()
        return self._f32[name]

    def q40_planes(self, name, n_blocks):
        if name not in self._q40:
            self._q40[name] = ops.repack_q40(self.raw(name), n_blocks)
        return self._q40[name]


def linear(w: DeviceWeights, x, name, N, bias=True, residual=None, out_f16=False):
    """CPULinear::execute (backends/cpu/op/CPULinear.cpp:98-234) on the tensor `name`.weight in its storage dtype."""
    b = w.f32(name + ".bias") if bias else None
    wn = name + ".weight"
    dt = w.dtype(wn)
    if dt == mf.Q4_K:
        return ops.linear_q4k(w.raw(wn), x, N, bias=b, residual=residual, out_f16=out_f16)
    if dt == mf.F32:
        y = ops.linear_f32(w.f32(wn), x, bias=b)
        if residual is not None:
            y = ops.add(y, residual)
        if out_f16:   # the fp16 store branch of mat_mul (Matmul.cpp:262-268)
            h = torch.empty(y.shape, dtype=torch.float16, device="cuda")
            check(L.load().mllm_hip_store_f16(vp(y), i64(N), vp(h), i64(N), C.c_int(y.shape[0]), C.c_int(N), ops._stream()), "store_f16")
            return h
        return y
    raise L.MllmHipError(f"linear: storage dtype {dt} of {wn} is not on the hot path")


class CausalLM:
    """Decoder-only LM with HF rotary, fp16 KV cache and FlashAttention2 (attn_implementation default,
    models/transformer/configuration_transformer.hpp:40).  `forward(ids)` = one Module::operator() call: a prefill when the cache is
    empty, a decode step after it; returns the last row's logits on the host."""

    def __init__(self, cfg, path: str):
        self.cfg, self.w = cfg, DeviceWeights(path)
        c = cfg
        self.k = [torch.zeros((c.cache_limit, c.kv_heads * c.head_dim), dtype=torch.float16, device="cuda") for _ in range(c.layers)]
        self.v = [torch.zeros((c.cache_limit, c.kv_heads * c.head_dim), dtype=torch.float16, device="cuda") for _ in range(c.layers)]
        self.T = 0
        s, co = L.rope_table_hf(c.rope_theta, c.head_dim, c.cache_limit)      # CPURoPE's static table (CPURoPE.cpp:100-128)
        self.sin, self.cos = torch.from_numpy(s).cuda(), torch.from_numpy(co).cuda()

    def clear_kvcache(self):
        self.T = 0

    PREFIX = ""

    def _embed(self, ids):
        c, w, name = self.cfg, self.w, self.PREFIX + "model.embed_tokens.weight"
        if w.dtype(name) == mf.Q4_0:
            qs, d = w.q40_planes(name, c.vocab * c.hidden // 32)
            idf = torch.from_numpy(np.asarray(ids, dtype=np.float32)).cuda()
            out = torch.empty((len(ids), c.hidden), dtype=torch.float32, device="cuda")
            check(L.load().mllm_hip_embedding_q40(vp(idf), vp(qs), vp(d), vp(out), C.c_int(len(ids)), C.c_int(c.hidden), C.c_int(c.vocab), ops._stream()),
                  "embedding_q40")
            return out
        # fp32 table: CPUEmbedding copies rows (CPUEmbedding.cpp:46-60) -- a gather, no arithmetic
        return w.f32(name).view(c.vocab, c.hidden)[torch.from_numpy(np.asarray(ids, dtype=np.int64)).cuda()].contiguous()

    def _head(self, x):
        c, w = self.cfg, self.w
        if not c.tie_embedding:
            return linear(w, x, self.PREFIX + "lm_head", c.vocab, bias=False)
        name = self.PREFIX + "model.embed_tokens.weight"      # Tensor::mm(x, embed^T) (modeling_qwen.hpp:158-159) = the same vec_dot per row
        if w.dtype(name) == mf.Q4_0:
            qs, d = w.q40_planes(name, c.vocab * c.hidden // 32)
            xqs, xd = ops.quantize_q80(x)
            y = torch.empty((x.shape[0], c.vocab), dtype=torch.float32, device="cuda")
            check(L.load().mllm_hip_linear_q40_q80(vp(qs), vp(d), None, vp(xqs), vp(xd), vp(y), i64(c.vocab), C.c_int(x.shape[0]), C.c_int(c.vocab),
                                                   C.c_int(c.hidden), ops._stream()), "linear_q40_q80")
            return y
        return ops.linear_f32(w.f32(name), x)

    def forward(self, ids) -> np.ndarray:
        return self._body(self._embed([int(v) for v in np.asarray(ids).ravel()]))

    def _body(self, x) -> np.ndarray:
        c, w, P = self.cfg, self.w, self.PREFIX
        S, T, D, heads, kvh = x.shape[0], self.T, c.head_dim, c.heads, c.kv_heads
        if T + S > c.cache_limit:
            raise L.MllmHipError("KV cache overflow")       # CPUKVCache.cpp:121-126 exits the process; the library reports an error
        sin, cos = self.sin[T:T + S], self.cos[T:T + S]
        for i in range(c.layers):
            p = P + f"model.layers.{i}."
            y = ops.rmsnorm(x, w.f32(p + "input_layernorm.weight"), c.rms_eps)
            q = linear(w, y, p + "self_attn.q_proj", heads * D, bias=c.qkv_bias)
            k = linear(w, y, p + "self_attn.k_proj", kvh * D, bias=c.qkv_bias)
            v = linear(w, y, p + "self_attn.v_proj", kvh * D, bias=c.qkv_bias, out_f16=True)
            q = ops.rope_apply(q, S, heads, D, sin, cos)
            self.k[i][T:T + S] = ops.rope_apply(k, S, kvh, D, sin, cos, out_f16=True)       # KVCache slab rows (CPUKVCache.cpp:253-275)
            self.v[i][T:T + S] = v
            o = ops.flash_attention2(q, self.k[i][:T + S], self.v[i][:T + S], S, T + S, heads, kvh, D, True)
            r = linear(w, o, p + "self_attn.o_proj", c.hidden, bias=False, residual=x)
            y = ops.rmsnorm(r, w.f32(p + "post_attention_layernorm.weight"), c.rms_eps)
            g = linear(w, y, p + "mlp.gate_proj", c.inter, bias=False)
            u = linear(w, y, p + "mlp.up_proj", c.inter, bias=False)
            x = linear(w, ops.mul(ops.silu(g), u), p + "mlp.down_proj", c.hidden, bias=False, residual=r)
        self.T = T + S
        x = ops.rmsnorm(x[S - 1:S].contiguous(), w.f32(P + "model.norm.weight"), c.rms_eps)
        return self._head(x)[0].cpu().numpy()

    def greedy(self, ids, steps):
        """The demos' loop: forward, host argmax (first maximum), feed the id back.  Returns (ids, [logits per step])."""
        toks, logits = [], []
        cur = ids
        for _ in range(steps):
            lg = self.forward(cur)
            logits.append(lg)
            toks.append(int(np.argmax(lg)))
            cur = [toks[-1]]
        return toks, logits


class ViT:
    """ViTModel::Forward (models/vit/modeling_vit.hpp:91-111): Conv2D patch embedding, cls token + position embeddings, pre-LN blocks
    (non-causal FlashAttention2 on fp32 K/V, GELU MLP), final LayerNorm of the cls row, classifier.  One image per call, like
    examples/demo_vit.cpp:33-38; `forward_batch` runs a list of images (the unit of the multi-GPU image shard)."""

    def __init__(self, cfg, path: str):
        self.cfg, self.w = cfg, DeviceWeights(path)

    def forward(self, img_hcw) -> torch.Tensor:
        c, w = self.cfg, self.w
        H, heads, D = c.hidden, c.heads, c.head_dim
        e = "vit.embeddings."
        g = c.img // c.patch
        pe = ops.conv2d_patch(img_hcw, c.img, 3, c.img, w.f32(e + "patch_embeddings.projection.weight"), H, c.patch,
                              w.f32(e + "patch_embeddings.projection.bias"))          # [oh][OC][ow]
        tok = pe.permute(0, 2, 1).reshape(g * g, H)                                    # transpose + flatten (metadata in the reference)
        x = torch.cat([w.f32(e + "cls_token").view(1, H), tok]).contiguous()          # Tensor::cat over SEQUENCE
        x = ops.add(w.f32(e + "position_embeddings").view(-1, H), x)
        N = x.shape[0]
        for i in range(c.blocks):
            b = f"vit.encoder.layer.{i}."
            y = ops.layernorm(x, w.f32(b + "layernorm_before.weight"), w.f32(b + "layernorm_before.bias"), 1e-5)
            q = linear(w, y, b + "attention.attention.query", H)
            k = linear(w, y, b + "attention.attention.key", H)
            v = linear(w, y, b + "attention.attention.value", H)
            o = ops.flash_attention2(q, k, v, N, N, heads, heads, D, False)
            r = linear(w, o, b + "attention.output.dense", H, residual=x)
            y = ops.layernorm(r, w.f32(b + "layernorm_after.weight"), w.f32(b + "layernorm_after.bias"), 1e-5)
            a = ops.gelu(linear(w, y, b + "intermediate.dense", c.ffn))
            x = linear(w, a, b + "output.dense", H, residual=r)
        y = ops.layernorm(x[:1].contiguous(), w.f32("vit.layernorm.weight"), w.f32("vit.layernorm.bias"), 1e-6)
        return linear(w, y, "classifier", c.classes, bias=False)[0]

    def forward_batch(self, imgs) -> torch.Tensor:
        return torch.stack([self.forward(im) for im in imgs])


class LLaVA(CausalLM):
    """LLaVAModel::Forward (models/llava/modeling_llava.hpp:101-137): text embedding; CLIP tower (LLaVAVisionModel, :62-98: Conv2D patch
    embedding without bias, class_embedding row, position Embedding, pre_layrnorm, ViTBlocks with QuickGELU, cls row clipped) ->
    multi_modal_projector (Linear, GELU, Linear); the <image> row replaced by the projected rows (index_put with accumulate,
    CPUIndexPutFunc.hpp:84-110); LLaMA body (LLaMABodyModel, :13-37)."""

    PREFIX = "language_model."

    def __init__(self, cfg, path: str):
        super().__init__(cfg.body(), path)
        self.lcfg = cfg

    def vision(self, img_hcw) -> torch.Tensor:
        c, w = self.lcfg, self.w
        V, heads, D = c.v_hidden, c.v_heads, c.v_head_dim
        base = "vision_tower.vision_model."
        e = base + "embeddings."
        g = c.img // c.patch
        pe = ops.conv2d_patch(img_hcw, c.img, 3, c.img, w.f32(e + "patch_embedding.weight"), V, c.patch)
        tok = pe.permute(0, 2, 1).reshape(g * g, V)
        x = torch.cat([w.f32(e + "class_embedding").view(1, V), tok]).contiguous()
        x = ops.add(w.f32(e + "position_embedding.weight").view(-1, V), x)       # Embedding over Tensor::range(0, N): rows 0..N-1 in order
        x = ops.layernorm(x, w.f32(base + "pre_layrnorm.weight"), w.f32(base + "pre_layrnorm.bias"), 1e-6)
        N = x.shape[0]
        for i in range(c.v_blocks):
            b = base + f"encoder.layers.{i}."
            y = ops.layernorm(x, w.f32(b + "layer_norm1.weight"), w.f32(b + "layer_norm1.bias"), 1e-5)
            q, k, v = (linear(w, y, b + f"self_attn.{nm}", V) for nm in ("q_proj", "k_proj", "v_proj"))
            o = ops.flash_attention2(q, k, v, N, N, heads, heads, D, False)
            r = linear(w, o, b + "self_attn.out_proj", V, residual=x)
            y = ops.layernorm(r, w.f32(b + "layer_norm2.weight"), w.f32(b + "layer_norm2.bias"), 1e-5)
            x = linear(w, ops.quickgelu(linear(w, y, b + "mlp.fc1", c.v_ffn)), b + "mlp.fc2", V, residual=r)
        y = ops.gelu(linear(w, x[1:].contiguous(), "multi_modal_projector.linear_1", c.v_ffn))
        return linear(w, y, "multi_modal_projector.linear_2", c.v_ffn)

    def forward(self, ids, img=None, vis=None) -> np.ndarray:
        """`vis`: visual rows already computed (e.g. gathered from the ranks of the image shard); otherwise `img` runs the tower here."""
        ids = [int(v) for v in np.asarray(ids).ravel()]
        x = self._embed(ids)
        if img is not None or vis is not None:
            if vis is None:
                vis = self.vision(img)
            at = ids.index(self.lcfg.image_token_id)
            x = torch.cat([x[:at], vis, x[at + 1:]]).contiguous()
        return self._body(x)
