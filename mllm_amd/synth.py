"""Seeded synthetic weights / inputs for the hot-path configs (SURVEY §8d "Synthetic inputs").

No model weights ship with the reference and there is no network, so every measurement and parity case runs on
synthetic tensors, each drawn from ``numpy.random.default_rng(20251031 + crc32(name))``: norm weights = 1, biases =
``0.01*N(0,1)`` fp32, fp32-stored weights ``N(0, 0.02^2)``.  Weights the file stores as Q4_K / Q4_0 are drawn
DIRECTLY IN THE QUANTISED DOMAIN (``q4k_blocks`` / ``q40_blocks``: seeded block scales, 6-bit sub-block scales / mins
and nibbles chosen so that the dequantised weights have zero mean and a standard deviation of about 0.02) -- there is
no fp32 original and no quantiser anywhere in this repository; what both the reference (for the goldens) and the HIP
path consume is the same file, byte for byte.  Tensor names follow the reference's name configs
(mllm/models/qwen/configuration_qwen.hpp:28-47, mllm/models/qwen2_vl/configuration_qwen2_vl.hpp:20-32).
The per-name storage dtype policy restates tools/quantizer/QuantWriter.cpp:10-35,123-157 for a Q4_K target; the block
layouts are mllm/DataType.hpp:75-78 (block_q4_0) and :93-98 (block_q4_K; 6-bit packing as read by
dequantize_row_q4_K, third_party/ggml/QuantizeQ4.cpp:295-333).
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Iterator, List, Tuple

import numpy as np

from . import mllmfile as mf

SEED0 = 20251031


@dataclass
class Qwen2VLConfig:
    """Shape subset of Qwen2VLConfig/QWenConfig (configuration_qwen.hpp:152-166 "1.5b", configuration_qwen2_vl.hpp:34-56)."""
    hidden: int = 1536
    inter: int = 8960
    layers: int = 28
    heads: int = 12
    kv_heads: int = 2
    vocab: int = 151936
    rms_eps: float = 1e-6
    rope_theta: float = 1000000.0
    max_pos: int = 32768
    mrope_section: Tuple[int, int, int] = (16, 24, 24)
    cache_limit: int = 800
    tie_embedding: bool = True
    # vision tower (modeling_qwen2_vl.hpp:371: 16 heads, mlp = 4*dim, QuickGELU, patch 14, 32 blocks, merge 2)
    v_dim: int = 1280
    v_heads: int = 16
    v_blocks: int = 32
    v_patch: int = 14
    v_merge: int = 2
    image_token_id: int = 151655
    vision_start_token_id: int = 151652
    vision_end_token_id: int = 151653
    video_token_id: int = 151656

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def v_head_dim(self) -> int:
        return self.v_dim // self.v_heads

    @property
    def v_mlp(self) -> int:
        return self.v_dim * 4

    @property
    def patch_elems(self) -> int:
        return 3 * 2 * self.v_patch * self.v_patch


def qwen2vl_2b() -> Qwen2VLConfig:
    return Qwen2VLConfig()


def qwen2vl_tiny() -> Qwen2VLConfig:
    """Smallest shape the reference's Qwen2VLModel accepts with Q4_K weights: head_dim must stay 128 (mrope_section sums
    to 64), every Linear K a multiple of 256, and the vision tower is hard-wired to 16 heads x 32 blocks."""
    return Qwen2VLConfig(hidden=256, inter=512, layers=2, heads=2, kv_heads=1, vocab=2048, cache_limit=96,
                         v_dim=256, image_token_id=2040, vision_start_token_id=2041, vision_end_token_id=2042,
                         video_token_id=2043)


FP32_LAYERS = ["norm", "rope", "bias", "rotary_emb", "_GN", "class_embedding", "embeddings", "logit_scale",
               "modality_preprocessors", "modality_heads", "modality_postprocessors", "pre_transformer_layer",
               "pos_embed.inv_freq", "ln_q", "patch_embed.proj"]
Q40_LAYERS = ["embed_tokens", "word_embeddings"]


def storage_dtype(name: str, target: int = mf.Q4_K) -> int:
    """QuantWriter::getQuantizationTypeFor (tools/quantizer/QuantWriter.cpp:123-157) for K-quant / Q4_0 targets."""
    if target == mf.F32:
        return mf.F32
    if any(s in name for s in Q40_LAYERS):
        return mf.Q4_0
    if any(s in name for s in FP32_LAYERS):
        return mf.F32
    return target


def tensor_f32(name: str, shape: Tuple[int, ...], kind: str) -> np.ndarray:
    rng = np.random.default_rng(SEED0 + zlib.crc32(name.encode()))
    n = int(np.prod(shape))
    if kind == "norm":
        return np.ones(n, dtype=np.float32)
    x = rng.standard_normal(n, dtype=np.float32)
    x *= np.float32(0.01 if kind == "bias" else 0.02)
    return x


def _bell_nibbles(rng: np.random.Generator, shape) -> np.ndarray:
    """Nibbles 0..15 with a triangular (bell-like) distribution, mean 7.75, standard deviation 3.3: the rounded mean of the two halves of a random byte."""
    b = rng.integers(0, 256, size=shape, dtype=np.uint8)
    return ((b & 15) + (b >> 4) + 1) >> 1


Q_MEAN = 7.75      # mean of _bell_nibbles
Q_STD = 3.3


def q4k_blocks(rng: np.random.Generator, nblk: int, std: float = 0.02, full_range: bool = False) -> np.ndarray:
    """`nblk` block_q4_K super-blocks (144 B / 256 weights) drawn in the quantised domain, uint8 [nblk][144].
    w = d * sc_j * q - dmin * m_j per 32-weight sub-block j: sc_j in 20..63, m_j ~ 7.75 * sc_j / 8 (+-2) with dmin ~ 8 d, so a sub-block's weights centre on
    zero; d is set for a standard deviation of `std` at the mean scale and jitters by +-25 % per super-block.
    full_range (op tests): every field over its whole range instead -- scales and mins 0..63, uniform nibbles, d and dmin from 0 (every 16th block) to 4x the nominal."""
    out = np.empty((nblk, 144), dtype=np.uint8)
    if full_range:
        sc = rng.integers(0, 64, size=(nblk, 8), dtype=np.uint8)
        m = rng.integers(0, 64, size=(nblk, 8), dtype=np.uint8)
        d = (np.float32(std / (Q_STD * 41.5)) * rng.uniform(0.0, 4.0, size=nblk).astype(np.float32)).astype(np.float16)
        dmin = (np.float32(8 * std / (Q_STD * 41.5)) * rng.uniform(0.0, 4.0, size=nblk).astype(np.float32)).astype(np.float16)
        d[::16] = 0
        dmin[8::16] = 0
    else:
        sc = rng.integers(20, 64, size=(nblk, 8), dtype=np.uint8)
        m = np.clip(np.rint(sc.astype(np.float32) * np.float32(Q_MEAN / 8)).astype(np.int32) + rng.integers(-2, 3, size=(nblk, 8)), 0, 63).astype(np.uint8)
        d = (np.float32(std / (Q_STD * 41.5)) * rng.uniform(0.8, 1.25, size=nblk).astype(np.float32)).astype(np.float16)
        dmin = (d.astype(np.float32) * np.float32(8) * rng.uniform(0.95, 1.05, size=nblk).astype(np.float32)).astype(np.float16)
    out[:, 0:2] = d.view(np.uint8).reshape(nblk, 2)
    out[:, 2:4] = dmin.view(np.uint8).reshape(nblk, 2)
    # 12 bytes of 6-bit fields: bytes 0..3 = sc[0..3] | (sc[4..7] >> 4) << 6, bytes 4..7 = m[0..3] | (m[4..7] >> 4) << 6, bytes 8..11 = sc[4..7] & 15 | (m[4..7] & 15) << 4
    out[:, 4:8] = sc[:, :4] | ((sc[:, 4:] >> 4) << 6)
    out[:, 8:12] = m[:, :4] | ((m[:, 4:] >> 4) << 6)
    out[:, 12:16] = (sc[:, 4:] & 15) | ((m[:, 4:] & 15) << 4)
    out[:, 16:] = rng.integers(0, 256, size=(nblk, 128), dtype=np.uint8) if full_range else _bell_nibbles(rng, (nblk, 128)) | (_bell_nibbles(rng, (nblk, 128)) << 4)
    return out


def q40_blocks(rng: np.random.Generator, nblk: int, std: float = 0.02, full_range: bool = False) -> np.ndarray:
    """`nblk` block_q4_0 blocks (fp16 d + 16 nibble bytes; w = (q - 8) d), uint8 [nblk][18], standard deviation about `std`.
    full_range (op tests): uniform nibbles, d of either sign (a real quantiser stores max / -8) from 0 (every 16th block) to 4x the nominal."""
    out = np.empty((nblk, 18), dtype=np.uint8)
    if full_range:
        d = (np.float32(std / Q_STD) * rng.uniform(-4.0, 4.0, size=nblk).astype(np.float32)).astype(np.float16)
        d[::16] = 0
        nib = rng.integers(0, 256, size=(nblk, 16), dtype=np.uint8)
    else:
        d = (np.float32(std / Q_STD) * rng.uniform(0.8, 1.25, size=nblk).astype(np.float32)).astype(np.float16)
        nib = _bell_nibbles(rng, (nblk, 16)) | (_bell_nibbles(rng, (nblk, 16)) << 4)
    out[:, 0:2] = d.view(np.uint8).reshape(nblk, 2)
    out[:, 2:] = nib
    return out


def quantized_blocks(dtype: int, rng: np.random.Generator, n_elem: int, std: float = 0.02, full_range: bool = False) -> np.ndarray:
    """Flat uint8 bytes of `n_elem` weights stored as `dtype` (Q4_K or Q4_0), drawn in the quantised domain."""
    be, _ = mf.BLOCK[dtype]
    if n_elem % be:
        raise ValueError(f"{n_elem} elements do not fill {mf.DTYPE_NAME[dtype]} blocks of {be}")
    if dtype == mf.Q4_K:
        return q4k_blocks(rng, n_elem // be, std, full_range).ravel()
    if dtype == mf.Q4_0:
        return q40_blocks(rng, n_elem // be, std, full_range).ravel()
    raise ValueError(f"no quantised-domain synthesiser for dtype {dtype}")


def tensor_stored(name: str, shape: Tuple[int, ...], kind: str, target: int = mf.Q4_K) -> Tuple[int, np.ndarray]:
    """(storage dtype, bytes / fp32 values) of one synthetic tensor as a `*-q4_k.mllm` (or fp32) file holds it."""
    dt = storage_dtype(name, target)
    if dt == mf.F32:
        return dt, tensor_f32(name, shape, kind)
    rng = np.random.default_rng(SEED0 + zlib.crc32(name.encode()))
    return dt, quantized_blocks(dt, rng, int(np.prod(shape)))


def qwen2vl_tensors(c: Qwen2VLConfig, vision: bool = True) -> Iterator[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) in file order. Linear weights are [out, in] row-major (CPULinear.cpp:50-51)."""
    H, I, D = c.hidden, c.inter, c.head_dim
    yield "model.embed_tokens.weight", (c.vocab, H), "w"
    for i in range(c.layers):
        p = f"model.layers.{i}."
        yield p + "input_layernorm.weight", (H,), "norm"
        yield p + "self_attn.q_proj.weight", (c.heads * D, H), "w"
        yield p + "self_attn.q_proj.bias", (c.heads * D,), "bias"
        yield p + "self_attn.k_proj.weight", (c.kv_heads * D, H), "w"
        yield p + "self_attn.k_proj.bias", (c.kv_heads * D,), "bias"
        yield p + "self_attn.v_proj.weight", (c.kv_heads * D, H), "w"
        yield p + "self_attn.v_proj.bias", (c.kv_heads * D,), "bias"
        yield p + "self_attn.o_proj.weight", (H, c.heads * D), "w"
        yield p + "post_attention_layernorm.weight", (H,), "norm"
        yield p + "mlp.gate_proj.weight", (I, H), "w"
        yield p + "mlp.up_proj.weight", (I, H), "w"
        yield p + "mlp.down_proj.weight", (H, I), "w"
    yield "model.norm.weight", (H,), "norm"
    if not c.tie_embedding:
        yield "lm_head.weight", (c.vocab, H), "w"
    if not vision:
        return
    V, M = c.v_dim, c.v_mlp
    yield "visual.patch_embed.proj.weight", (V, 3, 2, c.v_patch, c.v_patch), "w"
    for i in range(c.v_blocks):
        p = f"visual.blocks.{i}."
        yield p + "norm1.weight", (V,), "norm"
        yield p + "norm1.bias", (V,), "bias"
        yield p + "attn.qkv.weight", (3 * V, V), "w"
        yield p + "attn.qkv.bias", (3 * V,), "bias"
        yield p + "attn.proj.weight", (V, V), "w"
        yield p + "attn.proj.bias", (V,), "bias"
        yield p + "norm2.weight", (V,), "norm"
        yield p + "norm2.bias", (V,), "bias"
        yield p + "mlp.fc1.weight", (M, V), "w"
        yield p + "mlp.fc1.bias", (M,), "bias"
        yield p + "mlp.fc2.weight", (V, M), "w"
        yield p + "mlp.fc2.bias", (V,), "bias"
    m2 = V * c.v_merge * c.v_merge
    yield "visual.merger.ln_q.weight", (V,), "norm"
    yield "visual.merger.ln_q.bias", (V,), "bias"
    yield "visual.merger.mlp.0.weight", (m2, m2), "w"
    yield "visual.merger.mlp.0.bias", (m2,), "bias"
    yield "visual.merger.mlp.2.weight", (H, m2), "w"
    yield "visual.merger.mlp.2.bias", (H,), "bias"


# ---------------------------------------------------------------------------------------------------------------
# synthetic prompt / image
# ---------------------------------------------------------------------------------------------------------------

def qwen2vl_inputs(c: Qwen2VLConfig, grid_hw: Tuple[int, int] = (32, 32), n_text: int = 24):
    """pixel_values fp32 [N, 3*2*14*14] ~ N(0,1) seed 7; grid_thw [1,h,w]; ids = [vision_start] + N/4*[image_pad] +
    [vision_end] + n_text ids uniform in [0, min(vocab, 151643)-1) seed 11 (never a special id)."""
    gh, gw = grid_hw
    n_patch = gh * gw
    pix = np.random.default_rng(7).standard_normal((n_patch, c.patch_elems), dtype=np.float32)
    n_img_tok = n_patch // (c.v_merge * c.v_merge)
    hi = min(c.vocab, 151643)
    specials = {c.image_token_id, c.vision_start_token_id, c.vision_end_token_id, c.video_token_id}
    hi = min([hi] + [s for s in specials])
    text = np.random.default_rng(11).integers(0, hi, size=n_text)
    ids = np.concatenate([[c.vision_start_token_id], np.full(n_img_tok, c.image_token_id), [c.vision_end_token_id], text])
    return pix, np.array([1, gh, gw], dtype=np.int32), ids.astype(np.int32)


# ---------------------------------------------------------------------------------------------------------------
# the other BASELINE configs: text-only causal LMs (demo_qwen, demo_tinyllama) and ViT-B/16 (demo_vit)
# ---------------------------------------------------------------------------------------------------------------

@dataclass
class CausalLMConfig:
    """Shape subset of QWenConfig (configuration_qwen.hpp:78-245) / TinyLLaMAConfig (configuration_tinyllama.hpp:10-50): decoder
    blocks of MultiHeadAttention (HF rotary, fp16 KV cache, FlashAttention2) + gate/up/down SiLU MLP + RMSNorm."""
    family: str = "qwen"            # "qwen": q/k/v bias, tied lm_head allowed; "tinyllama": no bias, Linear lm_head
    hidden: int = 1024
    inter: int = 2816
    layers: int = 24
    heads: int = 16
    kv_heads: int = 16
    vocab: int = 151936
    rms_eps: float = 1e-6
    rope_theta: float = 1000000.0
    cache_limit: int = 400
    tie_embedding: bool = True
    target: int = mf.Q4_K           # storage dtype target of the .mllm (mf.F32 for the fp32 TinyLlama config)

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def qkv_bias(self) -> bool:
        return self.family == "qwen"


def qwen15_05b() -> CausalLMConfig:
    return CausalLMConfig()


def qwen15_tiny() -> CausalLMConfig:
    return CausalLMConfig(hidden=256, inter=768, layers=2, heads=4, kv_heads=4, vocab=2048, cache_limit=96)


def tinyllama_11b(target: int = mf.F32) -> CausalLMConfig:
    return CausalLMConfig(family="tinyllama", hidden=2048, inter=5632, layers=22, heads=32, kv_heads=4, vocab=32000, rope_theta=10000.0,
                          tie_embedding=False, target=target)


def tinyllama_tiny(target: int = mf.F32) -> CausalLMConfig:
    return CausalLMConfig(family="tinyllama", hidden=256, inter=512, layers=2, heads=4, kv_heads=2, vocab=1024, rope_theta=10000.0,
                          cache_limit=96, tie_embedding=False, target=target)


def causal_lm_tensors(c: CausalLMConfig) -> Iterator[Tuple[str, Tuple[int, ...], str]]:
    """HFHUBROPE tensor names (configuration_qwen.hpp:30-46, configuration_llama.hpp:39-56)."""
    H, I, D = c.hidden, c.inter, c.head_dim
    yield "model.embed_tokens.weight", (c.vocab, H), "w"
    for i in range(c.layers):
        p = f"model.layers.{i}."
        yield p + "input_layernorm.weight", (H,), "norm"
        for nm, rows in (("q_proj", c.heads * D), ("k_proj", c.kv_heads * D), ("v_proj", c.kv_heads * D)):
            yield p + f"self_attn.{nm}.weight", (rows, H), "w"
            if c.qkv_bias:
                yield p + f"self_attn.{nm}.bias", (rows,), "bias"
        yield p + "self_attn.o_proj.weight", (H, c.heads * D), "w"
        yield p + "post_attention_layernorm.weight", (H,), "norm"
        yield p + "mlp.gate_proj.weight", (I, H), "w"
        yield p + "mlp.up_proj.weight", (I, H), "w"
        yield p + "mlp.down_proj.weight", (H, I), "w"
    yield "model.norm.weight", (H,), "norm"
    if not c.tie_embedding:
        yield "lm_head.weight", (c.vocab, H), "w"


def causal_lm_ids(c: CausalLMConfig, n: int) -> np.ndarray:
    return np.random.default_rng(13).integers(0, c.vocab, size=n).astype(np.int32)


@dataclass
class ViTConfig:
    """ViTConfig("base", 16, 224, classes) (configuration_vit.hpp:86-111)."""
    hidden: int = 768
    heads: int = 12
    ffn: int = 3072
    blocks: int = 12
    patch: int = 16
    img: int = 224
    classes: int = 1000

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def tokens(self) -> int:
        return (self.img // self.patch) ** 2 + 1


def vit_b16() -> ViTConfig:
    return ViTConfig()


def vit_tiny() -> ViTConfig:
    return ViTConfig(hidden=256, heads=4, ffn=512, blocks=2, patch=16, img=64, classes=16)


def vit_tensors(c: ViTConfig) -> Iterator[Tuple[str, Tuple[int, ...], str]]:
    """Names of ViTNameConfig::init("vit") (configuration_vit.hpp:27-45) as ViTModel composes them (modeling_vit.hpp:21-104)."""
    H = c.hidden
    e = "vit.embeddings."
    yield e + "patch_embeddings.projection.weight", (H, 3, c.patch, c.patch), "w"
    yield e + "patch_embeddings.projection.bias", (H,), "bias"
    yield e + "cls_token", (H,), "w"
    yield e + "position_embeddings", (c.tokens, H), "w"
    for i in range(c.blocks):
        p = f"vit.encoder.layer.{i}."
        yield p + "layernorm_before.weight", (H,), "norm"
        yield p + "layernorm_before.bias", (H,), "bias"
        for nm in ("query", "key", "value"):
            yield p + f"attention.attention.{nm}.weight", (H, H), "w"
            yield p + f"attention.attention.{nm}.bias", (H,), "bias"
        yield p + "attention.output.dense.weight", (H, H), "w"
        yield p + "attention.output.dense.bias", (H,), "bias"
        yield p + "layernorm_after.weight", (H,), "norm"
        yield p + "layernorm_after.bias", (H,), "bias"
        yield p + "intermediate.dense.weight", (c.ffn, H), "w"
        yield p + "intermediate.dense.bias", (c.ffn,), "bias"
        yield p + "output.dense.weight", (H, c.ffn), "w"
        yield p + "output.dense.bias", (H,), "bias"
    yield "vit.layernorm.weight", (H,), "norm"
    yield "vit.layernorm.bias", (H,), "bias"
    yield "classifier.weight", (c.classes, H), "w"


def vit_images(c: ViTConfig, n: int) -> np.ndarray:
    """n images fp32 [n][H][C][W] ~ N(0,1), seed 17 (the tensor layout ViTProcessor::img2Tensor builds)."""
    return np.random.default_rng(17).standard_normal((n, c.img, 3, c.img), dtype=np.float32)


@dataclass
class LLaVAConfig:
    """LLaVAConfig(token_limit, "7B", 32064) (configuration_llava.hpp:14-40): LLaMA-7B body + CLIP-ViT-L/14-336 tower, 23 of its blocks."""
    hidden: int = 4096
    heads: int = 32
    inter: int = 11008
    layers: int = 32
    vocab: int = 32064
    cache_limit: int = 700
    rope_theta: float = 10000.0
    rms_eps: float = 1e-6
    v_hidden: int = 1024
    v_heads: int = 16
    v_ffn: int = 4096            # also the projector width; linear_2 is v_ffn x v_ffn, so it must equal `hidden`
    v_blocks: int = 23
    patch: int = 14
    img: int = 336
    image_token_id: int = 32000  # Tensor::where(32000, SEQUENCE) (modeling_llava.hpp:130)

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def v_head_dim(self) -> int:
        return self.v_hidden // self.v_heads

    @property
    def v_tokens(self) -> int:
        return (self.img // self.patch) ** 2          # rows kept after the cls row is clipped

    # the text body seen as a CausalLMConfig (LLaMABodyModel: kv heads == heads, no qkv bias, Linear lm_head)
    def body(self) -> CausalLMConfig:
        return CausalLMConfig(family="tinyllama", hidden=self.hidden, inter=self.inter, layers=self.layers, heads=self.heads, kv_heads=self.heads,
                              vocab=self.vocab, rms_eps=self.rms_eps, rope_theta=self.rope_theta, cache_limit=self.cache_limit, tie_embedding=False)


def llava_7b() -> LLaVAConfig:
    return LLaVAConfig()


def llava_tiny() -> LLaVAConfig:
    return LLaVAConfig(hidden=256, heads=2, inter=512, layers=2, cache_limit=96, v_hidden=256, v_heads=4, v_ffn=256, v_blocks=2, img=56)


def llava_tensors(c: LLaVAConfig) -> Iterator[Tuple[str, Tuple[int, ...], str]]:
    """language_model.* (configuration_llava.hpp:25-29) + vision_tower.vision_model.* with the "clip" names (configuration_vit.hpp:46-64)
    + multi_modal_projector.* (modeling_llava.hpp:81-83)."""
    H, I, D = c.hidden, c.inter, c.head_dim
    yield "language_model.model.embed_tokens.weight", (c.vocab, H), "w"
    for i in range(c.layers):
        p = f"language_model.model.layers.{i}."
        yield p + "input_layernorm.weight", (H,), "norm"
        for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
            yield p + f"self_attn.{nm}.weight", (H, H), "w"
        yield p + "post_attention_layernorm.weight", (H,), "norm"
        yield p + "mlp.gate_proj.weight", (I, H), "w"
        yield p + "mlp.up_proj.weight", (I, H), "w"
        yield p + "mlp.down_proj.weight", (H, I), "w"
    yield "language_model.model.norm.weight", (H,), "norm"
    yield "language_model.lm_head.weight", (c.vocab, H), "w"
    V, F = c.v_hidden, c.v_ffn
    e = "vision_tower.vision_model.embeddings."
    yield e + "patch_embedding.weight", (V, 3, c.patch, c.patch), "w"
    yield e + "class_embedding", (V,), "w"
    yield e + "position_embedding.weight", (c.v_tokens + 1, V), "w"
    yield "vision_tower.vision_model.pre_layrnorm.weight", (V,), "norm"
    yield "vision_tower.vision_model.pre_layrnorm.bias", (V,), "bias"
    for i in range(c.v_blocks):
        p = f"vision_tower.vision_model.encoder.layers.{i}."
        yield p + "layer_norm1.weight", (V,), "norm"
        yield p + "layer_norm1.bias", (V,), "bias"
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            yield p + f"self_attn.{nm}.weight", (V, V), "w"
            yield p + f"self_attn.{nm}.bias", (V,), "bias"
        yield p + "layer_norm2.weight", (V,), "norm"
        yield p + "layer_norm2.bias", (V,), "bias"
        yield p + "mlp.fc1.weight", (F, V), "w"
        yield p + "mlp.fc1.bias", (F,), "bias"
        yield p + "mlp.fc2.weight", (V, F), "w"
        yield p + "mlp.fc2.bias", (V,), "bias"
    yield "multi_modal_projector.linear_1.weight", (F, V), "w"
    yield "multi_modal_projector.linear_1.bias", (F,), "bias"
    yield "multi_modal_projector.linear_2.weight", (F, F), "w"
    yield "multi_modal_projector.linear_2.bias", (F,), "bias"


def llava_inputs(c: LLaVAConfig, n_text: int = 10):
    """ids = 3 text ids + [<image> 32000] + n_text text ids (seed 19); image fp32 [H][C][W] ~ N(0,1) seed 23."""
    text = np.random.default_rng(19).integers(0, c.image_token_id, size=3 + n_text)
    ids = np.concatenate([text[:3], [c.image_token_id], text[3:]]).astype(np.int32)
    img = np.random.default_rng(23).standard_normal((c.img, 3, c.img), dtype=np.float32)
    return ids, img


# ---------------------------------------------------------------------------------------------------------------
# a sparse-MoE feed-forward block (MiniCPM-MoE geometry; models/minicpm_moe/modeling_minicpm_moe.hpp:41-115, configuration_minicpm_moe.hpp:14-31)
# ---------------------------------------------------------------------------------------------------------------
@dataclass
class MoEConfig:
    hidden: int = 256
    inter: int = 512
    experts: int = 4
    per_tok: int = 2
    base: str = "model.layers.0.mlp."


def moe_tiny() -> MoEConfig:
    return MoEConfig()


def moe_tensors(c: MoEConfig) -> Iterator[Tuple[str, Tuple[int, ...], str]]:
    yield c.base + "gate.weight", (c.experts, c.hidden), "w"
    for e in range(c.experts):
        p = f"{c.base}experts.{e}."
        yield p + "w1.weight", (c.inter, c.hidden), "w"      # gate_proj
        yield p + "w3.weight", (c.inter, c.hidden), "w"      # up_proj
        yield p + "w2.weight", (c.hidden, c.inter), "w"      # down_proj


def moe_input(c: MoEConfig, n_tok: int, seed: int = 17) -> np.ndarray:
    return np.random.default_rng(seed).standard_normal((n_tok, c.hidden), dtype=np.float32)
