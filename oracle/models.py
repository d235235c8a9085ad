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
