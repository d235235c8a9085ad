"""Pins the oracle (oracle/restate.c) against golden vectors captured from the compiled reference itself
(tests/golden/ops.npz, made by oracle/make_golden.py with oracle/_ref/ref_ops).  CPU only.

Bar: bit-exact everywhere -- the oracle restates the reference's operation order (AVX2 lane order of the dot products,
the FA2 tile recurrence, GCC's fma contractions), op by op and composed into the whole Qwen2-VL graph."""
import os

import numpy as np
import pytest

from oracle import oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


def test_linear_q4k_bit_exact(ops_gold):
    g = ops_gold
    for x, y in ((g["lin_x5"], g["lin_y5"]), (g["lin_x1"], g["lin_y1"])):
        got = orc.linear(x, g["lin_w"], orc.Q4_K, 96, g["lin_b"])
        assert np.array_equal(got, y), maxdiff(got, y)


def test_linear_f32_bit_exact(ops_gold):
    g = ops_gold
    got = orc.linear(g["linf_x"], g["linf_w"].view(np.uint8), orc.F32, 24)
    assert np.array_equal(got, g["linf_y"]), maxdiff(got, g["linf_y"])


def test_tied_head_q40_and_embedding_bit_exact(ops_gold):
    g = ops_gold
    got = orc.linear(g["mm_x"], g["emb_w"], orc.Q4_0, 160)
    assert np.array_equal(got, g["mm_y"]), maxdiff(got, g["mm_y"])
    emb = orc.embedding(g["emb_ids"], g["emb_w"], orc.Q4_0, 512)
    assert np.array_equal(emb, g["emb_y"])


def test_norms(ops_gold):
    g = ops_gold
    r = orc.rmsnorm(g["norm_x"], g["norm_w"], 1e-6)
    assert np.array_equal(r, g["rms_y"]), maxdiff(r, g["rms_y"])
    l = orc.layernorm(g["norm_x"], g["norm_w"], g["norm_b"], 1e-6)
    assert maxdiff(l, g["ln_y"]) <= 2e-6, maxdiff(l, g["ln_y"])   # reference's a*b/c+d contraction is compiler-chosen


def test_activations_bit_exact(ops_gold):
    g = ops_gold
    assert np.array_equal(orc.silu(g["act_x"]), g["silu_y"])
    assert np.array_equal(orc.gelu(g["act_x"]), g["gelu_y"])
    assert np.array_equal(orc.quickgelu(g["act_x"]), g["quickgelu_y"])


def test_softmax(ops_gold):
    g = ops_gold
    got = orc.softmax(g["sm_x"])
    assert maxdiff(got, g["sm_y"]) <= 1e-7


def test_patch_convs(ops_gold):
    g = ops_gold
    got = orc.patch_gemm(g["conv3_x"], g["conv3_w"])
    assert np.array_equal(got, g["conv3_y"]), maxdiff(got, g["conv3_y"])
    got2 = orc.conv2d_patch(g["conv2_x"], 8, 3, 12, g["conv2_w"], 8, 4, g["conv2_b"])
    ref2 = g["conv2_y"].reshape(got2.shape)
    assert np.array_equal(got2, ref2), maxdiff(got2, ref2)


def test_rotary(ops_gold):
    g = ops_gold
    s, c = orc.mrope_table(1000000.0, 128, g["mrope_pos"])
    got = orc.rope_apply(g["mrope_x"], 5, 2, 128, s, c)
    assert np.array_equal(got, g["mrope_y"]), maxdiff(got, g["mrope_y"])
    s, c = orc.rope_table_hf(10000.0, 64, 64)
    got = orc.rope_apply(g["rope_x"], 5, 2, 64, s[:5], c[:5])
    assert np.array_equal(got, g["rope_y"]), maxdiff(got, g["rope_y"])
    ang = orc.vision_rope_angles(1, 4, 4, 2, 8)
    got = orc.vision_rope_apply(g["vrope_x"], 16, 2, 16, ang)
    assert np.array_equal(got, g["vrope_y"]), maxdiff(got, g["vrope_y"])


def test_attention_fp32_kv(ops_gold):
    g = ops_gold
    o = orc.attention(g["fa_q"], g["fa_k"], g["fa_v"], 40, 40, 2, 2, 16, False)
    assert np.array_equal(o, g["fa_o"]), maxdiff(o, g["fa_o"])
    o = orc.attention(g["fac_q"], g["fac_k"], g["fac_v"], 12, 12, 4, 2, 16, True)
    assert np.array_equal(o, g["fac_o"]), maxdiff(o, g["fac_o"])


def _attn_block(g, x, pos, k_cache, v_cache):
    """QWen2Attention (modeling_qwen2_vl.hpp:247-275) composed from oracle ops; caches are fp16 (uint16) arrays [T][128]."""
    H, D, heads, kvh = 256, 128, 2, 1
    q = orc.linear(x, g["blk_self_attn_q_proj_weight"], orc.Q4_K, heads * D, g["blk_self_attn_q_proj_bias"].view(np.float32))
    k = orc.linear(x, g["blk_self_attn_k_proj_weight"], orc.Q4_K, kvh * D, g["blk_self_attn_k_proj_bias"].view(np.float32))
    v = orc.linear(x, g["blk_self_attn_v_proj_weight"], orc.Q4_K, kvh * D, g["blk_self_attn_v_proj_bias"].view(np.float32), out_f16=True)
    S = x.shape[0]
    s, c = orc.mrope_table(1000000.0, D, pos)
    q = orc.rope_apply(q, S, heads, D, s, c)
    k16 = orc.rope_apply(k, S, kvh, D, s, c, out_f16=True)
    k_all = np.concatenate([k_cache, k16.reshape(S, kvh * D)]) if k_cache is not None else k16.reshape(S, kvh * D)
    v_all = np.concatenate([v_cache, v.reshape(S, kvh * D)]) if v_cache is not None else v.reshape(S, kvh * D)
    o = orc.attention(q, k_all, v_all, S, k_all.shape[0], heads, kvh, D, True)
    y = orc.linear(o, g["blk_self_attn_o_proj_weight"], orc.Q4_K, H)
    return y, k_all, v_all


def test_attention_block_prefill_and_decode(ops_gold):
    g = ops_gold
    pos8 = np.tile(np.arange(8, dtype=np.float32), (3, 1))
    y8, kc, vc = _attn_block(g, g["blk_x8"], pos8, None, None)
    assert np.array_equal(y8, g["blk_attn8"]), maxdiff(y8, g["blk_attn8"])
    y1, _, _ = _attn_block(g, g["blk_x1"], np.full((3, 1), 8, dtype=np.float32), kc, vc)
    assert np.array_equal(y1, g["blk_attn1"]), maxdiff(y1, g["blk_attn1"])


def test_mlp_block(ops_gold):
    g = ops_gold
    x = g["blk_x8"]
    gate = orc.linear(x, g["blk_mlp_gate_proj_weight"], orc.Q4_K, 512)
    up = orc.linear(x, g["blk_mlp_up_proj_weight"], orc.Q4_K, 512)
    y = orc.linear(orc.silu(gate) * up, g["blk_mlp_down_proj_weight"], orc.Q4_K, 256)
    assert np.array_equal(y, g["blk_mlp8"]), maxdiff(y, g["blk_mlp8"])


# ---- model-level pin: the oracle's ops composed into the reference's graphs reproduce the reference run bit for bit ----------
def _tiny():
    from mllm_amd import synth
    from mllm_amd import synthfile as weights
    from oracle import models
    cfg = synth.qwen2vl_tiny()
    return cfg, models, models.Weights(weights.qwen2vl_file(cfg)), np.load(os.path.join(GOLD, "qwen2vl_tiny.npz"))


def test_composed_vision_tower_bit_exact():
    from mllm_amd import synth
    cfg, models, w, g = _tiny()
    pix, grid, _ = synth.qwen2vl_inputs(cfg, (8, 8), 6)
    emb = models.vision_forward(w, cfg, pix, grid)
    assert np.array_equal(emb, g["image_embeds"]), maxdiff(emb, g["image_embeds"])


def test_composed_llm_text_and_image_bit_exact():
    from mllm_amd import synth
    cfg, models, w, g = _tiny()
    m = models.LLM(w, cfg)
    lg = m.prefill(g["ids_text"])
    rows, toks = [lg], [int(lg.argmax())]
    for _ in range(5):
        lg = m.decode(toks[-1]); rows.append(lg); toks.append(int(lg.argmax()))
    assert toks == g["tokens_text"].tolist() and np.array_equal(np.stack(rows), g["logits_text"])
    pix, grid, _ = synth.qwen2vl_inputs(cfg, (8, 8), 6)
    m = models.LLM(w, cfg)
    lg = m.prefill(g["ids"], pix, grid)
    rows, toks = [lg], [int(lg.argmax())]
    for _ in range(23):
        lg = m.decode(toks[-1]); rows.append(lg); toks.append(int(lg.argmax()))
    assert toks == g["tokens"].tolist() and np.array_equal(np.stack(rows), g["logits"])


def test_composed_graph_on_ragged_grids_bit_exact():
    """Non-square grids with patch / token counts off the kernels' tile sizes (6 x 10, 8 x 12, 4 x 18; tests/golden/qwen2vl_ragged.npz, the reference's own run): the
    oracle's tower and LLM reproduce every image embedding and every logit of 6 greedy steps."""
    from mllm_amd import synth
    cfg, models, w, _ = _tiny()
    g = np.load(os.path.join(GOLD, "qwen2vl_ragged.npz"))
    for k in range(3):
        grid = g[f"grid{k}"]
        pix, grid2, ids = synth.qwen2vl_inputs(cfg, (int(grid[1]), int(grid[2])), int(g[f"ntext{k}"]))
        assert np.array_equal(grid, grid2)
        emb = models.vision_forward(w, cfg, pix, grid)
        assert np.array_equal(emb, g[f"image_embeds{k}"]), (k, maxdiff(emb, g[f"image_embeds{k}"]))
        m = models.LLM(w, cfg)
        lg = m.prefill(ids, pix, grid)
        rows, toks = [lg], [int(lg.argmax())]
        for _ in range(5):
            lg = m.decode(toks[-1]); rows.append(lg); toks.append(int(lg.argmax()))
        assert toks == g[f"tokens{k}"].tolist() and np.array_equal(np.stack(rows), g[f"logits{k}"]), k
    m = models.LLM(w, cfg)                      # the shortest prompt the reference takes (two tokens)
    lg = m.prefill(np.array([17, 23], dtype=np.int32))
    rows, toks = [lg], [int(lg.argmax())]
    for _ in range(4):
        lg = m.decode(toks[-1]); rows.append(lg); toks.append(int(lg.argmax()))
    assert toks == g["one_tokens"].tolist() and np.array_equal(np.stack(rows), g["one_logits"])


def test_expf_restatement_equals_libm():
    """The attention kernels carry a restatement of glibc's expf (oracle/restate.c:orc_expf = mllm_amd/csrc/common.h:glibc_expf);
    it must agree bit for bit with the libm the reference links against, over the argument range attention produces and beyond."""
    import ctypes as C
    l = orc.lib()
    l.orc_expf_mismatches.restype = C.c_long
    l.orc_expf_mismatches.argtypes = [C.c_float, C.c_float, C.c_long, C.c_uint64]
    for lo, hi, n in ((-110.0, 0.0, 20_000_000), (-1.0, 0.0, 10_000_000), (-104.5, -103.0, 5_000_000), (0.0, 89.0, 5_000_000)):
        assert l.orc_expf_mismatches(lo, hi, n, 3) == 0
