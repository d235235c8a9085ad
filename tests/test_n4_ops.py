"""the extra ops of the other model families (sliding-window mask, NTK / LongRoPE rotary, MoE top-k / bincount / gather / scatter-add, Fuyu gather).

tests/golden/n4_ops.npz holds the REFERENCE's own outputs for each op (oracle/make_golden.py --n4 drives oracle/_ref/ref_ops, which calls the reference's Layer / Tensor
API); the numpy / C restatements in oracle/ are pinned against it here on the CPU, and the HIP kernels against both on the GPU -- bit equality everywhere."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as orc

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "n4_ops.npz"))


def eq(a, b):
    a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    return np.array_equal(a.reshape(-1).view(np.uint32) if a.dtype == np.float32 else a.reshape(-1), np.asarray(b, dtype=a.dtype).reshape(-1).view(np.uint32) if a.dtype == np.float32 else np.asarray(b).reshape(-1))


def _ntk(pp):
    theta, maxp, orig, heads, D = pp[0], int(pp[1]), int(pp[2]), int(pp[3]), int(pp[4])
    return theta, maxp, orig, heads, D, pp[5:5 + D // 2], pp[5 + D // 2:5 + D]


# ---- CPU: the restatements against the reference's outputs ----------------------------------------------------------------------------------------------
def test_oracle_sliding_window_mask_matches_reference():
    W, H, K = (int(v) for v in G["swmask_p"])
    assert eq(orc.sliding_window_mask(G["swmask_x"], H, K, W), G["swmask_y"])
    one = G["swmask_x"][:1]
    assert eq(orc.sliding_window_mask(one, H, K, W), one)          # a decode row passes through (CPUSlidingWindowMask.cpp:55-57)


def test_oracle_topk_ties_and_bincount_match_reference():
    v, i = orc.topk_rows(G["topk_x"], G["topk_v"].shape[1])
    assert eq(v, G["topk_v"]) and eq(i, G["topk_i"])
    assert G["topk_i"][4].tolist() == [15.0, 14.0, 13.0, 12.0]      # equal values: the larger index first
    # the HEAD-axis form ([1, H, S, 1]: in BSHD memory the same [S][H] rows, output [1, k, S, 1] = [S][k]) is the same selection over the same rows
    assert eq(G["topk_head_v"], G["topk_v"]) and eq(G["topk_head_i"], G["topk_i"])
    assert eq(orc.bincount(G["bincount_x"]), G["bincount_y"])


def test_oracle_gather_scatter_match_reference():
    assert eq(orc.scatter_add_rows(G["sa_dst"], G["sa_src"], G["sa_idx"]), G["sa_y"])        # row 2 is hit twice, in order
    assert eq(orc.gather_rows(G["sa_dst"], G["gr_idx"]), G["gr_y"])
    assert eq(orc.fuyu_gather(G["sa_dst"], G["fuyu_patches"], G["fuyu_idx"]), G["fuyu_y"])


def test_oracle_ntk_rope_matches_reference_long_short_and_decode_step():
    theta, maxp, orig, heads, D, lf, sf = _ntk(G["ntk_p"])
    s, c = orc.rope_table_ntk(theta, D, maxp, orig, lf, sf)
    assert eq(orc.rope_apply(G["ntk_x0"], 6, heads, D, s[:6], c[:6]), G["ntk_y0"])
    assert eq(orc.rope_apply(G["ntk_x1"], 1, heads, D, s[6:7], c[6:7]), G["ntk_y1"])       # the op's position counter moved on by the prefill length
    theta, maxp, orig, heads, D, lf, sf = _ntk(G["ntk_p_short"])
    s, c = orc.rope_table_ntk(theta, D, maxp, orig, lf, sf)
    assert maxp < orig                                                                     # the short factors
    assert eq(orc.rope_apply(G["ntk_x0"], 6, heads, D, s[:6], c[:6]), G["ntk_y_short"])


def test_product_ntk_table_is_host_code_and_matches_the_oracle():
    from mllm_amd import lib
    for key in ("ntk_p", "ntk_p_short"):
        theta, maxp, orig, heads, D, lf, sf = _ntk(G[key])
        s, c = lib.rope_table_ntk(theta, D, maxp, orig, lf, sf)
        so, co = orc.rope_table_ntk(theta, D, maxp, orig, lf, sf)
        assert eq(s, so) and eq(c, co)
    L = lib.load()
    buf = np.zeros(64, dtype=np.float32)
    p = buf.ctypes.data_as(C.c_void_p)
    assert L.mllm_hip_rope_table_ntk(C.c_float(1e4), C.c_int(7), C.c_int(4), C.c_int(32), p, p, p, p) == lib.ERR_ARG      # odd dim
    assert L.mllm_hip_rope_table_ntk(C.c_float(1e4), C.c_int(8), C.c_int(4), C.c_int(32), None, p, p, p) == lib.ERR_ARG


def test_n4_entry_points_validate_before_any_launch():
    from mllm_amd import lib
    L = lib.load()
    assert L.mllm_hip_sliding_window_mask(None, None, C.c_int(4), C.c_int(2), C.c_int(3), C.c_int(2), None) == lib.ERR_SHAPE      # fewer keys than rows
    assert L.mllm_hip_sliding_window_mask(None, None, C.c_int(0), C.c_int(2), C.c_int(3), C.c_int(2), None) == lib.OK
    assert L.mllm_hip_topk_rows(None, C.c_int64(8), None, None, C.c_int(2), C.c_int(8), C.c_int(9), None) == lib.ERR_SHAPE           # k > n
    assert L.mllm_hip_topk_rows(None, C.c_int64(8), None, None, C.c_int(2), C.c_int(8), C.c_int(2), None) == lib.ERR_ARG
    assert L.mllm_hip_bincount(None, C.c_int(0), None, C.c_int(4), None) == lib.ERR_ARG
    assert L.mllm_hip_gather_rows(None, C.c_int64(8), C.c_int(4), None, None, C.c_int64(8), C.c_int(0), C.c_int(8), C.c_int(0), None) == lib.OK
    assert L.mllm_hip_scatter_add_rows(None, C.c_int64(8), C.c_int(4), None, C.c_int64(8), None, C.c_int(3), C.c_int(8), None) == lib.ERR_ARG
    assert L.mllm_hip_scatter_add_rows(None, C.c_int64(8), C.c_int(-1), None, C.c_int64(8), None, C.c_int(3), C.c_int(8), None) == lib.ERR_SHAPE


# ---- GPU: the kernels against the reference's outputs and, at larger sizes, against the restatement ---------------------------------------------------------
@pytest.mark.gpu
def test_hip_n4_ops_match_the_reference_outputs():
    from mllm_amd import lib, ops
    ops.require_gpu()
    W, H, K = (int(v) for v in G["swmask_p"])
    assert eq(ops.sliding_window_mask(G["swmask_x"], H, K, W), G["swmask_y"])
    assert eq(ops.sliding_window_mask(G["swmask_x"][:1], H, K, W), G["swmask_x"][:1])
    v, i = ops.topk_rows(G["topk_x"], G["topk_v"].shape[1])
    assert eq(v, G["topk_v"]) and eq(i, G["topk_i"])
    assert eq(ops.bincount(G["bincount_x"], G["bincount_y"].size), G["bincount_y"])
    assert eq(ops.scatter_add_rows(G["sa_dst"], G["sa_src"], G["sa_idx"]), G["sa_y"])
    assert eq(ops.gather_rows(G["sa_dst"], G["gr_idx"]), G["gr_y"])
    assert eq(ops.fuyu_gather(G["sa_dst"], G["fuyu_patches"], G["fuyu_idx"]), G["fuyu_y"])
    theta, maxp, orig, heads, D, lf, sf = _ntk(G["ntk_p"])
    s, c = lib.rope_table_ntk(theta, D, maxp, orig, lf, sf)
    assert eq(ops.rope_apply(G["ntk_x0"], 6, heads, D, s[:6], c[:6]), G["ntk_y0"])
    assert eq(ops.rope_apply(G["ntk_x1"], 1, heads, D, s[6:7], c[6:7]), G["ntk_y1"])


@pytest.mark.gpu
def test_hip_n4_ops_at_model_sizes_match_the_restatement():
    from mllm_amd import ops
    ops.require_gpu()
    r = np.random.default_rng(5)
    # Mistral-like window over a longer cache: 8 heads, 96 new rows on 160 keys, window 64
    x = r.standard_normal((96, 8 * 160)).astype(np.float32)
    assert eq(ops.sliding_window_mask(x, 8, 160, 64), orc.sliding_window_mask(x, 8, 160, 64))
    # router: 300 tokens x 64 experts (quantised scores: many ties), k = 8; and a ragged n
    sc = (r.integers(0, 12, size=(300, 64)) / 16.0).astype(np.float32)
    v, i = ops.topk_rows(sc, 8)
    vo, io = orc.topk_rows(sc, 8)
    assert eq(v, vo) and eq(i, io)
    sc = r.standard_normal((33, 100)).astype(np.float32)
    v, i = ops.topk_rows(sc, 100)          # k = n: a full descending sort of every row
    vo, io = orc.topk_rows(sc, 100)
    assert eq(v, vo) and eq(i, io)
    ids = io[:, :6].reshape(-1)
    assert eq(ops.bincount(ids, 100), np.bincount(ids.astype(np.int64), minlength=100).astype(np.float32))
    # MoE combine at hidden 2048: 500 expert rows into 128 token rows, every token hit several times
    dst = r.standard_normal((128, 2048)).astype(np.float32); src = r.standard_normal((500, 2048)).astype(np.float32)
    idx = r.integers(0, 128, size=500).astype(np.float32)
    assert eq(ops.scatter_add_rows(dst, src, idx), orc.scatter_add_rows(dst, src, idx))
    assert eq(ops.gather_rows(src, idx[:77]), orc.gather_rows(src, idx[:77]))
    fi = np.where(r.random(128) < 0.4, r.integers(0, 500, size=128), -1).astype(np.float32)
    assert eq(ops.fuyu_gather(dst, src, fi), orc.fuyu_gather(dst, src, fi))
    # indices outside the table are never dereferenced: a bad destination row is skipped, a bad source row leaves its output row as it was (zeros here)
    bad = idx.copy(); bad[[3, 77, 400]] = [-1.0, 128.0, 1e9]
    keep = np.ones(500, dtype=bool); keep[[3, 77, 400]] = False
    assert eq(ops.scatter_add_rows(dst, src, bad), orc.scatter_add_rows(dst, src[keep], bad[keep]))
    gi = np.array([5, 499, 500, -3, 0], dtype=np.float32)
    want = np.zeros((5, 2048), dtype=np.float32); want[[0, 1, 4]] = src[[5, 499, 0]]
    assert eq(ops.gather_rows(src, gi), want)


# ---- the whole sparse-MoE block: the reference's MiniCPMMoE module run by the reference (tests/golden/moe.npz) --------------------------------------
def _moe_setup():
    import hashlib

    from mllm_amd import mllmfile as mf, synth
    from oracle import models as om
    from mllm_amd import synthfile as weights
    cfg = synth.moe_tiny()
    path = weights.moe_file(cfg)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "moe.npz"))
    f = mf.MllmFile(path)
    dig = "".join(hashlib.sha256(f.raw(n).tobytes()).hexdigest()[:16] for n in sorted(f.names()))
    f.close()
    assert dig.encode() == g["digest"].tobytes(), "the fixture quantiser's file differs from the one the reference's quantize wrote for the golden"
    return cfg, path, g, om


def test_moe_block_restatement_matches_the_reference_module():
    """oracle.models.moe_block against the reference's own MiniCPMMoE::Forward (models/minicpm_moe/modeling_minicpm_moe.hpp:52-105), bit for bit: 37 tokens (prefill: every
    expert has rows, fewer than 16 for some) and one token (decode: two experts with one row each)."""
    cfg, path, g, om = _moe_setup()
    w = om.Weights(path)
    for tag in ("p", "d"):
        y = om.moe_block(w, cfg, g["x_" + tag])
        assert y.shape == g["y_" + tag].shape and np.array_equal(y, g["y_" + tag]), (tag, float(np.abs(y - g["y_" + tag]).max()))


def test_moe_block_argument_checks():
    from mllm_amd import lib
    L = lib.load()
    P = C.c_void_p(0x1000)
    arr = (C.c_void_p * 2)(0x1000, 0x1000)
    assert L.mllm_hip_moe_block(P, P, C.c_int(4), C.c_int(200), C.c_int(512), C.c_int(2), C.c_int(1), P, arr, arr, arr, None) == lib.ERR_SHAPE      # hidden not in super-blocks
    assert L.mllm_hip_moe_block(P, P, C.c_int(4), C.c_int(256), C.c_int(512), C.c_int(2), C.c_int(3), P, arr, arr, arr, None) == lib.ERR_SHAPE      # more experts per token than experts
    assert L.mllm_hip_moe_block(P, P, C.c_int(0), C.c_int(256), C.c_int(512), C.c_int(2), C.c_int(1), P, arr, arr, arr, None) == lib.OK
    assert L.mllm_hip_moe_block(P, None, C.c_int(4), C.c_int(256), C.c_int(512), C.c_int(2), C.c_int(1), P, arr, arr, arr, None) == lib.ERR_ARG
    assert L.mllm_hip_scale_rows(P, C.c_int64(6), P, C.c_int(3), C.c_int(6), None) == lib.ERR_SHAPE


@pytest.mark.gpu
def test_hip_moe_block_matches_the_reference_module():
    """mllm_hip_moe_block against the reference's own MiniCPMMoE run (tests/golden/moe.npz), every output bit: the prefill case (37 tokens: experts with fewer and with
    more than 16 rows, i.e. both the GEMV and the GEMM form of the expert Linears) and the decode case (one token); then, at a size the golden does not cover (8 experts,
    3 per token, 300 tokens), against the restatement."""
    from mllm_amd import mllmfile as mf, ops, synth
    from mllm_amd import synthfile as weights
    ops.require_gpu()
    cfg, path, g, om = _moe_setup()

    def run(cfg, path, x):
        f = mf.MllmFile(path)
        raw = lambda n: np.array(f.raw(n))
        b = cfg.base
        y = ops.moe_block(x, raw(b + "gate.weight"), [raw(f"{b}experts.{e}.w1.weight") for e in range(cfg.experts)], [raw(f"{b}experts.{e}.w3.weight") for e in range(cfg.experts)],
                          [raw(f"{b}experts.{e}.w2.weight") for e in range(cfg.experts)], cfg.inter, cfg.per_tok).cpu().numpy()
        f.close()
        return y
    for tag in ("p", "d"):
        assert np.array_equal(run(cfg, path, g["x_" + tag]), g["y_" + tag]), tag
    big = synth.MoEConfig(hidden=512, inter=768, experts=8, per_tok=3)
    bpath = weights.moe_file(big)
    x = synth.moe_input(big, 300, seed=5)
    assert np.array_equal(run(big, bpath, x), om.moe_block(om.Weights(bpath), big, x))


# ---- A7, the eager-attention form of F_MM (CPUMatmulFunc.hpp:155-172 -> compute/GemmFp.hpp:104-150): tests/golden/mm_bhsd.npz is the reference's own Tensor::mm on BHSD tensors ----
MM_CASES = ("qk", "pv", "long", "full", "deep")


def test_gemm_fp32_bhsd_restatement_matches_the_reference():
    """oracle.gemm_fp32_bhsd against Tensor::mm on BHSD operands as the reference computes it: q k^T and p v shapes whose rows and columns do not fill the 8 x 8 micro-kernel
    (full tiles: one fma chain over K; edge tiles: per-256-block partial sums), K beyond one block on both kinds of tile."""
    from oracle import oracle as orc
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mm_bhsd.npz"))
    for n in MM_CASES:
        c = orc.gemm_fp32_bhsd(g[n + "_a"], g[n + "_b"])
        assert np.array_equal(c, g[n + "_y"]), n
    # the split matters: one chain over all of K differs from block partial sums on the deep edge case
    a, b = g["deep_a"], g["deep_b"]
    H, M, K = a.shape
    N = b.shape[2]
    pad_a = np.concatenate([a, np.zeros((H, 1, K), np.float32)], axis=1)          # one more row: M = 25 turns rows 24.. into an edge tile, rows 0..23 stay full
    assert np.array_equal(orc.gemm_fp32_bhsd(pad_a, b)[:, :M], g["deep_y"])


def test_gemm_f32_bhsd_argument_checks():
    from mllm_amd import lib
    L = lib.load()
    P = C.c_void_p(0x1000)
    assert L.mllm_hip_gemm_f32_bhsd(P, P, C.c_int(lib.Q4_K), P, C.c_int(1), C.c_int(4), C.c_int(4), C.c_int(4), None) == lib.ERR_DTYPE
    assert L.mllm_hip_gemm_f32_bhsd(P, P, C.c_int(lib.F32), P, C.c_int(1), C.c_int(4), C.c_int(4), C.c_int(0), None) == lib.ERR_SHAPE
    assert L.mllm_hip_gemm_f32_bhsd(P, P, C.c_int(lib.F32), P, C.c_int(0), C.c_int(4), C.c_int(4), C.c_int(4), None) == lib.OK
    assert L.mllm_hip_gemm_f32_bhsd(P, None, C.c_int(lib.F32), P, C.c_int(1), C.c_int(4), C.c_int(4), C.c_int(4), None) == lib.ERR_ARG


@pytest.mark.gpu
def test_hip_gemm_f32_bhsd_matches_the_reference():
    """mllm_hip_gemm_f32_bhsd against the reference's Tensor::mm on BHSD tensors (tests/golden/mm_bhsd.npz), every bit; then an attention-sized pair (12 heads, 300 x 300 x 128
    and 300 x 128 x 300) with fp32 and with fp16 right operands (gemm_fp32_fp16: the fp16 K/V cache of the eager branch) against the restatement."""
    from mllm_amd import ops
    from oracle import oracle as orc
    ops.require_gpu()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mm_bhsd.npz"))
    for n in MM_CASES:
        assert np.array_equal(ops.gemm_f32_bhsd(g[n + "_a"], g[n + "_b"]).cpu().numpy(), g[n + "_y"]), n
    r = np.random.default_rng(77)
    for (H, M, N, K) in ((12, 300, 300, 128), (12, 300, 128, 300), (2, 1, 555, 128)):
        a = r.standard_normal((H, M, K), dtype=np.float32)
        b = r.standard_normal((H, K, N), dtype=np.float32)
        assert np.array_equal(ops.gemm_f32_bhsd(a, b).cpu().numpy(), orc.gemm_fp32_bhsd(a, b)), (H, M, N, K)
        b16 = b.astype(np.float16)
        assert np.array_equal(ops.gemm_f32_bhsd(a, b16).cpu().numpy(), orc.gemm_fp32_bhsd(a, b16)), (H, M, N, K, "f16")
