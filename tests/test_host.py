"""CPU-side tests of the product's host logic: the C-ABI library loads and exports every symbol include/mllm_hip.h
declares, the .mllm container round-trips, the synthetic weight files are what mllm_amd/synth.py says they are (valid Q4_K / Q4_0
blocks drawn in the quantised domain, deterministic, the reference's per-name dtype policy), and the host-side table builders agree
with the oracle.  No GPU, no compute launch."""
import os

import numpy as np
import pytest

from mllm_amd import lib, mllmfile as mf, synth
from mllm_amd import synthfile as weights
from oracle import oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_library_exports_every_declared_symbol():
    so = lib.load()
    names = lib.declared_symbols()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(so, n)]
    assert not missing, missing


def test_quantised_domain_blocks_are_valid_and_centred():
    """q4k_blocks / q40_blocks: the right sizes, every 6-bit field within range by construction of the packing, dequantised weights (oracle restatement of
    dequantize_row_q4_K / _q4_0, QuantizeQ4.cpp:295-333, 74-93) with zero mean and the requested standard deviation, and the draw is a pure function of the seed."""
    r = np.random.default_rng(5)
    b = synth.q4k_blocks(r, 2048)
    assert b.shape == (2048, 144) and b.dtype == np.uint8
    w = orc.dequantize(b.ravel(), orc.Q4_K, 2048 * 256)
    assert abs(float(w.mean())) < 1e-3 and 0.017 < float(w.std()) < 0.025 and np.isfinite(w).all()
    q = synth.q40_blocks(r, 4096)
    assert q.shape == (4096, 18)
    w0 = orc.dequantize(q.ravel(), orc.Q4_0, 4096 * 32)
    assert abs(float(w0.mean())) < 3e-3 and 0.017 < float(w0.std()) < 0.025
    assert np.array_equal(synth.q4k_blocks(np.random.default_rng(5), 2048), b)
    # the 6-bit fields as the reference's reader unpacks them (get_scale_min_k4): scales 20..63, mins follow the scales
    sc_lo, m_lo = b[:, 4:8] & 63, b[:, 8:12] & 63
    assert sc_lo.min() >= 20 and np.all(np.abs(m_lo.astype(int) - np.rint(sc_lo * (7.75 / 8)).astype(int)) <= 2)
    assert synth.quantized_blocks(mf.Q4_K, r, 512).size == 288 and synth.quantized_blocks(mf.Q4_0, r, 64).size == 36
    with pytest.raises(ValueError):
        synth.quantized_blocks(mf.Q4_K, r, 100)


def test_no_quantiser_in_the_tree():
    """The synthetic files are drawn in the quantised domain: no fp32 -> Q4_K / Q4_0 fitter exists in the product library, the package or the tests."""
    so = lib.load()
    assert not hasattr(so, "mllm_hip_quantize_host") and not hasattr(so, "mllm_quant_rows")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert not os.path.exists(os.path.join(root, "tests", "fixtures"))
    for d in ("mllm_amd", "tests"):
        for f in os.listdir(os.path.join(root, d)):
            if f.endswith(".py") and f != "test_host.py":
                src = open(os.path.join(root, d, f)).read()
                assert "quantlib" not in src and "tests.fixtures" not in src, f


def test_mllm_file_roundtrip(tmp_path):
    p = str(tmp_path / "t.mllm")
    a = np.arange(12, dtype=np.float32)
    b = np.arange(36, dtype=np.uint8)
    mf.write_mllm(p, [("a.weight", mf.F32, a), ("b.weight", mf.Q4_0, b)])
    f = mf.MllmFile(p)
    assert f.names() == ["a.weight", "b.weight"]
    assert np.array_equal(f.f32("a.weight"), a) and f.dtype("b.weight") == mf.Q4_0 and np.array_equal(f.raw("b.weight"), b)
    f.close()
    with open(p, "r+b") as fh:
        fh.write(b"\0\0\0\0")
    with pytest.raises(ValueError):
        mf.MllmFile(p)


def test_storage_dtype_policy():
    assert synth.storage_dtype("model.embed_tokens.weight") == mf.Q4_0
    assert synth.storage_dtype("model.layers.0.self_attn.q_proj.weight") == mf.Q4_K
    assert synth.storage_dtype("model.layers.0.self_attn.q_proj.bias") == mf.F32
    assert synth.storage_dtype("model.layers.3.input_layernorm.weight") == mf.F32
    assert synth.storage_dtype("visual.patch_embed.proj.weight") == mf.F32
    assert synth.storage_dtype("visual.merger.ln_q.weight") == mf.F32
    assert synth.storage_dtype("visual.merger.mlp.0.weight") == mf.Q4_K


def test_synthetic_file_is_deterministic_and_follows_the_dtype_policy(tmp_path):
    """The tiny Qwen2-VL file twice (1 and 4 worker threads): identical bytes; every tensor has the dtype QuantWriter's policy gives its name and the size its shape
    needs; the reference reads exactly this file for the goldens (oracle/make_golden.py)."""
    cfg = synth.qwen2vl_tiny()
    p1 = weights.build_q4k_file(str(tmp_path / "a.mllm"), synth.qwen2vl_tensors(cfg), workers=1)
    p2 = weights.build_q4k_file(str(tmp_path / "b.mllm"), synth.qwen2vl_tensors(cfg), workers=4)
    assert open(p1, "rb").read() == open(p2, "rb").read()
    f = mf.MllmFile(p1)
    specs = {n: s for n, s, _ in synth.qwen2vl_tensors(cfg)}
    assert f.names() == list(specs)
    for n, shp in specs.items():
        dt = synth.storage_dtype(n)
        assert f.dtype(n) == dt and f.raw(n).size == mf.nbytes(dt, int(np.prod(shp))), n
    f.close()


def test_rotary_tables_match_oracle():
    pos = np.array([[0, 1, 2, 9], [0, 1, 5, 9], [0, 3, 2, 9]], dtype=np.float32)
    s, c = lib.mrope_table(1000000.0, 128, pos)
    so, co = orc.mrope_table(1000000.0, 128, pos)
    assert np.array_equal(s, so) and np.array_equal(c, co)
    s, c = lib.rope_table_hf(10000.0, 64, 33)
    so, co = orc.rope_table_hf(10000.0, 64, 33)
    assert np.array_equal(s, so) and np.array_equal(c, co)
    s, c = lib.vision_rope_table(1, 8, 6, 2, 40)
    ang = orc.vision_rope_angles(1, 8, 6, 2, 40)
    assert np.array_equal(s, np.sin(ang).astype(np.float32)) or np.allclose(s, np.sin(ang), atol=1e-7)
    assert np.allclose(c, np.cos(ang), atol=1e-7)


def test_act_luts_match_oracle():
    g, q = lib.build_act_luts()
    go, qo = orc.gelu_tables()
    assert np.array_equal(g, go) and np.array_equal(q, qo)


def test_ops_module_refuses_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mllm_amd import ops
    with pytest.raises(lib.MllmHipError):
        ops.require_gpu()


def test_vision_rope_tables_equal_libm_per_element():
    """mllm_hip_vision_rope_table memoises sinf / cosf per (position, frequency) -- 640 libm calls for a 448 x 448 image instead of 82 k; every table entry still equals libm's
    sinf / cosf of the angle mllm_hip_vision_rope_angles gives for it (CPUVisionRoPEFunc.hpp:21-60 evaluates them per use), bit for bit, square and non-square grids."""
    import ctypes as C

    from mllm_amd import lib
    l = lib.load()
    libm = C.CDLL("libm.so.6")
    libm.sinf.restype = libm.cosf.restype = C.c_float
    libm.sinf.argtypes = libm.cosf.argtypes = [C.c_float]
    for (t, h, w, rd) in ((1, 32, 32, 40), (1, 6, 10, 40), (2, 4, 18, 16)):
        n = t * h * w * rd
        ang, s, c = (np.empty(n, dtype=np.float32) for _ in range(3))
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        assert l.mllm_hip_vision_rope_angles(t, h, w, 2, rd, p(ang)) == 0
        assert l.mllm_hip_vision_rope_table(t, h, w, 2, rd, p(s), p(c)) == 0
        idx = np.unique(np.concatenate([np.arange(0, n, 97), np.arange(n - 64, n)]))
        for i in idx:
            assert s[i] == np.float32(libm.sinf(float(ang[i]))) and c[i] == np.float32(libm.cosf(float(ang[i]))), (t, h, w, i)
