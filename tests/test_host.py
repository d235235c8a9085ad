"""CPU-side tests of the product's host logic: the C-ABI library loads and exports every symbol include/mllm_hip.h
declares, the .mllm container round-trips, the host quantisers are byte-identical to the reference `quantize` tool (pinned by
digests captured from the reference, tests/golden/qwen2vl_tiny_q4k_digests.json), and the host-side table builders agree
with the oracle.  No GPU, no compute launch."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

from mllm_amd import lib, mllmfile as mf, synth
from tests.fixtures import quantlib, weights
from oracle import oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_library_exports_every_declared_symbol():
    so = lib.load()
    names = lib.declared_symbols()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(so, n)]
    assert not missing, missing


def test_quantized_nbytes_and_errors():
    assert quantlib.nbytes(quantlib.Q4_K, 512) == 288
    assert quantlib.nbytes(quantlib.Q4_0, 64) == 36
    assert quantlib.nbytes(quantlib.Q4_K, 100) == -1
    with pytest.raises(ValueError):
        quantlib.quantize(lib.Q4_K, np.zeros(100, dtype=np.float32))


def test_product_library_does_not_contain_the_fixture_quantiser():
    so = lib.load()
    assert not hasattr(so, "mllm_hip_quantize_host") and not hasattr(so, "mllm_quant_rows")
    # ... and nothing under mllm_amd/ imports the fixture tooling (it lives under tests/fixtures/)
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mllm_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert "quantlib" not in src and "quantize_host" not in src and "tests.fixtures" not in src, f


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


def test_host_quantizer_matches_reference_tool_digests(tmp_path):
    """Builds the tiny Qwen2-VL file with the product's quantiser and compares per-tensor sha256 with the digests of the file
    the reference's own `quantize ... Q4_K` produced from the same fp32 tensors."""
    want = json.load(open(os.path.join(GOLD, "qwen2vl_tiny_q4k_digests.json")))
    path = weights.qwen2vl_file(synth.qwen2vl_tiny(), cache_dir=str(tmp_path))
    got = weights.tensor_digests(path)
    assert set(got) == set(want)
    bad = [n for n in want if got[n] != want[n]]
    assert not bad, bad[:5]


def test_host_quantizer_full_size_digests():
    """The same at the benchmark's size: the 729 tensors of the Qwen2-VL-2B shaped synthetic model (2.2 B parameters) written by the fixture quantiser
    against the digests of the file the reference's `quantize ... Q4_K` wrote from the same fp32 tensors (that file also holds an untied lm_head the
    tied config never reads).  Uses the cached file when bench.py / the GPU tests have built it already; about a minute on 8 cores otherwise."""
    want = json.load(open(os.path.join(GOLD, "qwen2vl_2b_q4k_digests.json")))
    path = weights.qwen2vl_file(synth.qwen2vl_2b(), cache_dir=os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache"))
    got = weights.tensor_digests(path)
    assert len(got) == 729 and set(got) <= set(want)
    bad = [n for n in got if got[n] != want[n]]
    assert not bad, bad[:5]


def test_q80_host_quantizer_vs_oracle():
    x = np.random.default_rng(3).standard_normal(32 * 40).astype(np.float32)
    assert np.array_equal(quantlib.quantize(lib.Q8_0, x), orc.quantize_q8_0(x).ravel())


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
