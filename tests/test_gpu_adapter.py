"""SURVEY §8(b), executed: the reference's OWN Qwen2VLModel -- its Module / Layer / Tensor frontend compiled from the reference tree, unchanged -- runs on the MI355X
through the Backend / Op adapter of integration/hip/ (`model.to(<hip slot>)` then `model.load(path)`, examples/demo_qwen.cpp:43-59) and must give the bits the same
model gave on the reference's x86 CPU backend (tests/golden/qwen2vl_tiny.npz): greedy ids and every logit of every step, image + text and text only.
The driver (oracle/ref_drivers/ref_hip_qwen2vl.cpp) is built in the container by oracle/Makefile.ref into oracle/_ref/ and travels to the GPU box as a binary.  It also
reports which Ops the backend refused, i.e. ran on the CPU backend instead: the bar is none."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_hip_qwen2vl")


def _cfg_string(c):
    return (f"{c.hidden},{c.inter},{c.layers},{c.heads},{c.kv_heads},{c.vocab},{c.v_dim},{c.cache_limit},{c.image_token_id},{c.vision_start_token_id},"
            f"{c.vision_end_token_id},{c.video_token_id},{int(c.tie_embedding)}")


def _run(td, cfg, path, ids, steps, pix=None, grid=None, engine=0, env=None):
    ids.astype(np.int32).tofile(os.path.join(td, "ids.i32"))
    cmd = [DRIVER, "--model", path, "--ids", os.path.join(td, "ids.i32"), "--steps", str(steps), "--threads", "4", "--out", td, "--cfg", _cfg_string(cfg), "--dump-every", "1",
           "--engine", str(engine)]
    if pix is not None:
        pix.astype(np.float32).tofile(os.path.join(td, "pix.f32"))
        cmd += ["--pix", os.path.join(td, "pix.f32"), "--grid", ",".join(str(int(g)) for g in grid)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, **env) if env else None)
    assert out.returncode == 0, (out.returncode, out.stdout[-2000:], out.stderr[-4000:])
    report = json.loads(next(l for l in out.stdout.splitlines() if l.startswith('{"backend"')))
    toks = np.fromfile(os.path.join(td, "tokens.i32"), dtype=np.int32)
    logits = np.stack([np.fromfile(os.path.join(td, f"logits_{s}.f32"), dtype=np.float32) for s in range(steps)])
    return report, toks, logits


@pytest.fixture(scope="module")
def tiny(tmp_path_factory):
    if not os.path.exists(DRIVER):
        pytest.skip("oracle/_ref/ref_hip_qwen2vl was not built (make -f oracle/Makefile.ref, container only)")
    from mllm_amd import synth
    from mllm_amd import synthfile as weights
    cfg = synth.qwen2vl_tiny()
    return cfg, weights.qwen2vl_file(cfg, cache_dir=str(tmp_path_factory.mktemp("w")))


def test_reference_module_on_the_hip_backend_matches_its_cpu_run(tiny, tiny_gold, tmp_path):
    from mllm_amd import synth
    cfg, path = tiny
    g = tiny_gold
    pix, grid, ids = synth.qwen2vl_inputs(cfg, (8, 8), 6)
    assert np.array_equal(ids, g["ids"])
    steps = len(g["tokens"])
    report, toks, logits = _run(str(tmp_path), cfg, path, ids, steps, pix, grid)
    print("adapter report:", report)
    assert report["cpu_fallback_ops"] == 0 and report["refused"] == [], report
    assert report["hip_ops_run"] > 600
    assert toks.tolist() == g["tokens"].tolist(), (toks.tolist(), g["tokens"].tolist())
    assert np.array_equal(logits, g["logits"]), float(np.max(np.abs(logits - g["logits"])))


def test_lazy_window_fuses_the_decode_layer_and_changes_nothing(tiny, tiny_gold, tmp_path):
    """The backend's lazy window (HIPBackend::lazy) hands a decode layer's 17 element-wise / one-row Ops to five fused launches: 8 steps x 2 layers x 5 launches
    standing for 17 Ops each on the toy model -- and the logits of every step equal the reference's with the window on (default) and off (MLLM_HIP_NO_FUSE=1: one launch per Op,
    the path every other test of this file also ran on before the window existed)."""
    from mllm_amd import synth
    cfg, path = tiny
    g = tiny_gold
    pix, grid, ids = synth.qwen2vl_inputs(cfg, (8, 8), 6)
    steps = 9
    a, ta, la = _run(str(tmp_path), cfg, path, ids, steps, pix, grid)
    b, tb, lb = _run(str(tmp_path), cfg, path, ids, steps, pix, grid, env={"MLLM_HIP_NO_FUSE": "1"})
    assert b["fused_launches"] == 0 and a["fused_launches"] >= (steps - 1) * cfg.layers * 5, (a, b)
    assert a["fused_ops"] >= (steps - 1) * cfg.layers * 17 and a["hip_ops_run"] == b["hip_ops_run"], (a, b)
    assert np.array_equal(la, g["logits"][:steps]) and np.array_equal(lb, g["logits"][:steps]) and ta.tolist() == tb.tolist() == g["tokens"][:steps].tolist()


def test_reference_module_text_only_prompt(tiny, tiny_gold, tmp_path):
    cfg, path = tiny
    g = tiny_gold
    steps = len(g["tokens_text"])
    report, toks, logits = _run(str(tmp_path), cfg, path, g["ids_text"], steps)
    assert report["cpu_fallback_ops"] == 0, report
    assert toks.tolist() == g["tokens_text"].tolist()
    assert np.array_equal(logits, g["logits_text"]), float(np.max(np.abs(logits - g["logits_text"])))


@pytest.mark.parametrize("engine", [0, 1], ids=["op-by-op", "engine-module"])
def test_reference_module_on_ragged_grids(tiny, tmp_path, engine):
    """Non-square grids off the kernels' tile sizes (6 x 10, 8 x 12, 4 x 18 patches; tests/golden/qwen2vl_ragged.npz): the reference's Module through the adapter, and the
    engine-backed Module, give the reference's CPU logits for 6 greedy steps; no Op falls back."""
    from mllm_amd import synth
    cfg, path = tiny
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "qwen2vl_ragged.npz"))
    for k in range(3):
        grid = g[f"grid{k}"]
        pix, _, ids = synth.qwen2vl_inputs(cfg, (int(grid[1]), int(grid[2])), int(g[f"ntext{k}"]))
        d = tmp_path / f"g{k}"
        d.mkdir()
        report, toks, logits = _run(str(d), cfg, path, ids, 6, pix, grid, engine=engine)
        assert report["cpu_fallback_ops"] == 0 and report["refused"] == [], report
        assert toks.tolist() == g[f"tokens{k}"].tolist() and np.array_equal(logits, g[f"logits{k}"]), k
    d = tmp_path / "one"                        # the shortest prompt the reference takes (two tokens)
    d.mkdir()
    report, toks, logits = _run(str(d), cfg, path, np.array([17, 23], dtype=np.int32), 5, engine=engine)
    assert report["cpu_fallback_ops"] == 0, report
    assert toks.tolist() == g["one_tokens"].tolist() and np.array_equal(logits, g["one_logits"])


@pytest.mark.parametrize("engine", [0, 1], ids=["op-by-op", "engine-module"])
def test_untied_lm_head_through_the_boundary(tiny, tiny_gold, tmp_path, engine):
    """config.tie_embedding_words = false (demo_qwen2_vl's larger presets): the reference's Qwen2VLModel takes its `lm_head` Linear instead of the tied embedding table
    (modeling_qwen2_vl.hpp:375-401); both the Op-by-Op adapter and the engine-backed Module (which once hard-wired the tied head) must follow the flag: every logit of
    8 steps equals the reference's CPU run on the untied file."""
    from mllm_amd import synth
    from mllm_amd import synthfile as weights
    g = tiny_gold
    cfg = synth.qwen2vl_tiny()
    cfg.tie_embedding = False
    path = weights.qwen2vl_file(cfg, cache_dir=str(tmp_path))
    pix, grid, ids = synth.qwen2vl_inputs(cfg, (8, 8), 6)
    report, toks, logits = _run(str(tmp_path), cfg, path, ids, len(g["tokens_untied"]), pix, grid, engine=engine)
    assert report["cpu_fallback_ops"] == 0 and report["refused"] == [], report
    assert toks.tolist() == g["tokens_untied"].tolist() and np.array_equal(logits, g["logits_untied"])


def test_reference_module_at_the_2b_geometry(tmp_path):
    """The same, at BASELINE's geometry: the reference's Qwen2VLModel (28 layers, hidden 1536, 32 vision blocks) on the Q4_K file through the adapter, 448 x 448 image + 24 tokens
    prefilled, 64 decode steps: ids equal the reference's CPU run (tests/golden/qwen2vl_2b_ref.npz) and so do its sampled logits (top 64 + every 97th) at steps 0, 16, 32, 48, 64;
    no Op falls back.  The driver's report carries the reference's own Module::profiling() numbers (21 ms TTFT, 466 tok/s: the frontend's per-Op host work, 630 Ops per token,
    is the time -- the resident engine behind the same C ABI does 1,118 tok/s)."""
    if not os.path.exists(DRIVER):
        pytest.skip("oracle/_ref/ref_hip_qwen2vl was not built (make -f oracle/Makefile.ref, container only)")
    from mllm_amd import synth
    from mllm_amd import synthfile as weights
    cfg = synth.qwen2vl_2b()
    path = weights.qwen2vl_file(cfg, cache_dir=os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache"))
    g = np.load(os.path.join(ROOT, "tests", "golden", "qwen2vl_2b_ref.npz"))
    pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
    steps = len(g["tokens"])
    td = str(tmp_path)
    ids.astype(np.int32).tofile(os.path.join(td, "ids.i32"))
    pix.astype(np.float32).tofile(os.path.join(td, "pix.f32"))
    cmd = [DRIVER, "--model", path, "--ids", os.path.join(td, "ids.i32"), "--steps", str(steps), "--threads", "4", "--out", td, "--cfg", _cfg_string(cfg), "--dump-every", "16",
           "--pix", os.path.join(td, "pix.f32"), "--grid", ",".join(str(int(x)) for x in grid)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.returncode, out.stdout[-2000:], out.stderr[-4000:])
    report = json.loads(next(l for l in out.stdout.splitlines() if l.startswith('{"backend"')))
    print("adapter report (2B):", report)
    assert report["cpu_fallback_ops"] == 0 and report["refused"] == [], report
    toks = np.fromfile(os.path.join(td, "tokens.i32"), dtype=np.int32)
    assert toks.tolist() == g["tokens"].tolist()
    for i, s in enumerate(g["steps"]):
        lg = np.fromfile(os.path.join(td, f"logits_{int(s)}.f32"), dtype=np.float32)
        assert np.array_equal(lg[g["top_idx"][i]], g["top_val"][i]) and np.array_equal(lg[::97], g["strided"][i]), int(s)


def test_engine_module_behind_the_reference_frontend(tiny, tiny_gold, tmp_path):
    """integration/hip/HIPQwen2VLEngine.hpp: a mllm::Module with Qwen2VLModel's calling convention whose Forward is the resident engine (mllm_hip_model_*), driven by the same
    demo loop (model(input) -> host argmax -> chatPostProcessing): ids and every logit of every step equal the reference's CPU run, image + text and text only; then the 2B
    geometry: ids and sampled logits, and the reference's own profiling() of that run."""
    from mllm_amd import synth
    from mllm_amd import synthfile as weights
    cfg, path = tiny
    g = tiny_gold
    pix, grid, ids = synth.qwen2vl_inputs(cfg, (8, 8), 6)
    report, toks, logits = _run(str(tmp_path), cfg, path, ids, len(g["tokens"]), pix, grid, engine=1)
    assert toks.tolist() == g["tokens"].tolist() and np.array_equal(logits, g["logits"])
    report, toks, logits = _run(str(tmp_path), cfg, path, g["ids_text"], len(g["tokens_text"]), engine=1)
    assert toks.tolist() == g["tokens_text"].tolist() and np.array_equal(logits, g["logits_text"])
    big = synth.qwen2vl_2b()
    bpath = weights.qwen2vl_file(big, cache_dir=os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache"))
    gb = np.load(os.path.join(ROOT, "tests", "golden", "qwen2vl_2b_ref.npz"))
    pix, grid, ids = synth.qwen2vl_inputs(big, (32, 32), 24)
    steps = len(gb["tokens"])
    td = str(tmp_path)
    ids.astype(np.int32).tofile(os.path.join(td, "ids.i32"))
    pix.astype(np.float32).tofile(os.path.join(td, "pix.f32"))
    cmd = [DRIVER, "--model", bpath, "--ids", os.path.join(td, "ids.i32"), "--steps", str(steps), "--threads", "4", "--out", td, "--cfg", _cfg_string(big), "--dump-every", "16",
           "--pix", os.path.join(td, "pix.f32"), "--grid", ",".join(str(int(x)) for x in grid), "--engine", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.returncode, out.stdout[-2000:], out.stderr[-4000:])
    report = json.loads(next(l for l in out.stdout.splitlines() if l.startswith('{"backend"')))
    print("engine Module report (2B):", report)
    toks = np.fromfile(os.path.join(td, "tokens.i32"), dtype=np.int32)
    assert toks.tolist() == gb["tokens"].tolist()
    for i, s_ in enumerate(gb["steps"]):
        lg = np.fromfile(os.path.join(td, f"logits_{int(s_)}.f32"), dtype=np.float32)
        assert np.array_equal(lg[gb["top_idx"][i]], gb["top_val"][i]) and np.array_equal(lg[::97], gb["strided"][i]), int(s_)
