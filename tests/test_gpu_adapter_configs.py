"""SURVEY §8(b) for the OTHER four BASELINE configs: the reference's own Module graphs -- QWenForCausalLM (models/qwen/modeling_qwen.hpp:131-179), TinyLLaMAModel
(models/tinyllama/modeling_tinyllama.hpp:44-84), ViTModel (models/vit/modeling_vit.hpp:63-111) and the LLaVA graph (models/llava/modeling_llava.hpp:39-137, composed from
the reference's own modules because LLaVAModel itself cannot be loaded at this snapshot: see oracle/ref_drivers/ref_hip_llava.cpp) -- compiled from the reference tree,
unchanged, run on the MI355X through the Backend / Op adapter of integration/hip/ (`model.to(<hip slot>)` then `model.load(path)`, examples/demo_qwen.cpp:43-59).
Each must give the bits the same model gave on the reference's x86 CPU backend (tests/golden/configs_tiny.npz, llava_tiny.npz: every logit; configs_full.npz,
tinyllama_11b.npz, llava_7b.npz: greedy ids + the top-64 and every 97th logit of each step), with NO Op refused by the backend (= none run on the CPU backend).
The drivers (oracle/ref_drivers/ref_hip_{llm,vit,llava}.cpp) are built in the container by oracle/Makefile.ref and travel to the GPU box as binaries."""
import json
import os
import subprocess

import numpy as np
import pytest

from mllm_amd import mllmfile as mf, synth
from mllm_amd import synthfile as weights

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")
CACHE = os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache")


def _driver(name):
    exe = os.path.join(REF, name)
    if not os.path.exists(exe):
        pytest.skip(f"oracle/_ref/{name} was not built (make -f oracle/Makefile.ref, container only)")
    return exe


def _run(cmd, timeout=900):
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, (out.returncode, out.stdout[-2000:], out.stderr[-4000:])
    report = json.loads(next(l for l in out.stdout.splitlines() if l.startswith('{"backend"')))
    print(os.path.basename(cmd[0]), "report:", report)
    assert report["cpu_fallback_ops"] == 0 and report["refused"] == [], report
    return report


def _sample_err(g, s, lg, key=""):
    return float(max(np.abs(lg[g[key + "top_idx"][s]] - g[key + "top_val"][s]).max(), np.abs(lg[::97] - g[key + "strided"][s]).max()))


def _llm(td, c, family, ids, steps, path):
    ids.astype(np.int32).tofile(os.path.join(td, "ids.i32"))
    cfg = f"{c.hidden},{c.inter},{c.layers},{c.heads},{c.kv_heads},{c.vocab},{c.cache_limit},{int(c.tie_embedding)}"
    rep = _run([_driver("ref_hip_llm"), "--family", family, "--model", path, "--ids", os.path.join(td, "ids.i32"), "--steps", str(steps), "--threads", "4", "--out", td, "--cfg", cfg])
    toks = np.fromfile(os.path.join(td, "tokens.i32"), dtype=np.int32)
    logits = [np.fromfile(os.path.join(td, f"logits_{s}.f32"), dtype=np.float32) for s in range(steps)]
    return rep, toks, logits


@pytest.mark.parametrize("key,family,mk", [("qwen", "qwen", synth.qwen15_tiny), ("tlq", "tinyllama", lambda: synth.tinyllama_tiny(mf.Q4_K))], ids=["qwen", "tinyllama"])
def test_reference_causal_lm_modules_tiny(key, family, mk, tmp_path):
    """QWenForCausalLM (tied Q4_0 head through PARAMETER + F_TRANPOSE + F_MM, q/k/v bias) and TinyLLaMAModel (GQA 4/2, Linear head over ALL prompt rows: no last-token clip,
    SURVEY Q7) at toy shapes: every logit of 8 steps."""
    g = np.load(os.path.join(GOLD, "configs_tiny.npz"))
    c = mk()
    rep, toks, logits = _llm(str(tmp_path), c, family, g[key + "_ids"], len(g[key + "_tokens"]), weights.causal_lm_file(c, CACHE))
    assert rep["hip_ops_run"] > 100
    assert toks.tolist() == g[key + "_tokens"].tolist()
    for s, (lg, ref) in enumerate(zip(logits, g[key + "_logits"])):
        assert np.array_equal(lg, ref), (s, float(np.abs(lg - ref).max()))


def test_reference_qwen15_05b_module_full_size(tmp_path):
    """BASELINE config 2 at its real size (24 x 1024, 16 heads x 64, vocab 151,936 tied): 32-token prompt + 16 steps."""
    g = np.load(os.path.join(GOLD, "configs_full.npz"))
    c = synth.qwen15_05b()
    rep, toks, logits = _llm(str(tmp_path), c, "qwen", g["qwen_ids"], len(g["qwen_tokens"]), weights.causal_lm_file(c, CACHE))
    assert toks.tolist() == g["qwen_tokens"].tolist()
    assert max(_sample_err(g, s, lg, "qwen_") for s, lg in enumerate(logits)) == 0.0


def test_reference_tinyllama_11b_module_full_size(tmp_path):
    """BASELINE config 1's geometry (22 x 2048, 32 / 4 heads, untied 32000-row head), Q4_K: 24-token prompt + 12 steps."""
    g = np.load(os.path.join(GOLD, "tinyllama_11b.npz"))
    c = synth.tinyllama_11b(target=mf.Q4_K)
    rep, toks, logits = _llm(str(tmp_path), c, "tinyllama", g["ids"], len(g["tokens"]), weights.causal_lm_file(c, CACHE))
    assert toks.tolist() == g["tokens"].tolist()
    assert max(_sample_err(g, s, lg) for s, lg in enumerate(logits)) == 0.0


def _vit(td, c, n, path):
    synth.vit_images(c, n).tofile(os.path.join(td, "img.f32"))
    cfg = f"{c.hidden},{c.heads},{c.ffn},{c.blocks},{c.patch},{c.img},{c.classes}"
    rep = _run([_driver("ref_hip_vit"), "--model", path, "--img", os.path.join(td, "img.f32"), "--n", str(n), "--threads", "4", "--out", td, "--cfg", cfg])
    return rep, np.fromfile(os.path.join(td, "vit_logits.f32"), dtype=np.float32).reshape(n, c.classes)


def test_reference_vit_module_tiny_and_b16(tmp_path):
    """ViTModel: Convolution2D patch embedding -> F_TRANPOSE {(S,D),(H,S)} -> F_FLATTEN -> F_CAT with the class token -> + position embeddings -> blocks (LayerNorm,
    MultiHeadAttention without cache: fp32 K / V FlashAttention2, GELU MLP) -> clip({0}) -> LayerNorm -> classifier.  Toy shape (3 images) and ViT-B/16 at 224 x 224
    (2 images): every class logit."""
    g = np.load(os.path.join(GOLD, "configs_tiny.npz"))
    c = synth.vit_tiny()
    rep, got = _vit(str(tmp_path), c, 3, weights.vit_file(c, CACHE))
    assert np.array_equal(got, g["vit_logits"]), float(np.abs(got - g["vit_logits"]).max())
    gf = np.load(os.path.join(GOLD, "configs_full.npz"))
    c = synth.vit_b16()
    rep, got = _vit(str(tmp_path), c, 2, weights.vit_file(c, CACHE))
    assert np.array_equal(got, gf["vit_logits"]), float(np.abs(got - gf["vit_logits"]).max())


def _llava(td, c, steps, path, dump_vision=False):
    ids, img = synth.llava_inputs(c)
    ids.tofile(os.path.join(td, "ids.i32"))
    img.tofile(os.path.join(td, "img.f32"))
    cfg = f"{c.hidden},{c.heads},{c.inter},{c.layers},{c.vocab},{c.cache_limit},{c.v_hidden},{c.v_heads},{c.v_ffn},{c.v_blocks},{c.patch},{c.img}"
    base = [_driver("ref_hip_llava"), "--model", path, "--ids", os.path.join(td, "ids.i32"), "--img", os.path.join(td, "img.f32"), "--threads", "4", "--out", td, "--cfg", cfg]
    if dump_vision:
        _run(base + ["--dump-vision", "1"])
        return np.fromfile(os.path.join(td, "vision.f32"), dtype=np.float32).reshape(c.v_tokens, c.v_ffn)
    rep = _run(base + ["--steps", str(steps)])
    toks = np.fromfile(os.path.join(td, "tokens.i32"), dtype=np.int32)
    return ids, toks, [np.fromfile(os.path.join(td, f"logits_{s}.f32"), dtype=np.float32) for s in range(steps)]


def test_reference_llava_graph_tiny(tmp_path):
    """The LLaVA graph at a toy shape with the real head geometry: the projected visual rows (CLIP tower with an fp32 position EMBEDDING, class row clipped, projector)
    and every logit of 6 steps (F_WHERE + F_INDEX_PUT(accumulate) splice, LLaMA body)."""
    g = np.load(os.path.join(GOLD, "llava_tiny.npz"))
    c = synth.llava_tiny()
    path = weights.llava_file(c, CACHE)
    vis = _llava(str(tmp_path), c, 0, path, dump_vision=True)
    assert np.array_equal(vis, g["vision"]), float(np.abs(vis - g["vision"]).max())
    ids, toks, logits = _llava(str(tmp_path), c, len(g["tokens"]), path)
    assert np.array_equal(ids, g["ids"]) and toks.tolist() == g["tokens"].tolist()
    for s, (lg, ref) in enumerate(zip(logits, g["logits"])):
        assert np.array_equal(lg, ref), (s, float(np.abs(lg - ref).max()))


def test_reference_llava_graph_7b_geometry(tmp_path):
    """BASELINE config 5 at its real geometry: LLaMA-7B body + CLIP-ViT-L/14-336, S = 589 prefill + 5 decode steps: greedy ids, sampled logits and every 61st projected
    visual row of the reference's run."""
    g = np.load(os.path.join(GOLD, "llava_7b.npz"))
    c = synth.llava_7b()
    path = weights.llava_file(c, CACHE)
    vis = _llava(str(tmp_path), c, 0, path, dump_vision=True)
    assert np.array_equal(vis[::61], g["vision_rows"]), float(np.abs(vis[::61] - g["vision_rows"]).max())
    ids, toks, logits = _llava(str(tmp_path), c, len(g["tokens"]), path)
    assert toks.tolist() == g["tokens"].tolist(), (toks.tolist(), g["tokens"].tolist())
    assert max(_sample_err(g, s, lg) for s, lg in enumerate(logits)) == 0.0
