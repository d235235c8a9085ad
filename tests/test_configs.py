"""The other BASELINE.json configs as parity cases: demo_tinyllama (fp32 and Q4_K weights), demo_qwen
(Qwen1.5 geometry, Q4_K + tied Q4_0 lm_head) and demo_vit (ViT-B/16 geometry) -- golden outputs of the reference's own models
(oracle/_ref/ref_llm, ref_vit via oracle/make_golden.py --configs) on the same synthetic .mllm files.

CPU part: the oracle's composition of each graph (oracle/models.py) reproduces the reference bit for bit.
GPU part: the host graphs over the C-ABI launchers (mllm_amd/graphs.py) reproduce it bit for bit, ids and every logit."""
import os

import numpy as np
import pytest

from mllm_amd import mllmfile as mf, synth
from mllm_amd import synthfile as weights

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CACHE = os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache")

LM_CASES = [("qwen", synth.qwen15_tiny), ("tl", synth.tinyllama_tiny), ("tlq", lambda: synth.tinyllama_tiny(mf.Q4_K))]


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "configs_tiny.npz"))


@pytest.mark.parametrize("key,mk", LM_CASES, ids=[c[0] for c in LM_CASES])
def test_oracle_causal_lm_matches_reference(gold, key, mk):
    from oracle import models as om
    cfg = mk()
    assert np.array_equal(synth.causal_lm_ids(cfg, 20), gold[key + "_ids"])
    m = om.CausalLM(om.Weights(weights.causal_lm_file(cfg, CACHE)), cfg)
    cur = gold[key + "_ids"]
    for s, ref in enumerate(gold[key + "_logits"]):
        lg = m.forward(cur)
        assert np.array_equal(lg, ref), (s, float(np.abs(lg - ref).max()))
        assert int(np.argmax(lg)) == int(gold[key + "_tokens"][s])
        cur = [int(np.argmax(lg))]


def test_oracle_vit_matches_reference(gold):
    from oracle import models as om
    cfg = synth.vit_tiny()
    w = om.Weights(weights.vit_file(cfg, CACHE))
    for img, ref in zip(synth.vit_images(cfg, 3), gold["vit_logits"]):
        out = om.vit_forward(w, cfg, img)
        assert np.array_equal(out, ref), float(np.abs(out - ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("key,mk", LM_CASES, ids=[c[0] for c in LM_CASES])
def test_gpu_causal_lm_matches_reference(gold, key, mk):
    from mllm_amd import graphs
    cfg = mk()
    m = graphs.CausalLM(cfg, weights.causal_lm_file(cfg, CACHE))
    toks, logits = m.greedy(gold[key + "_ids"], len(gold[key + "_tokens"]))
    assert toks == gold[key + "_tokens"].tolist()
    for s, (lg, ref) in enumerate(zip(logits, gold[key + "_logits"])):
        assert np.array_equal(lg, ref), (s, float(np.abs(lg - ref).max()))
    # clear_kvcache + the same prompt again: bitwise idempotent
    m.clear_kvcache()
    assert np.array_equal(m.forward(gold[key + "_ids"]), logits[0])


@pytest.mark.gpu
def test_gpu_vit_matches_reference(gold):
    from mllm_amd import graphs
    cfg = synth.vit_tiny()
    m = graphs.ViT(cfg, weights.vit_file(cfg, CACHE))
    out = m.forward_batch(synth.vit_images(cfg, 3)).cpu().numpy()
    assert np.array_equal(out, gold["vit_logits"]), float(np.abs(out - gold["vit_logits"]).max())


# ---- the real sizes: Qwen1.5-0.5B (Q4_K, tied Q4_0 head, 151,936-way logits) and ViT-B/16 ---------------------------------------------------

@pytest.fixture(scope="module")
def gold_full():
    return np.load(os.path.join(GOLD, "configs_full.npz"))


def _sample_err(g, s, lg):
    return float(max(np.abs(lg[g["qwen_top_idx"][s]] - g["qwen_top_val"][s]).max(), np.abs(lg[::97] - g["qwen_strided"][s]).max()))


def test_oracle_full_size_qwen15_and_vitb_match_reference(gold_full):
    from oracle import models as om
    g = gold_full
    cfg = synth.qwen15_05b()
    m = om.CausalLM(om.Weights(weights.causal_lm_file(cfg, CACHE)), cfg)
    cur = g["qwen_ids"]
    for s in range(3):
        lg = m.forward(cur)
        assert _sample_err(g, s, lg) == 0.0 and int(np.argmax(lg)) == int(g["qwen_tokens"][s])
        cur = [int(np.argmax(lg))]
    vc = synth.vit_b16()
    out = om.vit_forward(om.Weights(weights.vit_file(vc, CACHE)), vc, synth.vit_images(vc, 2)[0])
    assert np.array_equal(out, g["vit_logits"][0]), float(np.abs(out - g["vit_logits"][0]).max())


@pytest.mark.gpu
def test_gpu_full_size_qwen15_matches_reference(gold_full):
    from mllm_amd import graphs
    g = gold_full
    cfg = synth.qwen15_05b()
    m = graphs.CausalLM(cfg, weights.causal_lm_file(cfg, CACHE))
    toks, logits = m.greedy(g["qwen_ids"], len(g["qwen_tokens"]))
    assert toks == g["qwen_tokens"].tolist()
    assert max(_sample_err(g, s, lg) for s, lg in enumerate(logits)) == 0.0


@pytest.mark.gpu
def test_gpu_full_size_vitb_matches_reference(gold_full):
    from mllm_amd import graphs
    cfg = synth.vit_b16()
    m = graphs.ViT(cfg, weights.vit_file(cfg, CACHE))
    out = m.forward_batch(synth.vit_images(cfg, 2)).cpu().numpy()
    assert np.array_equal(out, gold_full["vit_logits"]), float(np.abs(out - gold_full["vit_logits"]).max())


# ---- config 5 (demo_llava): the reference's LLaVAModel segfaults at this snapshot (Tensor::range, see oracle/ref_drivers/ref_llava.cpp), so
# ---- there is no whole-graph golden; the GPU graph is held to the oracle's composition, whose every op is pinned by the goldens above ----------

@pytest.mark.gpu
def test_gpu_llava_tiny_matches_oracle_composition():
    from mllm_amd import graphs
    from oracle import models as om
    cfg = synth.llava_tiny()
    path = weights.llava_file(cfg, CACHE)
    ids, img = synth.llava_inputs(cfg)
    ref = om.LLaVA(om.Weights(path), cfg)
    m = graphs.LLaVA(cfg, path)
    want, got = ref.forward(ids, img), m.forward(ids, img)
    assert np.array_equal(got, want), float(np.abs(got - want).max())
    for _ in range(4):
        tok = int(np.argmax(want))
        want, got = ref.forward([tok]), m.forward([tok])
        assert np.array_equal(got, want), float(np.abs(got - want).max())
    # the tower's rows handed in (the form the image shard uses) give the same logits as running it in place
    m.clear_kvcache()
    assert np.array_equal(m.forward(ids, vis=m.vision(img)), om.LLaVA(om.Weights(path), cfg).forward(ids, img))
