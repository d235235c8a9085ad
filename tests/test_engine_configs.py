"""Every BASELINE.json config on the resident C++ engine (csrc/engine.hip, mllm_hip_model_*) against goldens of the reference's own models:
configs 2 (Qwen1.5), 1 with Q4_K weights (TinyLlama geometry), 3 (ViT-B/16), 5 (LLaVA-1.5-7B + CLIP-L/336) -- toy shapes with every logit, and the
real geometries with greedy ids + sampled logits.  Config 4 (Qwen2-VL) has its own files (test_gpu_e2e.py, test_gpu_full.py).

The LLaVA goldens come from oracle/ref_drivers/ref_llava_parts.cpp: the reference's LLaVAModel graph composed from the reference's own modules with the
position ids fed in as a tensor (LLaVAModel itself crashes in Tensor::range at this snapshot, mllm/Op.hpp:61-68)."""
import os

import numpy as np
import pytest

from mllm_amd import mllmfile as mf, synth
from mllm_amd import synthfile as weights

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CACHE = os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache")


def _sample_err(g, s, lg, key=""):
    return float(max(np.abs(lg[g[key + "top_idx"][s]] - g[key + "top_val"][s]).max(), np.abs(lg[::97] - g[key + "strided"][s]).max()))


# ---- CPU: the oracle's LLaVA composition against the reference run (tiny shape, every logit and every projected visual row) --------------------------

def test_oracle_llava_matches_reference_tiny():
    from oracle import models as om
    g = np.load(os.path.join(GOLD, "llava_tiny.npz"))
    cfg = synth.llava_tiny()
    ids, img = synth.llava_inputs(cfg)
    assert np.array_equal(ids, g["ids"])
    m = om.LLaVA(om.Weights(weights.llava_file(cfg, CACHE)), cfg)
    lg = m.forward(ids, img)
    for s, ref in enumerate(g["logits"]):
        assert np.array_equal(lg, ref), (s, float(np.abs(lg - ref).max()))
        assert int(np.argmax(lg)) == int(g["tokens"][s])
        lg = m.forward([int(np.argmax(lg))])


def test_model_config_mapping():
    from mllm_amd import lib
    c = lib.model_config(synth.llava_7b())
    assert (c.arch, c.hidden, c.inter, c.heads, c.kv_heads, c.vocab, c.v_dim, c.v_heads, c.v_ffn, c.v_img) == (lib.ARCH_LLAVA, 4096, 11008, 32, 32, 32064, 1024, 16, 4096, 336)
    c = lib.model_config(synth.qwen15_05b())
    assert (c.arch, c.tie_embedding, c.qkv_bias) == (lib.ARCH_QWEN, 1, 1)
    c = lib.model_config(synth.tinyllama_11b(mf.Q4_K))
    assert (c.arch, c.tie_embedding, c.qkv_bias, c.kv_heads) == (lib.ARCH_LLAMA, 0, 0, 4)
    c = lib.model_config(synth.vit_b16())
    assert (c.arch, c.v_dim, c.v_classes, c.hidden) == (lib.ARCH_VIT, 768, 1000, 0)


# ---- GPU: the engine ------------------------------------------------------------------------------------------------------------------------------------

LM_CASES = [("qwen", synth.qwen15_tiny), ("tlq", lambda: synth.tinyllama_tiny(mf.Q4_K))]


@pytest.mark.gpu
@pytest.mark.parametrize("key,mk", LM_CASES, ids=[c[0] for c in LM_CASES])
def test_engine_causal_lm_tiny_matches_reference(key, mk):
    from mllm_amd import lib
    gold = np.load(os.path.join(GOLD, "configs_tiny.npz"))
    cfg = mk()
    m = lib.Model(cfg, weights.causal_lm_file(cfg, CACHE))
    toks, logits = m.greedy(gold[key + "_ids"], len(gold[key + "_tokens"]))
    assert toks == gold[key + "_tokens"].tolist()
    for s, (lg, ref) in enumerate(zip(logits, gold[key + "_logits"])):
        assert np.array_equal(lg, ref), (s, float(np.abs(lg - ref).max()))
    # clear_kvcache + the same prompt: bitwise idempotent; generate() (device argmax, captured graph) == the stepwise loop
    m.clear_kvcache()
    tok, lg, _ = m.prefill(gold[key + "_ids"])
    assert np.array_equal(lg, logits[0]) and tok == toks[0]
    gen, _ = m.generate(tok, len(toks) - 1)
    assert gen.tolist() == toks[1:]
    m.close()


@pytest.mark.gpu
def test_engine_vit_tiny_matches_reference():
    import torch
    from mllm_amd import lib
    gold = np.load(os.path.join(GOLD, "configs_tiny.npz"))
    cfg = synth.vit_tiny()
    m = lib.Model(cfg, weights.vit_file(cfg, CACHE))
    assert m.vision_shape() == (1, cfg.classes)
    imgs = synth.vit_images(cfg, 3)
    out = torch.empty((3, cfg.classes), dtype=torch.float32, device="cuda")
    m.vision(imgs, None, out.data_ptr(), 3)
    got = out.cpu().numpy()
    assert np.array_equal(got, gold["vit_logits"]), float(np.abs(got - gold["vit_logits"]).max())
    m.close()


@pytest.mark.gpu
def test_engine_llava_tiny_matches_reference():
    import torch
    from mllm_amd import lib
    g = np.load(os.path.join(GOLD, "llava_tiny.npz"))
    cfg = synth.llava_tiny()
    ids, img = synth.llava_inputs(cfg)
    m = lib.Model(cfg, weights.llava_file(cfg, CACHE))
    rows, cols = m.vision_shape()
    assert (rows, cols) == g["vision"].shape
    vis = torch.empty((rows, cols), dtype=torch.float32, device="cuda")
    m.vision(img, None, vis.data_ptr(), 1)
    assert np.array_equal(vis.cpu().numpy(), g["vision"])
    toks, logits = m.greedy(ids, len(g["tokens"]), image=img)
    assert toks == g["tokens"].tolist()
    for s, (lg, ref) in enumerate(zip(logits, g["logits"])):
        assert np.array_equal(lg, ref), (s, float(np.abs(lg - ref).max()))
    # the tower's rows handed in on the device (the form the image shard uses after its all-gather) give the same logits
    m.clear_kvcache()
    tok, lg, _ = m.prefill(ids, visual_dev=vis.data_ptr(), n_visual_rows=rows)
    assert np.array_equal(lg, g["logits"][0])
    m.close()


@pytest.mark.gpu
def test_engine_full_size_qwen15_matches_reference():
    from mllm_amd import lib
    g = np.load(os.path.join(GOLD, "configs_full.npz"))
    cfg = synth.qwen15_05b()
    m = lib.Model(cfg, weights.causal_lm_file(cfg, CACHE))
    toks, logits = m.greedy(g["qwen_ids"], len(g["qwen_tokens"]))
    assert toks == g["qwen_tokens"].tolist()
    assert max(_sample_err(g, s, lg, "qwen_") for s, lg in enumerate(logits)) == 0.0
    m.clear_kvcache()
    tok, _, _ = m.prefill(g["qwen_ids"], want_logits=False)
    gen, _ = m.generate(tok, len(toks) - 1)
    assert gen.tolist() == toks[1:]
    # the step launch by launch (bench.py's step_launches): same ids, and this shape's form of the step -- q|k|v + attention + o-projection as one launch from layer 1 on
    # (11 super-blocks per down row: the chain launch does not apply), gate|up and down on their own
    m.clear_kvcache()
    tok, _, _ = m.prefill(g["qwen_ids"], want_logits=False)
    kinds, last = m.time_step(tok, len(toks) - 1)
    assert last == toks[-1]
    assert {k: n for k, (_, n) in kinds.items()} == {"front": 23, "gateup": 24, "down": 24, "qkv": 1, "attn": 1, "head": 1, "next": 1}
    st = m.load_stats()
    assert st["file_bytes"] > 2.0e8 and st["total_ms"] > 0
    m.close()


@pytest.mark.gpu
def test_engine_full_size_tinyllama_matches_reference():
    """BASELINE config 2 at its real geometry (22 x 2048, 32 / 4 heads, untied Q4_K head of 32000 rows): 24-token prompt + 12 greedy steps, bit-identical to the reference's run
    (tests/golden/tinyllama_11b.npz, oracle/make_golden.py --tinyllama-full)."""
    from mllm_amd import lib
    g = np.load(os.path.join(GOLD, "tinyllama_11b.npz"))
    cfg = synth.tinyllama_11b(target=mf.Q4_K)
    m = lib.Model(cfg, weights.causal_lm_file(cfg, CACHE))
    toks, logits = m.greedy(g["ids"], len(g["tokens"]))
    assert toks == g["tokens"].tolist()
    assert max(_sample_err(g, s, lg) for s, lg in enumerate(logits)) == 0.0
    m.clear_kvcache()
    tok, _, _ = m.prefill(g["ids"], want_logits=False)
    gen, _ = m.generate(tok, len(toks) - 1)          # the captured decode graph gives the same ids
    assert gen.tolist() == toks[1:]
    # the step launch by launch: the chain launch applies here too (22 super-blocks per down row), D = 64 heads, the q|k|v role carried on by the down projection's workgroups
    m.clear_kvcache()
    tok, _, _ = m.prefill(g["ids"], want_logits=False)
    kinds, last = m.time_step(tok, len(toks) - 1)
    assert last == toks[-1]
    assert {k: n for k, (_, n) in kinds.items()} == {"chain": 21, "gateup": 22, "down": 1, "qkv": 1, "attn": 1, "head": 1, "next": 1}
    m.close()


@pytest.mark.gpu
def test_engine_full_size_vitb_matches_reference():
    import torch
    from mllm_amd import lib
    g = np.load(os.path.join(GOLD, "configs_full.npz"))
    cfg = synth.vit_b16()
    m = lib.Model(cfg, weights.vit_file(cfg, CACHE))
    out = torch.empty((2, cfg.classes), dtype=torch.float32, device="cuda")
    m.vision(synth.vit_images(cfg, 2), None, out.data_ptr(), 2)
    got = out.cpu().numpy()
    assert np.array_equal(got, g["vit_logits"]), float(np.abs(got - g["vit_logits"]).max())
    m.close()


@pytest.mark.gpu
def test_engine_llava_7b_geometry_matches_reference():
    """BASELINE config 5 at its real geometry: LLaMA-7B body (4096 / 11008 / 32 x 128, 32 layers, Linear lm_head 32064 x 4096) + CLIP-ViT-L/14-336
    (1024 / 4096 / 16 x 64, 23 blocks, 577 tokens) on synthetic Q4_K weights; S = 589 prefill + 5 decode steps, greedy ids and sampled logits of the
    reference's run (tests/golden/llava_7b.npz, oracle/make_golden.py --llava-full)."""
    import torch
    from mllm_amd import lib
    g = np.load(os.path.join(GOLD, "llava_7b.npz"))
    cfg = synth.llava_7b()
    ids, img = synth.llava_inputs(cfg)
    assert np.array_equal(ids, g["ids"])
    m = lib.Model(cfg, weights.llava_file(cfg, CACHE))
    rows, cols = m.vision_shape()
    assert (rows, cols) == (576, 4096)
    vis = torch.empty((rows, cols), dtype=torch.float32, device="cuda")
    m.vision(img, None, vis.data_ptr(), 1)
    got = vis.cpu().numpy()[::61]
    assert np.array_equal(got, g["vision_rows"]), float(np.abs(got - g["vision_rows"]).max())
    toks, logits = m.greedy(ids, len(g["tokens"]), image=img)
    assert toks == g["tokens"].tolist(), (toks, g["tokens"].tolist())
    assert max(_sample_err(g, s, lg) for s, lg in enumerate(logits)) == 0.0
    m.close()


@pytest.mark.gpu
def test_vision_batches_give_the_single_image_rows(monkeypatch):
    """mllm_hip_model_vision walks the images several at a time (row-wise kernels once over all their token rows, rotary and attention per image): a batch must give,
    bit for bit, the rows each image gives alone -- for the CLIP tower + projector (5 images: one full group of 4 rows-budget permitting, plus a ragged tail when the
    group size is forced to 2) and for the Qwen2-VL tower with its 2-D rotary and patch merger."""
    import torch
    from mllm_amd import lib
    r = np.random.default_rng(3)
    cfg = synth.llava_tiny()
    m = lib.Model(cfg, weights.llava_file(cfg, CACHE))
    rows, cols = m.vision_shape()
    imgs = r.standard_normal((5, cfg.img, 3, cfg.img)).astype(np.float32)
    def run(n_batch):
        lib.set_option("vision_batch", -1 if n_batch is None else n_batch)
        out = torch.empty((5 * rows, cols), dtype=torch.float32, device="cuda")
        m.vision(imgs, None, out.data_ptr(), 5)
        return out.cpu().numpy()
    single = run(1)
    assert np.array_equal(run(None), single) and np.array_equal(run(2), single) and np.array_equal(run(5), single)
    m.close()
    cfg = synth.qwen2vl_tiny()
    m = lib.Qwen2VL(cfg, weights.qwen2vl_file(cfg, CACHE))
    grid = np.array([1, 8, 8], dtype=np.int32)
    pix = r.standard_normal((3, 64, cfg.patch_elems)).astype(np.float32)
    outs = []
    for nb in (1, 3, 2):
        lib.set_option("vision_batch", nb)
        out = torch.empty((3 * 16, cfg.hidden), dtype=torch.float32, device="cuda")
        m.vision(pix, grid, out.data_ptr(), 3)
        outs.append(out.cpu().numpy())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    # image sizes and batch sizes that grow and shrink between calls (the buffers are re-sized for the largest of each seen so far)
    lib.set_option("vision_batch", -1)
    for (gh, gw, n) in ((8, 8, 3), (16, 8, 2), (8, 8, 5), (16, 16, 1), (8, 16, 4)):
        grid = np.array([1, gh, gw], dtype=np.int32)
        pix = r.standard_normal((n, gh * gw, cfg.patch_elems)).astype(np.float32)
        rows = gh * gw // 4
        got = torch.empty((n * rows, cfg.hidden), dtype=torch.float32, device="cuda")
        m.vision(pix, grid, got.data_ptr(), n)
        for b in range(n):
            one = torch.empty((rows, cfg.hidden), dtype=torch.float32, device="cuda")
            m.vision(pix[b], grid, one.data_ptr(), 1)
            assert np.array_equal(got[b * rows:(b + 1) * rows].cpu().numpy(), one.cpu().numpy()), (gh, gw, n, b)
    m.close()


@pytest.mark.gpu
def test_cabi_row_gather_single_rank():
    """The RCCL all-gather behind the C ABI (mllm_hip_comm_*, mllm_hip_all_gather_rows) on the one GPU of this box: a world of one rank returns the local
    rows, in order, padding dropped.  (More ranks need more GPUs; the partition / padding / order logic is covered over gloo in test_parallel_gloo.py.)"""
    import torch
    from mllm_amd import parallel
    g = parallel.RowGather(1, 0)
    x = torch.arange(3 * 5 * 8, dtype=torch.float32, device="cuda").reshape(3, 5, 8)
    y = g.gather(x, 2)
    torch.cuda.synchronize()
    assert y.shape == (2, 5, 8) and torch.equal(y, x[:2])
    g.close()
