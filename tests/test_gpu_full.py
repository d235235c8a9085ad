"""Full-size GPU parity; NOTE the 2 B goldens hold SAMPLED logits (the top-64 and every 97th logit of steps 0, 16, 32, 48, 64 -- not whole 151,936-wide rows) plus all 65 greedy ids.
BASELINE.json configs[3] shape (Qwen2-VL-2B, Q4_K, 448x448 image + 24-token prompt, KV limit 800) against golden outputs of the reference's own run on the same synthetic .mllm (tests/golden/qwen2vl_2b_ref*.npz, made by
oracle/make_golden.py --full from oracle/_ref/ref_qwen2vl / ref_ops): greedy ids identical, every sampled logit bit-identical
(the goldens keep the top-64 logits and every 97th logit of the dumped steps), the vision tower's image_embeds bit-identical."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from mllm_amd import lib, synth  # noqa: E402
from mllm_amd import synthfile as weights  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def full_model():
    cfg = synth.qwen2vl_2b()
    path = weights.qwen2vl_file(cfg, cache_dir=os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache"))
    m = lib.Qwen2VL(cfg, path)
    yield cfg, m
    m.close()


def _sample_err(g, i, logits):
    return float(max(np.abs(logits[g["top_idx"][i]] - g["top_val"][i]).max(), np.abs(logits[::97] - g["strided"][i]).max()))


def _run(m, g, first):
    steps = {int(s): i for i, s in enumerate(g["steps"])}
    tok, logits, _ = first()
    toks, errs = [tok], [_sample_err(g, steps[0], logits)]
    for s in range(1, len(g["tokens"])):
        tok, logits, _ = m.decode(tok)
        toks.append(tok)
        if s in steps:
            errs.append(_sample_err(g, steps[s], logits))
    return toks, errs


def test_full_vision_tower_bit_exact(full_model):
    cfg, m = full_model
    pix, grid, _ = synth.qwen2vl_inputs(cfg, (32, 32), 24)
    out = torch.empty((256, cfg.hidden), dtype=torch.float32, device="cuda")
    m.vision(pix, grid, out.data_ptr())
    ref = np.load(os.path.join(GOLD, "qwen2vl_2b_ref_vision.npz"))["image_embeds"]
    got = out.cpu().numpy()
    assert np.array_equal(got, ref), float(np.abs(got - ref).max())


def test_full_text_prompt_bit_exact(full_model):
    cfg, m = full_model
    g = np.load(os.path.join(GOLD, "qwen2vl_2b_ref_text.npz"))
    m.clear_kvcache()
    toks, errs = _run(m, g, lambda: m.prefill(g["ids"]))
    assert toks == g["tokens"].tolist(), sum(a == b for a, b in zip(toks, g["tokens"].tolist()))
    assert max(errs) == 0.0, errs


def test_full_image_prompt_bit_exact(full_model):
    cfg, m = full_model
    g = np.load(os.path.join(GOLD, "qwen2vl_2b_ref.npz"))
    pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
    m.clear_kvcache()
    toks, errs = _run(m, g, lambda: m.prefill(ids, pix, grid))
    assert toks == g["tokens"].tolist(), sum(a == b for a, b in zip(toks, g["tokens"].tolist()))
    assert max(errs) == 0.0, errs
    # generate() (device-side argmax, hipGraph replay) reproduces the stepwise ids
    m.clear_kvcache()
    tok, _, _ = m.prefill(ids, pix, grid)
    gen, _ = m.generate(tok, len(g["tokens"]) - 1)
    assert [tok] + gen.tolist() == g["tokens"].tolist()


def test_step_timed_launch_by_launch_is_the_same_step(full_model):
    """mllm_hip_model_time_step (bench.py's per-launch figures): the decode step issued eagerly with an event either side of every launch produces the golden greedy ids,
    and accounts for the step as the 2 B model runs it -- 27 chain launches (down + the next layer's q|k|v + attention + o-projection), 28 gate|up, layer 0's q|k|v and
    attention (+ o-projection) and layer 27's down on their own, the head, the state advance: 60 launches."""
    cfg, m = full_model
    g = np.load(os.path.join(GOLD, "qwen2vl_2b_ref_text.npz"))
    m.clear_kvcache()
    tok, _, _ = m.prefill(g["ids"])
    assert tok == int(g["tokens"][0])
    gen, _ = m.generate(tok, 7)
    assert gen.tolist() == g["tokens"][1:8].tolist()
    kinds, last = m.time_step(int(gen[-1]), 9)
    assert last == int(g["tokens"][16])
    assert {k: n for k, (_, n) in kinds.items()} == {"chain": 27, "gateup": 28, "qkv": 1, "attn": 1, "down": 1, "head": 1, "next": 1}
    assert all(0.5 < us < 200.0 for us, _ in kinds.values()), kinds
    tok2, _, _ = m.decode(last)      # the graph replay carries on from the state the eager steps left
    assert tok2 == int(g["tokens"][17])


def test_chain_launch_qkv_role_carried_on_or_on_its_own_workgroups():
    """Option chain_cont: in the chain launch the next layer's q|k|v role either gets workgroups of its own (0) or is carried on by the first down-projection workgroups once
    they are through (default).  Same arithmetic, same hand-overs: the golden ids and sampled logits of the reference's run either way."""
    cfg = synth.qwen2vl_2b()
    path = weights.qwen2vl_file(cfg, cache_dir=os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache"))
    g = np.load(os.path.join(GOLD, "qwen2vl_2b_ref_text.npz"))
    try:
        for mode in (0, 1):
            lib.set_option("chain_cont", mode)
            m = lib.Qwen2VL(cfg, path)
            try:
                toks, errs = _run(m, g, lambda: m.prefill(g["ids"]))
                assert toks == g["tokens"].tolist(), mode
                assert max(errs) == 0.0, (mode, errs)
            finally:
                m.close()
    finally:
        lib.set_option("chain_cont", -1)


def test_shared_launches_with_a_second_engine_busy_on_the_same_gpu(full_model):
    """The chain launch's roles wait for each other inside one launch; their producers are always dispatched first, so a role only ever waits for workgroups that are running or
    done -- also when another stream's kernels hold CUs.  Here a second engine runs the vision tower back to back from another thread (its own stream: GEMMs and attention
    kernels that fill the chip) while the first generates: the ids are those of the undisturbed run and no polled hand-over times out (generate would raise)."""
    import threading
    cfg, m = full_model
    path = weights.qwen2vl_file(cfg, cache_dir=os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache"))
    g = np.load(os.path.join(GOLD, "qwen2vl_2b_ref_text.npz"))
    m.clear_kvcache()
    tok, _, _ = m.prefill(g["ids"])
    alone, _ = m.generate(tok, 200)
    assert alone[:len(g["tokens"]) - 1].tolist() == g["tokens"][1:].tolist()
    other = lib.Qwen2VL(cfg, path)
    pix, grid, _ = synth.qwen2vl_inputs(cfg, (32, 32), 24)
    out = torch.empty((256, cfg.hidden), dtype=torch.float32, device="cuda")
    stop, passes, errors = threading.Event(), [0], []

    def tower():
        try:
            while not stop.is_set():
                other.vision(pix, grid, out.data_ptr())
                passes[0] += 1
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    th = threading.Thread(target=tower)
    th.start()
    try:
        for _ in range(6):
            m.clear_kvcache()
            tok, _, _ = m.prefill(g["ids"])
            busy, _ = m.generate(tok, 200)
            assert np.array_equal(busy, alone)
    finally:
        stop.set()
        th.join()
        other.close()
    assert not errors and passes[0] >= 6, (errors, passes)


def test_pipelined_decode_attention_equals_the_unpipelined_kernel_over_a_long_context(tmp_path):
    """dec_attn_pipe_kernel (scores of later key blocks computed while the walk over the first ones runs; kernels_attn_core.h: fa2_decode_head_pipe) against
    dec_attn_kernel (attn_flags bit 2: phases A -> B -> C one after the other, the form pinned against the oracle to T = 1500 by test_fa2_on_the_engine_kv_layout):
    the same greedy ids and bit-identical logits along a decode that takes the cache from 30 to 760 keys -- every block count from 1 to 24, blocks that wrap the
    producers' LDS regions (more than 14 blocks), partial last blocks, the appended key alone in its block."""
    from mllm_amd import lib, synth
    from mllm_amd import synthfile as weights
    cfg = synth.qwen2vl_tiny()
    cfg.cache_limit = 800
    path = weights.qwen2vl_file(cfg, cache_dir=os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache"))
    ids = np.random.default_rng(21).integers(0, 2000, size=30).astype(np.int32)
    runs = {}
    for flags in (7, 3):
        lib.set_option("attn_flags", flags)
        try:
            m = lib.Qwen2VL(cfg, path)
            tok, lg0, _ = m.prefill(ids)
            toks, _ = m.generate(tok, 700)
            nxt, lg1, _ = m.decode(int(toks[-1]))
            steps = [m.decode(nxt)[1]]
            for _ in range(27):
                steps.append(m.decode(int(np.argmax(steps[-1])))[1])
            runs[flags] = (toks.copy(), lg1.copy(), np.stack(steps))
            m.close()
        finally:
            lib.set_option("attn_flags", -1)
    assert runs[7][0].tolist() == runs[3][0].tolist()
    assert np.array_equal(runs[7][1], runs[3][1]) and np.array_equal(runs[7][2], runs[3][2])


def test_weight_warming_workgroups_change_nothing_but_time():
    """The attention launch's warming workgroups (attn_flags bits 4 / 6: the CUs that hold no head read the layer's gate|up and o-projection rows through LDS-DMA into a
    landing pad, so that the XCD L2s hold them when those launches arrive) at the 2B shape -- where every region of decode_warm_table exists -- against the same model
    without them: identical ids over 200 steps and bit-identical logits of the step behind them."""
    from mllm_amd import synthfile as weights
    cfg = synth.qwen2vl_2b()
    path = weights.qwen2vl_file(cfg, cache_dir=os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache"))
    pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
    runs = {}
    for flags in (11, 91, 251):
        lib.set_option("attn_flags", flags)
        try:
            m = lib.Qwen2VL(cfg, path)
            tok, _, _ = m.prefill(ids, pix, grid, want_logits=False)
            toks, _ = m.generate(tok, 200)
            _, lg, _ = m.decode(int(toks[-1]))
            runs[flags] = (toks.copy(), lg.copy())
            m.close()
        finally:
            lib.set_option("attn_flags", -1)
    for flags in (91, 251):
        assert runs[flags][0].tolist() == runs[11][0].tolist() and np.array_equal(runs[flags][1], runs[11][1]), flags
