"""End-to-end GPU parity: the engine's Qwen2-VL graph (vision tower + LLM, prefill + greedy decode) against golden outputs of
the reference's own Qwen2VLModel run on its x86 CPU backend on the same synthetic Q4_K .mllm (tests/golden/qwen2vl_tiny.npz,
made by oracle/make_golden.py).  BASELINE.json's north_star asks for identical greedy token ids and logits within 1e-3; the
kernels keep the reference's operation order, so the bar held here is stricter: every logit bit-identical."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from mllm_amd import lib, synth  # noqa: E402
from mllm_amd import synthfile as weights  # noqa: E402


@pytest.fixture(scope="module")
def tiny_model(tmp_path_factory):
    cfg = synth.qwen2vl_tiny()
    path = weights.qwen2vl_file(cfg, cache_dir=str(tmp_path_factory.mktemp("w")))
    m = lib.Qwen2VL(cfg, path)
    m.path = path
    yield cfg, m
    m.close()


def test_tiny_prefill_and_decode_match_reference(tiny_model, tiny_gold):
    cfg, m = tiny_model
    g = tiny_gold
    pix, grid, ids = synth.qwen2vl_inputs(cfg, (8, 8), 6)
    assert np.array_equal(ids, g["ids"])
    m.clear_kvcache()
    tok, logits, ms = m.prefill(ids, pix, grid)
    toks, maxerr = [tok], float(np.max(np.abs(logits - g["logits"][0])))
    for s in range(1, len(g["tokens"])):
        tok, logits, _ = m.decode(tok)
        toks.append(tok)
        maxerr = max(maxerr, float(np.max(np.abs(logits - g["logits"][s]))))
    assert toks == g["tokens"].tolist(), (toks, g["tokens"].tolist())
    assert maxerr == 0.0, maxerr


def test_tiny_text_only_prompt(tiny_model, tiny_gold):
    cfg, m = tiny_model
    g = tiny_gold
    m.clear_kvcache()
    tok, logits, _ = m.prefill(g["ids_text"])
    toks, maxerr = [tok], float(np.max(np.abs(logits - g["logits_text"][0])))
    for s in range(1, len(g["tokens_text"])):
        tok, logits, _ = m.decode(tok)
        toks.append(tok)
        maxerr = max(maxerr, float(np.max(np.abs(logits - g["logits_text"][s]))))
    assert toks == g["tokens_text"].tolist()
    assert maxerr == 0.0, maxerr


def test_untied_lm_head_and_cache_len(tiny_gold, tmp_path):
    """tie_embedding_words = false (a separate Q4_K lm_head Linear, modeling_qwen2_vl.hpp:375-401): every logit of 8 steps equals the reference's run on the untied file;
    mllm_hip_model_cache_len follows the prefill, the decode steps and clear_kvcache; and a tiny 2 x 2 grid (4 patches: below the tower's 16-row minimum) is refused with a
    message instead of failing inside the first block."""
    g = tiny_gold
    cfg = synth.qwen2vl_tiny()
    cfg.tie_embedding = False
    m = lib.Qwen2VL(cfg, weights.qwen2vl_file(cfg, cache_dir=str(tmp_path)))
    pix, grid, ids = synth.qwen2vl_inputs(cfg, (8, 8), 6)
    assert m.cache_len() == 0
    tok, logits, _ = m.prefill(ids, pix, grid)
    assert m.cache_len() == len(ids)
    toks, rows = [tok], [logits]
    for s in range(1, len(g["tokens_untied"])):
        tok, logits, _ = m.decode(tok)
        toks.append(tok)
        rows.append(logits)
    assert m.cache_len() == len(ids) + len(toks) - 1
    assert toks == g["tokens_untied"].tolist() and np.array_equal(np.stack(rows), g["logits_untied"])
    m.clear_kvcache()
    assert m.cache_len() == 0
    small = np.zeros((4, cfg.patch_elems), dtype=np.float32)
    out = torch.empty((1, cfg.hidden), dtype=torch.float32, device="cuda")
    with pytest.raises(lib.MllmHipError, match="16 patches"):
        m.vision(small, np.array([1, 2, 2], dtype=np.int32), out.data_ptr())
    m.close()


def test_vision_tower_matches_reference(tiny_model, tiny_gold):
    cfg, m = tiny_model
    pix, grid, _ = synth.qwen2vl_inputs(cfg, (8, 8), 6)
    out = torch.empty((16, cfg.hidden), dtype=torch.float32, device="cuda")
    m.vision(pix, grid, out.data_ptr())
    err = float(np.max(np.abs(out.cpu().numpy() - tiny_gold["image_embeds"])))
    assert err == 0.0, err


def test_ragged_grids_match_reference(tiny_model):
    """Non-square grids whose patch and token counts sit off every tile size (6 x 10 = 60 patches / 24 prompt tokens, 8 x 12 = 96 / 33, 4 x 18 = 72 / 21): image embeddings and
    every logit of 6 greedy steps equal the reference's own run (tests/golden/qwen2vl_ragged.npz, oracle/make_golden.py --ragged)."""
    cfg, m = tiny_model
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "qwen2vl_ragged.npz"))
    for k in range(3):
        grid = g[f"grid{k}"]
        pix, _, ids = synth.qwen2vl_inputs(cfg, (int(grid[1]), int(grid[2])), int(g[f"ntext{k}"]))
        n_tok = pix.shape[0] // 4
        out = torch.empty((n_tok, cfg.hidden), dtype=torch.float32, device="cuda")
        m.vision(pix, grid, out.data_ptr())
        assert np.array_equal(out.cpu().numpy(), g[f"image_embeds{k}"]), k
        m.clear_kvcache()
        tok, logits, _ = m.prefill(ids, pix, grid)
        toks, rows = [tok], [logits]
        for _ in range(5):
            tok, logits, _ = m.decode(tok)
            toks.append(tok)
            rows.append(logits)
        assert toks == g[f"tokens{k}"].tolist(), (k, toks)
        assert np.array_equal(np.stack(rows), g[f"logits{k}"]), (k, float(np.abs(np.stack(rows) - g[f"logits{k}"]).max()))
    m.clear_kvcache()                           # the shortest prompt the reference takes (two tokens)
    tok, logits, _ = m.prefill(np.array([17, 23], dtype=np.int32))
    toks, rows = [tok], [logits]
    for _ in range(4):
        tok, logits, _ = m.decode(tok)
        toks.append(tok)
        rows.append(logits)
    assert toks == g["one_tokens"].tolist() and np.array_equal(np.stack(rows), g["one_logits"])
    m.clear_kvcache()


def test_one_token_prompt_is_a_prefill(tiny_model):
    """S == 1 on an empty cache: the reference's own model faults here (its get_position_ids reads S == 1 as a decode step), so there is no golden; the engine treats it as a
    prefill of one token and must equal the oracle's composition of the graph."""
    from oracle import models as omodels
    cfg, m = tiny_model
    ref = omodels.LLM(omodels.Weights(m.path), cfg)
    m.clear_kvcache()
    tok, logits, _ = m.prefill(np.array([17], dtype=np.int32))
    want = ref.prefill(np.array([17], dtype=np.int32))
    assert np.array_equal(logits, want) and m.cache_len() == 1
    for _ in range(3):
        want = ref.decode(tok)
        tok, logits, _ = m.decode(tok)
        assert np.array_equal(logits, want)
    m.clear_kvcache()


def test_merged_attention_and_o_projection_launch_changes_nothing(tiny_gold, tmp_path):
    """A layer's down projection and the next layer's q|k|v projection, attention and o-projection share one launch by default (option merge_o = 4; 3: without the down projection: the attention's workgroups ride behind the projection's and
    poll its q | k | v rows, the o-projection's ride behind the attention's and poll its output row, each handed over as {value, epoch} pairs; 2 / 1: attention + o-projection only;
    0: five launches per layer): ids and every logit of 24 steps equal the reference's golden run in every form, and a second generation on the re-armed state (epochs start
    over) repeats the first."""
    g = tiny_gold
    cfg = synth.qwen2vl_tiny()
    path = weights.qwen2vl_file(cfg, cache_dir=str(tmp_path))
    pix, grid, ids = synth.qwen2vl_inputs(cfg, (8, 8), 6)
    try:
        for mode in (0, 1, 2, 3, 4):
            lib.set_option("merge_o", mode)
            m = lib.Qwen2VL(cfg, path)
            for _ in range(2):
                m.clear_kvcache()
                tok, logits, _ = m.prefill(ids, pix, grid)
                toks, rows = [tok], [logits]
                for s in range(1, len(g["tokens"])):
                    tok, logits, _ = m.decode(tok)
                    toks.append(tok)
                    rows.append(logits)
                assert toks == g["tokens"].tolist() and np.array_equal(np.stack(rows), g["logits"]), mode
            m.clear_kvcache()
            tok, _, _ = m.prefill(ids, pix, grid)
            gen, _ = m.generate(tok, 20)
            assert gen.tolist() == g["tokens"][1:21].tolist(), mode
            m.close()
    finally:
        lib.set_option("merge_o", -1)


def test_decode_without_the_captured_graph(tmp_path):
    """MLLM_HIP_NO_GRAPH=1 (what the profiling scripts set): the decode step's launches -- the shared ones included -- issued one by one instead of replayed from the captured
    hipGraph give the same ids and logits (own process: the variable is read when the model is created)."""
    import subprocess
    import sys
    code = (
        "import numpy as np, os, sys\n"
        "from mllm_amd import lib, synth\n"
        "from mllm_amd import synthfile as weights\n"
        "cfg = synth.qwen2vl_tiny()\n"
        "g = np.load(os.path.join(sys.argv[1], 'qwen2vl_tiny.npz'))\n"
        "m = lib.Qwen2VL(cfg, weights.qwen2vl_file(cfg, cache_dir=sys.argv[2]))\n"
        "pix, grid, ids = synth.qwen2vl_inputs(cfg, (8, 8), 6)\n"
        "tok, logits, _ = m.prefill(ids, pix, grid)\n"
        "toks, rows = [tok], [logits]\n"
        "for s in range(1, 12):\n"
        "    tok, logits, _ = m.decode(tok); toks.append(tok); rows.append(logits)\n"
        "assert toks == g['tokens'][:12].tolist() and np.array_equal(np.stack(rows), g['logits'][:12])\n"
        "gen, _ = m.generate(tok, 8)\n"
        "assert gen.tolist() == g['tokens'][12:20].tolist()\n"
        "print('eager ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code, os.path.join(root, "tests", "golden"), str(tmp_path)], capture_output=True, text=True, timeout=600, cwd=root,
                         env=dict(os.environ, MLLM_HIP_NO_GRAPH="1", PYTHONPATH=root))
    assert out.returncode == 0 and "eager ok" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


def test_generate_equals_stepwise_decode_and_clear_kvcache_resets(tiny_model):
    cfg, m = tiny_model
    pix, grid, ids = synth.qwen2vl_inputs(cfg, (8, 8), 6)
    m.clear_kvcache()
    t0, l0, _ = m.prefill(ids, pix, grid)
    step = [t0]
    for _ in range(10):
        t, _, _ = m.decode(step[-1])
        step.append(t)
    m.clear_kvcache()
    t1, l1, _ = m.prefill(ids, pix, grid)
    assert t1 == t0 and np.array_equal(l0, l1)          # idempotent after clear_kvcache (bitwise: deterministic kernels)
    gen, _ = m.generate(t1, 10)
    assert gen.tolist() == step[1:]


def test_kv_overflow_is_an_error_not_a_crash(tiny_model):
    cfg, m = tiny_model
    m.clear_kvcache()
    ids = np.arange(cfg.cache_limit - 1, dtype=np.int32) % 1000
    m.prefill(ids)
    m.decode(5)
    with pytest.raises(lib.MllmHipError):
        m.decode(5)
    m.clear_kvcache()


def test_long_context_crosses_the_score_pass_and_ring_boundaries(tmp_path):
    """A 530-token text prompt + decode steps on a 640-entry cache: prefill attention over 17 key chunks, then decode attention with
    T > 512 (second score pass, V ring refills), against the oracle's composition of the same graph (oracle/models.py)."""
    from oracle import models as omodels
    cfg = synth.qwen2vl_tiny()
    path = weights.qwen2vl_file(cfg, cache_dir=str(tmp_path))
    m = lib.Qwen2VL(cfg, path, cache_limit=640)
    try:
        ids = (np.arange(530, dtype=np.int64) * 7919 % 2000).astype(np.int32)
        ref = omodels.LLM(omodels.Weights(path), cfg)
        want = ref.prefill(ids)
        tok, logits, _ = m.prefill(ids)
        assert np.array_equal(logits, want), float(np.abs(logits - want).max())
        for _ in range(5):
            want = ref.decode(tok)
            tok, logits, _ = m.decode(tok)
            assert np.array_equal(logits, want), float(np.abs(logits - want).max())
            assert tok == int(np.argmax(want))
    finally:
        m.close()


def test_large_cache_limit_uses_a_three_slot_ring(tmp_path):
    """cache_limit 4096: the score / mask arrays take 36 KB of LDS and the V ring drops to three slots (modulo slot index, refills from the
    first chunk on) -- prefill + decode steps of the toy model against the oracle's composition."""
    from oracle import models as omodels
    cfg = synth.qwen2vl_tiny()
    path = weights.qwen2vl_file(cfg, cache_dir=str(tmp_path))
    m = lib.Qwen2VL(cfg, path, cache_limit=4096)
    try:
        ids = (np.arange(530, dtype=np.int64) * 104729 % 2000).astype(np.int32)
        ref = omodels.LLM(omodels.Weights(path), cfg)
        want = ref.prefill(ids)
        tok, logits, _ = m.prefill(ids)
        assert np.array_equal(logits, want), float(np.abs(logits - want).max())
        for _ in range(3):
            want = ref.decode(tok)
            tok, logits, _ = m.decode(tok)
            assert np.array_equal(logits, want), float(np.abs(logits - want).max())
    finally:
        m.close()


def test_mid_shape_engine_matches_oracle_composition(tmp_path):
    """hidden 512 / inter 1280: two super-blocks per row and an intermediate size divisible by 5, i.e. the one-lane-per-super-block
    gate|up kernel (dec_gateup_blk) on a shape other than the 2B model's (the tiny golden config falls back to the 8-lane kernel)."""
    from oracle import models as omodels
    cfg = synth.Qwen2VLConfig(hidden=512, inter=1280, layers=2, heads=4, kv_heads=2, vocab=2048, cache_limit=64, v_dim=256, image_token_id=2040,
                              vision_start_token_id=2041, vision_end_token_id=2042, video_token_id=2043)
    path = weights.qwen2vl_file(cfg, cache_dir=str(tmp_path), vision=False, tag="-mid")
    m = lib.Qwen2VL(cfg, path)
    try:
        ids = (np.arange(20, dtype=np.int64) * 7919 % 2000).astype(np.int32)
        ref = omodels.LLM(omodels.Weights(path), cfg)
        want = ref.prefill(ids)
        tok, logits, _ = m.prefill(ids)
        assert np.array_equal(logits, want), float(np.abs(logits - want).max())
        for _ in range(6):
            want = ref.decode(tok)
            tok, logits, _ = m.decode(tok)
            assert np.array_equal(logits, want), float(np.abs(logits - want).max())
    finally:
        m.close()


def test_lane_per_superblock_projection_on_short_rows(tmp_path):
    """dec_proj_blk serves rows of >= 17 super-blocks by default (the 2B model's down projection); forced onto the toy model's o / down
    projections (mllm_hip_set_option("pjb_min_ns", 1)) it must reproduce the golden run as well."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "qwen2vl_tiny.npz"))
    cfg = synth.qwen2vl_tiny()
    lib.set_option("pjb_min_ns", 1)      # before the model captures its decode graph
    try:
        m = lib.Qwen2VL(cfg, weights.qwen2vl_file(cfg, cache_dir=str(tmp_path)))
        tok, logits, _ = m.prefill(g["ids_text"])
        assert np.array_equal(logits, g["logits_text"][0])
        for s in range(1, len(g["tokens_text"])):
            tok, logits, _ = m.decode(tok)
            assert np.array_equal(logits, g["logits_text"][s]), s
        m.close()
    finally:
        lib.set_option("pjb_min_ns", -1)
