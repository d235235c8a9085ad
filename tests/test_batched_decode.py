"""Batched decode (mllm_hip_model_batch_*; the reference's hook is KVCache_batch, mllm/Types.hpp:26-33): B independent sequences stepped together share one pass over the
weights.  The contract tested here is the one that makes it safe to report beside the batch-1 headline: ROW b OF A BATCHED STEP IS, BIT FOR BIT, WHAT SEQUENCE b PRODUCES
STEPPING ALONE -- greedy ids and every logit -- for prompts of different lengths (one with an image), for both head forms (tied Q4_0 table, Linear Q4_K head), and a
sequence can leave the batch and carry on with the fused single-sequence step."""
import os

import numpy as np
import pytest

from mllm_amd import mllmfile as mf, synth
from mllm_amd import synthfile as weights

pytestmark = pytest.mark.gpu
CACHE = os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache")


def _alone(lib, cfg, path, prompt, steps, image=None, meta=None):
    """batch-1 reference run on a fresh model: prefill, then `steps` single decode steps (the fused kernels / captured graph)"""
    m = lib.Model(cfg, path)
    tok, lg, _ = m.prefill(prompt, image, meta)
    toks, rows = [tok], [lg]
    for _ in range(steps):
        tok, lg, _ = m.decode(tok)
        toks.append(tok)
        rows.append(lg)
    m.close()
    return toks, np.stack(rows)


def test_batched_rows_equal_their_batch1_runs_qwen2vl_tiny():
    from mllm_amd import lib
    cfg = synth.qwen2vl_tiny()
    path = weights.qwen2vl_file(cfg, CACHE)
    pix, grid, ids_img = synth.qwen2vl_inputs(cfg, (8, 8), 6)
    r = np.random.default_rng(77)
    prompts = [(ids_img, pix, grid), (r.integers(0, 2000, size=9).astype(np.int32), None, None), (r.integers(0, 2000, size=17).astype(np.int32), None, None),
               (r.integers(0, 2000, size=5).astype(np.int32), None, None)]
    steps = 7
    want = [_alone(lib, cfg, path, p, steps, im, me) for p, im, me in prompts]
    m = lib.Model(cfg, path)
    B = len(prompts)
    m.batch_begin(B)
    cur = []
    for b, (p, im, me) in enumerate(prompts):
        m.batch_select(b)
        tok, lg, _ = m.prefill(p, im, me)
        assert tok == want[b][0][0] and np.array_equal(lg, want[b][1][0]), b
        assert m.cache_len() == len(p)
        cur.append(tok)
    for s in range(1, steps + 1):
        nxt, lg, ms = m.batch_decode(cur)
        for b in range(B):
            assert int(nxt[b]) == want[b][0][s], (s, b)
            assert np.array_equal(lg[b], want[b][1][s]), (s, b, float(np.abs(lg[b] - want[b][1][s]).max()))
        cur = nxt.tolist()
    # a sequence leaves the batch: the fused single-sequence step (re-armed) continues it exactly like a run that never was batched
    ref_toks, ref_rows = _alone(lib, cfg, path, prompts[2][0], steps + 3)
    m.batch_select(2)
    assert m.cache_len() == len(prompts[2][0]) + steps
    tok = cur[2]
    for s in range(steps + 1, steps + 4):
        tok, lg, _ = m.decode(tok)
        assert tok == ref_toks[s] and np.array_equal(lg, ref_rows[s]), s
    # clear_kvcache acts on the selected sequence only
    m.clear_kvcache()
    assert m.cache_len() == 0
    m.batch_select(1)
    assert m.cache_len() == len(prompts[1][0]) + steps
    with pytest.raises(lib.MllmHipError):
        m.batch_decode(cur)          # sequence 2 has no prefill any more
    m.close()


def test_batched_rows_linear_head_and_b2():
    """TinyLlama geometry (GQA 4 / 2, Linear Q4_K head over the batch rows, HF rotary positions = tokens in each cache), B = 2 then B = 3 on the same model."""
    from mllm_amd import lib
    cfg = synth.tinyllama_tiny(mf.Q4_K)
    path = weights.causal_lm_file(cfg, CACHE)
    r = np.random.default_rng(5)
    prompts = [r.integers(0, cfg.vocab, size=n).astype(np.int32) for n in (20, 6, 11)]
    steps = 5
    want = [_alone(lib, cfg, path, p, steps) for p in prompts]
    m = lib.Model(cfg, path)
    m.batch_begin(3)
    cur = []
    for b, p in enumerate(prompts):
        m.batch_select(b)
        tok, _, _ = m.prefill(p)
        cur.append(tok)
    nxt, lg, _ = m.batch_decode(cur[:2])          # B = 2: sequences 0 and 1 step, sequence 2 waits
    for b in range(2):
        assert int(nxt[b]) == want[b][0][1] and np.array_equal(lg[b], want[b][1][1])
    nxt3, lg3, _ = m.batch_decode([int(nxt[0]), int(nxt[1]), cur[2]])
    assert int(nxt3[0]) == want[0][0][2] and int(nxt3[1]) == want[1][0][2] and int(nxt3[2]) == want[2][0][1]
    assert np.array_equal(lg3[0], want[0][1][2]) and np.array_equal(lg3[2], want[2][1][1])
    with pytest.raises(lib.MllmHipError):
        m.batch_begin(16)
    m.close()
