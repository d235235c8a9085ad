"""Soak of the shared decode launches: 30 x (image prefill + 500 generated tokens up to the cache limit) on the 2 B model, every run must repeat the first one's ids exactly
(deterministic kernels) and no polled hand-off may time out (the engine would return an error)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from mllm_amd import lib, synth
from mllm_amd import synthfile as weights
cfg = synth.qwen2vl_2b()
m = lib.Qwen2VL(cfg, weights.qwen2vl_file(cfg))
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
first = None
t0 = time.time()
for r in range(30):
    m.clear_kvcache()
    tok, _, _ = m.prefill(ids, pix, grid, want_logits=False)
    toks, ms = m.generate(tok, 500)
    if first is None:
        first = toks.copy()
    assert np.array_equal(first, toks), r
    if r % 5 == 0:
        print(f"run {r}: {500e3 / ms:.1f} tok/s", flush=True)
print(f"30 runs x 500 tokens identical, {time.time() - t0:.1f} s")
m.close()
