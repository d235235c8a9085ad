"""tok/s of generate_sampled (top-k 5 / 50, top-p 0.92) against greedy generate, 128 steps"""
import sys
sys.path.insert(0, '.')
import numpy as np
from mllm_amd import lib, synth
from tests.fixtures import weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
m = lib.Qwen2VL(cfg, path)
u = np.random.default_rng(3).random(128).astype(np.float32)
import time
for name, kw in (("greedy generate", None), ("top-k 5", dict(method=1, top_k=5)), ("top-k 50", dict(method=1, top_k=50)), ("top-p 0.92", dict(method=2, top_p=0.92))):
    m.clear_kvcache()
    tok, _, _ = m.prefill(ids, pix, grid, want_logits=False)
    t = time.time()
    if kw is None: m.generate(tok, 128)
    else: m.generate_sampled(tok, 128, kw["method"], u, top_k=kw.get("top_k", 5), top_p=kw.get("top_p", 0.92))
    dt = time.time() - t
    print(f"{name}: {128 / dt:.0f} tok/s")
m.close()
