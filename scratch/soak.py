import sys, os, numpy as np, time
sys.path.insert(0, '.')
from mllm_amd import lib, synth
from tests.fixtures import weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
m = lib.Qwen2VL(cfg, path)
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
ref = None
t0 = time.time()
for it in range(12):
    m.clear_kvcache()
    tok, _, _ = m.prefill(ids, pix, grid, want_logits=False)
    toks, _ = m.generate(tok, 500)
    h = hash(toks.tobytes())
    if ref is None: ref = h
    assert h == ref, 'run %d differs' % it
print('12 x (image prefill + 500 tokens) identical, %.1f s' % (time.time() - t0))
