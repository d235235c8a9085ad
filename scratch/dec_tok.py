import sys, os, numpy as np
sys.path.insert(0, '.')
from mllm_amd import lib
if os.environ.get('DBGSO'): lib.SO_PATH = os.path.abspath(os.environ['DBGSO'])
from mllm_amd import synth, weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
m = lib.Qwen2VL(cfg, path)
ids = (np.arange(282) * 7919 % 150000).astype(np.int32)
tok, _, _ = m.prefill(ids, want_logits=False)
toks, _ = m.generate(tok, 16)
import time
t0 = time.perf_counter(); toks, dev_ms = m.generate(int(toks[-1]), 256); dt = time.perf_counter() - t0
print(os.environ.get('DBGSO', 'main'), 'tok/s %.1f  (%.4f ms/token)' % (256 / dt, dt * 1000 / 256))
