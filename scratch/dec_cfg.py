import sys, os, numpy as np
sys.path.insert(0, '.')
from mllm_amd import lib
lib.SO_PATH = os.path.abspath(sys.argv[1])
from mllm_amd import synth, weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
m = lib.Qwen2VL(cfg, path)
ids = (np.arange(282) * 7919 % 150000).astype(np.int32)
tok, _, _ = m.prefill(ids, want_logits=False)
gen, _ = m.generate(tok, 8)
r = []
for which, name in ((10, 'qkv'), (12, 'oproj'), (13, 'gateup'), (14, 'down')):
    ms, nb = m.time_gemv(which, 112); r.append('%s %.2f' % (name, ms * 1000))
import time
t0 = time.perf_counter(); gen, _ = m.generate(int(gen[-1]), 128); dt = time.perf_counter() - t0
print(sys.argv[1], ' | '.join(r), '| decode tok/s %.0f' % (128 / dt))
