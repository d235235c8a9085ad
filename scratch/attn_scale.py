import sys, numpy as np
sys.path.insert(0, '.')
from mllm_amd import lib
import os
if len(sys.argv) > 1: lib.SO_PATH = os.path.abspath(sys.argv[1])
from mllm_amd import synth, weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
m = lib.Qwen2VL(cfg, path)
for n in (8, 64, 136, 264, 392, 520, 776):
    m.clear_kvcache()
    ids = (np.arange(n) * 7919 % 150000).astype(np.int32)
    tok, _, _ = m.prefill(ids, want_logits=False)
    gen, _ = m.generate(tok, 4)
    ms, nb = m.time_gemv(11, 56)
    print('T', n + 4, 'dec_attn us %.2f' % (ms * 1000))
