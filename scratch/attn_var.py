import sys, os, numpy as np
sys.path.insert(0, '.')
from mllm_amd import lib
lib.SO_PATH = os.path.abspath(sys.argv[1])
from mllm_amd import synth, weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
m = lib.Qwen2VL(cfg, path)
out = []
for n in (8, 264, 520, 776):
    m.clear_kvcache()
    ids = (np.arange(n) * 7919 % 150000).astype(np.int32)
    tok, _, _ = m.prefill(ids, want_logits=False)
    gen, _ = m.generate(tok, 4)
    ms, nb = m.time_gemv(11, 56)
    out.append('T %d: %.2f' % (n + 4, ms * 1000))
print(sys.argv[1], ' | '.join(out))
