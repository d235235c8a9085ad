import sys
sys.path.insert(0, '.')
import numpy as np
from mllm_amd import lib, synth
from mllm_amd import synthfile as weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg, cache_dir="/tmp/mllm_amd_cache")
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
lib.set_option("merge_o", 5)
m = lib.Qwen2VL(cfg, path)
tok, _, _ = m.prefill(ids, pix, grid, want_logits=False)
try:
    t2, _, ms = m.decode(tok)
    print('one step ok', t2, ms)
    gen, ms = m.generate(t2, 4)
    print('generate ok', gen, ms)
except Exception as e:
    print('ERR', e)
