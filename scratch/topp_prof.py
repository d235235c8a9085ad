import sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from mllm_amd import lib, synth
from tests.fixtures import weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
ids = (np.arange(24) * 7919 % 150000).astype(np.int32)
m = lib.Qwen2VL(cfg, path)
u = np.random.default_rng(3).random(32).astype(np.float32)
tok, _, _ = m.prefill(ids, want_logits=False)
m.generate_sampled(tok, 32, 2, u, top_p=0.92)
m.close()
