"""decode attention alone (mllm_hip_model_time_kernel 11) at several context lengths on the production library: the slope is the walker's cost per 32-key block"""
import sys
sys.path.insert(0, '.')
import numpy as np
from mllm_amd import lib, synth
import os
if os.environ.get('MLLM_SO'): lib.SO_PATH = os.path.abspath(os.environ['MLLM_SO'])
from mllm_amd import synthfile as weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg, cache_dir="/tmp/mllm_amd_cache")
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
m = lib.Qwen2VL(cfg, path)
tok, _, _ = m.prefill(ids, pix, grid, want_logits=False)
T = 282
prev = None
for steps in (6, 96, 96, 96, 96, 96):
    gen, _ = m.generate(tok, steps); tok = int(gen[-1]); T += steps
    us = [m.time_kernel(11, 56)[0] * 1e3 for _ in range(3)]
    u = float(np.median(us))
    print('T = %d: attention alone %.2f us' % (T, u) + ('' if prev is None else '   slope %.3f us per 32 keys' % ((u - prev[1]) / (T - prev[0]) * 32)), flush=True)
    prev = (T, u)
