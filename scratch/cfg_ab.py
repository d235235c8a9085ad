"""decode tok/s of the small causal-LM configs (Qwen1.5-0.5B, TinyLlama-1.1B Q4_K) over 256 steps; MLLM_SO selects a variant library (same-box A/B)"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from mllm_amd import lib, synth
from mllm_amd import mllmfile as mf
if os.environ.get('MLLM_SO'): lib.SO_PATH = os.environ['MLLM_SO']
from tests.fixtures import weights
cache = os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache")
for name, cfg in (("qwen15", synth.qwen15_05b()), ("tinyllama", synth.tinyllama_11b(mf.Q4_K))):
    path = weights.causal_lm_file(cfg, cache); ids = synth.causal_lm_ids(cfg, 64)
    m = lib.Model(cfg, path)
    rates = []
    for rep in range(3):
        m.clear_kvcache()
        tok, _, _ = m.prefill(ids, None, None, want_logits=False)
        gen, ms = m.generate(tok, 256)
        rates.append(1000 * len(gen) / ms)
    print(os.environ.get('MLLM_SO', 'default'), name, 'tok/s', ['%.0f' % r for r in rates], 'ids hash', hash(np.asarray(gen).tobytes()) & 0xffff, flush=True)
    m.close()
