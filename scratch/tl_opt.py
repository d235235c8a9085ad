"""TinyLlama-1.1B (Q4_K) decode for values of one option: tok/s over 256 steps, ids crc, launch by launch"""
import sys, zlib
sys.path.insert(0, '.')
import numpy as np
from mllm_amd import lib, synth
from mllm_amd import mllmfile as mf
from mllm_amd import synthfile as weights
import os
cfg = synth.qwen15_05b() if os.environ.get("CFG") == "qwen15" else synth.tinyllama_11b(mf.Q4_K)
path = weights.causal_lm_file(cfg, "/tmp/mllm_amd_cache"); ids = synth.causal_lm_ids(cfg, 64)
name = sys.argv[1]
for v in [int(a) for a in sys.argv[2:]]:
    lib.set_option(name, v)
    m = lib.Model(cfg, path)
    r = []
    for rep in range(3):
        m.clear_kvcache()
        tok, _, _ = m.prefill(ids, None, None, want_logits=False)
        gen, ms = m.generate(tok, 256)
        r.append(1000 * 256 / ms)
    kinds, _ = m.time_step(int(gen[-1]), 8)
    print(name, v, ' '.join('%.1f' % x for x in r), 'ids crc', zlib.crc32(gen.tobytes()), {k: round(u, 2) for k, (u, n) in kinds.items()}, flush=True)
    m.close()
lib.set_option(name, -1)
