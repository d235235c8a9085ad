"""LLaVA-1.5-7B decode for values of one option (python scratch/llava_opt.py <option> v1 v2 ...): tok/s over 80 steps, ids crc, launch by launch"""
import sys, zlib
sys.path.insert(0, '.')
import numpy as np
from mllm_amd import lib, synth
from mllm_amd import synthfile as weights
cfg = synth.llava_7b()
path = weights.llava_file(cfg, "/tmp/mllm_amd_cache")
ids, pix = synth.llava_inputs(cfg)
name = sys.argv[1]
for v in [int(a) for a in sys.argv[2:]]:
    lib.set_option(name, v)
    m = lib.Model(cfg, path)
    r = []
    for rep in range(2):
        m.clear_kvcache()
        tok, _, _ = m.prefill(ids, pix, None, want_logits=False)
        gen, ms = m.generate(tok, 80)
        r.append(1000 * 80 / ms)
    kinds, _ = m.time_step(int(gen[-1]), 6)
    print(name, v, ' '.join('%.1f' % x for x in r), 'ids crc', zlib.crc32(gen.tobytes()), {k: round(u, 2) for k, (u, n) in kinds.items()}, flush=True)
    m.close()
lib.set_option(name, -1)
