"""LLaVA-1.5-7B decode on two builds of the library (MLLM_SO_A / MLLM_SO_B), alternating processes: tok/s over the steps the 700-entry cache leaves, ids crc, launch by launch"""
import os, sys, subprocess
code = '''
import sys, zlib
sys.path.insert(0, '.')
import numpy as np
from mllm_amd import lib, synth
lib.SO_PATH = sys.argv[1]
from mllm_amd import synthfile as weights
cfg = synth.llava_7b()
path = weights.llava_file(cfg, "/tmp/mllm_amd_cache")
ids, pix = synth.llava_inputs(cfg)
m = lib.Model(cfg, path)
r = []
for rep in range(2):
    m.clear_kvcache()
    tok, _, _ = m.prefill(ids, pix, None, want_logits=False)
    gen, ms = m.generate(tok, 80)
    r.append(1000 * 80 / ms)
kinds, _ = m.time_step(int(gen[-1]), 6)
print(' '.join('%.1f' % x for x in r), 'ids crc', zlib.crc32(gen.tobytes()), {k: round(u, 2) for k, (u, n) in kinds.items()})
'''
a, b = os.environ['MLLM_SO_A'], os.environ['MLLM_SO_B']
for rnd in range(2):
    for name, so in (('A', a), ('B', b)):
        out = subprocess.run([sys.executable, '-c', code, os.path.abspath(so)], capture_output=True, text=True)
        print(name, os.path.basename(so), out.stdout.strip()[-400:], out.stderr.strip()[-400:] if out.returncode else '', flush=True)
