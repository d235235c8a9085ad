"""Same-box A/B of two builds of the library on the bench's decode (2 B model, 448 x 448 image + 24-token prompt, 16 + 256 steps): MLLM_SO_A / MLLM_SO_B are the two .so
paths; each runs in its own process, alternating, three times; prints tok/s and a hash of the generated ids (must be equal)."""
import os, sys, subprocess
code = '''
import sys, zlib
sys.path.insert(0, '.')
import numpy as np
from mllm_amd import lib, synth
lib.SO_PATH = sys.argv[1]
from mllm_amd import synthfile as weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg, cache_dir="/tmp/mllm_amd_cache")
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
m = lib.Qwen2VL(cfg, path)
r = []
for rep in range(3):
    m.clear_kvcache()
    tok, _, _ = m.prefill(ids, pix, grid, want_logits=False)
    g0, _ = m.generate(tok, 16)
    gen, ms = m.generate(int(g0[-1]), 256)
    r.append(1000 * 256 / ms)
print(' '.join('%.1f' % x for x in r), 'ids crc', zlib.crc32(np.concatenate([g0, gen]).tobytes()))
'''
a, b = os.environ['MLLM_SO_A'], os.environ['MLLM_SO_B']
for rnd in range(3):
    for name, so in (('A', a), ('B', b)):
        out = subprocess.run([sys.executable, '-c', code, os.path.abspath(so)], capture_output=True, text=True)
        print(name, os.path.basename(so), out.stdout.strip()[-200:], out.stderr.strip()[-300:] if out.returncode else '', flush=True)
