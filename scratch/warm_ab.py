"""A/B of the weight-warming workgroups of the decode attention launch (attn_flags bits 4 / 5): generate() over the bench's 256 steps, ids compared with the default."""
import sys
sys.path.insert(0, '.')
import numpy as np
from mllm_amd import lib, synth
import os
if os.environ.get('MLLM_SO'): lib.SO_PATH = os.environ['MLLM_SO']
from tests.fixtures import weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
flag_list = [int(a) for a in sys.argv[1:]] or [11, 27, 59, 43, 11, 27]
ref = None
for flags in flag_list:
    lib.set_option("attn_flags", flags)
    m = lib.Qwen2VL(cfg, path)
    tok, _, _ = m.prefill(ids, pix, grid, want_logits=False)
    gen, ms = m.generate(tok, 16)
    rates = []
    for rep in range(3):
        m.clear_kvcache()
        tok, _, _ = m.prefill(ids, pix, grid, want_logits=False)
        gen, ms = m.generate(tok, 256)
        rates.append(1000 * len(gen) / ms)
    g = np.asarray(gen)
    if ref is None: ref = g
    print(f"flags {flags}: generate 256 steps {np.median(rates):.1f} tok/s  (runs {', '.join(f'{r:.0f}' for r in rates)})  ids equal default: {np.array_equal(g, ref)}", flush=True)
    m.close()
lib.set_option("attn_flags", -1)
