"""A/B timing of the decode attention launch (HIP events, cycling over the 28 layers) at several context lengths: pipelined (default) vs un-pipelined (attn_flags bit 2)."""
import sys
sys.path.insert(0, '.')
import numpy as np
from mllm_amd import lib, synth
from tests.fixtures import weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
for flags in (7, 15, 3, 11):
    lib.set_option("attn_flags", flags)
    m = lib.Qwen2VL(cfg, path)
    tok, _, _ = m.prefill(ids, pix, grid, want_logits=False)
    T = 282
    for target in (290, 432, 554, 700):
        gen, ms = m.generate(tok, target - T); tok = int(gen[-1]); T = target
        us = np.median([m.time_kernel(11, 28)[0] * 1000 for _ in range(5)])
        print(f"flags {flags} T {T}: dec_attn {us:.2f} us   (generate {1000 * len(gen) / ms:.0f} tok/s)", flush=True)
    m.close()
lib.set_option("attn_flags", -1)
