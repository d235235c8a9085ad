"""generate() tok/s over the bench's 256 steps (T 298 -> 554) for attention flag sets: 7 old kernel, 3 pipelined, 11 pipelined + L2 warm-up from dec_qkv (default)."""
import sys
sys.path.insert(0, '.')
import numpy as np
from mllm_amd import lib, synth
from tests.fixtures import weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
for flags in (7, 3, 11, 7, 11):
    lib.set_option("attn_flags", flags)
    m = lib.Qwen2VL(cfg, path)
    tok, _, _ = m.prefill(ids, pix, grid, want_logits=False)
    gen, ms = m.generate(tok, 16)
    gen, ms = m.generate(int(gen[-1]), 256)
    print(f"flags {flags}: 256 steps {1000 * 256 / ms:.1f} tok/s", flush=True)
    m.close()
lib.set_option("attn_flags", -1)
