"""256 + 256 generated tokens of the 2 B model with option merge_o = argv[1] (for rocprofv3 --kernel-trace --stats)."""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mllm_amd import lib, synth
from mllm_amd import synthfile as weights
cfg = synth.qwen2vl_2b()
path = weights.qwen2vl_file(cfg)
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
steps = 64
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "steps":
        steps = int(v)
    elif int(v) >= 0:
        lib.set_option(k, int(v))
m = lib.Qwen2VL(cfg, path)
tok, _, _ = m.prefill(ids, pix, grid)
toks, ms = m.generate(tok, steps)
toks, ms = m.generate(int(toks[-1]), steps)
print(f"{sys.argv[1:]}: {steps * 1e3 / ms:.1f} tok/s")
m.close()
