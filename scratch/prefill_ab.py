"""Prefill A/B on the 2 B model: option name=value pairs given on the command line are applied for the second model."""
import sys
import numpy as np
sys.path.insert(0, ".")
from mllm_amd import lib, synth
from mllm_amd import synthfile as weights
cfg = synth.qwen2vl_2b()
path = weights.qwen2vl_file(cfg)
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
opts = [kv.split("=") for kv in sys.argv[1:]]
ref = None
for variant in (0, 1, 0, 1):
    for k, v in opts:
        lib.set_option(k, int(v) if variant else -1)
    m = lib.Qwen2VL(cfg, path)
    ts = []
    for _ in range(7):
        m.clear_kvcache()
        tok, logits, ms = m.prefill(ids, pix, grid)
        ts.append(ms)
    if ref is None:
        ref = logits.copy()
    print(f"variant {variant}: prefill median {np.median(ts):.3f} ms  (min {min(ts):.3f})  logits equal: {np.array_equal(ref, logits)}", flush=True)
    m.close()
