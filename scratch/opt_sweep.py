"""tok/s of generate() over 256 steps for values of one library option: python scratch/opt_sweep.py <option> v1 v2 ..."""
import sys
sys.path.insert(0, '.')
import numpy as np
from mllm_amd import lib, synth
from mllm_amd import synthfile as weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg, cache_dir="/tmp/mllm_amd_cache")
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
name = sys.argv[1]
ref = None
for v in [int(a) for a in sys.argv[2:]]:
    lib.set_option(name, v)
    m = lib.Qwen2VL(cfg, path)
    rates = []
    for rep in range(3):
        m.clear_kvcache()
        tok, _, _ = m.prefill(ids, pix, grid, want_logits=False)
        gen, ms = m.generate(tok, 256)
        rates.append(1000 * len(gen) / ms)
    kinds, _ = m.time_step(int(gen[-1]), 8)
    g = np.asarray(gen)
    if ref is None: ref = g
    print(f"{name} = {v}: {np.median(rates):.1f} tok/s  ids equal first: {np.array_equal(g, ref)}  head {kinds['head'][0]:.2f} us", flush=True)
    m.close()
lib.set_option(name, -1)
