"""merge_o 4 (chain launch + gate|up launch) against 5 (the layer launch: gate|up -> down -> q|k|v carried on by the same workgroups): 2 B model, the bench's decode"""
import sys, zlib
sys.path.insert(0, '.')
import numpy as np
from mllm_amd import lib, synth
from mllm_amd import synthfile as weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg, cache_dir="/tmp/mllm_amd_cache")
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
modes = [int(a) for a in sys.argv[1:]] or [4, 5, 4, 5]
for mode in modes:
    lib.set_option("merge_o", mode)
    m = lib.Qwen2VL(cfg, path)
    r = []
    for rep in range(3):
        m.clear_kvcache()
        tok, _, _ = m.prefill(ids, pix, grid, want_logits=False)
        g0, _ = m.generate(tok, 16)
        gen, ms = m.generate(int(g0[-1]), 256)
        r.append(1000 * 256 / ms)
    kinds, _ = m.time_step(int(gen[-1]), 8)
    print('merge_o', mode, ' '.join('%.1f' % x for x in r), 'ids crc', zlib.crc32(np.concatenate([g0, gen]).tobytes()), {k: (round(u, 2), n) for k, (u, n) in kinds.items()}, flush=True)
    m.close()
lib.set_option("merge_o", -1)
