"""chain_cont A/B in one process: the q|k|v role of the chain launch as workgroups of its own (0) or carried on by the first down-projection workgroups (1)"""
import sys, zlib
sys.path.insert(0, '.')
import numpy as np
from mllm_amd import lib, synth
from mllm_amd import synthfile as weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg, cache_dir="/tmp/mllm_amd_cache")
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
for mode in (0, 1, 0, 1, 0, 1):
    lib.set_option("chain_cont", mode)
    m = lib.Qwen2VL(cfg, path)
    r = []
    for rep in range(3):
        m.clear_kvcache()
        tok, _, _ = m.prefill(ids, pix, grid, want_logits=False)
        g0, _ = m.generate(tok, 16)
        gen, ms = m.generate(int(g0[-1]), 256)
        r.append(1000 * 256 / ms)
    kinds, _ = m.time_step(int(gen[-1]), 8)
    print('chain_cont', mode, ' '.join('%.1f' % x for x in r), 'ids crc', zlib.crc32(np.concatenate([g0, gen]).tobytes()), 'chain us %.2f' % kinds['chain'][0], flush=True)
    m.close()
lib.set_option("chain_cont", -1)
