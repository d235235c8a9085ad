"""Per-kernel decode timings (HIP events, cold-HBM cycling over the layers) and generate() throughput at two context lengths.
usage: python scratch/dec_time.py [config]   config: qwen2vl (default) | qwen15 | llava | tinyllama"""
import sys, os, numpy as np
sys.path.insert(0, '.')
from mllm_amd import lib, synth, mllmfile as mf
from tests.fixtures import weights
which = sys.argv[1] if len(sys.argv) > 1 else 'qwen2vl'
if which == 'qwen2vl':
    cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg); pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
    m = lib.Qwen2VL(cfg, path); tok, _, ms = m.prefill(ids, pix, grid, want_logits=False)
elif which == 'qwen15':
    cfg = synth.qwen15_05b(); path = weights.causal_lm_file(cfg); m = lib.Model(cfg, path); tok, _, ms = m.prefill(synth.causal_lm_ids(cfg, 64), want_logits=False)
elif which == 'tinyllama':
    cfg = synth.tinyllama_11b(mf.Q4_K); path = weights.causal_lm_file(cfg); m = lib.Model(cfg, path); tok, _, ms = m.prefill(synth.causal_lm_ids(cfg, 64), want_logits=False)
else:
    cfg = synth.llava_7b(); path = weights.llava_file(cfg); ids, img = synth.llava_inputs(cfg); m = lib.Model(cfg, path); tok, _, ms = m.prefill(ids, img, want_logits=False)
print(which, 'prefill ms %.2f' % ms, 'load', m.load_stats())
names = {10: 'dec_qkv', 11: 'dec_attn', 12: 'dec_oproj', 13: 'dec_gateup', 14: 'dec_down'}
gen, _ = m.generate(tok, 8)
for rnd in range(2):
    tot = 0.0
    for k in [10, 11, 12, 13, 14]:
        ms, nb = m.time_kernel(k, 280); tot += ms * 1000
        print('  %-12s us %6.2f  GB/s %6.0f' % (names[k], ms * 1000, nb / ms / 1e6))
    print('  sum of five: %.2f us/layer' % tot)
    n = 128 if which != 'llava' else 32
    gen, ms = m.generate(int(gen[-1]), n); print('generate %d: ms/token %.4f tok/s %.1f' % (n, ms / n, n * 1000 / ms))
m.close()
