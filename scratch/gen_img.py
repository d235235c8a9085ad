import sys, numpy as np
sys.path.insert(0, '.')
from mllm_amd import lib, synth, weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32,32), 24)
m = lib.Qwen2VL(cfg, path)
tok,_,ms = m.prefill(ids, pix, grid, want_logits=False)
gen,_ = m.generate(tok, 8)
names={10:'dec_qkv',11:'dec_attn',12:'dec_oproj',13:'dec_gateup',14:'dec_down'}
for which in [10,11,12,13,14]:
    ms, nb = m.time_gemv(which, 280); print('%-14s us %6.2f  GB/s %6.0f'%(names[which], ms*1000, nb/ms/1e6))
gen, ms = m.generate(int(gen[-1]), 128); print('generate 128 at T~300: ms/token', ms/128, 'tok/s', 128000/ms)
