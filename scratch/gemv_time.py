import sys, numpy as np
sys.path.insert(0, '.')
from mllm_amd import lib, synth
from tests.fixtures import weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32,32), 24)
m = lib.Qwen2VL(cfg, path)
tok,_,ms = m.prefill(ids, None, None, want_logits=False)
gen,_ = m.generate(tok, 8)
names={0:'gemv gate|up',1:'gemv down',2:'gemv qkv',3:'gemv o',10:'dec_qkv',11:'dec_attn',12:'dec_oproj',13:'dec_gateup',14:'dec_down'}
for which in [0,1,2,3,10,11,12,13,14]:
    ms, nb = m.time_gemv(which, 280); print('%-14s us %6.2f  GB/s %6.0f'%(names[which], ms*1000, nb/ms/1e6))
gen, ms = m.generate(int(gen[-1]), 64); print('generate 64: ms/token', ms/64, 'tok/s', 64000/ms)
