import sys, numpy as np
sys.path.insert(0, '.')
from mllm_amd import lib, synth, weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32,32), 24)
m = lib.Qwen2VL(cfg, path)
tok,_,ms = m.prefill(ids, pix, grid, want_logits=False)
for which,name in enumerate(['gate|up 17920x1536','down 1536x8960','qkv 2048x1536','o 1536x1536']):
    ms, nb = m.time_gemv(which, 200); print(name, 'us %.2f'%(ms*1000), 'GB/s %.0f'%(nb/ms/1e6))
gen, ms = m.generate(tok, 64); print('generate 64: ms/token', ms/64, 'tok/s', 64000/ms)
