import sys, numpy as np
sys.path.insert(0, '.')
from mllm_amd import lib, synth, weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32,32), 24)
m = lib.Qwen2VL(cfg, path)
tok,_,ms = m.prefill(ids, None, None, want_logits=False) if len(sys.argv)>1 and sys.argv[1]=='text' else m.prefill(ids, pix, grid, want_logits=False)
gen, ms = m.generate(tok, 16)
gen, ms = m.generate(int(gen[-1]), 64); print('generate 64: ms/token', ms/64, 'tok/s', 64000/ms)
