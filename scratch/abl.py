import sys, numpy as np
sys.path.insert(0, '.')
from mllm_amd import lib, synth, weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32,32), 24)
m = lib.Qwen2VL(cfg, path)
tok,_,ms = m.prefill(ids, None, None, want_logits=False)
gen,_ = m.generate(tok, 8)
for which in [int(a) for a in sys.argv[1:]]:
    ms, nb = m.time_gemv(which, 280); print('which %d us %6.2f  GB/s %6.0f'%(which, ms*1000, nb/ms/1e6))
