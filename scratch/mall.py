import sys, os, numpy as np
sys.path.insert(0, '.')
from mllm_amd import lib
if os.environ.get('DBGSO'): lib.SO_PATH = os.path.abspath(os.environ['DBGSO'])
from mllm_amd import synth, weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
m = lib.Qwen2VL(cfg, path)
ids = (np.arange(280) * 7919 % 150000).astype(np.int32)
tok, _, _ = m.prefill(ids, want_logits=False); m.generate(tok, 4)
for nl in (28, 8, 2, 1):
    os.environ['MLLM_HIP_TIME_LAYERS'] = str(nl)
    print('layers cycled', nl, ' '.join('%s %.2f' % (n, m.time_gemv(w, 112)[0] * 1000) for n, w in (('qkv', 10), ('attn', 11), ('oproj', 12), ('gateup', 13), ('down', 14))))
