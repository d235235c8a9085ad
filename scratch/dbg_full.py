import sys, numpy as np, time
sys.path.insert(0, '.')
from mllm_amd import lib, synth, weights
from oracle import models
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
g = np.load('tests/golden/qwen2vl_2b_ref_text.npz')
m = lib.Qwen2VL(cfg, path)
w = models.Weights(path); o = models.LLM(w, cfg)
tok, lg, _ = m.prefill(g['ids'])
t = time.time(); ol = o.prefill(g['ids']); print('oracle prefill s', time.time() - t)
print('step 0 tok', tok, int(ol.argmax()), 'maxdiff', float(np.abs(lg - ol).max()), 'ndiff', int((lg != ol).sum()))
for s in range(1, 8):
    tok_o = int(ol.argmax())
    tok, lg, _ = m.decode(tok_o)
    ol = o.decode(tok_o)
    d = np.abs(lg - ol)
    print('step', s, 'tok', tok, int(ol.argmax()), 'maxdiff', float(d.max()), 'ndiff', int((lg != ol).sum()), 'ref', int(g['tokens'][s]))
