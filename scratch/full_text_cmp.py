import sys, numpy as np
sys.path.insert(0, '.')
from mllm_amd import lib, synth, weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
g = np.load('tests/golden/qwen2vl_2b_ref_text.npz')
m = lib.Qwen2VL(cfg, path)
tok, logits, ms = m.prefill(g['ids'])
steps = {int(s):i for i,s in enumerate(g['steps'])}
def cmp(step, logits):
    i = steps[step]; idx=g['top_idx'][i]; val=g['top_val'][i]
    return float(max(np.abs(logits[idx]-val).max(), np.abs(logits[::97]-g['strided'][i]).max()))
errs=[cmp(0,logits)]; toks=[tok]
for s in range(1, len(g['tokens'])):
    tok, logits, _ = m.decode(tok); toks.append(tok)
    if s in steps: errs.append(cmp(s, logits))
print('text-only full size: tokens match', toks == g['tokens'].tolist(), sum(a==b for a,b in zip(toks,g['tokens'].tolist())),'/',len(toks))
print('logit errs', errs, 'ref top1-top2 margins', [float(g['top_val'][i][0]-g['top_val'][i][1]) for i in range(len(errs))])
