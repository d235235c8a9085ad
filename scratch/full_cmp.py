import sys, time, numpy as np, os
sys.path.insert(0, '.')
from mllm_amd import lib, synth, weights
t=time.time()
cfg = synth.qwen2vl_2b()
path = weights.qwen2vl_file(cfg)
print('weights', time.time()-t, os.path.getsize(path)/1e6, 'MB', 'cpus', os.cpu_count(), flush=True)
g = np.load('tests/golden/qwen2vl_2b_ref.npz')
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32,32), 24)
t=time.time(); m = lib.Qwen2VL(cfg, path); print('load', time.time()-t, flush=True)
for rep in range(2):
    m.clear_kvcache()
    tok, logits, ms = m.prefill(ids, pix, grid)
    print('prefill ms', ms, 'tok', tok, flush=True)
toks=[tok]; errs=[]
steps = {int(s):i for i,s in enumerate(g['steps'])}
def cmp(step, logits):
    i = steps[step]; idx=g['top_idx'][i]; val=g['top_val'][i]
    e1 = np.abs(logits[idx]-val).max(); e2 = np.abs(logits[::97]-g['strided'][i]).max()
    return max(e1,e2), int(np.argsort(-logits,kind='stable')[0]==idx[0])
errs.append(cmp(0, logits))
dec=[]
for s in range(1, len(g['tokens'])):
    tok, logits, ms = m.decode(tok); toks.append(tok); dec.append(ms)
    if s in steps: errs.append(cmp(s, logits))
print('tokens match', toks == g['tokens'].tolist(), sum(a==b for a,b in zip(toks,g['tokens'].tolist())), '/', len(toks))
print('mine', toks[:50]); print('ref ', g['tokens'].tolist()[:50])
print('logit errs at dumped steps', errs)
print('decode ms mean', np.mean(dec), 'tok/s', 1000/np.mean(dec))
m.clear_kvcache(); tok,_,_ = m.prefill(ids, pix, grid)
gen, ms = m.generate(tok, 64); print('generate 64: ms/token', ms/64, 'tok/s', 64000/ms)
for which in range(4):
    ms, nb = m.time_gemv(which, 50); print('gemv', which, 'us', ms*1000, 'GB/s', nb/ms/1e6)
