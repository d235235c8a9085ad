import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '.')
from mllm_amd import lib
lib.SO_PATH = '/tmp/libmllm_hip_stamps.so'
from mllm_amd import synth
from tests.fixtures import weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32,32), 24)
m = lib.Qwen2VL(cfg, path)
tok,_,ms = m.prefill(ids, None, None, want_logits=False)
gen,_ = m.generate(tok, 8)
which = int(sys.argv[1]); nwg = int(sys.argv[2])
ms, nb = m.time_gemv(which, 56); print('which', which, 'us %.2f'%(ms*1000))
buf = np.zeros(8192*8, dtype=np.uint64)
assert lib.load().mllm_hip_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size)) == 0
st = buf.reshape(-1,8)[:nwg].astype(np.int64)
t0 = st[:,0].min()
rel = (st - t0) * 10.0 / 1000.0   # us (100 MHz)
names = ['start','issued','x landed','prologue done','weights landed','dot done']
for i,n in enumerate(names):
    c = rel[:,i]; print('%-16s min %.2f  median %.2f  p90 %.2f  max %.2f us'%(n, c.min(), np.median(c), np.percentile(c,90), c.max()))
d = rel[:,1:6]-rel[:,0:5]
for i,n in enumerate(['issue','wait x','prologue','wait weights','dot']): print('  seg %-14s median %.2f  max %.2f'%(n, np.median(d[:,i]), d[:,i].max()))
