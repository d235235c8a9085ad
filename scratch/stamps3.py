import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '.')
from mllm_amd import lib
lib.SO_PATH = os.path.abspath(os.environ.get('STAMPSO', 'scratch/libmllm_hip_stamps.so'))
from mllm_amd import synth, weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
ids = (np.arange(int(os.environ.get('NTOK','8'))) * 7919 % 150000).astype(np.int32)
m = lib.Qwen2VL(cfg, path)
tok,_,ms = m.prefill(ids, want_logits=False)
gen,_ = m.generate(tok, 8)
which = int(sys.argv[1]); nwg = int(sys.argv[2]); names = sys.argv[3].split(',')
ms, nb = m.time_gemv(which, 56); print('which', which, 'us %.2f'%(ms*1000))
buf = np.zeros(8192*8, dtype=np.uint64)
assert lib.load().mllm_hip_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size)) == 0
st = buf.reshape(-1,8)[:nwg].astype(np.int64)
t0 = st[:,1].min()
rel = (st - t0) * 10.0 / 1000.0
for i,n in enumerate(names):
    c = rel[:,i]; print('%-18s min %.2f  median %.2f  p90 %.2f  max %.2f us'%(n, c.min(), np.median(c), np.percentile(c,90), c.max()))
if len(names) >= 8:
    clk = (st[:, 7] - st[:, 6]).astype(np.float64); rt = (st[:, 5] - st[:, 0]).astype(np.float64) * 10.0
    print('shader clocks per ns (GHz):', np.median(clk / rt), 'clk', np.median(clk), 'ns', np.median(rt))
