"""In-kernel stamps of the decode attention (diagnostic build, scratch/stamps.sh): where a workgroup's time goes.
stamp slots: 1 entry, 0 head start (after rotary + barrier), 2 scores done, 3 softmax done, 4 V parked, 5/6 first/last walker done, 7 logsum lane done"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '.')
from mllm_amd import lib
lib.SO_PATH = '/tmp/libmllm_hip_stamps.so'
from mllm_amd import synth
from tests.fixtures import weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
m = lib.Qwen2VL(cfg, path)
tok, _, ms = m.prefill(ids, pix, grid, want_logits=False)
gen, _ = m.generate(tok, int(sys.argv[1]) if len(sys.argv) > 1 else 8)
ms, nb = m.time_kernel(11, 28); print('dec_attn us %.2f' % (ms * 1000))
buf = np.zeros(8192 * 16, dtype=np.uint64)
assert lib.load().mllm_hip_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size)) == 0
st = buf.reshape(-1, 16)[:12].astype(np.int64)
order = [1, 0, 2, 3, 4, 5, 6, 7]
names = ['entry', 'head start', 'scores done', 'softmax done', 'parked', 'walker0 done', 'walkerN done', 'logsum done']
t0 = st[:, 1].min()
rel = (st - t0) / 100.0
for i, n in zip(order, names):
    c = rel[:, i]; print('%-14s min %.2f  median %.2f  max %.2f us' % (n, c.min(), np.median(c), c.max()))
print('fast/slow blocks per chunk (wg 0..2):', st[:3, 8:16].tolist())
