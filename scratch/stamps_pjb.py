"""In-kernel stamps of dec_proj_blk on the down projection (diagnostic build: scratch/variant.sh pjb "-DMLLM_HIP_STAMPS_PJB"): 0 entry, 1 activation row + weight rows landed, 2 prologue (RMSNorm + Q8_K) done,
3 weights waited for, 4 tables emitted, 5 chains walked, 6 end.  Thread 0 of every workgroup, last launch of the replayed step (layer 27)."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '.')
from mllm_amd import lib
lib.SO_PATH = os.environ["MLLM_SO"]
from mllm_amd import synth
from tests.fixtures import weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
m = lib.Qwen2VL(cfg, path)
tok, _, ms = m.prefill(ids, pix, grid, want_logits=False)
gen, ms = m.generate(tok, 32)
print('tok/s %.0f' % (1000 * len(gen) / ms))
buf = np.zeros(8192 * 16, dtype=np.uint64)
assert lib.load().mllm_hip_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size)) == 0
st = buf.reshape(-1, 16).astype(np.int64)[:256, :8]
t0 = st[:, 0].min()
rel = (st - t0) / 100.0
names = ["entry", "x row landed", "Q8_K of the row", "weights waited", "barrier", "tables emitted", "barrier", "chains walked, stored"]
for i, n in enumerate(names):
    c = rel[:, i]; print('%-16s min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f us' % (n, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
d = np.diff(rel, axis=1)
print('per-workgroup intervals (median):', ['%.2f' % np.median(d[:, i]) for i in range(7)])
