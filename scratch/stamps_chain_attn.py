"""The attention role inside the chain launch, from the attention's own stamps (diagnostic build scratch/stamps.sh -DMLLM_HIP_STAMPS_CHAIN, which also defines MLLM_HIP_STAMPS):
rows = blockIdx.x of the launch; slots: 1 entry, 0 prologue barrier passed (q|k|v arrived, rotary + append done), 8 producer 0: scores of its first block done, 9 carry taken,
10 block READY, 12 / 13 the same for producer 4, 2 walker: block 0 READY and its first reads issued, 3 walk done, 7 logsum lane done, 4 walker at the final barrier, 5 end.
Rows 4096 + blockIdx.x: the walker's block-end times.  Chain rows (2048 + blockIdx.x): 0 entry, 2 q|k|v pairs arrived."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '.')
from mllm_amd import lib
lib.SO_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libmllm_hip_stamps.so')
from mllm_amd import synth
from mllm_amd import synthfile as weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg, cache_dir="/tmp/mllm_amd_cache")
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
m = lib.Qwen2VL(cfg, path)
tok, _, _ = m.prefill(ids, pix, grid, want_logits=False)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
gen, ms = m.generate(tok, steps)
print('us per token %.1f, T = %d' % (ms * 1e3 / steps, 282 + steps))
buf = np.zeros(8192 * 16, dtype=np.uint64)
assert lib.load().mllm_hip_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size)) == 0
st = buf.reshape(-1, 16).astype(np.int64)
nz = [r for r in range(0, 1024) if st[r, 3] > 0 and st[r, 2] > 0]
print('rows with walker stamps:', nz[:40], len(nz))
base = 256      # chain_cont: the attention region follows the 256 down-projection workgroups
live = [r for r in nz if r >= base][:24]
t0 = st[2048:2048 + 256, 0].min()      # the launch's first entry
rel = lambda a: (a - t0) / 100.0
A = st[live]; Cn = st[[2048 + r for r in live]]
def line(name, col):
    c = rel(col); print('   %-46s min %6.2f  median %6.2f  max %6.2f us' % (name, c.min(), np.median(c), c.max()))
line('entry', A[:, 1]); line('q|k|v pairs arrived (chain stamp)', Cn[:, 2]); line('prologue barrier passed', A[:, 0])
line('producer 0: key rows staged (ds_write issued)', A[:, 6]); line('producer 0: its key pieces read back', A[:, 11]); line('producer 0: scores of its first block done', A[:, 8]); line('producer 0: carry taken', A[:, 9]); line('producer 0: block READY', A[:, 10])
line('producer 4: carry taken', A[:, 12]); line('producer 4: block READY', A[:, 13])
line('walker: block 0 READY, first reads issued', A[:, 2]); line('walk done', A[:, 3]); line('logsum lane done', A[:, 7]); line('walker at the final barrier', A[:, 4]); line('end', A[:, 5])
for r in live[:2]:
    w = st[4096 + r, :13]
    print('walker block-end times, workgroup', r, [round(float(rel(x)), 2) for x in w.tolist() if x > 0])
