"""In-kernel stamps of the pipelined decode attention (diagnostic build, scratch/stamps.sh): where a workgroup's time goes.
slots: 1 entry, 0 prologue barrier passed, 8 producer 0: scores of its first block done, 9 carry taken, 10 block READY, 12/13 the same for producer 4,
2 walker: block 0 READY and its first reads issued, 3 walk over the slab keys done, 4 walker at the final barrier, 7 logsum lane done, 5 kernel end"""
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
st = buf.reshape(-1, 16).astype(np.int64)
st = st[st[:, 1] > 0][:24]
order = [1, 0, 8, 9, 10, 12, 13, 2, 3, 7, 4, 5]
names = ['entry', 'prologue done', 'P0 scores', 'P0 carry', 'P0 READY', 'P4 carry', 'P4 READY', 'walker starts', 'walk done', 'logsum done', 'walker at barrier', 'kernel end']
t0 = st[:, 1].min()
rel = (st - t0) / 100.0
for i, n in zip(order, names):
    c = rel[:, i]; print('%-18s min %.2f  median %.2f  max %.2f us' % (n, c.min(), np.median(c), c.max()))

allst = buf.reshape(-1, 16).astype(np.int64)
print('XCC of dec_attn blocks 0..23:', allst[:24, 14].tolist())
print('XCC of dec_qkv  blocks 0..23:', allst[:24, 15].tolist())
print('same XCD for equal block index (first 96):', int((allst[:96, 14] == allst[:96, 15]).sum()), 'of 96')

blk = allst[4096:4096 + 2, :12]
for w in range(2):
    print('walker block-end times, wg', w, ':', [round((x - t0) / 100.0, 2) for x in blk[w].tolist()], 'masks of moved blocks not shown')
