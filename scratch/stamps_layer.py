"""Timeline of the layer launch (dec_layer_kernel, merge_o = 5) from in-kernel stamps -- diagnostic build scratch/stamps.sh -DMLLM_HIP_STAMPS_CHAIN.
first-role workgroups (gate|up -> down -> q|k|v): 0 entry, 5 gate|up done (act pairs written), 6 act pairs arrived, 7 quantised + weight stage landed (barrier), 1 down exit
(x pairs written), 2 x pairs arrived (q|k|v), 3 RMSNorm + Q8_K done, 8 q|k|v exit.  attention: 2 q|k|v arrived, 3 rotary done, 4 walk done, 1 exit.  o-projection: 2 pairs
arrived, 3 quantised, 4 dots done, 1 exit."""
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
try:
    gen, ms = m.generate(tok, steps)
    print('us per token %.1f' % (ms * 1e3 / steps))
except Exception as e:
    print('ERR', e)
buf = np.zeros(8192 * 16, dtype=np.uint64)
assert lib.load().mllm_hip_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size)) == 0
st = buf.reshape(-1, 16).astype(np.int64)[2048:4096]
n = int((st[:, 0] > 0).sum()); st = st[:n]
t0 = st[:, 0].min()
rel = (st - t0) / 100.0
rel[st == 0] = np.nan
print('workgroups stamped', n, ' launch span %.2f us' % np.nanmax(rel))
def show(name, lo, hi, slots, names):
    r = rel[lo:hi]
    print('%s: workgroups %d..%d' % (name, lo, hi - 1))
    for sl, nm in zip(slots, names):
        c = r[:, sl]
        if np.all(np.isnan(c)): continue
        print('   %-40s min %6.2f  median %6.2f  max %6.2f us' % (nm, np.nanmin(c), np.nanmedian(c), np.nanmax(c)))
show('gate|up -> down (-> q|k|v: first 128)', 0, 256, [0, 5, 6, 7, 1, 2, 3, 8],
     ['entry', 'gate|up done (act pairs written)', 'act pairs arrived', 'Q8_K + weight stage landed (barrier)', 'down exit (x pairs written)', 'x pairs arrived (q|k|v)', 'RMSNorm + Q8_K done', 'q|k|v exit'])
show('attention region', 256, 512, [0, 2, 3, 4, 1], ['entry', 'q|k|v pairs arrived', 'rotary + append done', 'walk done', 'exit'])
show('o-projection', 512, n, [0, 2, 3, 4, 1], ['entry', 'attention pairs arrived', 'quantised', 'dots done', 'exit'])
