"""Timeline of the layer launch (merge_o = 5: down -> q|k|v | tail: attention / o-projection / nothing -> gate|up) from in-kernel stamps (scratch/stamps.sh -DMLLM_HIP_STAMPS_CHAIN)."""
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
print('us per token %.1f' % (ms * 1e3 / steps))
buf = np.zeros(8192 * 16, dtype=np.uint64)
assert lib.load().mllm_hip_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size)) == 0
st = buf.reshape(-1, 16).astype(np.int64)[2048:4096]
n = int((st[:, 0] > 0).sum()); st = st[:n]
t0 = st[:, 0].min()
rel = (st - t0) / 100.0
rel[st == 0] = np.nan
print('workgroups stamped', n, ' launch span %.2f us' % np.nanmax(rel))
def show(name, rows, slots, names):
    r = rel[rows]
    print('%s: %d workgroups' % (name, len(rows)))
    for sl, nm in zip(slots, names):
        c = r[:, sl]
        if np.all(np.isnan(c)): continue
        print('   %-40s min %6.2f  median %6.2f  max %6.2f us' % (nm, np.nanmin(c), np.nanmedian(c), np.nanmax(c)))
show('down (-> q|k|v: first 128)', list(range(256)), [0, 2, 3, 4, 1, 5], ['entry', 'activation quantised', 'weight DMA landed', 'sums emitted', 'down exit (x pairs written)', 'q|k|v exit'])
tail = list(range(256, n))
live = [256 + r * 8 + c for c in range(2) for r in range(12)]
rest = [t for t in tail if t not in live]
opj = [t for t in rest if not np.isnan(rel[t, 4])]
non = [t for t in rest if np.isnan(rel[t, 4])]
show('tail: attention -> gate|up', live, [0, 2, 3, 4, 1, 6, 7], ['entry', 'q|k|v pairs arrived', 'rotary + append done', 'walk done', 'attention end', 't pairs arrived', 'gate|up done'])
show('tail: o-projection -> gate|up', opj, [0, 2, 3, 4, 1, 6, 7], ['entry', 'attention pairs arrived', 'quantised', 'dots done', 'o-projection end', 't pairs arrived', 'gate|up done'])
show('tail: gate|up only', non, [0, 6, 7], ['entry', 't pairs arrived', 'gate|up done'])
