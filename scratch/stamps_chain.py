"""Timeline of the chain launch (dec_down_front_kernel: down(l) | q|k|v(l+1) | attention(l+1) | o-projection(l+1)) from in-kernel stamps -- diagnostic build
scratch/stamps.sh -DMLLM_HIP_STAMPS_CHAIN.  s_memrealtime (100 MHz) of thread 0 of every workgroup: slot 0 entry, 1 exit, role milestones:
  down: 2 activation quantised, 3 weight DMA landed (barrier), 4 super-block sums emitted (barrier)
  qkv : 2 x pairs arrived, 3 RMSNorm + Q8_K done
  attn: 2 q|k|v pairs arrived, 3 rotary + append done (prologue barrier), 4 attention walk done
  o   : 2 attention pairs arrived (wave 0), 3 quantised (barrier), 4 dots done
The stamps are those of the LAST chain launch of the last step (layer 26 -> 27)."""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
from mllm_amd import lib
import os; lib.SO_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libmllm_hip_stamps.so')
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
live = st[:, 0] > 0
n = int(live.sum()); st = st[:n]
H, I = cfg.hidden, cfg.inter
# role extents as the launcher computes them (2 B model): down 256 workgroups of 6 rows, q|k|v ((2048 + 1) / 2 + 7) / 8 = 128, attention 24 + warmers, o-projection the rest
t0 = st[:, 0].min()
rel = (st - t0) / 100.0
rel[st == 0] = np.nan
print('workgroups stamped', n, ' launch span %.2f us (first entry -> last exit)' % np.nanmax(rel[:, 1]))
# find the role boundaries from the stamps themselves: roles differ in which slots they fill; print per contiguous range given on the command line or guessed
def show(name, lo, hi, slots, names):
    r = rel[lo:hi]
    print('%s: workgroups %d..%d' % (name, lo, hi - 1))
    for sl, nm in zip(slots, names):
        c = r[:, sl]
        print('   %-34s min %6.2f  median %6.2f  max %6.2f us' % (nm, np.nanmin(c), np.nanmedian(c), np.nanmax(c)))
gd = int(sys.argv[2]) if len(sys.argv) > 2 else 256
gq = int(sys.argv[3]) if len(sys.argv) > 3 else 128
ga = int(sys.argv[4]) if len(sys.argv) > 4 else n - gd - gq - 192
show('down', 0, gd, [0, 2, 3, 4, 1], ['entry', 'activation quantised', 'weight DMA landed', 'super-block sums emitted', 'exit (pairs written)'])
show('q|k|v', gd, gd + gq, [0, 2, 3, 1], ['entry', 'x pairs arrived', 'RMSNorm + Q8_K done', 'exit (pairs written)'])
show('attention (first 24)', gd + gq, gd + gq + 24, [0, 2, 3, 4, 1], ['entry', 'q|k|v pairs arrived', 'rotary + append done', 'walk done', 'exit (pairs written)'])
if ga > 24: show('warmers', gd + gq + 24, gd + gq + ga, [0, 1], ['entry', 'exit'])
show('o-projection', gd + gq + ga, n, [0, 2, 3, 4, 1], ['entry', 'attention pairs arrived', 'quantised', 'dots done', 'exit (pairs written)'])
