"""A/B of the merged attention + o-projection decode launch (option merge_o) on the 2 B model: same prompt, ids and logits compared, tok/s of 256 generated tokens."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from mllm_amd import lib, synth
from mllm_amd import synthfile as weights

cfg = synth.qwen2vl_2b()
path = weights.qwen2vl_file(cfg)
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
res = {}
for mode in (0, 2, 12, 22, 32):
    lib.set_option("merge_o", mode)
    m = lib.Qwen2VL(cfg, path)
    tok, logits, _ = m.prefill(ids, pix, grid)
    rows = []
    t = tok
    for _ in range(6):
        t, lg, _ = m.decode(t)
        rows.append(lg.copy())
    toks, ms = m.generate(t, 256)
    toks2, ms2 = m.generate(int(toks[-1]), 256)
    print(f"merge_o={mode}: {256e3 / ms:.1f} tok/s, then {256e3 / ms2:.1f} tok/s  (first ids {toks[:6].tolist()})", flush=True)
    res.setdefault(mode, []).append((np.stack(rows), toks, toks2))
    m.close()
a, b = res[0][0], res[2][0]
print("logits equal:", np.array_equal(a[0], b[0]), " ids equal:", np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]))
