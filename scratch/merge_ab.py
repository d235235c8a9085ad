"""A/B of the merged decode launches (options merge_o: attention + o-projection; merge_qkv: down + next layer's q|k|v) on the 2 B model: same prompt, ids and logits compared,
tok/s of 256 + 256 generated tokens."""
import sys
import numpy as np
sys.path.insert(0, ".")
from mllm_amd import lib, synth
from mllm_amd import synthfile as weights

cfg = synth.qwen2vl_2b()
path = weights.qwen2vl_file(cfg)
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
res = {}
for mo, mq in ((0, 0), (4, 0), (40, 0), (4, 0), (40, 0)):
    lib.set_option("merge_o", mo)
    m = lib.Qwen2VL(cfg, path)
    tok, logits, _ = m.prefill(ids, pix, grid)
    rows, t = [], tok
    for _ in range(6):
        t, lg, _ = m.decode(t)
        rows.append(lg.copy())
    toks, ms = m.generate(t, 256)
    toks2, ms2 = m.generate(int(toks[-1]), 256)
    print(f"merge_o={mo} merge_qkv={mq}: {256e3 / ms:.1f} tok/s, then {256e3 / ms2:.1f} tok/s", flush=True)
    res[(mo, mq)] = (np.stack(rows), toks, toks2)
    m.close()
base = res[(0, 0)]
for k, v in res.items():
    print(k, "logits equal:", np.array_equal(base[0], v[0]), " ids equal:", np.array_equal(base[1], v[1]) and np.array_equal(base[2], v[2]))
