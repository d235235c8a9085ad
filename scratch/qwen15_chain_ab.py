"""Qwen1.5-0.5B (inter 2816: 11 super-blocks per down row, NS = 2): the down projection as the one-lane-per-super-block kernel (option pjb_min_ns = 2) so that the chain launch
(down + next layer's q|k|v + attention + o-projection) applies, against the default (down on the register form, q|k|v + attention + o-projection as one launch)."""
import sys
sys.path.insert(0, '.')
import numpy as np
from mllm_amd import lib, synth
from mllm_amd import synthfile as weights
cfg = synth.qwen15_05b()
path = weights.causal_lm_file(cfg, "/tmp/mllm_amd_cache"); ids = synth.causal_lm_ids(cfg, 64)
ref = None
for mn in (-1, 2, -1, 2):
    lib.set_option("pjb_min_ns", mn)
    m = lib.Model(cfg, path)
    rates = []
    for rep in range(3):
        m.clear_kvcache()
        tok, _, _ = m.prefill(ids, None, None, want_logits=False)
        gen, ms = m.generate(tok, 256)
        rates.append(1000 * len(gen) / ms)
    kinds, _ = m.time_step(int(gen[-1]), 8)
    if ref is None: ref = gen.copy()
    print('pjb_min_ns', mn, 'tok/s', ['%.0f' % r for r in rates], 'same ids', bool(np.array_equal(gen, ref)), {k: (round(u, 2), n) for k, (u, n) in kinds.items()}, flush=True)
    m.close()
