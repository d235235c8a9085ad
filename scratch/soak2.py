"""Stability soak of the round-2 paths: repeated image prefill + 500-token decode (identical ids every time), interleaved with batched vision passes of changing
batch size (identical rows every time), device memory in use unchanged after the first iteration."""
import sys, os, numpy as np, time, torch
sys.path.insert(0, '.')
from mllm_amd import lib, synth
from tests.fixtures import weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
m = lib.Qwen2VL(cfg, path)
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
imgs = np.random.default_rng(1).standard_normal((6, 1024, cfg.patch_elems)).astype(np.float32)
out = torch.empty((6 * 256, cfg.hidden), dtype=torch.float32, device='cuda')
ref = refv = None
t0 = time.time(); mem = None
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    m.clear_kvcache()
    tok, _, _ = m.prefill(ids, pix, grid, want_logits=False)
    toks, _ = m.generate(tok, 500)
    h = hash(toks.tobytes())
    n = (6, 3, 5, 1)[it % 4]
    m.vision(imgs[:n], grid, out.data_ptr(), n)
    hv = hash(out[:256].cpu().numpy().tobytes())
    if ref is None: ref, refv = h, hv
    assert h == ref and hv == refv, 'run %d differs' % it
    free, total = torch.cuda.mem_get_info()
    if it == 4: mem = free
    if it > 4: assert abs(free - mem) < (64 << 20), ('device memory moved', free, mem)
print('%d x (image prefill + 500 tokens + batched vision) identical, memory steady, %.1f s' % (it + 1, time.time() - t0))
