# (run with a temporary MLLM_HIP_STEP_EVENT_FLAGS override in mllm_hip_model_time_step, since removed: the library reads no such variable)
# event flags for mllm_hip_model_time_step's markers: default (0), DisableSystemFence, ReleaseToDevice -- how far above the kernel trace does each sit
import os, sys, subprocess
code = '''
import numpy as np
from mllm_amd import lib, synth
from mllm_amd import synthfile as weights
cfg = synth.qwen2vl_2b()
m = lib.Qwen2VL(cfg, weights.qwen2vl_file(cfg, cache_dir="/tmp/mllm_amd_cache"))
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
tok, _, _ = m.prefill(ids, pix, grid)
gen, ms = m.generate(tok, 64)
kinds, last = m.time_step(int(gen[-1]), 16)
print("graph us/token", ms * 1e3 / 64, "sum", sum(u * n for u, n in kinds.values()), {k: round(u, 2) for k, (u, n) in kinds.items()})
'''
for fl in ("0", "0x20000000", "0x40000000"):
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, MLLM_HIP_STEP_EVENT_FLAGS=fl))
    print(fl, out.stdout.strip()[-600:], out.stderr.strip()[-300:] if out.returncode else "")
