"""the reference's Qwen2VLModel at the 2B geometry through the HIP adapter: ids against the reference's CPU run (tests/golden/qwen2vl_2b_ref.npz) and the rates the driver reports"""
import os, sys, json, subprocess, tempfile
import numpy as np
sys.path.insert(0, '.')
from mllm_amd import synth
from mllm_amd import synthfile as weights
from tests.test_gpu_adapter import _cfg_string, DRIVER
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
g = np.load('tests/golden/qwen2vl_2b_ref.npz')
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
steps = len(g["tokens"])
td = tempfile.mkdtemp()
ids.astype(np.int32).tofile(td + "/ids.i32"); pix.astype(np.float32).tofile(td + "/pix.f32")
cmd = [DRIVER, "--model", path, "--ids", td + "/ids.i32", "--steps", str(steps), "--threads", "4", "--out", td, "--cfg", _cfg_string(cfg), "--dump-every", "0",
       "--pix", td + "/pix.f32", "--grid", ",".join(str(int(x)) for x in grid)]
out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
print(out.returncode, out.stdout[-1500:], out.stderr[-1500:])
toks = np.fromfile(td + "/tokens.i32", dtype=np.int32)
print("ids equal the reference's CPU run:", toks.tolist() == g["tokens"].tolist(), len(toks))
