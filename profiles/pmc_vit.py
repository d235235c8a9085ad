#!/usr/bin/env python3
"""Workload for the MFMA-utilisation PMC pass (run under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace`): two forwards of
the Qwen2-VL vision tower (448x448, 1024 patches) on the resident engine."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mllm_amd import lib, synth  # noqa: E402
from mllm_amd import synthfile as weights  # noqa: E402

cfg = synth.qwen2vl_2b()
m = lib.Qwen2VL(cfg, weights.qwen2vl_file(cfg, cache_dir=os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache")))
pix, grid, _ = synth.qwen2vl_inputs(cfg, (32, 32), 24)
out = torch.empty((256, cfg.hidden), dtype=torch.float32, device="cuda")
for _ in range(2):
    m.vision(pix, grid, out.data_ptr())
torch.cuda.synchronize()
print("checksum", float(out.sum()))
m.close()
