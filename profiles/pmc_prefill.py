#!/usr/bin/env python3
"""Workload for the prefill PMC pass (run under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace`): three prefills of BASELINE configs[3]'s prompt
(448 x 448 image + 24 tokens, S = 282: vision tower + splice + LLM prefill) on the resident engine."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mllm_amd import lib, synth  # noqa: E402
from mllm_amd import synthfile as weights  # noqa: E402

cfg = synth.qwen2vl_2b()
m = lib.Qwen2VL(cfg, weights.qwen2vl_file(cfg, cache_dir=os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache")))
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
for _ in range(3):
    m.clear_kvcache()
    tok, _, ms = m.prefill(ids, pix, grid, want_logits=False)
print("prefill ms", ms, "token", tok)
m.close()
