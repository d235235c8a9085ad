#!/usr/bin/env python3
"""Workload for the PMC pass (run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`): text prefill + 24 greedy decode steps of the
Qwen2-VL-2B Q4_K engine, launched eagerly (MLLM_HIP_NO_GRAPH=1) so every kernel is its own dispatch record."""
import os
import sys

import numpy as np

os.environ.setdefault("MLLM_HIP_NO_GRAPH", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mllm_amd import lib, synth  # noqa: E402
from mllm_amd import synthfile as weights  # noqa: E402

lib.set_option("merge_o", 0)      # the five kernels of a layer one by one (the step itself folds the o-projection into the attention's launch; the bytes are the same)
cfg = synth.qwen2vl_2b()
m = lib.Qwen2VL(cfg, weights.qwen2vl_file(cfg, cache_dir=os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache")))
ids = (np.arange(40) * 7919 % 150000).astype(np.int32)
tok, _, _ = m.prefill(ids, want_logits=False)
gen, _ = m.generate(tok, 24)
print("tokens", gen[:8].tolist())
m.close()
