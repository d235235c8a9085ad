"""three image+text prefills of the 2B engine (for a rocprofv3 kernel trace: per-kernel totals / 3 = one prefill)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mllm_amd import lib, synth
if os.environ.get('MLLM_SO'): lib.SO_PATH = os.environ['MLLM_SO']
from tests.fixtures import weights
cfg = synth.qwen2vl_2b()
m = lib.Qwen2VL(cfg, weights.qwen2vl_file(cfg, cache_dir=os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache")))
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
for _ in range(3):
    m.clear_kvcache()
    tok, _, ms = m.prefill(ids, pix, grid, want_logits=False)
    print("prefill ms", ms)
m.close()
