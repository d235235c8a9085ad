"""Build synthetic `*-q4_k.mllm` files with the product's own quantiser (no dependency on the reference tool).

Same per-name dtype policy and block formats as `quantize <in> <out> Q4_K` of the reference
(tools/quantizer/QuantWriter.cpp:123-157,288-300); byte-for-byte agreement with it is pinned by tests/test_quantizer.py.
"""
from __future__ import annotations

import hashlib
import os

from . import lib, mllmfile as mf, synth


def build_q4k_file(path: str, specs, target: int = mf.Q4_K) -> str:
    def gen():
        for name, shape, kind in specs:
            x = synth.tensor_f32(name, shape, kind)
            dt = synth.storage_dtype(name, target)
            yield name, dt, (x if dt == mf.F32 else lib.quantize_host(dt, x))
    tmp = path + ".tmp"
    mf.write_mllm(tmp, list(_stream(gen())))
    os.replace(tmp, path)
    return path


def _stream(it):
    for item in it:
        yield item


def qwen2vl_file(cfg: synth.Qwen2VLConfig, cache_dir: str = "/tmp/mllm_amd_cache", tag: str = "", vision: bool = True) -> str:
    os.makedirs(cache_dir, exist_ok=True)
    key = f"q2vl-h{cfg.hidden}-i{cfg.inter}-l{cfg.layers}-v{cfg.vocab}-vd{cfg.v_dim}-vb{cfg.v_blocks}{'' if vision else '-novis'}{tag}-q4k.mllm"
    path = os.path.join(cache_dir, key)
    if not os.path.exists(path):
        build_q4k_file(path, synth.qwen2vl_tensors(cfg, vision=vision))
    return path


def tensor_digests(path: str) -> dict:
    f = mf.MllmFile(path)
    out = {n: hashlib.sha256(f.raw(n).tobytes()).hexdigest()[:16] for n in f.names()}
    f.close()
    return out
