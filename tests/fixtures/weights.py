"""TEST FIXTURE TOOLING, not product: builds synthetic `*-q4_k.mllm` files with the fixture quantiser (tests/fixtures/quantlib.py; no dependency on the reference tool at run time).

Same per-name dtype policy and block formats as `quantize <in> <out> Q4_K` of the reference
(tools/quantizer/QuantWriter.cpp:123-157,288-300); byte-for-byte agreement with it is pinned by tests/test_host.py
(digests of files the reference tool wrote, tests/golden/*_digests.json).
"""
from __future__ import annotations

import hashlib
import os

import numpy as np

from mllm_amd import mllmfile as mf, synth

from . import quantlib


def _make_tensor(args):
    name, shape, kind, target = args
    x = synth.tensor_f32(name, shape, kind)
    dt = synth.storage_dtype(name, target)
    return name, dt, (x if dt == mf.F32 else quantlib.quantize(dt, x))


def build_q4k_file(path: str, specs, target: int = mf.Q4_K, workers: int | None = None) -> str:
    """Synthesise + quantise every tensor and write the .mllm.  Tensors are made by a thread pool (numpy's Generator and the
    ctypes call into the C quantiser both release the GIL); the output bytes do not depend on `workers`."""
    jobs = [(n, s, k, target) for n, s, k in specs]
    total = sum(int(np.prod(s)) for _, s, _, _ in jobs)
    if workers is None:
        workers = min(16, os.cpu_count() or 1)
    tmp = path + f".tmp{os.getpid()}"
    if workers > 1 and total > 50_000_000:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as pool:
            items = list(pool.map(_make_tensor, jobs))
    else:
        items = [_make_tensor(j) for j in jobs]
    mf.write_mllm(tmp, items)
    os.replace(tmp, path)
    return path


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


def causal_lm_file(cfg: synth.CausalLMConfig, cache_dir: str = "/tmp/mllm_amd_cache") -> str:
    os.makedirs(cache_dir, exist_ok=True)
    key = (f"{cfg.family}-h{cfg.hidden}-i{cfg.inter}-l{cfg.layers}-a{cfg.heads}k{cfg.kv_heads}-v{cfg.vocab}-t{int(cfg.tie_embedding)}"
           f"-{'f32' if cfg.target == mf.F32 else 'q4k'}.mllm")
    path = os.path.join(cache_dir, key)
    if not os.path.exists(path):
        build_q4k_file(path, synth.causal_lm_tensors(cfg), target=cfg.target)
    return path


def vit_file(cfg: synth.ViTConfig, cache_dir: str = "/tmp/mllm_amd_cache") -> str:
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, f"vit-h{cfg.hidden}-f{cfg.ffn}-b{cfg.blocks}-p{cfg.patch}-i{cfg.img}-c{cfg.classes}-q4k.mllm")
    if not os.path.exists(path):
        build_q4k_file(path, synth.vit_tensors(cfg))
    return path


def llava_file(cfg: synth.LLaVAConfig, cache_dir: str = "/tmp/mllm_amd_cache") -> str:
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, f"llava-h{cfg.hidden}-i{cfg.inter}-l{cfg.layers}-v{cfg.vocab}-vh{cfg.v_hidden}-vb{cfg.v_blocks}-img{cfg.img}-q4k.mllm")
    if not os.path.exists(path):
        build_q4k_file(path, synth.llava_tensors(cfg))
    return path


def moe_file(cfg: synth.MoEConfig, cache_dir: str = "/tmp/mllm_amd_cache") -> str:
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, f"moe-h{cfg.hidden}-i{cfg.inter}-e{cfg.experts}-k{cfg.per_tok}-q4k.mllm")
    if not os.path.exists(path):
        build_q4k_file(path, synth.moe_tensors(cfg))
    return path


def write_fp32_mllm(path: str, specs) -> None:
    """The fp32 `.mllm` of the synthetic tensors -- the input of the reference's own `quantize` tool (oracle/make_golden.py)."""
    mf.write_mllm(path, ((n, mf.F32, synth.tensor_f32(n, s, k)) for n, s, k in specs))
