"""Writes the synthetic `*-q4_k.mllm` (or fp32) weight files every test, the bench and the golden generator run on.

The tensors come from mllm_amd/synth.py -- Q4_K / Q4_0 tensors drawn directly in the quantised domain, fp32 ones as seeded normals -- under the per-name dtype policy
of the reference's `quantize <in> <out> Q4_K` (tools/quantizer/QuantWriter.cpp:123-157), in the container format of mllm_amd/mllmfile.py (mllm/ParamLoader.cpp:14-31).
No quantiser is involved: the reference (oracle/make_golden.py, in the development container) and the HIP path read the same bytes.
"""
from __future__ import annotations

import hashlib
import os

import numpy as np

from . import mllmfile as mf, synth


TAG = "q4k-qd1"      # file-name tag of the synthesis scheme (quantised-domain draw, version 1): a cached file of another scheme is never picked up


def _make_tensor(args):
    name, shape, kind, target = args
    dt, data = synth.tensor_stored(name, shape, kind, target)
    return name, dt, data


def build_q4k_file(path: str, specs, target: int = mf.Q4_K, workers: int | None = None) -> str:
    """Synthesise every tensor and write the .mllm.  Tensors are made by a thread pool (numpy's Generator and array arithmetic release the GIL);
    the output bytes do not depend on `workers`."""
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
    key = f"q2vl-h{cfg.hidden}-i{cfg.inter}-l{cfg.layers}-v{cfg.vocab}-vd{cfg.v_dim}-vb{cfg.v_blocks}{'' if vision else '-novis'}{'' if cfg.tie_embedding else '-untied'}{tag}-{TAG}.mllm"
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
           f"-{'f32' if cfg.target == mf.F32 else TAG}.mllm")
    path = os.path.join(cache_dir, key)
    if not os.path.exists(path):
        build_q4k_file(path, synth.causal_lm_tensors(cfg), target=cfg.target)
    return path


def vit_file(cfg: synth.ViTConfig, cache_dir: str = "/tmp/mllm_amd_cache") -> str:
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, f"vit-h{cfg.hidden}-f{cfg.ffn}-b{cfg.blocks}-p{cfg.patch}-i{cfg.img}-c{cfg.classes}-{TAG}.mllm")
    if not os.path.exists(path):
        build_q4k_file(path, synth.vit_tensors(cfg))
    return path


def llava_file(cfg: synth.LLaVAConfig, cache_dir: str = "/tmp/mllm_amd_cache") -> str:
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, f"llava-h{cfg.hidden}-i{cfg.inter}-l{cfg.layers}-v{cfg.vocab}-vh{cfg.v_hidden}-vb{cfg.v_blocks}-img{cfg.img}-{TAG}.mllm")
    if not os.path.exists(path):
        build_q4k_file(path, synth.llava_tensors(cfg))
    return path


def moe_file(cfg: synth.MoEConfig, cache_dir: str = "/tmp/mllm_amd_cache") -> str:
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, f"moe-h{cfg.hidden}-i{cfg.inter}-e{cfg.experts}-k{cfg.per_tok}-{TAG}.mllm")
    if not os.path.exists(path):
        build_q4k_file(path, synth.moe_tensors(cfg))
    return path
