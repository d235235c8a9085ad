#!/usr/bin/env python3
"""bench.py -- the hot path's headline metric on MI355X: Qwen2-VL-2B INT4 (Q4_K) decode tok/s + 448x448 image prefill ms.

    python bench.py --gpus 1 --steps 256 --warmup 16
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one greedy decode forward (one token) of the Qwen2-VL-2B shaped model on synthetic Q4_K weights after a
synthetic 448x448 image + 24-token prompt prefill (BASELINE.json configs[3]).  Timed exactly like Module::profiling()
(mllm/Module.cpp:35-42): TTFT = the prefill forward, decode tok/s = steps / time of the following forwards -- here the K
forwards run back to back on the device (argmax on device), bracketed by barrier + synchronize, max over ranks.
N > 1: the LLM path does not shard ("replicas only"): every rank decodes its own replica (weak scaling, value = N*K/t).  The
vision prefill does shard: a batch of 8 images is split over the ranks, each rank runs the ViT on its images and one RCCL
all-gather reassembles the visual-token sequence (reported as vit_images_per_s, not part of `value`).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def cpu_baseline(path, cfg, rank):
    """The compiled reference (oracle/_ref/ref_qwen2vl, kind "reference") on the host cores, on a bounded sample of the same
    workload: a 16-token text prompt + 24 greedy decode steps of the same .mllm (decode is the metric's unit)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_qwen2vl")
    if rank != 0 or not os.path.exists(exe):
        return None
    import tempfile
    cores = min(16, os.cpu_count() or 1)
    td = tempfile.mkdtemp()
    ids = np.random.default_rng(11).integers(0, 151000, size=16).astype(np.int32)
    ids.tofile(os.path.join(td, "ids.i32"))
    env = dict(os.environ, OMP_NUM_THREADS=str(cores))
    try:
        out = subprocess.run([exe, "--model", path, "--ids", os.path.join(td, "ids.i32"), "--steps", "25", "--threads", str(cores), "--out", td,
                              "--dump-every", "0"], capture_output=True, text=True, timeout=600, env=env)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][0]
        r = json.loads(line)
        return {"value": round(r["decode_tok_s"], 3), "unit": "tok/s", "cores": cores, "kind": "reference",
                "sample": "reference x86 AVX2 CPU backend (oracle/_ref, -t %d): 16-token text prompt + 24 greedy decode steps on the same Q4_K .mllm" % cores,
                "prefill_tok_s": round(1000.0 * r["prefill_tokens"] / r["prefill_ms"], 3)}
    except Exception as e:  # the baseline is reported, never required
        return {"value": None, "unit": "tok/s", "cores": cores, "kind": "reference", "sample": f"failed: {e}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--vit-batch", type=int, default=8)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from mllm_amd import lib, synth, weights

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback for the HIP path)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    cfg = synth.qwen2vl_2b()
    cache = os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache")
    t0 = time.time()
    if rank == 0:
        path = weights.qwen2vl_file(cfg, cache_dir=cache)
    barrier()
    path = weights.qwen2vl_file(cfg, cache_dir=cache)
    t_weights = time.time() - t0
    pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)      # 448x448 -> 1024 patches -> 256 visual tokens, S = 282

    t0 = time.time()
    m = lib.Qwen2VL(cfg, path, device=local_rank)
    t_load = time.time() - t0
    # warm prefill (first-touch, code load), then the measured TTFT
    m.prefill(ids, pix, grid, want_logits=False)
    m.clear_kvcache()
    barrier()
    tok, _, prefill_ms = m.prefill(ids, pix, grid, want_logits=False)
    pre = torch.tensor([prefill_ms], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(pre, op=dist.ReduceOp.MAX)
    prefill_ms = float(pre.item())

    # ---- decode: W warmup steps, then exactly K timed steps ----------------------------------------------------------
    K, W = args.steps, args.warmup
    assert ids.size + K + W + 1 <= cfg.cache_limit, "steps + warmup exceed the KV slab (-l 800 of the demo)"
    if W > 0:
        toks, _ = m.generate(tok, W)
        tok = int(toks[-1])
    barrier()
    t0 = time.perf_counter()
    toks, dev_ms = m.generate(tok, K)
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    value = world * K / dt

    # ---- roofline of the dominant kernel (gate/up Q4_K GEMV, 17920 x 1536): live HIP-event timing on the engine's stream ----
    ms_launch, bytes_launch = m.time_gemv(13, 280)
    achieved = bytes_launch / (ms_launch * 1e-3) / 1e9
    # traffic: HBM-side bytes per launch from the PMC pass of this round (rocprofv3 --pmc FETCH_SIZE in its own run, x2 gfx950 correction;
    # profiles/r01_pmc_fetch_size.md) -- a profiler pass cannot run inside the timed process, so the committed figure is reported
    traffic = None
    try:
        traffic = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["dec_gateup_kernel"]["fetch_bytes_per_launch"]
    except Exception:
        pass
    roofline = {"bound": "hbm", "kernel": "dec_gateup_blk_kernel<5,1,7> (fused RMSNorm+Q8_K+gate|up GEMV+SiLU*mul, 17920 x 1536 Q4_K rows, one lane per super-block, LDS-DMA weight stream; launches cycle over the 28 layers so every launch streams cold HBM)", "achieved": round(achieved, 1),
                "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                "us_per_launch": round(ms_launch * 1e3, 3), "algorithmic_bytes_per_launch": int(bytes_launch)}
    wbytes = m.decode_weight_bytes()
    e2e_gbs = wbytes * (K / dt) / 1e9

    # ---- sharded vision prefill (N > 1: image batch split over ranks + one all-gather) ---------------------------------
    vit = None
    nb = args.vit_batch
    if nb > 0:
        from mllm_amd import parallel
        n_tok = (grid[0] * grid[1] * grid[2]) // 4
        batch = torch.from_numpy(np.stack([np.roll(pix, i, axis=0) for i in range(nb)]))   # [nb, 1024, 1176] host pixels

        def run_vision(px):
            out = torch.empty((px.shape[0], n_tok, cfg.hidden), dtype=torch.float32, device="cuda")
            m.vision(px.numpy(), grid, out.data_ptr(), px.shape[0])
            return out

        run_vision(batch[:1])  # warm
        barrier()
        t0 = time.perf_counter()
        full = parallel.sharded_vision(run_vision, batch)     # image shard per rank + ONE all-gather (RCCL over xGMI)
        barrier()
        dtv = time.perf_counter() - t0
        tv = torch.tensor([dtv], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(tv, op=dist.ReduceOp.MAX)
        assert tuple(full.shape) == (nb, n_tok, cfg.hidden)
        vit = {"images": nb, "ms": round(float(tv.item()) * 1e3, 3), "images_per_s": round(nb / float(tv.item()), 2),
               "tflops": round(1.48 * nb / float(tv.item()), 2), "checksum": round(float(full.double().sum().item()), 3),
               "note": "ViT 448x448 (1024 patches) incl. H2D of the pixels; image batch sharded over ranks, 1 all-gather"}

    base = None if args.no_cpu_baseline or world > 1 else cpu_baseline(path, cfg, rank)
    if rank == 0:
        out = {
            "metric": "decode tok/s + prefill ms (448px img), Qwen2-VL-2B INT4, 1 GPU", "value": round(value, 2), "unit": "tok/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt * 1e3 / K, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int8xint4->int32, fp32 accumulate (Q8_K x Q4_K)", "data": "synthetic",
            "config": {"workload": "demo_qwen2_vl Qwen2-VL-2B Q4_K: 448x448 image (1024 patches) + 24-token prompt prefill (S=282), then greedy decode, KV limit 800",
                       "prefill_tokens": int(ids.size), "parallelism": "replicas" if world > 1 else "single"},
            "prefill_ms": round(prefill_ms, 3), "prefill_tok_s": round(1000.0 * ids.size / prefill_ms, 1),
            "decode_weight_bytes_per_token": int(wbytes), "decode_hbm_GBps_algorithmic": round(e2e_gbs, 1), "decode_hbm_frac_of_8TBps": round(e2e_gbs / 8000.0, 4),
            "roofline": roofline, "cpu_baseline": base, "vit_prefill": vit,
            "setup_s": {"weights": round(t_weights, 1), "load": round(t_load, 2)},
        }
        print(json.dumps(out))
    m.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
