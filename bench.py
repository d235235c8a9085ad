#!/usr/bin/env python3
"""bench.py -- the hot path's headline metric on MI355X: Qwen2-VL-2B INT4 (Q4_K) decode tok/s + 448x448 image prefill ms.

    python bench.py --gpus 1 --steps 256 --warmup 16                 # headline (BASELINE.json configs[3])
    python bench.py --config qwen15|tinyllama|llava|vit              # the other BASELINE configs on the same resident engine
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one greedy decode forward (one token) after the config's synthetic prompt prefill (qwen2vl: 448x448 image + 24 tokens, S = 282).
Timed exactly like Module::profiling() (mllm/Module.cpp:35-42): TTFT = the prefill forward (median of 5 here), decode tok/s = steps / time of
the following forwards -- the K forwards run back to back on the device (argmax on device), bracketed by barrier + synchronize, max over ranks.
N > 1: the LLM path does not shard ("replicas only"): every rank decodes its own replica (weak scaling, value = N*K/t).  The vision prefill does
shard: a batch of images is split over the ranks, each rank runs the tower on its images and one RCCL all-gather reassembles the visual-token
sequence (reported as vit_prefill, not part of `value`).  config vit has no decode: a step is one image through the tower (value = images/s).
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

METRIC = "decode tok/s + prefill ms (448px img), Qwen2-VL-2B INT4, 1 GPU"
WORKLOADS = {
    "qwen2vl": "demo_qwen2_vl Qwen2-VL-2B Q4_K: 448x448 image (1024 patches) + 24-token prompt prefill (S=282), then greedy decode, KV limit 800",
    "qwen15": "demo_qwen Qwen-1.5-0.5B Q4_K: 64-token prompt prefill, then greedy decode, KV limit 400",
    "tinyllama": "demo_tinyllama TinyLLaMA-1.1B geometry with Q4_K weights: 64-token prompt prefill, then greedy decode, KV limit 400",
    "llava": "demo_llava LLaVA-1.5-7B Q4_K + CLIP-ViT-L/14-336: one 336x336 image + 13 text tokens (S=589), then greedy decode, KV limit 700",
    "vit": "demo_vit ViT-B/16 224x224 Q4_K: image batch through patch embedding + 12 encoder blocks + classifier",
}


def cpu_baseline(path, rank, ids, pix):
    """The compiled reference (oracle/_ref/ref_qwen2vl, kind "reference") on the host cores, on a bounded sample of the SAME workload: the S = 282 image
    prompt (vision tower + LLM prefill) and 24 greedy decode steps on the same .mllm."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_qwen2vl")
    if rank != 0 or not os.path.exists(exe):
        return None
    import tempfile
    host_cpus = os.cpu_count() or 1
    cores = min(16, host_cpus)
    td = tempfile.mkdtemp()
    ids.astype(np.int32).tofile(os.path.join(td, "ids.i32"))
    pix.astype(np.float32).tofile(os.path.join(td, "pix.f32"))
    env = dict(os.environ, OMP_NUM_THREADS=str(cores))
    try:
        out = subprocess.run([exe, "--model", path, "--ids", os.path.join(td, "ids.i32"), "--pix", os.path.join(td, "pix.f32"), "--grid", "1,32,32", "--steps", "25",
                              "--threads", str(cores), "--out", td, "--dump-every", "0"], capture_output=True, text=True, timeout=900, env=env)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][0]
        r = json.loads(line)
        # the host's clocks and neighbours move this number by +-15 % from run to run (15.7 .. 21.4 tok/s across round 2's records): the per-step times say so
        return {"value": round(r["decode_tok_s"], 3), "unit": "tok/s", "cores": cores, "host_cpus": host_cpus, "kind": "reference", "run_to_run_spread": "about +-15 % (24 steps on shared host cores)",
                "sample": "reference x86 AVX2 CPU backend (oracle/_ref/ref_qwen2vl, -t %d) on the same Q4_K .mllm: the same 448x448 image + 24-token prompt (S=%d) prefill, "
                          "then 24 greedy decode steps" % (cores, r["prefill_tokens"]),
                "prefill_ms": round(r["prefill_ms"], 1), "prefill_tok_s": round(1000.0 * r["prefill_tokens"] / r["prefill_ms"], 3)}
    except Exception as e:  # the baseline is reported, never required
        return {"value": None, "unit": "tok/s", "cores": cores, "host_cpus": host_cpus, "kind": "reference", "sample": f"failed: {e}"}


def pmc_mfma():
    """Matrix-pipe busy fraction per prefill kernel from this round's rocprofv3 PMC pass (`--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -- python3
    profiles/pmc_prefill.py`, summarised by profiles/pmc_mfma_summarize.py into profiles/r04_pmc_mfma.json)."""
    for name in ("r04_pmc_mfma.json",):
        try:
            return name, json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
    return None, {}


PEAK_F16_MFMA_TF = 2500.0      # dense bf16 / fp16 MFMA peak of MI355X (MI355X_MICROARCH.md); the 5 PF headline figures are 2:1 sparse
PEAK_F32_MFMA_TF = 157.3       # v_mfma_f32_32x32x2_f32 / 16x16x4: the exact fp32 fma-chain forms the attention is confined to


def prefill_kernels(cfgname):
    """Live HIP-event timings (torch events on torch's current stream, which is the stream mllm_amd.ops launches on) of the two kernels that are the prefill: the Q4_K GEMM
    on the shapes of the config's prefill, and the exact-order FlashAttention2 on its attention shapes.  Algorithmic FLOPs: GEMM 2 M N K; attention 4 H S^2 D (causal: half)."""
    import ctypes as C
    import torch
    from mllm_amd import lib, ops, synth
    L = lib.load()
    r = np.random.default_rng(0)
    shapes = {"qwen2vl": [("vit fc1", 1024, 5120, 1280, 32), ("vit fc2", 1024, 1280, 5120, 32), ("vit qkv", 1024, 3840, 1280, 32), ("vit proj", 1024, 1280, 1280, 32),
                          ("llm gate|up", 282, 17920, 1536, 28), ("llm down", 282, 1536, 8960, 28), ("llm q|k|v", 282, 2048, 1536, 28), ("llm o", 282, 1536, 1536, 28)],
              "llava": [("clip fc1", 577, 4096, 1024, 23), ("clip fc2", 577, 1024, 4096, 23), ("llm gate|up", 589, 22016, 4096, 32), ("llm down", 589, 4096, 11008, 32)]}.get(cfgname, [])
    attn = {"qwen2vl": [("vit block (S=1024, 16 x 80, fp32 K/V)", 1024, 16, 16, 80, False, False, 32), ("llm layer (S=282, 12/2 x 128, causal, fp16 K/V)", 282, 12, 2, 128, True, True, 28)],
            "llava": [("clip block (S=577, 16 x 64)", 577, 16, 16, 64, False, False, 23), ("llm layer (S=589, 32 x 128, causal)", 589, 32, 32, 128, True, True, 32)]}.get(cfgname, [])

    def timed(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n      # us

    out = []
    for label, M, N, K, calls in shapes:
        Wd = torch.from_numpy(synth.quantized_blocks(lib.Q4_K, r, N * K)).cuda()
        wp = torch.empty(int(L.mllm_hip_q4k_wpack_bytes(C.c_int(N), C.c_int(K))), dtype=torch.uint8, device="cuda")
        xp = torch.empty(int(L.mllm_hip_q4k_prepack_bytes(C.c_int(M), C.c_int(K))), dtype=torch.uint8, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        lib.check(L.mllm_hip_q4k_prepack(C.c_void_p(Wd.data_ptr()), C.c_int(N), C.c_int(K), C.c_void_p(wp.data_ptr()), st))
        x = torch.from_numpy(r.standard_normal((M, K)).astype(np.float32)).cuda()
        lib.check(L.mllm_hip_quantize_q8k_packed(C.c_void_p(x.data_ptr()), C.c_void_p(xp.data_ptr()), C.c_int(M), C.c_int(K), st))
        y = torch.empty((M, N), dtype=torch.float32, device="cuda")
        us = timed(lambda: lib.check(L.mllm_hip_linear_q4kp_packed(C.c_void_p(wp.data_ptr()), None, C.c_void_p(xp.data_ptr()), C.c_void_p(y.data_ptr()), C.c_int(lib.F32), C.c_int64(N),
                                                                   None, C.c_int(M), C.c_int(N), C.c_int(K), st)), 20)
        fl = 2.0 * M * N * K
        out.append({"kernel": "gemm_q4k " + label, "shape_MNK": [M, N, K], "calls_per_prefill": calls, "us_per_launch": round(us, 2), "flops_per_launch": fl,
                    "achieved_TFps": round(fl / us / 1e6, 1), "peak_TFps": PEAK_F16_MFMA_TF, "frac": round(fl / us / 1e6 / PEAK_F16_MFMA_TF, 4)})
        del Wd, wp, xp, x, y
    for label, S, H, Hkv, D, causal, f16, calls in attn:
        q = torch.from_numpy(r.standard_normal((S, H * D)).astype(np.float32)).cuda()
        k = torch.from_numpy(r.standard_normal((S, Hkv * D)).astype(np.float32)).cuda()
        v = torch.from_numpy(r.standard_normal((S, Hkv * D)).astype(np.float32)).cuda()
        if f16:
            k, v = k.half(), v.half()
        us = timed(lambda: ops.flash_attention2(q, k, v, S, S, H, Hkv, D, causal), 10)
        fl = 4.0 * H * S * S * D * (0.5 if causal else 1.0)
        out.append({"kernel": "fa2_prefill " + label, "calls_per_prefill": calls, "us_per_launch": round(us, 2), "flops_per_launch": fl, "achieved_TFps": round(fl / us / 1e6, 1),
                    "peak_TFps": PEAK_F32_MFMA_TF, "frac": round(fl / us / 1e6 / PEAK_F32_MFMA_TF, 4)})
    return out


def pmc_traffic():
    """HBM-side bytes per launch of the decode kernels from this round's rocprofv3 PMC pass (own run, FETCH_SIZE x2 gfx950 correction):
    profiles/r04_pmc_traffic.json, regenerated by `rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 profiles/pmc_decode.py` + pmc_summarize.py."""
    for name in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            return name, json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
    return None, {}


class StubEngine:
    """What --dry-run puts in the engine's place: the same calls bench.py makes, each taking a FIXED, rank-dependent time (rank r is (1 + r / 4) x slower), so a test
    can predict the line: value = world * K / (K * step_s of the slowest rank).  No arithmetic, no GPU, never a measurement."""
    step_s = 1.0e-3

    def __init__(self, rank):
        self.k = 1.0 + 0.25 * rank

    def prefill(self):
        time.sleep(4 * self.step_s * self.k)
        return 4 * self.step_s * self.k * 1e3

    def generate(self, steps):
        time.sleep(steps * self.step_s * self.k)

    def vision(self, n):
        time.sleep(n * 2 * self.step_s * self.k)


def dry_run(args):
    """bench.py's distributed skeleton -- rendezvous from the torchrun environment, barrier on both sides of the timed region, MAX over ranks, value = world * K / t,
    the sharded vision prefill through mllm_amd.parallel with its per-pass all-gather -- on gloo and a stub engine, so the N > 1 path is exercised where no GPU is
    (tests/test_bench_contract.py).  The GPU path below runs the same steps with "nccl" (= RCCL) and the resident engine."""
    import torch
    import torch.distributed as dist
    from mllm_amd import parallel
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    if world > 1:
        dist.init_process_group("gloo")

    def barrier():
        if world > 1:
            dist.barrier()

    def rmax(v):
        t = torch.tensor([v], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    eng = StubEngine(rank)
    nb = max(args.vit_batch, world)
    batch = torch.arange(nb * 4 * 3, dtype=torch.float32).reshape(nb, 4, 3)

    def run_vision(px):
        eng.vision(px.shape[0])
        return px * 2.0
    barrier()
    t0 = time.perf_counter()
    full = parallel.sharded_vision(run_vision, batch, pass_images=1)
    barrier()
    dtv = rmax(time.perf_counter() - t0)
    assert torch.equal(full, batch * 2.0)
    prefill_ms = rmax(eng.prefill())
    K, W = args.steps, args.warmup
    eng.generate(W)
    barrier()
    t0 = time.perf_counter()
    eng.generate(K)
    barrier()
    dt = rmax(time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": round(world * K / dt, 2), "unit": "tok/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt * 1e3 / K, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "none (dry run: stub engine on gloo, plumbing check only)",
                          "dry_run": True, "config": {"workload": "stub", "parallelism": "replicas" if world > 1 else "single", "rccl_world": world},
                          "prefill_ms": round(prefill_ms, 3), "roofline": None, "cpu_baseline": None,
                          "vit_prefill": {"images": nb, "ms": round(dtv * 1e3, 3), "world": world, "all_gathers": len(parallel.pass_groups((nb + world - 1) // world, 1))}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--config", default="qwen2vl", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--vit-batch", type=int, default=8)
    ap.add_argument("--no-batched", dest="batched", action="store_false", help="skip the batched-decode extra (B = 2, 4, 8 sequences per step)")
    ap.add_argument("--dry-run", action="store_true", help="plumbing check without a GPU: gloo + a stub engine with fixed per-rank step times; the line says dry_run and is no measurement")
    args = ap.parse_args()
    if args.dry_run:
        return dry_run(args)

    import torch
    import torch.distributed as dist
    from mllm_amd import lib, mllmfile as mf, parallel, synth
    from mllm_amd import synthfile as weights      # writes the synthetic .mllm (drawn in the quantised domain, mllm_amd/synth.py) before anything is timed

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

    def rmax(v):
        t = torch.tensor([v], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    cache = os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache")
    cfgname = args.config
    image = meta = None
    if cfgname == "qwen2vl":
        cfg = synth.qwen2vl_2b(); mk = lambda: weights.qwen2vl_file(cfg, cache_dir=cache)
        image, meta, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)      # 448x448 -> 1024 patches -> 256 visual tokens, S = 282
    elif cfgname == "qwen15":
        cfg = synth.qwen15_05b(); mk = lambda: weights.causal_lm_file(cfg, cache); ids = synth.causal_lm_ids(cfg, 64)
    elif cfgname == "tinyllama":
        cfg = synth.tinyllama_11b(mf.Q4_K); mk = lambda: weights.causal_lm_file(cfg, cache); ids = synth.causal_lm_ids(cfg, 64)
    elif cfgname == "llava":
        cfg = synth.llava_7b(); mk = lambda: weights.llava_file(cfg, cache); ids, image = synth.llava_inputs(cfg)
    else:
        cfg = synth.vit_b16(); mk = lambda: weights.vit_file(cfg, cache); ids = None
    t0 = time.time()
    if rank == 0:
        path = mk()
    barrier()
    path = mk()
    t_weights = time.time() - t0

    t0 = time.time()
    m = lib.Model(cfg, path, device=local_rank)
    t_load = time.time() - t0
    load = m.load_stats()

    # ---- sharded vision prefill (N > 1: image batch split over ranks + one all-gather); config vit: this IS the step -------------------------
    def tower_batch(nb):
        if cfgname == "qwen2vl":
            return torch.from_numpy(np.stack([np.roll(image, i, axis=0) for i in range(nb)]))          # [nb, 1024, 1176] host pixels
        if cfgname == "llava":
            return torch.from_numpy(np.stack([np.roll(image, i, axis=0) for i in range(nb)]))          # [nb, 336, 3, 336]
        return torch.from_numpy(synth.vit_images(cfg, nb))

    vit = None
    has_tower = cfgname in ("qwen2vl", "llava", "vit")
    if has_tower and (args.vit_batch > 0 or cfgname == "vit"):
        nb = max(args.vit_batch, world) if cfgname != "vit" else max(args.vit_batch, 1) * 8
        rows, cols = m.vision_shape(meta)
        batch = tower_batch(nb)

        def run_vision(px):
            out = torch.empty((px.shape[0], rows, cols), dtype=torch.float32, device="cuda")
            m.vision(px.numpy(), meta, out.data_ptr(), px.shape[0])
            return out

        tower_tokens = {"qwen2vl": int(batch.shape[1]), "llava": rows + 1, "vit": 197}[cfgname]
        vis_pass = max(1, min(32, 6400 // tower_tokens))      # images per tower pass: the engine's own pass size (about 6400 token rows)
        run_vision(batch[:1])  # warm
        reps = max(1, args.steps // nb) if cfgname == "vit" else 1
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            # image shard per rank; with more than one rank the all-gather of tower pass g (RCCL over xGMI) travels while pass g + 1 is computed
            full = parallel.sharded_vision(run_vision, batch, pass_images=vis_pass if world > 1 else 0)
        barrier()
        dtv = rmax(time.perf_counter() - t0) / reps
        assert tuple(full.shape) == (nb, rows, cols)
        flops = {"qwen2vl": 1.48e12, "llava": 0.39e12, "vit": 35e9}[cfgname]      # SURVEY §8(d) algorithmic FLOPs per image
        vit = {"images": nb, "ms": round(dtv * 1e3, 3), "images_per_s": round(nb / dtv, 2), "tflops": round(flops * nb / dtv / 1e12, 2),
               "frac_of_bf16_mfma_peak_2500TF": round(flops * nb / dtv / 2.5e15, 4), "checksum": round(float(full.double().sum().item()), 3),
               "world": world, "note": "tower incl. H2D of the pixels; a rank walks its images several per pass (about 6400 token rows), the upload of pass g+1 overlaps the tower on pass g; image batch sharded over ranks, 1 all-gather"}

    if cfgname == "vit":
        if rank == 0:
            print(json.dumps({
                "metric": METRIC, "value": vit["images_per_s"], "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(vit["ms"] / vit["images"], 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "int8xint4->int32, fp32 accumulate (Q8_K x Q4_K)", "data": "synthetic", "config": {"workload": WORKLOADS[cfgname], "parallelism": f"image shard x{world}"},
                "roofline": {"bound": "mfma", "achieved": vit["tflops"], "peak": 2500.0, "unit": "TFLOP/s", "frac": vit["frac_of_bf16_mfma_peak_2500TF"], "traffic": None},
                "cpu_baseline": None, "vit_prefill": vit, "setup_s": {"weights": round(t_weights, 1), "load": round(t_load, 2)}, "load": load}))
        m.close()
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- prefill: one warm run (first touch, code load), then the median of 5 timed runs (TTFT; inputs resident in HBM when the clock starts) ----
    m.prefill(ids, image, meta, want_logits=False)
    pre = []
    for _ in range(5):
        m.clear_kvcache()
        barrier()
        tok, _, ms = m.prefill(ids, image, meta, want_logits=False)
        pre.append(ms)
    prefill_ms = rmax(float(np.median(pre)))
    S = int(ids.size) if cfgname != "llava" else int(ids.size) - 1 + cfg.v_tokens

    # ---- decode: W warmup steps, then exactly K timed steps ----------------------------------------------------------
    K, W = args.steps, args.warmup
    if cfgname == "llava" and S + K + W + 1 > cfg.cache_limit:
        K = max(8, cfg.cache_limit - S - W - 1 - 4)      # (four slots stay free for the launch-by-launch steps of roofline.step_launches)
    assert S + K + W + 1 <= cfg.cache_limit, "steps + warmup exceed the KV slab of the config"
    if W > 0:
        toks, _ = m.generate(tok, W)
        tok = int(toks[-1])
    barrier()
    t0 = time.perf_counter()
    toks, dev_ms = m.generate(tok, K)
    barrier()
    dt = rmax(time.perf_counter() - t0)
    value = world * K / dt

    # ---- roofline (decode is HBM-bound): the whole token first -- algorithmic bytes (weights + KV read) / device time of the K steps -- then
    # the two kernels that matter, each from live HIP events on the engine's stream over 280 launches cycling over the layers (cold HBM like the real step)
    wbytes = m.decode_weight_bytes()
    T_mid = S + W + K // 2
    kv_bytes = cfg.layers * 2 * T_mid * cfg.kv_heads * cfg.head_dim * 2 if cfgname != "llava" else cfg.layers * 2 * T_mid * cfg.heads * cfg.head_dim * 2
    tok_bytes = wbytes + kv_bytes
    tok_gbs = tok_bytes * (K / (dev_ms * 1e-3)) / 1e9
    tname, traffic = pmc_traffic()
    kernels = []
    for which, label, tkey in ((13, "dec_gateup (fused RMSNorm + Q8_K + gate|up Q4_K GEMV + SiLU*mul): the dominant HBM stream; timed alone here (cold rows) -- inside the step the attention launch has warmed the L2s with them", "dec_gateup"),
                               (11, "dec_attn (rotary + KV append + exact-order attention over the cache; its idle workgroups request the layer's gate|up (+ o-projection, when that runs as its own launch) rows early: the traffic figure is theirs; timed alone here -- in the step the o-projection's workgroups ride in this launch): the time-dominant kernel", "dec_attn"),
                               (14, "dec_down (Q8_K + down Q4_K GEMV + residual)", "dec_down")):
        ms_l, b_l = m.time_kernel(which, 280)
        ach = b_l / (ms_l * 1e-3) / 1e9
        kernels.append({"kernel": label, "us_per_launch": round(ms_l * 1e3, 3), "algorithmic_bytes_per_launch": int(b_l), "achieved_GBps": round(ach, 1),
                        "frac": round(ach / 8000.0, 4), "traffic_bytes_per_launch": (traffic.get(tkey) or {}).get("fetch_bytes_per_launch") if cfgname == "qwen2vl" else None})
    # ---- the step launch by launch: a few more greedy steps run eagerly with a HIP event either side of every launch (mllm_hip_model_time_step).  Since round 4 most of
    # the token is ONE kind of launch -- a layer's down projection + the next layer's q|k|v + attention + o-projection handing their rows over in flight -- which cannot
    # be timed alone (its roles poll each other); this is its live figure, to be read beside the kernel trace's duration of the same kernel in the captured graph
    step_launches, room = [], cfg.cache_limit - (S + W + K) - 1
    if rank == 0 and min(8, room) >= 2:
        b = {k: m.time_kernel(w, 1)[1] for k, w in (("qkv", 10), ("kv", 11), ("o_proj", 12), ("gateup", 13), ("down", 14))}
        b["attn"] = b["kv"]
        per_layer = b["qkv"] + b["o_proj"] + b["gateup"] + b["down"]
        kinds, _ = m.time_step(int(toks[-1]), min(8, room))
        if "o_proj" not in kinds and "attn" in kinds:
            b["attn"] = b["kv"] + b["o_proj"]
        b["front"] = b["qkv"] + b["kv"] + b["o_proj"]
        b["chain"] = b["down"] + b["front"]
        b["head"] = wbytes - cfg.layers * per_layer
        b["next"] = 0
        what = {"chain": "dec_down_front: down projection of layer l + q|k|v, attention and o-projection of layer l + 1 as roles of one launch",
                "front": "q|k|v + attention + o-projection as roles of one launch", "attn": "attention (+ the o-projection's workgroups when merged)",
                "qkv": "RMSNorm + Q8_K + q|k|v GEMV", "o_proj": "o-projection GEMV + residual", "gateup": "RMSNorm + Q8_K + gate|up GEMV + SiLU*mul",
                "down": "Q8_K + down GEMV + residual", "head": "model.norm + lm_head GEMV + argmax partials", "next": "argmax + state advance"}
        for k, (us_l, n_l) in sorted(kinds.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
            ach = b[k] / (us_l * 1e-6) / 1e9 if us_l > 0 else 0.0
            step_launches.append({"launch": k, "what": what[k], "launches_per_token": n_l, "us_per_launch": round(us_l, 3), "us_per_token": round(us_l * n_l, 1),
                                  "algorithmic_bytes_per_launch": int(b[k]), "achieved_GBps": round(ach, 1), "frac": round(ach / 8000.0, 4)})
    roofline = {"bound": "hbm", "achieved": round(tok_gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(tok_gbs / 8000.0, 4),
                "traffic": (traffic.get("whole_token") or {}).get("fetch_bytes_per_token") if cfgname == "qwen2vl" else None,
                "scope": "whole decode token: algorithmic bytes = all Linear weights + lm_head + KV read at the mean context of the timed steps, over the device time of "
                         "the K steps (HIP events on the engine's stream)",
                "algorithmic_bytes_per_token": int(tok_bytes), "us_per_token_device": round(dev_ms * 1e3 / K, 2), "traffic_source": tname,
                "traffic_note": "FETCH_SIZE x2 (L2 memory-side requests; Infinity-Cache hits are counted): every kernel's own figure is its algorithmic bytes within 2 %; the "
                                "total is higher by the 17.1 MB per layer that the attention launch's warming workgroups request early (gate|up + o-projection rows), which the "
                                "GEMV launches then find in the Infinity Cache (profiles/r04_pmc_fetch_size.md; mechanism: r03_warm_workgroups.md)",
                "kernels": kernels,
                "step_launches": step_launches,
                "step_launches_note": "every launch of the decode step, HIP events (no system fence) either side of each on the engine's stream over eager steps: "
                                      "event-to-event = the kernel plus the command processor's hand-over, 0.3-3 us above the kernel trace's duration of the same kernel "
                                      "in profiles/r04_bench_kernel_stats.md; their sum sits the same way above us_per_token_device, which is the captured graph's",
                "step_launches_us_per_token": round(sum(e["us_per_token"] for e in step_launches), 1) if step_launches else None}

    # ---- the prefill half of the metric: MFMA-bound.  Algorithmic FLOPs of the forward (2 M N K of every Linear, 4 H S^2 D of every attention; the vision
    # tower's 1.48 TF per 448 x 448 image + the S = 282 LLM prefill's 0.74 TF) over the median device time, against the dense bf16 MFMA peak; then the two kernels that
    # are the prefill, each timed alone on its real shapes and priced against the peak of the MFMA form it runs on, with the matrix-pipe busy fraction of the PMC pass
    pre_flops = {"qwen2vl": 1.48e12 + 0.74e12, "llava": 0.39e12 + 2.0 * 589 * 6.6e9, "qwen15": 2.0 * 64 * 0.31e9, "tinyllama": 2.0 * 64 * 1.03e9}[cfgname]
    pk = prefill_kernels(cfgname) if rank == 0 else []
    mname, mfma = pmc_mfma()
    for kdesc in pk:
        key = "gemm_q4k" if kdesc["kernel"].startswith("gemm_q4k") else ("fa2_prefill_80" if "x 80" in kdesc["kernel"] else "fa2_prefill_128")
        kdesc["mfma_busy_frac"] = (mfma.get(key) or {}).get("mfma_busy_frac") if cfgname == "qwen2vl" else None
    pre_tf = pre_flops / (prefill_ms * 1e-3) / 1e12
    prefill_roofline = {"bound": "mfma", "achieved": round(pre_tf, 1), "peak": PEAK_F16_MFMA_TF, "unit": "TFLOP/s", "frac": round(pre_tf / PEAK_F16_MFMA_TF, 4),
                        "algorithmic_flops": pre_flops, "ms": round(prefill_ms, 3), "mfma_busy_source": mname,
                        "scope": "whole prefill forward (vision tower + splice + LLM prefill): algorithmic FLOPs / median device time, against the dense bf16 / fp16 MFMA peak; the "
                                 "attention runs on fp32 MFMAs (exact fma chains), whose peak is 157 TF: its rows are priced against that",
                        "kernels": pk}

    # ---- batched decode, a reported extra (never `value`): B sequences with the same prompt stepped together share one pass over the weights.  The token is launch- and
    # latency-bound, not byte-bound, so the aggregate rate grows with B; each row is bit-identical to its batch-1 run (tests/test_batched_decode.py)
    batched = None
    if args.batched and rank == 0 and cfgname in ("qwen2vl", "qwen15", "tinyllama"):
        batched = []
        bsteps = 32
        for Bn in (2, 4, 8, 15):
            m.batch_begin(Bn)
            cur = []
            for b in range(Bn):
                m.batch_select(b)
                m.clear_kvcache()
                tk, _, _ = m.prefill(ids, image, meta, want_logits=False)
                cur.append(tk)
            for _ in range(4):
                cur = m.batch_decode(cur, want_logits=False)[0].tolist()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dev = 0.0
            for _ in range(bsteps):
                nxt, _, ms_b = m.batch_decode(cur, want_logits=False)
                cur = nxt.tolist()
                dev += ms_b
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            T_b = S + 4 + bsteps // 2
            step_bytes = wbytes + Bn * cfg.layers * 2 * T_b * cfg.kv_heads * cfg.head_dim * 2
            batched.append({"B": Bn, "steps": bsteps, "tok_s_aggregate": round(Bn * bsteps / wall, 1), "ms_per_step_wall": round(wall * 1e3 / bsteps, 4),
                            "ms_per_step_device": round(dev / bsteps, 4), "algorithmic_bytes_per_step": int(step_bytes),
                            "frac": round(step_bytes / (dev / bsteps * 1e-3) / 1e9 / 8000.0, 4)})
        m.batch_select(0)

    base = None
    if cfgname == "qwen2vl" and not args.no_cpu_baseline and world == 1:
        base = cpu_baseline(path, rank, ids, image)
    if rank == 0:
        out = {
            "metric": METRIC, "value": round(value, 2), "unit": "tok/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt * 1e3 / K, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int8xint4->int32, fp32 accumulate (Q8_K x Q4_K)", "data": "synthetic",
            "config": {"workload": WORKLOADS[cfgname], "prefill_tokens": S, "parallelism": "replicas" if world > 1 else "single", "rccl_world": world},
            "prefill_ms": round(prefill_ms, 3), "prefill_ms_runs": [round(v, 3) for v in pre], "prefill_tok_s": round(1000.0 * S / prefill_ms, 1),
            "decode_weight_bytes_per_token": int(wbytes), "roofline": roofline, "prefill_roofline": prefill_roofline, "cpu_baseline": base, "vit_prefill": vit,
            "batched_decode": batched,
            "setup_s": {"weights": round(t_weights, 1), "load": round(t_load, 2)}, "load": load,
        }
        print(json.dumps(out))
    m.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
