"""The measurement contract: the committed bench line carries every field the driver and the judge read, its roofline
arithmetic is self-consistent, and bench.py refuses to run without a HIP device (there is no CPU fallback to measure by accident)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line_path():
    import glob
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_line.json")))[-1]


def _line():
    """The newest committed bench line (profiles/rNN_bench_line.json)."""
    return json.load(open(_line_path()))


def test_committed_bench_line_has_the_contract_fields():
    d = _line()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"]
    for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "tok/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None          # BASELINE.md holds no published number for this metric
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1000.0 * d["n_gpus"] / d["ms_per_step"]) / d["value"] < 1e-3
    assert d["prefill_ms"] > 0


def test_roofline_and_cpu_baseline_objects():
    d = _line()
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    if "kernels" in r:
        # round 2 on: the headline is the whole decode token (algorithmic bytes per token / device time per token) ...
        assert abs(r["achieved"] - r["algorithmic_bytes_per_token"] / (r["us_per_token_device"] * 1e-6) / 1e9) / r["achieved"] < 1e-2
        assert d["decode_weight_bytes_per_token"] == 868257792 and r["algorithmic_bytes_per_token"] > d["decode_weight_bytes_per_token"]      # + the KV read
        # ... and each named kernel: achieved = algorithmic bytes per launch / measured launch duration
        g = r["kernels"][0]
        assert "gateup" in g["kernel"] and g["algorithmic_bytes_per_launch"] == 17920 * 6 * 144      # gate|up of Qwen2-VL-2B: 17920 rows x 6 super-blocks x 144 B
        for k in r["kernels"]:
            assert abs(k["achieved_GBps"] - k["algorithmic_bytes_per_launch"] / (k["us_per_launch"] * 1e-6) / 1e9) / k["achieved_GBps"] < 1e-2
        t = g["traffic_bytes_per_launch"]
        assert t is None or 0.95 < t / g["algorithmic_bytes_per_launch"] < 1.10      # PMC traffic within a few percent of the algorithmic bytes: no wasted re-reads
    else:
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["us_per_launch"] * 1e-6) / 1e9) / r["achieved"] < 1e-2
        assert r["algorithmic_bytes_per_launch"] == 17920 * 6 * 144
        assert r["traffic"] is None or 0.95 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.10
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["unit"] == "tok/s" and c["cores"] >= 1


def test_step_launches_recompute():
    """Round 4: the decode step launch by launch (mllm_hip_model_time_step) -- every entry's rate follows from its bytes and its time, the launches add up to the step the
    engine runs (60 on the 2 B model, 27 of them the chain launch), and their summed time sits above the captured graph's token by the markers' cost only."""
    d = _line()
    r = d["roofline"]
    sl = r.get("step_launches")
    assert sl, "the committed line predates step_launches"
    for e in sl:
        if e["algorithmic_bytes_per_launch"]:
            ach = e["algorithmic_bytes_per_launch"] / (e["us_per_launch"] * 1e-6) / 1e9
            assert abs(ach - e["achieved_GBps"]) <= 0.002 * ach + 0.1 and abs(ach / 8000.0 - e["frac"]) < 2e-4
        assert abs(e["us_per_launch"] * e["launches_per_token"] - e["us_per_token"]) < 0.1
    by = {e["launch"]: e for e in sl}
    assert by["chain"]["launches_per_token"] == 27 and sum(e["launches_per_token"] for e in sl) == 60
    assert sl[0]["launch"] == "chain"      # sorted by time per token: the chain launch is the step's dominant one
    total = sum(e["us_per_token"] for e in sl)
    assert abs(total - r["step_launches_us_per_token"]) < 0.5
    assert r["us_per_token_device"] < total < 1.12 * r["us_per_token_device"]
    # (the KV share is priced at the context these extra steps run at, past the timed steps' mean)
    assert 0.99 * r["algorithmic_bytes_per_token"] < sum(e["algorithmic_bytes_per_launch"] * e["launches_per_token"] for e in sl) < 1.01 * r["algorithmic_bytes_per_token"]


def test_prefill_roofline_and_host_cpus():
    """Round 4 on: the prefill half of the metric carries its own roofline object (MFMA-bound: algorithmic FLOPs of the forward over the median device time against the dense
    bf16 peak, and the GEMM / attention kernels each against the peak of the MFMA form they run on, with the PMC pass's matrix-pipe busy fraction), every fraction is
    recomputable from the numbers beside it, and cpu_baseline states the box's core count next to the threads used."""
    if os.path.basename(_line_path()) < "r04":
        return
    d = _line()
    p = d["prefill_roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "algorithmic_flops", "ms", "kernels"):
        assert k in p, k
    assert p["bound"] == "mfma" and p["unit"] == "TFLOP/s" and p["peak"] == 2500.0
    assert abs(p["ms"] - d["prefill_ms"]) < 1e-6
    assert abs(p["achieved"] - p["algorithmic_flops"] / (p["ms"] * 1e-3) / 1e12) / p["achieved"] < 1e-2 and abs(p["frac"] - p["achieved"] / p["peak"]) < 1e-3
    names = [k["kernel"] for k in p["kernels"]]
    assert any(n.startswith("gemm_q4k") for n in names) and any(n.startswith("fa2_prefill") for n in names)
    for k in p["kernels"]:
        assert abs(k["achieved_TFps"] - k["flops_per_launch"] / (k["us_per_launch"] * 1e-6) / 1e12) / k["achieved_TFps"] < 2e-2
        assert abs(k["frac"] - k["achieved_TFps"] / k["peak_TFps"]) < 1e-3
        assert k["peak_TFps"] == (2500.0 if k["kernel"].startswith("gemm_q4k") else 157.3)
        assert k["mfma_busy_frac"] is None or 0.0 < k["mfma_busy_frac"] < 1.0
    fc1 = next(k for k in p["kernels"] if k["kernel"] == "gemm_q4k vit fc1")
    assert fc1["shape_MNK"] == [1024, 5120, 1280] and fc1["flops_per_launch"] == 2.0 * 1024 * 5120 * 1280
    c = d["cpu_baseline"]
    assert c["host_cpus"] >= c["cores"] >= 1


def test_bench_fails_loudly_without_a_hip_device():
    import torch
    if torch.cuda.is_available():
        return   # on the GPU box this is exercised by the bench run itself
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "no HIP device" in (p.stderr + p.stdout)
    assert not any(l.startswith("{") for l in p.stdout.splitlines())      # and prints no measurement


def test_bench_two_ranks_dry_run_under_torchrun(tmp_path):
    """bench.py's N > 1 skeleton, launched the way the driver launches it (python -m torch.distributed.run, rendezvous on 127.0.0.1), on gloo with the stub engine of
    --dry-run: rank 1 is 1.25x slower than rank 0 by construction, so the barriers + MAX over ranks must give value = world * K / (K * 1.25 ms) = 1600 tok/s (sleep
    granularity allows a few per cent), one JSON line from rank 0 only, rccl_world = n_gpus = 2, the vision prefill sharded with one all-gather per tower pass."""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "4", "--dry-run"], capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["dry_run"] is True and r["n_gpus"] == 2 and r["config"]["rccl_world"] == 2 and r["config"]["parallelism"] == "replicas" and r["scaling"] == "weak"
    assert r["steps"] == 40 and r["warmup"] == 4 and r["roofline"] is None and r["cpu_baseline"] is None
    assert 1250.0 <= r["value"] <= 1620.0, r["value"]                      # 2 * 40 / (40 * 1.25 ms), never the faster rank's 2000
    assert abs(r["ms_per_step"] - 1000.0 * 2 / r["value"]) < 1e-3
    assert r["vit_prefill"]["world"] == 2 and r["vit_prefill"]["images"] == 8 and r["vit_prefill"]["all_gathers"] == 4
    # without --dry-run the same command refuses to run where there is no HIP device
    if not __import__("torch").cuda.is_available():
        bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "0"], capture_output=True, text=True, timeout=300, cwd=root)
        assert bad.returncode != 0 and "no HIP device" in (bad.stderr + bad.stdout)
