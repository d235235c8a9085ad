"""Builds libmllm_hip.so (HIP kernels + C-ABI + host engine) in-tree for gfx950.

hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
SO = os.path.join(HERE, "libmllm_hip.so")
HIP_SOURCES = ["runtime.hip", "kernels_elem.hip", "kernels_linear.hip", "kernels_attn.hip", "kernels_decode.hip", "kernels_sample.hip", "kernels_image.hip", "kernels_n4.hip", "moe.hip", "engine.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in ("common.h", "decode_launch.h", "q4k_dot.h", "q40_dot.h", "kernels_attn_core.h")]
    headers.append(os.path.join(HERE, "..", "include", "mllm_hip.h"))
    objs = []
    procs = []
    for s in HIP_SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("build failed: " + " ".join(cmd))
    if force or procs or _stale(SO, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs + ["-L/opt/rocm/lib", "-lrccl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
