"""TEST FIXTURE TOOLING, not product: the host-side fp32 -> Q4_K / Q4_0 / Q8_0 block quantiser used to write synthetic `.mllm` files
(tests/fixtures/quantizer/host_quantize.cpp -> tests/fixtures/libmllm_quant.so).  It follows the reference's quantiser statement for statement (it has to:
the files must be byte-identical to what `quantize ... Q4_K` writes, tests/test_host.py), which is why it lives under tests/ and nothing in mllm_amd/ imports it."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "quantizer", "host_quantize.cpp")
SO = os.path.join(HERE, "libmllm_quant.so")
F32, F16, Q4_0, Q8_0, Q4_K, Q8_K = 0, 1, 2, 8, 12, 15
_lib = None


def build(force: bool = False) -> str:
    """g++ with the reference's x86 flags (CMakeLists.txt:172-176 minus -march=native); explicit fmaf + -ffp-contract=off fix the rounding."""
    if force or not os.path.exists(SO) or os.path.getmtime(SRC) > os.path.getmtime(SO):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-mavx2", "-mf16c", "-mfma", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", SRC, "-o", SO])
    return SO


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.mllm_quant_nbytes.restype = C.c_int64
        _lib.mllm_quant_nbytes.argtypes = [C.c_int, C.c_int64]
        _lib.mllm_quant_rows.restype = C.c_int
        _lib.mllm_quant_rows.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
    return _lib


def nbytes(dtype: int, n_elem: int) -> int:
    return int(load().mllm_quant_nbytes(dtype, n_elem))


def quantize(dtype: int, x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32).ravel()
    n = nbytes(dtype, x.size)
    if n < 0:
        raise ValueError(f"cannot quantize {x.size} elements to dtype {dtype}")
    out = np.empty(n, dtype=np.uint8)
    rc = load().mllm_quant_rows(dtype, x.ctypes.data, out.ctypes.data, x.size)
    if rc:
        raise ValueError(f"mllm_quant_rows failed with code {rc}")
    return out
