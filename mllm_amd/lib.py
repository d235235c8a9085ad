"""ctypes binding of libmllm_hip.so (include/mllm_hip.h).  No CPU fallback: if the library is missing this raises."""
from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "libmllm_hip.so")
HEADER = os.path.join(HERE, "..", "include", "mllm_hip.h")

OK = 0
F32, F16, Q4_0, Q8_0, Q4_K, Q8_K = 0, 1, 2, 8, 12, 15


class MllmHipError(RuntimeError):
    pass


class Qwen2VLConfigC(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("hidden", "inter", "layers", "heads", "kv_heads", "vocab")] + [
        ("rms_eps", C.c_float), ("rope_theta", C.c_float), ("mrope_section", C.c_int * 3), ("cache_limit", C.c_int),
        ("tie_embedding", C.c_int)] + [(n, C.c_int) for n in (
            "v_dim", "v_heads", "v_blocks", "v_patch", "v_merge", "image_token_id", "vision_start_token_id",
            "vision_end_token_id", "video_token_id")]


_lib = None


def declared_symbols() -> list[str]:
    """Every function the C header declares (used by the CPU-side export test)."""
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"\b(mllm_hip_[a-z0-9_]+)\s*\(", txt)))


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise MllmHipError(f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback for the HIP path)")
        _lib = C.CDLL(SO_PATH)
        _lib.mllm_hip_last_error.restype = C.c_char_p
        _lib.mllm_hip_quantized_nbytes.restype = C.c_int64
        _lib.mllm_hip_quantized_nbytes.argtypes = [C.c_int, C.c_int64]
        _lib.mllm_hip_linear_workspace_bytes.restype = C.c_size_t
        _lib.mllm_hip_fa2_workspace_bytes.restype = C.c_size_t
        _lib.mllm_hip_qwen2vl_decode_weight_bytes.restype = C.c_int64
        _lib.mllm_hip_qwen2vl_decode_weight_bytes.argtypes = [C.c_void_p]
        _lib.mllm_hip_q4k_prepack_bytes.restype = C.c_size_t
        _lib.mllm_hip_qwen2vl_stream.restype = C.c_void_p
        _lib.mllm_hip_qwen2vl_stream.argtypes = [C.c_void_p]
        _lib.mllm_hip_qwen2vl_destroy.restype = None
        _lib.mllm_hip_qwen2vl_destroy.argtypes = [C.c_void_p]
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != OK:
        raise MllmHipError(f"{what} failed with code {rc}: {load().mllm_hip_last_error().decode()}")


def vp(x):
    """void* of a torch tensor / numpy array / int / None."""
    if x is None:
        return C.c_void_p(0)
    if isinstance(x, int):
        return C.c_void_p(x)
    if isinstance(x, np.ndarray):
        return x.ctypes.data_as(C.c_void_p)
    return C.c_void_p(x.data_ptr())


def i64(v):
    return C.c_int64(int(v))


# ---- host-side helpers (no GPU needed) --------------------------------------------------------------------------------
def quantize_host(dtype: int, x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32).ravel()
    n = load().mllm_hip_quantized_nbytes(dtype, x.size)
    if n < 0:
        raise MllmHipError(f"cannot quantize {x.size} elements to dtype {dtype}")
    out = np.empty(n, dtype=np.uint8)
    check(load().mllm_hip_quantize_host(C.c_int(dtype), vp(x), vp(out), i64(x.size)), "quantize_host")
    return out


def build_act_luts():
    g = np.empty(65536, dtype=np.uint16)
    q = np.empty(65536, dtype=np.uint16)
    check(load().mllm_hip_build_act_luts(vp(g), vp(q)), "build_act_luts")
    return g, q


def rope_table_hf(base, dim, n_pos):
    s = np.empty((n_pos, dim), dtype=np.float32)
    c = np.empty((n_pos, dim), dtype=np.float32)
    check(load().mllm_hip_rope_table_hf(C.c_float(base), C.c_int(dim), C.c_int(n_pos), vp(s), vp(c)))
    return s, c


def mrope_table(base, dim, pos, section=(16, 24, 24)):
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    S = pos.shape[1]
    sec = np.asarray(section, dtype=np.int32)
    s = np.zeros((S, dim // 2), dtype=np.float32)
    c = np.zeros((S, dim // 2), dtype=np.float32)
    check(load().mllm_hip_mrope_table(C.c_float(base), C.c_int(dim), vp(pos), C.c_int(S), vp(sec), C.c_int(len(sec)), vp(s), vp(c)))
    return s, c


def vision_rope_table(t, h, w, merge, rot_dim):
    s = np.empty((t * h * w, rot_dim), dtype=np.float32)
    c = np.empty((t * h * w, rot_dim), dtype=np.float32)
    check(load().mllm_hip_vision_rope_table(C.c_int(t), C.c_int(h), C.c_int(w), C.c_int(merge), C.c_int(rot_dim), vp(s), vp(c)))
    return s, c


# ---- engine wrapper ------------------------------------------------------------------------------------------------------
def make_config(c, cache_limit=None) -> Qwen2VLConfigC:
    cc = Qwen2VLConfigC()
    for n in ("hidden", "inter", "layers", "heads", "kv_heads", "vocab", "v_dim", "v_heads", "v_blocks", "v_patch", "v_merge",
              "image_token_id", "vision_start_token_id", "vision_end_token_id", "video_token_id"):
        setattr(cc, n, int(getattr(c, n)))
    cc.rms_eps = c.rms_eps
    cc.rope_theta = c.rope_theta
    cc.mrope_section = (C.c_int * 3)(*c.mrope_section)
    cc.cache_limit = int(cache_limit or c.cache_limit)
    cc.tie_embedding = int(c.tie_embedding)
    return cc


class Qwen2VL:
    """Host mirror of demo_qwen2_vl.cpp's use of Qwen2VLModel: load, prefill, decode, generate, clear_kvcache."""

    def __init__(self, cfg, mllm_path: str, device: int = 0, cache_limit=None):
        lib = load()
        check(lib.mllm_hip_init(C.c_int(device)), "mllm_hip_init")
        self.cfg = cfg
        self._cc = make_config(cfg, cache_limit)
        self._h = C.c_void_p()
        check(lib.mllm_hip_qwen2vl_create(C.byref(self._cc), mllm_path.encode(), C.byref(self._h)), "qwen2vl_create")

    def close(self):
        if self._h:
            load().mllm_hip_qwen2vl_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear_kvcache(self):
        check(load().mllm_hip_qwen2vl_clear_kvcache(self._h))

    def prefill(self, ids, pixel_values=None, grid_thw=None, want_logits=True):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        pix = np.ascontiguousarray(pixel_values, dtype=np.float32) if pixel_values is not None else None
        grid = np.ascontiguousarray(grid_thw, dtype=np.int32) if grid_thw is not None else None
        logits = np.empty(self.cfg.vocab, dtype=np.float32) if want_logits else None
        tok = C.c_int32()
        ms = C.c_float()
        check(load().mllm_hip_qwen2vl_prefill(self._h, vp(ids), C.c_int(ids.size), vp(pix), vp(grid), vp(logits), C.byref(tok), C.byref(ms)), "prefill")
        return tok.value, logits, ms.value

    def decode(self, token, want_logits=True):
        logits = np.empty(self.cfg.vocab, dtype=np.float32) if want_logits else None
        tok = C.c_int32()
        ms = C.c_float()
        check(load().mllm_hip_qwen2vl_decode(self._h, C.c_int32(int(token)), vp(logits), C.byref(tok), C.byref(ms)), "decode")
        return tok.value, logits, ms.value

    def generate(self, first_token, steps):
        toks = np.empty(steps, dtype=np.int32)
        ms = C.c_float()
        check(load().mllm_hip_qwen2vl_generate(self._h, C.c_int32(int(first_token)), C.c_int(steps), vp(toks), C.byref(ms)), "generate")
        return toks, ms.value

    def vision(self, pixel_values, grid_thw, embeds_dev_ptr, n_img=1):
        pix = np.ascontiguousarray(pixel_values, dtype=np.float32)
        grid = np.ascontiguousarray(grid_thw, dtype=np.int32)
        ms = C.c_float()
        check(load().mllm_hip_qwen2vl_vision(self._h, vp(pix), vp(grid), C.c_int(n_img), vp(embeds_dev_ptr), C.byref(ms)), "vision")
        return ms.value

    def decode_weight_bytes(self) -> int:
        return load().mllm_hip_qwen2vl_decode_weight_bytes(self._h)

    def stream(self) -> int:
        return load().mllm_hip_qwen2vl_stream(self._h)

    def time_gemv(self, which=0, iters=50):
        ms = C.c_float()
        nbytes = C.c_int64()
        check(load().mllm_hip_qwen2vl_time_gemv(self._h, C.c_int(which), C.c_int(iters), C.byref(ms), C.byref(nbytes)), "time_gemv")
        return ms.value, nbytes.value
