"""ctypes binding of libmllm_hip.so (include/mllm_hip.h).  No CPU fallback: if the library is missing this raises."""
from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "libmllm_hip.so")
HEADER = os.path.join(HERE, "..", "include", "mllm_hip.h")

OK, ERR_HIP, ERR_SHAPE, ERR_DTYPE, ERR_IO, ERR_ARG = 0, -1, -2, -3, -4, -5          # MLLM_HIP_* of include/mllm_hip.h
F32, F16, Q4_0, Q8_0, Q4_K, Q8_K = 0, 1, 2, 8, 12, 15


class MllmHipError(RuntimeError):
    pass


ARCH_QWEN2VL, ARCH_QWEN, ARCH_LLAMA, ARCH_LLAVA, ARCH_VIT = 0, 1, 2, 3, 4


class ModelConfigC(C.Structure):
    """mllm_hip_model_config (include/mllm_hip.h)."""
    _fields_ = [(n, C.c_int) for n in ("arch", "hidden", "inter", "layers", "heads", "kv_heads", "vocab")] + [
        ("rms_eps", C.c_float), ("final_eps", C.c_float), ("rope_theta", C.c_float), ("mrope_section", C.c_int * 3)] + [(n, C.c_int) for n in (
            "cache_limit", "tie_embedding", "qkv_bias", "v_dim", "v_heads", "v_blocks", "v_patch", "v_merge", "v_ffn", "v_img", "v_classes",
            "image_token_id", "vision_start_token_id", "vision_end_token_id", "video_token_id")]


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
        _lib.mllm_hip_linear_workspace_bytes.restype = C.c_size_t
        _lib.mllm_hip_fa2_workspace_bytes.restype = C.c_size_t
        _lib.mllm_hip_qwen2vl_decode_weight_bytes.restype = C.c_int64
        _lib.mllm_hip_qwen2vl_decode_weight_bytes.argtypes = [C.c_void_p]
        _lib.mllm_hip_q4k_prepack_bytes.restype = C.c_size_t
        _lib.mllm_hip_q4k_wpack_bytes.restype = C.c_size_t
        _lib.mllm_hip_qwen2vl_stream.restype = C.c_void_p
        _lib.mllm_hip_qwen2vl_stream.argtypes = [C.c_void_p]
        _lib.mllm_hip_qwen2vl_destroy.restype = None
        _lib.mllm_hip_qwen2vl_destroy.argtypes = [C.c_void_p]
        _lib.mllm_hip_model_decode_weight_bytes.restype = C.c_int64
        _lib.mllm_hip_model_decode_weight_bytes.argtypes = [C.c_void_p]
        _lib.mllm_hip_model_stream.restype = C.c_void_p
        _lib.mllm_hip_model_stream.argtypes = [C.c_void_p]
        _lib.mllm_hip_model_destroy.restype = None
        _lib.mllm_hip_model_destroy.argtypes = [C.c_void_p]
        _lib.mllm_hip_sort_desc_workspace_bytes.restype = C.c_size_t
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
def set_option(name: str, value: int) -> None:
    """mllm_hip_set_option: measurement / bring-up switches of the launch paths (-1 = unset); they live in the library, not in the environment."""
    check(load().mllm_hip_set_option(name.encode(), int(value)), "set_option")


def get_option(name: str) -> int:
    v = C.c_int(0)
    check(load().mllm_hip_get_option(name.encode(), C.byref(v)), "get_option")
    return v.value


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


def rope_table_hf_llama3(base, dim, n_pos, factor, low_freq_factor, high_freq_factor, original_max_pos):
    s = np.empty((n_pos, dim), dtype=np.float32)
    c = np.empty((n_pos, dim), dtype=np.float32)
    check(load().mllm_hip_rope_table_hf_llama3(C.c_float(base), C.c_int(dim), C.c_int(n_pos), C.c_float(factor), C.c_float(low_freq_factor), C.c_float(high_freq_factor),
                                               C.c_float(original_max_pos), vp(s), vp(c)), "rope_table_hf_llama3")
    return s, c


def rope_table_ntk(theta, dim, n_pos, original_max_pos, long_factor, short_factor):
    """NTKROPE tables (CPUNTKRoPE.cpp:27-80), [n_pos][dim]; rotate with ops.rope_apply."""
    s = np.empty((n_pos, dim), dtype=np.float32)
    c = np.empty((n_pos, dim), dtype=np.float32)
    lf, sf = np.ascontiguousarray(long_factor, dtype=np.float32), np.ascontiguousarray(short_factor, dtype=np.float32)
    if lf.size != dim // 2 or sf.size != dim // 2:
        raise ValueError("long_factor / short_factor must hold dim / 2 values")
    check(load().mllm_hip_rope_table_ntk(C.c_float(theta), C.c_int(dim), C.c_int(n_pos), C.c_int(original_max_pos), vp(lf), vp(sf), vp(s), vp(c)), "rope_table_ntk")
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
def model_config(c, cache_limit=None) -> ModelConfigC:
    """mllm_hip_model_config from one of the synth.* config dataclasses (Qwen2VLConfig, CausalLMConfig, LLaVAConfig, ViTConfig)."""
    from . import synth
    cc = ModelConfigC()
    if isinstance(c, synth.Qwen2VLConfig):
        cc.arch = ARCH_QWEN2VL
        for n in ("hidden", "inter", "layers", "heads", "kv_heads", "vocab", "v_dim", "v_heads", "v_blocks", "v_patch", "v_merge",
                  "image_token_id", "vision_start_token_id", "vision_end_token_id", "video_token_id"):
            setattr(cc, n, int(getattr(c, n)))
        cc.rms_eps, cc.final_eps, cc.rope_theta = c.rms_eps, 1e-6, c.rope_theta       # model.norm: 1e-6 hard-coded (modeling_qwen2_vl.hpp:374)
        cc.mrope_section = (C.c_int * 3)(*c.mrope_section)
        cc.tie_embedding, cc.qkv_bias = int(c.tie_embedding), 1
    elif isinstance(c, synth.CausalLMConfig):
        cc.arch = ARCH_QWEN if c.family == "qwen" else ARCH_LLAMA
        for n in ("hidden", "inter", "layers", "heads", "kv_heads", "vocab"):
            setattr(cc, n, int(getattr(c, n)))
        # QWen: config.rms_norm_eps everywhere (modeling_qwen.hpp:73-105); TinyLLaMA / LLaMA: 1e-6 literals (modeling_tinyllama.hpp:30-64)
        cc.rms_eps, cc.final_eps, cc.rope_theta = c.rms_eps, c.rms_eps if c.family == "qwen" else 1e-6, c.rope_theta
        cc.tie_embedding, cc.qkv_bias = int(c.tie_embedding), int(c.qkv_bias)
    elif isinstance(c, synth.LLaVAConfig):
        cc.arch = ARCH_LLAVA
        cc.hidden, cc.inter, cc.layers, cc.heads, cc.kv_heads, cc.vocab = c.hidden, c.inter, c.layers, c.heads, c.heads, c.vocab
        cc.rms_eps, cc.final_eps, cc.rope_theta = c.rms_eps, 1e-6, c.rope_theta
        cc.tie_embedding, cc.qkv_bias = 0, 0
        cc.v_dim, cc.v_heads, cc.v_blocks, cc.v_patch, cc.v_ffn, cc.v_img = c.v_hidden, c.v_heads, c.v_blocks, c.patch, c.v_ffn, c.img
        cc.image_token_id = c.image_token_id
    elif isinstance(c, synth.ViTConfig):
        cc.arch = ARCH_VIT
        cc.v_dim, cc.v_heads, cc.v_blocks, cc.v_patch, cc.v_ffn, cc.v_img, cc.v_classes = c.hidden, c.heads, c.blocks, c.patch, c.ffn, c.img, c.classes
    else:
        raise MllmHipError(f"no engine config for {type(c).__name__}")
    if cc.arch != ARCH_VIT:
        cc.cache_limit = int(cache_limit or c.cache_limit)
    return cc


class Model:
    """Host mirror of the demos' use of a reference model (examples/demo_qwen2_vl.cpp, demo_qwen.cpp, demo_llava.cpp, demo_vit.cpp): load, prefill,
    decode, generate, clear_kvcache -- on the resident C++ engine (csrc/engine.hip) through the C ABI."""

    def __init__(self, cfg, mllm_path: str, device: int = 0, cache_limit=None):
        lib = load()
        check(lib.mllm_hip_init(C.c_int(device)), "mllm_hip_init")
        self.cfg = cfg
        self._cc = model_config(cfg, cache_limit)
        self._h = C.c_void_p()
        check(lib.mllm_hip_model_create(C.byref(self._cc), mllm_path.encode(), C.byref(self._h)), "model_create")

    @property
    def vocab(self) -> int:
        return int(self._cc.vocab)

    def close(self):
        if self._h:
            load().mllm_hip_model_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear_kvcache(self):
        check(load().mllm_hip_model_clear_kvcache(self._h))

    def batch_begin(self, B: int):
        """B independent sequences (own KV slabs) on this model; sequence 0 is the model's own cache."""
        check(load().mllm_hip_model_batch_begin(self._h, C.c_int(B)), "batch_begin")

    def batch_select(self, seq: int):
        """Make `seq` the sequence prefill / decode / generate / clear_kvcache / cache_len act on."""
        check(load().mllm_hip_model_batch_select(self._h, C.c_int(seq)), "batch_select")

    def batch_decode(self, tokens, want_logits=True):
        """One step for sequences 0 .. len(tokens) - 1 together: (next greedy ids [B], logits [B][vocab] or None, device ms)."""
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        B = int(t.size)
        nxt = np.empty(B, dtype=np.int32)
        lg = np.empty((B, self.vocab), dtype=np.float32) if want_logits else None
        ms = C.c_float()
        check(load().mllm_hip_model_batch_decode(self._h, C.c_int(B), vp(t), vp(lg), vp(nxt), C.byref(ms)), "batch_decode")
        return nxt, lg, ms.value

    def cache_len(self) -> int:
        """Tokens the KV cache holds (0 on a fresh or cleared model)."""
        return int(load().mllm_hip_model_cache_len(self._h))

    def load_stats(self):
        tot, h2d, tail, nb = C.c_float(), C.c_float(), C.c_float(), C.c_int64()
        check(load().mllm_hip_model_load_stats(self._h, C.byref(tot), C.byref(nb), C.byref(h2d), C.byref(tail)), "load_stats")
        return {"total_ms": tot.value, "file_bytes": nb.value, "h2d_ms": h2d.value, "repack_tail_ms": tail.value}

    def memory_stats(self):
        res, rel = C.c_int64(), C.c_int64()
        check(load().mllm_hip_model_memory_stats(self._h, C.byref(res), C.byref(rel)), "memory_stats")
        return {"resident_bytes": res.value, "released_bytes": rel.value}

    def prefill(self, ids, image=None, image_meta=None, want_logits=True, visual_dev=None, n_visual_rows=0):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        img = np.ascontiguousarray(image, dtype=np.float32) if image is not None else None
        meta = np.ascontiguousarray(image_meta, dtype=np.int32) if image_meta is not None else None
        logits = np.empty(self.vocab, dtype=np.float32) if want_logits else None
        tok = C.c_int32()
        ms = C.c_float()
        check(load().mllm_hip_model_prefill(self._h, vp(ids), C.c_int(ids.size), vp(img), vp(meta), vp(visual_dev), C.c_int(n_visual_rows), vp(logits),
                                            C.byref(tok), C.byref(ms)), "prefill")
        return tok.value, logits, ms.value

    def decode(self, token, want_logits=True):
        logits = np.empty(self.vocab, dtype=np.float32) if want_logits else None
        tok = C.c_int32()
        ms = C.c_float()
        check(load().mllm_hip_model_decode(self._h, C.c_int32(int(token)), vp(logits), C.byref(tok), C.byref(ms)), "decode")
        return tok.value, logits, ms.value

    def generate(self, first_token, steps):
        toks = np.empty(steps, dtype=np.int32)
        ms = C.c_float()
        check(load().mllm_hip_model_generate(self._h, C.c_int32(int(first_token)), C.c_int(steps), vp(toks), C.byref(ms)), "generate")
        return toks, ms.value

    def generate_sampled(self, first_token, steps, method, u01, top_k=5, top_p=0.92, temperature=0.7, eos=-1):
        """Module::generate with LlmTextGeneratorOpts (mllm/Generate.hpp:26-36 defaults): method 0 greedy, 1 top-k, 2 top-p."""
        toks = np.empty(steps, dtype=np.int32)
        u = np.ascontiguousarray(u01, dtype=np.float32)
        assert u.size >= steps
        n, ms = C.c_int(), C.c_float()
        check(load().mllm_hip_model_generate_sampled(self._h, C.c_int32(int(first_token)), C.c_int(steps), C.c_int(method), C.c_int(top_k), C.c_float(top_p),
                                                     C.c_float(temperature), vp(u), C.c_int32(eos), vp(toks), C.byref(n), C.byref(ms)), "generate_sampled")
        return toks[:n.value], ms.value

    def greedy(self, ids, steps, image=None, image_meta=None):
        """The demos' loop through single forwards: prefill, then `steps - 1` decode calls; returns (ids, [logits per step])."""
        tok, lg, _ = self.prefill(ids, image, image_meta)
        toks, logits = [tok], [lg]
        for _ in range(steps - 1):
            tok, lg, _ = self.decode(tok)
            toks.append(tok)
            logits.append(lg)
        return toks, logits

    def vision_shape(self, image_meta=None):
        r, c = C.c_int(), C.c_int()
        meta = np.ascontiguousarray(image_meta, dtype=np.int32) if image_meta is not None else None
        check(load().mllm_hip_model_vision_shape(self._h, vp(meta), C.byref(r), C.byref(c)), "vision_shape")
        return r.value, c.value

    def vision(self, images, image_meta, out_dev_ptr, n_img=1):
        img = np.ascontiguousarray(images, dtype=np.float32)
        meta = np.ascontiguousarray(image_meta, dtype=np.int32) if image_meta is not None else None
        ms = C.c_float()
        check(load().mllm_hip_model_vision(self._h, vp(img), vp(meta), C.c_int(n_img), vp(out_dev_ptr), C.byref(ms)), "vision")
        return ms.value

    def decode_weight_bytes(self) -> int:
        return load().mllm_hip_model_decode_weight_bytes(self._h)

    def stream(self) -> int:
        return load().mllm_hip_model_stream(self._h)

    def time_kernel(self, which=0, iters=50):
        ms = C.c_float()
        nbytes = C.c_int64()
        check(load().mllm_hip_model_time_kernel(self._h, C.c_int(which), C.c_int(iters), C.byref(ms), C.byref(nbytes)), "time_kernel")
        return ms.value, nbytes.value

    time_gemv = time_kernel

    STEP_KINDS = ("qkv", "attn", "o_proj", "gateup", "down", "chain", "front", "head", "next")

    def time_step(self, first_token, steps=8):
        """The decode step launch by launch (HIP events either side of every launch, eager): {kind: (us per launch, launches per step)}, last token."""
        us = (C.c_float * 9)()
        n = (C.c_int32 * 9)()
        tok = C.c_int32()
        check(load().mllm_hip_model_time_step(self._h, C.c_int32(int(first_token)), C.c_int(steps), us, n, C.byref(tok)), "time_step")
        return {k: (float(us[i]), int(n[i])) for i, k in enumerate(self.STEP_KINDS) if n[i]}, tok.value


class Qwen2VL(Model):
    """demo_qwen2_vl.cpp's model: the generic engine with the Qwen2-VL graph (pixel_values + grid_thw as the image)."""

    def prefill(self, ids, pixel_values=None, grid_thw=None, want_logits=True, **kw):
        return super().prefill(ids, pixel_values, grid_thw, want_logits, **kw)


def make_config(c, cache_limit=None) -> Qwen2VLConfigC:
    """mllm_hip_qwen2vl_config of the round-1 entry points (kept as forwards onto the generic engine)."""
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
