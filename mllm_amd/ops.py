"""Python face of the C-ABI launchers over torch CUDA(HIP) tensors: one function per reference Op on the hot path
(names follow the reference's OpType / Layer names).  torch is plumbing only (device memory + current stream);
every computation runs in libmllm_hip.so.  There is no CPU fallback: calling these without a GPU raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import lib as L
from .lib import F16, F32, Q4_0, Q4_K, check, i64, vp


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(x, dtype=None):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if dtype is not None:
        x = x.to(dtype)
    return x.contiguous().cuda()


def require_gpu():
    if not torch.cuda.is_available():
        raise L.MllmHipError("mllm_amd.ops needs an MI355X: no HIP device is visible (no CPU fallback exists)")
    check(L.load().mllm_hip_init(C.c_int(torch.cuda.current_device())), "mllm_hip_init")


class Q8K:
    """q8k planes of `[M][K]` activations."""

    def __init__(self, M, K):
        self.M, self.K = M, K
        self.qs = torch.empty((M, K), dtype=torch.int8, device="cuda")
        self.d = torch.empty((M, K // 256), dtype=torch.float32, device="cuda")
        self.bsums = torch.empty((M, K // 16), dtype=torch.int16, device="cuda")


def quantize_q8k(x) -> Q8K:
    x = _dev(x, torch.float32)
    M, K = x.shape
    q = Q8K(M, K)
    check(L.load().mllm_hip_quantize_q8k(vp(x), vp(q.qs), vp(q.d), vp(q.bsums), C.c_int(M), C.c_int(K), _stream()), "quantize_q8k")
    return q


def quantize_q80(x):
    x = _dev(x, torch.float32)
    M, K = x.shape
    qs = torch.empty((M, K), dtype=torch.int8, device="cuda")
    d = torch.empty((M, K // 32), dtype=torch.int16, device="cuda")
    check(L.load().mllm_hip_quantize_q80(vp(x), vp(qs), vp(d), C.c_int(M), C.c_int(K), _stream()), "quantize_q80")
    return qs, d


def linear_q4k(W_raw, x, N, bias=None, residual=None, out_f16=False, xq: Q8K | None = None):
    """CPULinear with Q4_K weights: W_raw uint8 `[N * K/256 * 144]`, x fp32 `[M][K]`."""
    W = _dev(W_raw, torch.uint8)
    if xq is None:
        xq = quantize_q8k(x)
    M, K = xq.M, xq.K
    b = _dev(bias, torch.float32) if bias is not None else None
    r = _dev(residual, torch.float32) if residual is not None else None
    y = torch.empty((M, N), dtype=torch.float16 if out_f16 else torch.float32, device="cuda")
    check(L.load().mllm_hip_linear_q4k_q8k(vp(W), vp(b), vp(xq.qs), vp(xq.d), vp(xq.bsums), vp(y), C.c_int(F16 if out_f16 else F32), i64(N), vp(r),
                                           C.c_int(M), C.c_int(N), C.c_int(K), _stream()), "linear_q4k_q8k")
    return y


def repack_q40(raw, n_blocks):
    raw = _dev(raw, torch.uint8)
    qs = torch.empty(n_blocks * 16, dtype=torch.uint8, device="cuda")
    d = torch.empty(n_blocks, dtype=torch.int16, device="cuda")
    check(L.load().mllm_hip_repack_q40(vp(raw), vp(qs), vp(d), i64(n_blocks), _stream()), "repack_q40")
    return qs, d


def linear_q40(W_raw, x, N, bias=None):
    x = _dev(x, torch.float32)
    M, K = x.shape
    Wqs, Wd = repack_q40(W_raw, N * K // 32)
    xqs, xd = quantize_q80(x)
    b = _dev(bias, torch.float32) if bias is not None else None
    y = torch.empty((M, N), dtype=torch.float32, device="cuda")
    check(L.load().mllm_hip_linear_q40_q80(vp(Wqs), vp(Wd), vp(b), vp(xqs), vp(xd), vp(y), i64(N), C.c_int(M), C.c_int(N), C.c_int(K), _stream()), "linear_q40_q80")
    return y


def linear_f32(W, x, bias=None):
    W, x = _dev(W, torch.float32), _dev(x, torch.float32)
    M, K = x.shape
    N = W.numel() // K
    b = _dev(bias, torch.float32) if bias is not None else None
    y = torch.empty((M, N), dtype=torch.float32, device="cuda")
    check(L.load().mllm_hip_linear_f32(vp(W), vp(b), vp(x), vp(y), i64(N), C.c_int(M), C.c_int(N), C.c_int(K), _stream()), "linear_f32")
    return y


def gemm_f32_bhsd(a, b):
    """F_MM on BHSD operands (CPUMatmulFunc.hpp:155-172 -> GemmFp.hpp:104-150): a [heads, M, K] fp32, b [heads, K, N] fp32 or fp16 -> [heads, M, N] fp32."""
    a = _dev(a, torch.float32)
    b16 = (b.dtype == torch.float16) if isinstance(b, torch.Tensor) else (np.asarray(b).dtype == np.float16)
    b = _dev(b, torch.float16 if b16 else torch.float32)
    H, M, K = a.shape
    N = b.shape[2]
    assert tuple(b.shape) == (H, K, N)
    c = torch.empty((H, M, N), dtype=torch.float32, device="cuda")
    check(L.load().mllm_hip_gemm_f32_bhsd(vp(a), vp(b), C.c_int(L.F16 if b16 else L.F32), vp(c), C.c_int(H), C.c_int(M), C.c_int(N), C.c_int(K), _stream()), "gemm_f32_bhsd")
    return c


def embedding_q40(ids, W_raw, vocab, hidden):
    Wqs, Wd = repack_q40(W_raw, vocab * hidden // 32)
    ids = _dev(np.asarray(ids, dtype=np.float32))
    out = torch.empty((ids.numel(), hidden), dtype=torch.float32, device="cuda")
    check(L.load().mllm_hip_embedding_q40(vp(ids), vp(Wqs), vp(Wd), vp(out), C.c_int(ids.numel()), C.c_int(hidden), C.c_int(vocab), _stream()), "embedding_q40")
    return out


def rmsnorm(x, w, eps, add_unit_offset=False, quant=False):
    x, w = _dev(x, torch.float32), _dev(w, torch.float32)
    M, dim = x.shape
    y = torch.empty_like(x)
    q = Q8K(M, dim) if quant else None
    check(L.load().mllm_hip_rmsnorm(vp(x), vp(w), vp(y), vp(q.qs if q else None), vp(q.d if q else None), vp(q.bsums if q else None),
                                    C.c_int(M), C.c_int(dim), C.c_float(eps), C.c_int(int(add_unit_offset)), _stream()), "rmsnorm")
    return (y, q) if quant else y


def layernorm(x, w, b, eps, quant=False):
    x, w = _dev(x, torch.float32), _dev(w, torch.float32)
    b = _dev(b, torch.float32) if b is not None else None
    M, dim = x.shape
    y = torch.empty_like(x)
    q = Q8K(M, dim) if quant else None
    check(L.load().mllm_hip_layernorm(vp(x), vp(w), vp(b), vp(y), vp(q.qs if q else None), vp(q.d if q else None), vp(q.bsums if q else None),
                                      C.c_int(M), C.c_int(dim), C.c_float(eps), _stream()), "layernorm")
    return (y, q) if quant else y


def _unary(fn, x, *extra):
    x = _dev(x, torch.float32)
    y = torch.empty_like(x)
    check(getattr(L.load(), fn)(vp(x), vp(y), i64(x.numel()), *extra, _stream()), fn)
    return y


def silu(x):
    return _unary("mllm_hip_silu", x)


def silu_rows(x):
    """CPUSiLU on rows `[rows][dim]`: the dim % 8 trailing values of every row take libm's expf (mllm_hip_silu_rows)."""
    x = _dev(x, torch.float32)
    y = torch.empty_like(x)
    check(L.load().mllm_hip_silu_rows(vp(x), vp(y), i64(x.shape[0]), C.c_int(x.shape[1]), _stream()), "silu_rows")
    return y


_luts = None


def _act_luts():
    global _luts
    if _luts is None:
        g, q = L.build_act_luts()
        _luts = (_dev(g.view(np.int16)), _dev(q.view(np.int16)))
    return _luts


def gelu(x):
    return _unary("mllm_hip_act_lut", x, vp(_act_luts()[0]))


def quickgelu(x):
    return _unary("mllm_hip_act_lut", x, vp(_act_luts()[1]))


def silu_mul(gu, I):
    gu = _dev(gu, torch.float32)
    M = gu.shape[0]
    y = torch.empty((M, I), dtype=torch.float32, device="cuda")
    check(L.load().mllm_hip_silu_mul(vp(gu), vp(y), C.c_int(M), C.c_int(I), _stream()), "silu_mul")
    return y


def add(a, b):
    a, b = _dev(a, torch.float32), _dev(b, torch.float32)
    y = torch.empty_like(a)
    check(L.load().mllm_hip_add(vp(a), vp(b), vp(y), i64(a.numel()), _stream()), "add")
    return y


def mul(a, b):
    a, b = _dev(a, torch.float32), _dev(b, torch.float32)
    y = torch.empty_like(a)
    check(L.load().mllm_hip_mul(vp(a), vp(b), vp(y), i64(a.numel()), _stream()), "mul")
    return y


def softmax(x, valid=None):
    x = _dev(x, torch.float32)
    n = x.shape[-1]
    rows = x.numel() // n
    y = torch.empty_like(x)
    v = _dev(np.asarray(valid, dtype=np.int32)) if valid is not None else None
    check(L.load().mllm_hip_softmax(vp(x), vp(y), C.c_int(rows), C.c_int(n), vp(v), _stream()), "softmax")
    return y


def argmax(x):
    x = _dev(x, torch.float32)
    out = torch.empty(1, dtype=torch.int32, device="cuda")
    check(L.load().mllm_hip_argmax(vp(x), C.c_int(x.numel()), vp(out), _stream()), "argmax")
    return int(out.item())


def index_put_rows(dst, value, idx):
    dst, value = _dev(dst, torch.float32).clone(), _dev(value, torch.float32)
    idx = _dev(np.asarray(idx, dtype=np.int32))
    check(L.load().mllm_hip_index_put_rows(vp(dst), vp(value), vp(idx), C.c_int(idx.numel()), C.c_int(dst.shape[-1]), _stream()), "index_put_rows")
    return dst


def rope_apply(x, S, H, D, sin_t, cos_t, out_f16=False):
    x, sin_t, cos_t = _dev(x, torch.float32), _dev(sin_t, torch.float32), _dev(cos_t, torch.float32)
    out = torch.empty((S, H * D), dtype=torch.float16 if out_f16 else torch.float32, device="cuda")
    check(L.load().mllm_hip_rope_apply(vp(x), i64(H * D), vp(sin_t), vp(cos_t), C.c_int(sin_t.shape[-1]), vp(out), C.c_int(F16 if out_f16 else F32),
                                       i64(H * D), C.c_int(S), C.c_int(H), C.c_int(D), _stream()), "rope_apply")
    return out


def qkv_rope_append(qkv, S, Hq, Hkv, D, sin_t, cos_t, vt_ld):
    """mllm_hip_qkv_rope_append on rows [S][(Hq + 2 Hkv) D]: returns (q rotated fp32 [S][Hq D], k fp16 [S][Hkv D], v fp16 transposed [Hkv D][vt_ld])."""
    qkv, sin_t, cos_t = _dev(qkv, torch.float32).clone(), _dev(sin_t, torch.float32), _dev(cos_t, torch.float32)
    W = (Hq + 2 * Hkv) * D
    k = torch.zeros((S, Hkv * D), dtype=torch.float16, device="cuda")
    vt = torch.zeros((Hkv * D, vt_ld), dtype=torch.float16, device="cuda")
    check(L.load().mllm_hip_qkv_rope_append(vp(qkv), i64(W), vp(sin_t), vp(cos_t), C.c_int(sin_t.shape[-1]), vp(k), i64(Hkv * D), vp(vt), i64(vt_ld), C.c_int(S), C.c_int(Hq),
                                            C.c_int(Hkv), C.c_int(D), _stream()), "qkv_rope_append")
    return qkv[:, :Hq * D].contiguous(), k, vt


def flash_attention2(q, k, v, Sq, Sk, Hq, Hkv, D, causal, sk_dev=None):
    """Tensor::flash_attention2_forward: q fp32 [Sq][Hq*D]; k, v fp16 or fp32 [Sk][Hkv*D]."""
    q = _dev(q, torch.float32)
    k, v = _dev(k), _dev(v)
    kv_dt = F16 if k.dtype == torch.float16 else F32
    o = torch.empty((Sq, Hq * D), dtype=torch.float32, device="cuda")
    wsb = L.load().mllm_hip_fa2_workspace_bytes(C.c_int(Sq), C.c_int(Hq), C.c_int(D), C.c_int(Sk))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
    check(L.load().mllm_hip_fa2(vp(q), i64(Hq * D), vp(k), i64(Hkv * D), vp(v), i64(Hkv * D), C.c_int(kv_dt), vp(o), i64(Hq * D), C.c_int(Sq), C.c_int(Sk),
                                C.c_int(Hq), C.c_int(Hkv), C.c_int(D), C.c_int(int(causal)), vp(sk_dev), vp(ws), _stream()), "fa2")
    return o


def flash_attention2_batch(q, k, v, Sq, Sk, Hq, Hkv, D, causal):
    """nb independent attentions in one launch (mllm_hip_fa2_batch): q fp32 [nb][Sq][Hq*D]; k, v fp16 or fp32 [nb][Sk][Hkv*D]."""
    q = _dev(q, torch.float32)
    k, v = _dev(k), _dev(v)
    nb = q.shape[0]
    kv_dt = F16 if k.dtype == torch.float16 else F32
    o = torch.empty((nb, Sq, Hq * D), dtype=torch.float32, device="cuda")
    check(L.load().mllm_hip_fa2_batch(vp(q), i64(Hq * D), vp(k), i64(Hkv * D), vp(v), i64(Hkv * D), C.c_int(kv_dt), vp(o), i64(Hq * D), C.c_int(Sq), C.c_int(Sk),
                                      C.c_int(Hq), C.c_int(Hkv), C.c_int(D), C.c_int(int(causal)), C.c_int(nb), i64(Sq * Hq * D), i64(Sk * Hkv * D), i64(Sk * Hkv * D),
                                      i64(Sq * Hq * D), _stream()), "fa2_batch")
    return o


def linear_q4k_packed_producers(Wq, x, N, mode="quant", w=None, b=None, eps=1e-6):
    """Prefill path of the resident engine: the producer (quantiser / RMSNorm / LayerNorm) writes the packed activation operand, the
    GEMM consumes it.  Returns y = Linear(producer(x))."""
    x = _dev(x, torch.float32)
    M, K = x.shape
    Wd = _dev(np.asarray(Wq).view(np.uint8))
    lib_ = L.load()
    lib_.mllm_hip_q4k_prepack_bytes.restype = C.c_size_t
    lib_.mllm_hip_q4k_wpack_bytes.restype = C.c_size_t
    Kw = K // 2 if mode == "silu_mul" else K
    wp = torch.empty(lib_.mllm_hip_q4k_wpack_bytes(C.c_int(N), C.c_int(Kw)), dtype=torch.uint8, device="cuda")
    xp = torch.empty(lib_.mllm_hip_q4k_prepack_bytes(C.c_int(M), C.c_int(Kw)), dtype=torch.uint8, device="cuda")
    check(lib_.mllm_hip_q4k_prepack(vp(Wd), C.c_int(N), C.c_int(Kw), vp(wp), _stream()), "q4k_prepack")
    if mode == "quant":
        check(lib_.mllm_hip_quantize_q8k_packed(vp(x), vp(xp), C.c_int(M), C.c_int(K), _stream()), "quantize_q8k_packed")
    elif mode == "rms":
        check(lib_.mllm_hip_rmsnorm_packed(vp(x), vp(_dev(w, torch.float32)), None, vp(xp), C.c_int(M), C.c_int(K), C.c_float(eps), C.c_int(0), _stream()), "rmsnorm_packed")
    elif mode in ("gelu", "quickgelu"):
        lut = _act_luts()[0 if mode == "gelu" else 1]
        check(lib_.mllm_hip_quantize_q8k_packed_act(vp(x), vp(lut), vp(xp), C.c_int(M), C.c_int(K), _stream()), "quantize_q8k_packed_act")
    elif mode == "silu_mul":     # x is the fused [M][2 K] gate|up buffer
        K = K // 2
        xp = torch.empty(lib_.mllm_hip_q4k_prepack_bytes(C.c_int(M), C.c_int(K)), dtype=torch.uint8, device="cuda")
        check(lib_.mllm_hip_quantize_q8k_packed_silu_mul(vp(x), vp(xp), C.c_int(M), C.c_int(K), _stream()), "quantize_q8k_packed_silu_mul")
    else:
        wb = _dev(b, torch.float32) if b is not None else None
        check(lib_.mllm_hip_layernorm_packed(vp(x), vp(_dev(w, torch.float32)), vp(wb), None, vp(xp), C.c_int(M), C.c_int(K), C.c_float(eps), _stream()), "layernorm_packed")
    y = torch.empty((M, N), dtype=torch.float32, device="cuda")
    check(lib_.mllm_hip_linear_q4kp_packed(vp(wp), None, vp(xp), vp(y), C.c_int(F32), i64(N), None, C.c_int(M), C.c_int(N), C.c_int(K), _stream()), "linear_q4kp_packed")
    return y


def flash_attention2_vt(q, k16, v_f32, Sq, Sk, Hq, Hkv, D, causal):
    """Same attention on the resident engine's KV layout: K fp16 rows, V stored transposed (mllm_hip_store_f16_t) with padded rows."""
    q, k16, v_f32 = _dev(q, torch.float32), _dev(k16, torch.float16), _dev(v_f32, torch.float32)
    ld = ((Sk + 63) // 64) * 64 + 128
    vt = torch.zeros((Hkv * D, ld), dtype=torch.float16, device="cuda")
    check(L.load().mllm_hip_store_f16_t(vp(v_f32), i64(Hkv * D), vp(vt), i64(ld), C.c_int(Sk), C.c_int(Hkv * D), _stream()), "store_f16_t")
    o = torch.empty((Sq, Hq * D), dtype=torch.float32, device="cuda")
    check(L.load().mllm_hip_fa2_vt(vp(q), i64(Hq * D), vp(k16), i64(Hkv * D), vp(vt), i64(ld), vp(o), i64(Hq * D), C.c_int(Sq), C.c_int(Sk), C.c_int(Hq),
                                   C.c_int(Hkv), C.c_int(D), C.c_int(int(causal)), _stream()), "fa2_vt")
    return o


def patch_gemm(patches, W, bias=None):
    patches, W = _dev(patches, torch.float32), _dev(W, torch.float32)
    N, KK = patches.shape
    OC = W.numel() // KK
    b = _dev(bias, torch.float32) if bias is not None else None
    out = torch.empty((N, OC), dtype=torch.float32, device="cuda")
    check(L.load().mllm_hip_patch_gemm_f32(vp(patches), vp(W), vp(b), vp(out), C.c_int(N), C.c_int(KK), C.c_int(OC), _stream()), "patch_gemm")
    return out


def conv2d_patch(img_hcw, H, Cc, Wd, Wt, OC, p, bias=None):
    img = _dev(img_hcw, torch.float32)
    KK = p * Cc * p
    patches = torch.empty(((H // p) * (Wd // p), KK), dtype=torch.float32, device="cuda")
    check(L.load().mllm_hip_im2patch_hcw(vp(img), vp(patches), C.c_int(H), C.c_int(Cc), C.c_int(Wd), C.c_int(p), _stream()), "im2patch")
    out = patch_gemm(patches, Wt, bias)  # [oh*ow][OC]
    return out.reshape(H // p, Wd // p, OC).permute(0, 2, 1).contiguous()


def topk(logits, k):
    """Top-k candidate set of _LlmTextGenerateTopkSamplingMethod (mllm/Generate.cpp:56-67) on device: (values desc, indices)."""
    x = _dev(logits, torch.float32).reshape(-1)
    val = torch.empty(k, dtype=torch.float32, device="cuda")
    idx = torch.empty(k, dtype=torch.int32, device="cuda")
    check(L.load().mllm_hip_topk(vp(x), C.c_int(x.numel()), C.c_int(k), vp(val), vp(idx), _stream()), "topk")
    return val.cpu().numpy(), idx.cpu().numpy()


def topk_probs(top_val, temperature):
    """Temperature softmax + renormalisation over the k candidates (Generate.cpp:69-87), host arithmetic with the host's libm."""
    v = np.ascontiguousarray(top_val, dtype=np.float32)
    p = np.empty_like(v)
    check(L.load().mllm_hip_topk_probs_host(vp(v), C.c_int(v.size), C.c_float(temperature), vp(p)), "topk_probs_host")
    return p


# ---- the extra ops of the other model families (csrc/kernels_n4.hip); index tensors are fp32 like the reference's ------------------------------
def sliding_window_mask(x, H, keys, window):
    x = _dev(x, torch.float32)
    S = x.numel() // (H * keys)
    y = torch.empty_like(x)
    check(L.load().mllm_hip_sliding_window_mask(vp(x), vp(y), C.c_int(S), C.c_int(H), C.c_int(keys), C.c_int(window), _stream()), "sliding_window_mask")
    return y


def topk_rows(x, k):
    x = _dev(x, torch.float32)
    rows, n = x.shape
    v = torch.empty((rows, k), dtype=torch.float32, device="cuda")
    i = torch.empty((rows, k), dtype=torch.float32, device="cuda")
    check(L.load().mllm_hip_topk_rows(vp(x), i64(n), vp(v), vp(i), C.c_int(rows), C.c_int(n), C.c_int(k), _stream()), "topk_rows")
    return v, i


def bincount(ids, nbins):
    ids = _dev(ids, torch.float32)
    out = torch.empty(nbins, dtype=torch.float32, device="cuda")
    check(L.load().mllm_hip_bincount(vp(ids), C.c_int(ids.numel()), vp(out), C.c_int(nbins), _stream()), "bincount")
    return out


def gather_rows(src, idx):
    src, idx = _dev(src, torch.float32), _dev(idx, torch.float32)
    D = src.shape[-1]
    out = torch.zeros((idx.numel(), D), dtype=torch.float32, device="cuda")      # rows whose index is out of range stay zero
    check(L.load().mllm_hip_gather_rows(vp(src), i64(D), C.c_int(src.reshape(-1, D).shape[0]), vp(idx), vp(out), i64(D), C.c_int(idx.numel()), C.c_int(D), C.c_int(0), _stream()), "gather_rows")
    return out


def fuyu_gather(word, patches, idx):
    word, patches, idx = _dev(word, torch.float32).clone(), _dev(patches, torch.float32), _dev(idx, torch.float32)
    D = word.shape[-1]
    check(L.load().mllm_hip_gather_rows(vp(patches), i64(D), C.c_int(patches.reshape(-1, D).shape[0]), vp(idx), vp(word), i64(D), C.c_int(idx.numel()), C.c_int(D), C.c_int(1), _stream()), "fuyu_gather")
    return word


def scatter_add_rows(dst, src, idx):
    dst, src, idx = _dev(dst, torch.float32).clone(), _dev(src, torch.float32), _dev(idx, torch.float32)
    D = dst.shape[-1]
    check(L.load().mllm_hip_scatter_add_rows(vp(dst), i64(D), C.c_int(dst.reshape(-1, D).shape[0]), vp(src), i64(D), vp(idx), C.c_int(idx.numel()), C.c_int(D), _stream()), "scatter_add_rows")
    return dst


def moe_block(x, router_raw, w1_raw, w3_raw, w2_raw, inter, per_tok):
    """MiniCPMMoE::Forward (models/minicpm_moe/modeling_minicpm_moe.hpp:52-105) on token rows x `[S][hidden]`: router / expert weights as the raw Q4_K bytes of the
    .mllm (one array per expert)."""
    x = _dev(x, torch.float32)
    S, H = x.shape
    E = len(w1_raw)
    router = _dev(np.asarray(router_raw).view(np.uint8))
    keep = [[_dev(np.asarray(w).view(np.uint8)) for w in ws] for ws in (w1_raw, w3_raw, w2_raw)]
    arrs = [(C.c_void_p * E)(*[t.data_ptr() for t in ws]) for ws in keep]
    out = torch.empty((S, H), dtype=torch.float32, device="cuda")
    check(L.load().mllm_hip_moe_block(vp(x), vp(out), C.c_int(S), C.c_int(H), C.c_int(inter), C.c_int(E), C.c_int(per_tok), vp(router), arrs[0], arrs[1], arrs[2],
                                      _stream()), "moe_block")
    return out


class _RowSeg(C.Structure):
    _fields_ = [("W", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p), ("post_add", C.c_void_p), ("post_out", C.c_void_p), ("N", C.c_int), ("wg0_", C.c_int)]


class _RowFused(C.Structure):
    _fields_ = [("xa", C.c_void_p), ("xb", C.c_void_p), ("sum_out", C.c_void_p), ("norm_w", C.c_void_p), ("norm_out", C.c_void_p), ("eps", C.c_float), ("K", C.c_int),
                ("nseg", C.c_int), ("mode", C.c_int), ("rpw_", C.c_int), ("seg", _RowSeg * 3), ("silu_out", C.c_void_p), ("mul_out", C.c_void_p)]


def row_fused(xa, segs, xb=None, norm_w=None, eps=1e-6, gateup=False):
    """mllm_hip_row_fused_launch (include/mllm_hip.h): one activation row `xa [K]` (+ `xb`) -> [RMSNorm] -> up to three Q4_K Linears `segs = [(W_raw, N, bias, post_add), ...]` with the
    F_TTADD behind each (mode 0) or the SiLU / F_TTMUL of a gate / up pair (mode 1).  Returns a dict of every Op output the launch wrote."""
    xa = _dev(xa, torch.float32).reshape(-1)
    K = xa.numel()
    a = _RowFused()
    keep = [xa]

    def dv(t):
        t = _dev(t, torch.float32).reshape(-1)
        keep.append(t)
        return t

    out = {}
    a.xa = xa.data_ptr()
    if xb is not None:
        a.xb = dv(xb).data_ptr()
        out["sum"] = torch.full((K,), float("nan"), dtype=torch.float32, device="cuda")
        a.sum_out = out["sum"].data_ptr()
    if norm_w is not None:
        a.norm_w = dv(norm_w).data_ptr()
        out["norm"] = torch.full((K,), float("nan"), dtype=torch.float32, device="cuda")
        a.norm_out = out["norm"].data_ptr()
    a.eps, a.K, a.nseg, a.mode = eps, K, len(segs), 1 if gateup else 0
    out["y"], out["post"] = [], []
    for i, (W, N, bias, post_add) in enumerate(segs):
        Wd = _dev(W, torch.uint8)
        keep.append(Wd)
        y = torch.full((N,), float("nan"), dtype=torch.float32, device="cuda")
        out["y"].append(y)
        a.seg[i].W, a.seg[i].y, a.seg[i].N = Wd.data_ptr(), y.data_ptr(), N
        if bias is not None:
            a.seg[i].bias = dv(bias).data_ptr()
        if post_add is not None:
            a.seg[i].post_add = dv(post_add).data_ptr()
            p = torch.full((N,), float("nan"), dtype=torch.float32, device="cuda")
            out["post"].append(p)
            a.seg[i].post_out = p.data_ptr()
        else:
            out["post"].append(None)
    if gateup:
        out["silu"] = torch.full((segs[0][1],), float("nan"), dtype=torch.float32, device="cuda")
        out["mul"] = torch.full((segs[0][1],), float("nan"), dtype=torch.float32, device="cuda")
        a.silu_out, a.mul_out = out["silu"].data_ptr(), out["mul"].data_ptr()
    check(L.load().mllm_hip_row_fused_launch(C.byref(a), _stream()), "row_fused")
    torch.cuda.synchronize()
    return out


def rope2_store2(q, k, v, S, Hq, Hkv, D, sin_q, cos_q, sin_k, cos_k):
    """mllm_hip_rope2_store2: returns (q_out fp32, k_out fp32, k16, v16)."""
    q, k, v = _dev(q, torch.float32), _dev(k, torch.float32), _dev(v, torch.float32)
    sq, cq, sk, ck = (_dev(t, torch.float32) for t in (sin_q, cos_q, sin_k, cos_k))
    qo, ko = torch.empty_like(q), torch.empty_like(k)
    k16 = torch.empty(k.shape, dtype=torch.float16, device="cuda")
    v16 = torch.empty(v.shape, dtype=torch.float16, device="cuda")
    check(L.load().mllm_hip_rope2_store2(vp(q), vp(sq), vp(cq), C.c_int(sq.shape[-1]), vp(qo), C.c_int(Hq), vp(k), vp(sk), vp(ck), C.c_int(sk.shape[-1]), vp(ko), vp(k16), vp(v),
                                         vp(v16), C.c_int(Hkv), C.c_int(S), C.c_int(D), _stream()), "rope2_store2")
    return qo, ko, k16, v16


def fa2_decode_step(q_raw, k_raw, v_raw, kslab, vslab, T, Hq, Hkv, D, sin_q, cos_q, sin_k, cos_k):
    """mllm_hip_fa2_decode_step: slabs fp16 `[>= T + 1][Hkv * D]` (modified in place: row T).  Returns (q_out, k_out, O)."""
    q, k, v = _dev(q_raw, torch.float32), _dev(k_raw, torch.float32), _dev(v_raw, torch.float32)
    sq, cq, sk, ck = (_dev(t, torch.float32) for t in (sin_q, cos_q, sin_k, cos_k))
    qo, ko = torch.empty_like(q), torch.empty_like(k)
    o = torch.empty((Hq * D,), dtype=torch.float32, device="cuda")
    check(L.load().mllm_hip_fa2_decode_step(vp(q), vp(sq), vp(cq), vp(qo), vp(k), vp(sk), vp(ck), vp(ko), vp(v), vp(kslab), vp(vslab), C.c_int(T), vp(o), C.c_int(Hq), C.c_int(Hkv),
                                            C.c_int(D), _stream()), "fa2_decode_step")
    return qo, ko, o
