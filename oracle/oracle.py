"""ctypes face of oracle/restate.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing under
mllm_amd/ does.  See the header of restate.c for what it restates and how it is pinned.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "liboracle.so")
SRC = os.path.join(HERE, "restate.c")


def build(force: bool = False) -> str:
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        subprocess.check_call(["gcc", "-O2", "-mavx2", "-mf16c", "-mfma", "-ffp-contract=off", "-fopenmp", "-shared",
                               "-fPIC", SRC, "-o", SO, "-lm"])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_vec_dot_q4_K_q8_K.restype = C.c_float
        _lib.orc_vec_dot_q4_0_q8_0.restype = C.c_float
        _lib.orc_vec_dot_f32.restype = C.c_float
        _lib.orc_f16_to_f32.restype = C.c_float
        _lib.orc_v_expf.restype = C.c_float
        _lib.orc_v_expf.argtypes = [C.c_float]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


F32, Q4_0, Q4_K = 0, 2, 12


def quantize_q8_K(x: np.ndarray) -> np.ndarray:
    x = _f32(x)
    k = x.shape[-1]
    rows = x.reshape(-1, k)
    out = np.zeros((rows.shape[0], k // 256 * 292), dtype=np.uint8)
    for i in range(rows.shape[0]):
        lib().orc_quantize_row_q8_K(_p(rows[i]), _p(out[i]), C.c_int(k))
    return out


def quantize_q8_0(x: np.ndarray) -> np.ndarray:
    x = _f32(x)
    k = x.shape[-1]
    rows = x.reshape(-1, k)
    out = np.zeros((rows.shape[0], k // 32 * 34), dtype=np.uint8)
    for i in range(rows.shape[0]):
        lib().orc_quantize_row_q8_0(_p(rows[i]), _p(out[i]), C.c_int(k))
    return out


def dequantize(raw: np.ndarray, dtype: int, n: int) -> np.ndarray:
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    out = np.empty(n, dtype=np.float32)
    if dtype == Q4_0:
        lib().orc_dequantize_row_q4_0(_p(raw), _p(out), C.c_int(n))
    elif dtype == Q4_K:
        lib().orc_dequantize_row_q4_K(_p(raw), _p(out), C.c_int(n))
    else:
        out[:] = raw.view(np.float32)[:n]
    return out


def linear(x, W_raw, wdtype, N, bias=None, out_f16=False):
    x = _f32(x)
    K = x.shape[-1]
    M = x.size // K
    W_raw = np.ascontiguousarray(W_raw)
    b = _f32(bias) if bias is not None else None
    y = np.empty((M, N), dtype=np.uint16 if out_f16 else np.float32)
    lib().orc_linear(_p(x), C.c_int(M), C.c_int(K), _p(W_raw), C.c_int(wdtype), C.c_int(N), _p(b), _p(y), C.c_int(int(out_f16)))
    return y


def embedding(ids, W_raw, wdtype, hidden):
    ids = _f32(ids).ravel()
    out = np.empty((ids.size, hidden), dtype=np.float32)
    lib().orc_embedding(_p(ids), C.c_int(ids.size), _p(np.ascontiguousarray(W_raw)), C.c_int(wdtype), C.c_int(hidden), _p(out))
    return out


def rmsnorm(x, w, eps, add_unit_offset=False):
    x = _f32(x)
    dim = x.shape[-1]
    y = np.empty_like(x)
    lib().orc_rmsnorm(_p(x), _p(_f32(w)), _p(y), C.c_int(x.size // dim), C.c_int(dim), C.c_float(eps), C.c_int(int(add_unit_offset)))
    return y


def layernorm(x, w, b, eps):
    x = _f32(x)
    dim = x.shape[-1]
    y = np.empty_like(x)
    lib().orc_layernorm(_p(x), _p(_f32(w)), _p(_f32(b)) if b is not None else None, _p(y), C.c_int(x.size // dim), C.c_int(dim), C.c_float(eps))
    return y


def _unary(name, x):
    x = _f32(x)
    y = np.empty_like(x)
    getattr(lib(), name)(_p(x), _p(y), C.c_int(x.size))
    return y


def silu(x):
    return _unary("orc_silu", x)


def gelu(x):
    return _unary("orc_gelu", x)


def quickgelu(x):
    return _unary("orc_quickgelu", x)


def gelu_tables():
    g = np.empty(65536, dtype=np.uint16)
    q = np.empty(65536, dtype=np.uint16)
    lib().orc_gelu_tables(_p(g), _p(q))
    return g, q


def softmax(x, valid=None):
    x = _f32(x)
    n = x.shape[-1]
    rows = x.reshape(-1, n)
    y = np.empty_like(rows)
    for i in range(rows.shape[0]):
        v = n if valid is None else int(valid[i])
        lib().orc_softmax_row(_p(rows[i]), _p(y[i]), C.c_int(n), C.c_int(v))
    return y.reshape(x.shape)


def rope_table_hf(base, dim, n_pos):
    s = np.empty((n_pos, dim), dtype=np.float32)
    c = np.empty((n_pos, dim), dtype=np.float32)
    lib().orc_rope_table_hf(C.c_float(base), C.c_int(dim), C.c_int(n_pos), _p(s), _p(c))
    return s, c


def rope_table_hf_llama3(base, dim, n_pos, factor, low_freq_factor, high_freq_factor, original_max_pos):
    s = np.empty((n_pos, dim), dtype=np.float32)
    c = np.empty((n_pos, dim), dtype=np.float32)
    lib().orc_rope_table_hf_llama3(C.c_float(base), C.c_int(dim), C.c_int(n_pos), C.c_float(factor), C.c_float(low_freq_factor), C.c_float(high_freq_factor),
                                   C.c_float(original_max_pos), _p(s), _p(c))
    return s, c


def rope_table_ntk(theta, dim, n_pos, original_max_pos, long_factor, short_factor):
    """CPUNTKRoPE.cpp:27-80; tables [n_pos][dim]."""
    s = np.empty((n_pos, dim), dtype=np.float32)
    c = np.empty((n_pos, dim), dtype=np.float32)
    lf, sf = _f32(long_factor), _f32(short_factor)
    lib().orc_rope_table_ntk(C.c_float(theta), C.c_int(dim), C.c_int(n_pos), C.c_int(original_max_pos), _p(lf), _p(sf), _p(s), _p(c))
    return s, c


# ---- the small ops of the other model families, restated in numpy (each pinned by tests/golden/n4_ops.npz) -------------------------------
def sliding_window_mask(x, H, keys, window):
    """CPUSlidingWindowMask.cpp:30-58 on scores [S][H][keys]: key d of row s is kept iff s - (window - 1) <= d <= s + (keys - S); S == 1 passes through."""
    x = _f32(x).reshape(-1, H, keys)
    S = x.shape[0]
    if S == 1:
        return x.reshape(S, H * keys).copy()
    old = keys - S
    s = np.arange(S)[:, None, None]
    d = np.arange(keys)[None, None, :]
    out = np.where((d > s + old) | (d < s - (window - 1)), np.float32(np.finfo(np.float32).min), x)
    return out.reshape(S, H * keys).astype(np.float32)


def topk_rows(x, k):
    """CPUTopkFunc.hpp:48-70: a min-heap of (value, index) pairs keeps the k largest pairs; output descending by (value, index) -- among equal values the LARGER index first.
    Returns (values, indices as float)."""
    x = _f32(x)
    R, n = x.shape
    v = np.empty((R, k), dtype=np.float32)
    ix = np.empty((R, k), dtype=np.float32)
    for r in range(R):
        order = sorted(range(n), key=lambda j: (x[r, j], j), reverse=True)[:k]
        v[r] = x[r, order]
        ix[r] = order
    return v, ix


def bincount(ids):
    """CPUBinCountFunc.hpp:20-35: counts of the non-negative integer parts, length max + 1."""
    ii = _f32(ids).astype(np.int64)
    return np.bincount(ii[ii >= 0], minlength=int(ii.max()) + 1 if ii.size else 0).astype(np.float32)


def scatter_add_rows(dst, src, idx):
    """CPUScatterAddFunc.hpp:38-52: dst[idx[r]] += src[r], r ascending (a repeated destination row accumulates in that order)."""
    out = _f32(dst).copy()
    src = _f32(src)
    for r, i in enumerate(_f32(idx).astype(np.int64)):
        out[i] = out[i] + src[r]
    return out


def gather_rows(src, idx):
    """Tensor::clip(index, SEQUENCE) (CPUClipFunc.hpp:309-323): out[r] = src[idx[r]]."""
    return _f32(src)[_f32(idx).astype(np.int64)].copy()


def fuyu_gather(word, patches, idx):
    """CPUFuyuGatherEmbdFunc.hpp:45-62: word[s] = patches[idx[s]] where idx[s] >= 0."""
    out = _f32(word).copy()
    patches = _f32(patches)
    for s, i in enumerate(_f32(idx).astype(np.int64)):
        if i >= 0:
            out[s] = patches[i]
    return out


def mrope_table(base, dim, pos, section=(16, 24, 24)):
    pos = _f32(pos)  # [3, S]
    S = pos.shape[1]
    sec = np.asarray(section, dtype=np.int32)
    s = np.zeros((S, dim // 2), dtype=np.float32)
    c = np.zeros((S, dim // 2), dtype=np.float32)
    lib().orc_mrope_table(C.c_float(base), C.c_int(dim), _p(pos), C.c_int(S), _p(sec), C.c_int(len(sec)), _p(s), _p(c))
    return s, c


def vision_rope_angles(t, h, w, merge, rot_dim):
    a = np.empty((t * h * w, rot_dim), dtype=np.float32)
    lib().orc_vision_rope_angles(C.c_int(t), C.c_int(h), C.c_int(w), C.c_int(merge), C.c_int(rot_dim), _p(a))
    return a


def rope_apply(x, S, H, D, sin_t, cos_t, out_f16=False):
    x = _f32(x)
    sin_t, cos_t = _f32(sin_t), _f32(cos_t)
    out = np.empty(x.shape, dtype=np.uint16 if out_f16 else np.float32)
    lib().orc_rope_apply(_p(x), C.c_int(S), C.c_int(H), C.c_int(D), _p(sin_t), _p(cos_t), C.c_int(sin_t.shape[-1]), _p(out), C.c_int(int(out_f16)))
    return out


def vision_rope_apply(x, S, H, D, angle):
    x = _f32(x)
    out = np.empty_like(x)
    lib().orc_vision_rope_apply(_p(x), C.c_int(S), C.c_int(H), C.c_int(D), _p(_f32(angle)), _p(out))
    return out


def attention(q, k, v, Sq, Sk, Hq, Hkv, D, causal):
    q = _f32(q)
    kv_f16 = k.dtype == np.uint16
    k = np.ascontiguousarray(k)
    v = np.ascontiguousarray(v)
    o = np.empty((Sq, Hq * D), dtype=np.float32)
    lib().orc_attention(_p(q), _p(k), _p(v), C.c_int(int(kv_f16)), _p(o), C.c_int(Sq), C.c_int(Sk), C.c_int(Hq), C.c_int(Hkv), C.c_int(D), C.c_int(int(causal)))
    return o


def patch_gemm(patches, W, bias=None):
    patches, W = _f32(patches), _f32(W)
    N, KK = patches.shape
    OC = W.size // KK
    out = np.empty((N, OC), dtype=np.float32)
    lib().orc_patch_gemm(_p(patches), C.c_int(N), C.c_int(KK), _p(W), C.c_int(OC), _p(_f32(bias)) if bias is not None else None, _p(out))
    return out


def conv2d_patch(img, H, Cc, Wd, Wt, OC, p, bias=None):
    img, Wt = _f32(img), _f32(Wt)
    out = np.empty((H // p, OC, Wd // p), dtype=np.float32)
    lib().orc_conv2d_patch(_p(img), C.c_int(H), C.c_int(Cc), C.c_int(Wd), _p(Wt), C.c_int(OC), C.c_int(p), _p(_f32(bias)) if bias is not None else None, _p(out))
    return out


def f16_to_f32(a):
    return np.asarray(a, dtype=np.uint16).view(np.float16).astype(np.float32)


def f32_to_f16(a):
    return _f32(a).astype(np.float16).view(np.uint16)


def topk_sampling_probs(logits, k, temperature):
    """N2 restatement (pure Python/numpy) of _LlmTextGenerateTopkSamplingMethod::generate, mllm/Generate.cpp:56-87; pinned by tests/golden/sampling.npz
    (the reference's own candidates and pre-draw probabilities, oracle/ref_drivers/ref_sampling.cpp).  Returns (indices, values, probabilities)."""
    import math
    x = np.asarray(logits, dtype=np.float32).ravel()
    order = np.argsort(-x.astype(np.float64), kind="stable")[:k]          # partial_sort by descending logit; equal logits by index
    top = x[order]
    max_logit = float(top[int(np.argmax(top))])                            # double max_logit = top[argmax(top)]
    soft = np.empty(k, dtype=np.float32)
    sum_exp = 0.0
    for i in range(k):
        soft[i] = np.float32(math.exp((float(top[i]) - max_logit) / float(np.float32(temperature))))
        sum_exp += float(soft[i])
    for i in range(k):
        soft[i] = np.float32(float(soft[i]) / sum_exp)
    fsum = np.float32(sum(float(v) for v in soft))                         # float _sum = std::accumulate(..., 0.0)
    for i in range(k):
        soft[i] = np.float32(soft[i] / fsum)
    return order.astype(np.int32), top, soft


def topp_sampling_probs(probs, p, temperature):
    """N2 restatement of _LlmTextGenerateToppSamplingMethod::generate (mllm/Generate.cpp:93-142): sort all (score, index) descending, keep the prefix whose
    running FLOAT sum reaches p (`while (p < m_p)`), then the same temperature softmax over the kept scores; pinned by tests/golden/sampling.npz.
    Returns (indices, values, probabilities); a one-element nucleus returns probability 1."""
    x = np.asarray(probs, dtype=np.float32).ravel()
    order = np.argsort(-x.astype(np.float64), kind="stable")
    acc = np.float32(0.0)
    n = 0
    while acc < np.float32(p):
        acc = np.float32(acc + x[order[n]])
        n += 1
    idx = order[:n].astype(np.int32)
    if n == 1:
        return idx, x[idx], np.ones(1, dtype=np.float32)
    _, _, soft = topk_sampling_probs(x[idx], n, temperature)      # the kept scores are already in descending order
    return idx, x[idx], soft


# ---- Qwen2-VL image preprocessing (mllm/models/qwen2_vl/processing_qwen2_vl.hpp:84-235, mllm/processor/PreProcess.cpp:37-43,84-154,233-262) -------------
# TEST INFRASTRUCTURE.  The resize is third-party code the reference vendors: stb_image_resize2.h (third_party/stb, v2.x) with STBIR_FILTER_CUBICBSPLINE, STBIR_EDGE_CLAMP,
# float RGB.  Its published algorithm is restated here: per axis a gather with coefficients B(distance) (cubic B-spline, support 2, stretched by 1/scale when shrinking),
# normalised to sum 1, out-of-range taps folded into the edge pixel (stb_image_resize2.h:2821-2832 kernel, :3176-3283 upsample taps, :3287-3380 downsample taps, :3382-3500
# normalisation + clamp).  The library's SIMD gather loops and its horizontal/vertical ordering heuristic fix the ORDER of the float additions; that order is not restated, so
# this function is pinned to the reference's output (tests/golden/preprocess.npz) within a stated tolerance, not bit for bit.
QWEN2VL_MEAN = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)
QWEN2VL_STD = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)


def smart_resize(height, width, factor=28, min_pixels=4 * 28 * 28, max_pixels=16384 * 28 * 28):
    """processing_qwen2_vl.hpp:84-109 (float beta, integer rounding by factor)."""
    import math
    if max(height, width) / np.float32(min(height, width)) > 200:
        raise ValueError("absolute aspect ratio must be smaller than 200")
    rnd = lambda v: ((v + factor // 2) // factor) * factor
    h_bar, w_bar = max(factor, rnd(height)), max(factor, rnd(width))
    if h_bar * w_bar > max_pixels:
        beta = np.float32(math.sqrt(np.float32(height * width) / np.float32(max_pixels)))
        h_bar = int(math.floor(np.float32(height) / beta / factor)) * factor
        w_bar = int(math.floor(np.float32(width) / beta / factor)) * factor
    elif h_bar * w_bar < min_pixels:
        beta = np.float32(math.sqrt(np.float32(min_pixels) / np.float32(height * width)))
        h_bar = int(math.ceil(np.float32(height) * beta / factor)) * factor
        w_bar = int(math.ceil(np.float32(width) * beta / factor)) * factor
    return h_bar, w_bar


def _bspline(x):
    f = np.float32
    x = f(abs(x))
    if x < f(1.0):
        return f((f(4.0) + x * x * (f(3.0) * x - f(6.0))) / f(6.0))
    if x < f(2.0):
        return f((f(8.0) + x * (f(-12.0) + x * (f(6.0) - x))) / f(6.0))
    return f(0.0)


def stb_axis_taps(in_size, out_size):
    """Per output index: (first input pixel, float32 coefficients), normalised, clamped to [0, in_size)."""
    import math
    f = np.float32
    small = f(1.0 / (1 << 20))
    scale = f(out_size / in_size)
    inv_scale = f(1.0 / (out_size / in_size))
    taps = [dict() for _ in range(out_size)]
    if scale >= f(1.0) - small:
        radius = f(2.0) * scale
        for n in range(out_size):
            oc = f(n) + f(0.5)
            centre = f(oc * inv_scale)
            first = int(math.floor(f((oc - radius) * inv_scale) + f(0.5)))
            last = int(math.floor(f((oc + radius) * inv_scale) - f(0.5)))
            for p in range(first, last + 1):
                c = _bspline(centre - (f(p) + f(0.5)))
                if abs(c) >= small:
                    taps[n][p] = c
    else:
        radius = f(2.0) * inv_scale
        margin = int(math.ceil(2.0 * 2.0 / float(scale))) // 2 + 1
        for p in range(-margin, in_size + margin):
            ic = f(p) + f(0.5)
            out_centre = f(ic * scale)
            first = max(0, int(math.floor(f((ic - radius) * scale) + f(0.5))))
            last = min(out_size - 1, int(math.floor(f((ic + radius) * scale) - f(0.5))))
            for o in range(first, last + 1):
                c = f(_bspline((f(o) + f(0.5)) - out_centre) * scale)
                if abs(c) >= small:
                    taps[o][p] = c
    out = []
    for t in taps:
        ps = sorted(t)
        cs = np.array([t[p] for p in ps], dtype=np.float32)
        total = f(0.0)
        for c in cs:
            total = f(total + c)
        if total < f(1.0) - small or total > f(1.0) + small:
            cs = (cs * f(f(1.0) / total)).astype(np.float32)
        folded = {}
        for p, c in zip(ps, cs):                      # STBIR_EDGE_CLAMP: taps outside the image act on the edge pixel
            q = min(max(p, 0), in_size - 1)
            folded[q] = f(folded.get(q, f(0.0)) + c)
        qs = sorted(folded)
        out.append((qs[0], np.array([folded.get(q, f(0.0)) for q in range(qs[0], qs[-1] + 1)], dtype=np.float32)))
    return out


def stb_resize_bspline(img, out_h, out_w):
    """img float32 [H][W][C] -> [out_h][out_w][C]: horizontal gather then vertical gather, float32 sums in tap order."""
    H, W, C = img.shape
    tw, th = stb_axis_taps(W, out_w), stb_axis_taps(H, out_h)
    tmp = np.zeros((H, out_w, C), dtype=np.float32)
    for x, (p0, cs) in enumerate(tw):
        acc = np.zeros((H, C), dtype=np.float32)
        for k, c in enumerate(cs):
            acc = (acc + img[:, p0 + k, :] * c).astype(np.float32)
        tmp[:, x, :] = acc
    out = np.zeros((out_h, out_w, C), dtype=np.float32)
    for y, (p0, cs) in enumerate(th):
        acc = np.zeros((out_w, C), dtype=np.float32)
        for k, c in enumerate(cs):
            acc = (acc + tmp[p0 + k] * c).astype(np.float32)
        out[y] = acc
    return out


def qwen2vl_patchify(chw, patch=14, merge=2, tps=2):
    """convertPatches (processing_qwen2_vl.hpp:119-177) for one image whose single frame is doubled: chw float32 [3][H][W] -> [gh*gw][3*tps*patch*patch], grid."""
    C, H, W = chw.shape
    gh, gw = H // patch, W // patch
    x = chw.reshape(C, gh // merge, merge, patch, gw // merge, merge, patch)          # c, d1, d3, d7, d2, d4, d8
    x = x.transpose(1, 4, 2, 5, 0, 3, 6)                                               # d1, d2, d3, d4, c, d7, d8
    x = np.repeat(x[:, :, :, :, :, None], tps, axis=5)                                 # ..., c, t, d7, d8
    return np.ascontiguousarray(x.reshape(gh * gw, C * tps * patch * patch)), np.array([1, gh, gw], dtype=np.int32)


def qwen2vl_preprocess(rgb_u8):
    """preprocess_images for an already decoded RGB image uint8 [H][W][3] -> (patches float32 [N][1176], grid_thw)."""
    H, W, _ = rgb_u8.shape
    x = (rgb_u8.astype(np.float32) / np.float32(255.0)).astype(np.float32)             # RescaleImage
    nh, nw = smart_resize(H, W)
    x = stb_resize_bspline(x, nh, nw)                                                  # fetch_image: always resampled, even at equal size
    x = ((x - QWEN2VL_MEAN) / QWEN2VL_STD).astype(np.float32)                          # NormalizeImages(means, stds)
    return qwen2vl_patchify(np.ascontiguousarray(x.transpose(2, 0, 1)))


def gemm_fp32_bhsd(a, b):
    """F_MM on BHSD operands (CPUMatmulFunc.hpp:155-172 -> compute/GemmFp.hpp:104-150 / :233-283): a [heads, M, K] fp32, b [heads, K, N] fp32 or fp16 -> [heads, M, N]."""
    a = _f32(a)
    b16 = np.asarray(b).dtype == np.float16
    b = np.ascontiguousarray(b, dtype=np.float16 if b16 else np.float32)
    H, M, K = a.shape
    N = b.shape[2]
    assert b.shape == (H, K, N)
    c = np.empty((H, M, N), dtype=np.float32)
    lib().orc_gemm_fp32_bhsd(_p(a), b.ctypes.data_as(C.c_void_p), C.c_int(int(b16)), _p(c), C.c_int(H), C.c_int(M), C.c_int(N), C.c_int(K))
    return c
