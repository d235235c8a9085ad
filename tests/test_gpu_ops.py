"""GPU parity tests proper: every HIP launcher of the hot path (called through the C ABI) against the oracle on seeded inputs
and against the committed golden vectors of the reference.  Bar: bit-exact -- integer / byte / index results and fp32 results alike: the kernels keep the reference's operation order
(AVX2 lane chains of the dot products, FA2 tile recurrence with glibc expf, sequential LayerNorm sums).  The one stated
exception is RMSNorm's double-precision sum of squares (tree instead of sequential: the fp32 mean can move by one ulp with
probability ~1e-9 per row)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from mllm_amd import lib, mllmfile as mf, ops  # noqa: E402
from mllm_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    ops.require_gpu()


def rng(seed):
    return np.random.default_rng(seed)


def md(a, b):
    a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    return float(np.max(np.abs(a.astype(np.float64) - np.asarray(b, dtype=np.float64))))


def eq(a, b):
    a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    return np.array_equal(a, np.asarray(b))


# ---- A4 ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,K", [(1, 256), (3, 1536), (5, 8960), (282, 1536)])
def test_quantize_q8k_bit_exact(M, K):
    x = rng(M * 7 + K).standard_normal((M, K)).astype(np.float32) * 3
    x[0, :7] = [0.5, -0.5, 1.5, 2.5, -2.5, 1e-30, 0]        # ties for nearest_int
    if M > 1:
        x[1, :256] = 0                                        # all-zero block
        x[2, 5] = 7.25; x[2, 9] = -7.25                       # equal |x| with opposite sign: first one wins
        if K >= 512:                                          # ... also inside one lane's four values (both orders), at the block's last value, and a value that clamps at 127
            x[2, 256 + 40] = -30.0; x[2, 256 + 41] = 30.0
            x[0, 256 + 42] = 30.0; x[0, 256 + 43] = -30.0; x[0, 256 + 3] = -30.0
        if K >= 768:
            x[2, 767] = -50.0; x[2, 600] = 49.9999
    q = ops.quantize_q8k(x)
    blocks = orc.quantize_q8_K(x).reshape(M, K // 256, 292)
    d = blocks[:, :, :4].copy().view(np.float32).reshape(M, K // 256)
    qs = blocks[:, :, 4:260].reshape(M, K).view(np.int8)
    bs = blocks[:, :, 260:].copy().view(np.int16).reshape(M, K // 16)
    assert eq(q.qs, qs) and eq(q.d, d) and eq(q.bsums, bs)


@pytest.mark.parametrize("M,K", [(1, 1536), (4, 512)])
def test_quantize_q80_bit_exact(M, K):
    x = rng(K).standard_normal((M, K)).astype(np.float32)
    x[0, :32] = 0
    qs, d = ops.quantize_q80(x)
    blocks = orc.quantize_q8_0(x).reshape(M, K // 32, 34)
    assert eq(qs, blocks[:, :, 2:].reshape(M, K).view(np.int8))
    assert eq(d.view(torch.int16), blocks[:, :, :2].copy().view(np.int16).reshape(M, K // 32))


# ---- A1/A2/A5: Q4_K Linear -----------------------------------------------------------------------------------------------
def _q4k_case(M, K, N, seed, bias=True):
    r = rng(seed)
    Wq = synth.quantized_blocks(lib.Q4_K, r, N * K, std=0.05, full_range=True)      # Q4_K rows drawn in the quantised domain, every field over its whole range (mllm_amd/synth.py)
    x = r.standard_normal((M, K)).astype(np.float32)
    b = (r.standard_normal(N) * 0.1).astype(np.float32) if bias else None
    return Wq, x, b


# M = 1 goes through the one-launch form (dec_linear_row_q4k: up to 32 rows per workgroup, the last workgroup re-doing rows of its neighbour); K beyond 10240 and M = 3 the two-launch one
@pytest.mark.parametrize("M,K,N", [(1, 1536, 2048), (1, 8960, 1536), (1, 256, 64), (1, 1280, 3840), (3, 512, 96), (1, 11008, 128), (1, 1536, 8960), (1, 1536, 8950), (1, 512, 1000),
                                   (1, 256, 8229), (1, 2560, 31), (1, 256, 1)])
def test_linear_q4k_gemv_vs_oracle(M, K, N):
    Wq, x, b = _q4k_case(M, K, N, K + N)
    y = ops.linear_q4k(Wq, x, N, bias=b)
    ref = orc.linear(x, Wq, orc.Q4_K, N, b)
    assert eq(y, ref), md(y, ref)


# the decode quantiser finds the SIGNED first maximum of a 256-block from the lanes' max / min (4 consecutive values per lane); ties of +amax and -amax decide the sign of
# iscale by index order: in different lanes, inside one lane in both orders, the maximum at the very end, an all-zero block, and a block that clamps at 127
@pytest.mark.parametrize("K,N", [(1536, 96), (8960, 64)])
def test_linear_q4k_gemv_signed_maximum_ties(K, N):
    Wq, x, _ = _q4k_case(1, K, N, K * 3 + N, bias=False)
    nb = K // 256
    x = np.clip(x, -2.0, 2.0)
    cases = [((5, 3.0), (77, -3.0)), ((5, -3.0), (77, 3.0)), ((40, 3.0), (41, -3.0)), ((40, -3.0), (41, 3.0)), ((42, -3.0), (43, 3.0), (3, 3.0)),
             ((255, -3.0),), ((252, 3.0), (253, 3.0), (254, -3.0), (255, -3.0)), ((0, -3.0), (1, 3.0), (2, -3.0), (3, 3.0))]
    for b in range(nb):
        for j, val in cases[b % len(cases)]:
            x[0, b * 256 + j] = val
    if nb > 8:
        x[0, 8 * 256:9 * 256] = 0.0
        x[0, 9 * 256:10 * 256] = -0.0
    y = ops.linear_q4k(Wq, x, N)
    ref = orc.linear(x, Wq, orc.Q4_K, N)
    assert eq(y, ref), md(y, ref)


def test_linear_q4k_gemm_signed_maximum_ties():      # the same ties through the prefill quantiser that writes the GEMM's packed operand
    M, K, N = 40, 768, 96
    Wq, x, _ = _q4k_case(M, K, N, 99, bias=False)
    x = np.clip(x, -2.0, 2.0)
    for m in range(M):
        for b in range(K // 256):
            j = (m * 7 + b * 13) % 252
            x[m, b * 256 + j] = 3.0 if (m + b) % 2 else -3.0
            x[m, b * 256 + j + (1 if m % 3 else 9)] = -3.0 if (m + b) % 2 else 3.0
    x[5, 256:512] = 0.0
    y = ops.linear_q4k(Wq, x, N)
    ref = orc.linear(x, Wq, orc.Q4_K, N)
    assert eq(y, ref), md(y, ref)


# the GEMM's K loop is peeled (first blocks / block nb-2 / last block): nb = 1, 2, 3, 4 and more; ragged M and N edges, an odd number of 32-column tiles (the second
# tile of a workgroup's 32 x 64 is then idle), weights expanded from nibbles in registers
@pytest.mark.parametrize("M,K,N", [(16, 256, 128), (64, 1536, 256), (282, 1536, 2048), (100, 8960, 192), (1024, 1280, 384), (33, 512, 96),
                                   (512, 256, 512), (448, 512, 448), (300, 768, 700), (129, 1024, 1100), (20, 768, 100), (47, 1024, 70)])
def test_linear_q4k_gemm_vs_oracle(M, K, N):
    Wq, x, b = _q4k_case(M, K, N, M + K + N)
    y = ops.linear_q4k(Wq, x, N, bias=b)
    ref = orc.linear(x, Wq, orc.Q4_K, N, b)
    assert eq(y, ref), md(y, ref)


def test_linear_q4k_fp16_out_and_residual():
    Wq, x, b = _q4k_case(1, 1536, 256, 5)
    y16 = ops.linear_q4k(Wq, x, 256, bias=b, out_f16=True)
    ref16 = orc.f16_to_f32(orc.linear(x, Wq, orc.Q4_K, 256, b, out_f16=True))
    assert eq(y16.float(), ref16), md(y16.float(), ref16)
    res = rng(1).standard_normal((1, 256)).astype(np.float32)
    y = ops.linear_q4k(Wq, x, 256, bias=None, residual=res)
    ref = orc.linear(x, Wq, orc.Q4_K, 256) + res
    assert eq(y, ref), md(y, ref)
    Wq2, x2, _ = _q4k_case(40, 512, 128, 6)
    res2 = rng(2).standard_normal((40, 128)).astype(np.float32)
    y2 = ops.linear_q4k(Wq2, x2, 128, residual=res2)
    assert eq(y2, orc.linear(x2, Wq2, orc.Q4_K, 128) + res2)
    # bias + residual on ragged edges, and the fp16 output (the epilogue is split over a wave pair)
    Wq3, x3, b3 = _q4k_case(300, 768, 700, 7)
    res3 = rng(3).standard_normal((300, 700)).astype(np.float32)
    y3 = ops.linear_q4k(Wq3, x3, 700, bias=b3, residual=res3)
    assert eq(y3, orc.linear(x3, Wq3, orc.Q4_K, 700, b3) + res3)
    y3h = ops.linear_q4k(Wq3, x3, 700, bias=b3, out_f16=True)
    ref3h = orc.f16_to_f32(orc.linear(x3, Wq3, orc.Q4_K, 700, b3, out_f16=True))
    assert eq(y3h.float(), ref3h), md(y3h.float(), ref3h)


@pytest.mark.parametrize("M,K,N", [(40, 1536, 256), (282, 1536, 2048), (1024, 1280, 384), (33, 512, 96), (16, 256, 64)])
def test_linear_on_packed_producers(M, K, N):
    """Prefill path: quantiser / RMSNorm / LayerNorm write the GEMM's packed activation operand directly; same bits as norm -> Linear."""
    Wq, x, _ = _q4k_case(M, K, N, M + 3 * K + N, bias=False)
    r = rng(M + K)
    w = (1 + 0.1 * r.standard_normal(K)).astype(np.float32)
    b = (0.1 * r.standard_normal(K)).astype(np.float32)
    assert eq(ops.linear_q4k_packed_producers(Wq, x, N, "quant"), orc.linear(x, Wq, orc.Q4_K, N))
    assert eq(ops.linear_q4k_packed_producers(Wq, x, N, "rms", w=w), orc.linear(orc.rmsnorm(x, w, 1e-6), Wq, orc.Q4_K, N))
    assert eq(ops.linear_q4k_packed_producers(Wq, x, N, "ln", w=w, b=b), orc.linear(orc.layernorm(x, w, b, 1e-6), Wq, orc.Q4_K, N))


@pytest.mark.parametrize("M,K,N", [(64, 512, 96), (282, 1536, 128), (17, 256, 64)])
def test_linear_on_activation_fused_producers(M, K, N):
    """The quantiser with GELU / QuickGELU (LUT) or silu(gate)*up folded in feeds the GEMM the same bits as the separate ops."""
    Wq, x, _ = _q4k_case(M, K, N, 5 * M + K + N, bias=False)
    x = x * 3
    assert eq(ops.linear_q4k_packed_producers(Wq, x, N, "gelu"), orc.linear(orc.gelu(x), Wq, orc.Q4_K, N))
    assert eq(ops.linear_q4k_packed_producers(Wq, x, N, "quickgelu"), orc.linear(orc.quickgelu(x), Wq, orc.Q4_K, N))
    u = rng(M * K).standard_normal((M, K)).astype(np.float32)
    gu = np.concatenate([x, u], axis=1)
    assert eq(ops.linear_q4k_packed_producers(Wq, gu, N, "silu_mul"), orc.linear(orc.silu(x) * u, Wq, orc.Q4_K, N))


def test_linear_golden_reference(ops_gold):
    g = ops_gold
    for x, yref in ((g["lin_x5"], g["lin_y5"]), (g["lin_x1"], g["lin_y1"])):
        y = ops.linear_q4k(g["lin_w"], x, 96, bias=g["lin_b"])
        assert eq(y, yref), md(y, yref)


def test_linear_is_linear_in_weights_rows_and_zero_input():
    """size-independent properties at the full gate/up shape: zero activations give exactly the bias; duplicated weight rows
    give bit-identical outputs whichever wave computes them."""
    K, N = 1536, 17920
    r = rng(11)
    blk = synth.quantized_blocks(lib.Q4_K, r, 64 * K).reshape(64, -1)
    Wq = np.tile(blk, (N // 64, 1)).ravel()
    x = r.standard_normal((1, K)).astype(np.float32)
    y = ops.linear_q4k(Wq, x, N).cpu().numpy().reshape(N // 64, 64)
    assert np.array_equal(y, np.tile(y[0], (N // 64, 1)))
    b = r.standard_normal(N).astype(np.float32)
    y0 = ops.linear_q4k(Wq, np.zeros((1, K), dtype=np.float32), N, bias=b)
    assert eq(y0.reshape(-1), b)


# ---- A6/A7/A8 -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K,N", [(1536, 4096), (512, 160), (1024, 999), (4096, 64)])
def test_linear_q40_vs_oracle(K, N):
    r = rng(K + N)
    Wq = synth.quantized_blocks(lib.Q4_0, r, N * K, std=0.05, full_range=True)
    x = r.standard_normal((1, K)).astype(np.float32)
    y = ops.linear_q40(Wq, x, N)
    ref = orc.linear(x, Wq, orc.Q4_0, N)
    assert eq(y, ref), md(y, ref)


def test_tied_head_and_embedding_golden(ops_gold):
    g = ops_gold
    y = ops.linear_q40(g["emb_w"], g["mm_x"], 160)
    assert eq(y, g["mm_y"]), md(y, g["mm_y"])
    e = ops.embedding_q40(g["emb_ids"], g["emb_w"], 160, 512)
    assert eq(e, g["emb_y"])


def test_linear_f32_and_patch_convs(ops_gold):
    g = ops_gold
    y = ops.linear_f32(g["linf_w"], g["linf_x"])
    assert eq(y, g["linf_y"]), md(y, g["linf_y"])
    y = ops.patch_gemm(g["conv3_x"], g["conv3_w"])
    assert eq(y, g["conv3_y"]), md(y, g["conv3_y"])
    y = ops.conv2d_patch(g["conv2_x"], 8, 3, 12, g["conv2_w"], 8, 4, g["conv2_b"])
    assert eq(y, g["conv2_y"].reshape(2, 8, 3)), md(y, g["conv2_y"].reshape(2, 8, 3))
    r = rng(9)
    px, W = r.standard_normal((1024, 1176)).astype(np.float32), (r.standard_normal((1280, 1176)) * 0.02).astype(np.float32)
    y = ops.patch_gemm(px, W)
    assert eq(y, orc.patch_gemm(px, W))


# the fp32 Linear on the matrix cores (gemm_f32_mfma_kernel): ragged M / N tiles, K / 32 not a multiple of four (whole links left for the VALU), K % 32 leftovers, K % 4 != 0
# and M < 16 (the VALU kernel), with and without bias
@pytest.mark.parametrize("M,K,N,bias", [(64, 2048, 96, False), (50, 1176, 70, True), (17, 160, 33, True), (33, 224, 16, False), (40, 1000, 48, True), (16, 128, 16, False),
                                        (20, 130, 24, True), (5, 2048, 64, True), (100, 96, 40, False)])
def test_linear_f32_shapes_vs_oracle(M, K, N, bias):
    r = rng(M * 31 + K + N)
    W = (r.standard_normal((N, K)) * 0.05).astype(np.float32)
    x = r.standard_normal((M, K)).astype(np.float32)
    b = (r.standard_normal(N) * 0.1).astype(np.float32) if bias else None
    y = ops.linear_f32(W, x, bias=b)
    ref = orc.linear(x, W, orc.F32, N, b)
    assert eq(y, ref), md(y, ref)


def test_host_register_round_trip():      # a page-locked host row is an ordinary copy target (the engine-backed Module keeps its logits Tensor registered)
    import ctypes as C
    L = lib.load()
    host = np.zeros(151936, dtype=np.float32)
    src = torch.arange(151936, dtype=torch.float32, device="cuda")
    assert L.mllm_hip_host_register(C.c_void_p(host.ctypes.data), C.c_size_t(host.nbytes)) == 0
    try:
        assert L.mllm_hip_d2h(C.c_void_p(host.ctypes.data), C.c_void_p(src.data_ptr()), C.c_size_t(host.nbytes), None) == 0
        assert L.mllm_hip_sync(None) == 0
        assert np.array_equal(host, np.arange(151936, dtype=np.float32))
    finally:
        assert L.mllm_hip_host_unregister(C.c_void_p(host.ctypes.data)) == 0
    assert L.mllm_hip_host_register(None, C.c_size_t(16)) != 0


# ---- A9 / A18 --------------------------------------------------------------------------------------------------------------
def test_rmsnorm(ops_gold):
    g = ops_gold
    y = ops.rmsnorm(g["norm_x"], g["norm_w"], 1e-6)
    assert md(y, g["rms_y"]) <= 3e-7 * float(np.abs(g["rms_y"]).max())      # double-sum order may move the fp32 mean by 1 ulp
    x = rng(4).standard_normal((7, 1536)).astype(np.float32) * 5
    w = (1 + 0.1 * rng(5).standard_normal(1536)).astype(np.float32)
    y, q = ops.rmsnorm(x, w, 1e-6, quant=True)
    ref = orc.rmsnorm(x, w, 1e-6)
    assert md(y, ref) <= 3e-7 * float(np.abs(ref).max())
    # the fused quantisation is exactly the quantisation of the kernel's own fp32 output
    q2 = ops.quantize_q8k(y)
    assert eq(q.qs, q2.qs.cpu().numpy()) and eq(q.d, q2.d.cpu().numpy()) and eq(q.bsums, q2.bsums.cpu().numpy())
    y1 = ops.rmsnorm(x, w, 1e-6, add_unit_offset=True)
    assert md(y1, orc.rmsnorm(x, w, 1e-6, True)) <= 6e-7 * float(np.abs(ref).max()) * 2


def test_layernorm(ops_gold):
    g = ops_gold
    y = ops.layernorm(g["norm_x"], g["norm_w"], g["norm_b"], 1e-6)
    assert eq(y, g["ln_y"]), md(y, g["ln_y"])
    x = rng(6).standard_normal((9, 1280)).astype(np.float32) * 2 + 0.3
    y, q = ops.layernorm(x, g["norm_w"][:1].repeat(1280), None, 1e-6, quant=True)
    assert eq(y, orc.layernorm(x, g["norm_w"][:1].repeat(1280), None, 1e-6))
    x = rng(7).standard_normal((1024, 1280)).astype(np.float32) * 3 - 0.2       # the vision tower's shape: 64 workgroups of 16 rows
    wv, bv = rng(8).standard_normal(1280).astype(np.float32), rng(9).standard_normal(1280).astype(np.float32)
    assert eq(ops.layernorm(x, wv, bv, 1e-6), orc.layernorm(x, wv, bv, 1e-6))
    x = rng(10).standard_normal((5, 100)).astype(np.float32)                    # ragged dim (no float4 tail, one partial chunk)
    assert eq(ops.layernorm(x, wv[:100], bv[:100], 1e-5), orc.layernorm(x, wv[:100], bv[:100], 1e-5))
    q2 = ops.quantize_q8k(y)
    assert eq(q.qs, q2.qs.cpu().numpy())


# ---- A14 / A18 / A20 / A15 ------------------------------------------------------------------------------------------------
def test_activations_bit_exact(ops_gold):
    g = ops_gold
    assert eq(ops.silu(g["act_x"]), g["silu_y"])
    assert eq(ops.gelu(g["act_x"]), g["gelu_y"])
    assert eq(ops.quickgelu(g["act_x"]), g["quickgelu_y"])
    x = (rng(8).standard_normal(8960 * 3) * 4).astype(np.float32)
    assert eq(ops.silu(x), orc.silu(x))
    assert eq(ops.quickgelu(x), orc.quickgelu(x))


def test_qkv_rope_append_equals_the_separate_launches():
    """mllm_hip_qkv_rope_append (q rotated in place, k rotated into fp16 rows, v into the transposed fp16 slab: one launch) against mllm_hip_rope_apply twice + the fp16 stores,
    and against the restatement of the rotary, on a GQA shape with a ragged row count."""
    r = rng(61)
    S, Hq, Hkv, D = 37, 12, 2, 128
    qkv = r.standard_normal((S, (Hq + 2 * Hkv) * D)).astype(np.float32)
    sin_t, cos_t = orc.rope_table_hf(1000000.0, D, 64)
    sin_t, cos_t = np.ascontiguousarray(sin_t[5:5 + S, :D // 2]), np.ascontiguousarray(cos_t[5:5 + S, :D // 2])
    q, k, vt = ops.qkv_rope_append(qkv, S, Hq, Hkv, D, sin_t, cos_t, 64)
    q_ref = ops.rope_apply(np.ascontiguousarray(qkv[:, :Hq * D]), S, Hq, D, sin_t, cos_t)
    k_ref = ops.rope_apply(np.ascontiguousarray(qkv[:, Hq * D:(Hq + Hkv) * D]), S, Hkv, D, sin_t, cos_t, out_f16=True)
    assert eq(q, q_ref.cpu().numpy()) and np.array_equal(k.cpu().numpy().view(np.uint16), k_ref.cpu().numpy().view(np.uint16))
    v16 = qkv[:, (Hq + Hkv) * D:].astype(np.float16)
    assert np.array_equal(vt.cpu().numpy()[:, :S].view(np.uint16), np.ascontiguousarray(v16.T).view(np.uint16)) and not vt.cpu().numpy()[:, S:].any()


def test_silu_mul_add_mul_bit_exact():
    r = rng(10)
    gu = r.standard_normal((5, 2 * 8960)).astype(np.float32) * 2
    y = ops.silu_mul(gu, 8960)
    assert eq(y, orc.silu(gu[:, :8960]) * gu[:, 8960:])
    a, b = r.standard_normal(10007).astype(np.float32), r.standard_normal(10007).astype(np.float32)
    assert eq(ops.add(a, b), a + b) and eq(ops.mul(a, b), a * b)


@pytest.mark.parametrize("rows,dim", [(1, 1030), (5, 13), (3, 8), (7, 4), (2, 8967)])
def test_silu_rows_with_the_scalar_tail(rows, dim):
    """CPUSiLU applies mllm_vec_silu_f32 per row: the dim % 8 trailing values of EVERY row go through mllm_silu_f32 (libm expf), the rest through the polynomial."""
    x = (rng(rows * dim).standard_normal((rows, dim)) * 3).astype(np.float32)
    want = np.stack([orc.silu(x[r]) for r in range(rows)])
    assert eq(ops.silu_rows(x), want)


def test_softmax_argmax_index_put(ops_gold):
    g = ops_gold
    y = ops.softmax(g["sm_x"].reshape(6, 24))
    assert np.array_equal(y.cpu().numpy(), g["sm_y"].reshape(6, 24))       # every bit: chunk sums in the hsum order, libm expf on the < 8 trailing columns
    yv = ops.softmax(g["sm_x"].reshape(6, 24), valid=[1, 5, 24, 7, 8, 9])
    assert np.array_equal(yv.cpu().numpy(), orc.softmax(g["sm_x"].reshape(6, 24), valid=[1, 5, 24, 7, 8, 9]))
    for n in (16384, 151936, 50003):                                       # one long row: the chip-wide form (order-free parts everywhere, the row sum by one wave)
        xs = (rng(60 + n % 89).standard_normal((1, n)) * 4).astype(np.float32)
        assert np.array_equal(ops.softmax(xs).cpu().numpy(), orc.softmax(xs)), n
    for n in (4, 67, 200, 1031):                                           # widths with full passes, a partial pass and a tail
        xs = (rng(40 + n).standard_normal((5, n)) * 4).astype(np.float32)
        assert np.array_equal(ops.softmax(xs).cpu().numpy(), orc.softmax(xs)), n
    x = rng(12).standard_normal(151936).astype(np.float32)
    x[77] = x[150000] = 9.5                                   # tie: first index wins (std::max_element)
    assert ops.argmax(x) == 77
    # the chip-wide form (rows of 16384 values and more): ties inside one workgroup's slice, across slices, a ragged last slice holding the maximum, a constant row
    for n in (16384, 50001, 151936):
        r = rng(200 + n % 97)
        y = r.standard_normal(n).astype(np.float32)
        assert ops.argmax(y) == int(np.argmax(y)), n
        y[n - 1] = 11.0
        assert ops.argmax(y) == n - 1, n
        y[n // 2] = y[n // 2 + 3] = 11.0
        assert ops.argmax(y) == n // 2, n
        y[5] = 11.0
        assert ops.argmax(y) == 5, n
        assert ops.argmax(np.full(n, -3.5, dtype=np.float32)) == 0, n
    dst = rng(13).standard_normal((10, 64)).astype(np.float32)
    val = rng(14).standard_normal((3, 64)).astype(np.float32)
    out = ops.index_put_rows(dst, val, [7, 0, 4]).cpu().numpy()
    exp = dst.copy(); exp[[7, 0, 4]] = val
    assert np.array_equal(out, exp)


# ---- A10 / A11 / A19 ----------------------------------------------------------------------------------------------------------
def test_rotary_bit_exact(ops_gold):
    g = ops_gold
    s, c = lib.mrope_table(1000000.0, 128, g["mrope_pos"])
    assert eq(ops.rope_apply(g["mrope_x"], 5, 2, 128, s, c), g["mrope_y"])
    k16 = ops.rope_apply(g["mrope_x"], 5, 2, 128, s, c, out_f16=True)
    assert eq(k16.view(torch.int16), orc.rope_apply(g["mrope_x"], 5, 2, 128, s, c, out_f16=True).view(np.int16))
    s, c = lib.rope_table_hf(10000.0, 64, 64)
    assert eq(ops.rope_apply(g["rope_x"], 5, 2, 64, s[:5], c[:5]), g["rope_y"])
    s, c = lib.vision_rope_table(1, 4, 4, 2, 8)
    assert eq(ops.rope_apply(g["vrope_x"], 16, 2, 16, s, c), g["vrope_y"])


def test_rotary_fp16_store_rounds_twice_like_the_reference():
    """fp32 rotate result, THEN fp16: a fused fma->fp16 (one rounding) differs in ~2^-13 of the values; 400k values catch it."""
    S, H, D = 400, 8, 128
    x = rng(31).standard_normal((S, H * D)).astype(np.float32)
    pos = np.tile(np.arange(S, dtype=np.float32), (3, 1))
    s, c = lib.mrope_table(1000000.0, D, pos)
    k16 = ops.rope_apply(x, S, H, D, s, c, out_f16=True)
    assert eq(k16.view(torch.int16), orc.rope_apply(x, S, H, D, s, c, out_f16=True).view(np.int16))
    Wq, xx, b = _q4k_case(64, 512, 256, 77)
    y16 = ops.linear_q4k(Wq, xx, 256, bias=b, out_f16=True)       # GEMM epilogue (bias add, then fp16)
    assert eq(y16.view(torch.int16), orc.linear(xx, Wq, orc.Q4_K, 256, b, out_f16=True).view(np.int16))
    Wq, xx, b = _q4k_case(1, 1536, 4096, 78)
    y16 = ops.linear_q4k(Wq, xx, 4096, bias=b, out_f16=True)      # GEMV epilogue
    assert eq(y16.view(torch.int16), orc.linear(xx, Wq, orc.Q4_K, 4096, b, out_f16=True).view(np.int16))


# ---- A13 ---------------------------------------------------------------------------------------------------------------------
def test_fa2_golden_fp32_kv(ops_gold):
    g = ops_gold
    o = ops.flash_attention2(g["fa_q"], g["fa_k"], g["fa_v"], 40, 40, 2, 2, 16, False)
    assert eq(o, g["fa_o"]), md(o, g["fa_o"])
    o = ops.flash_attention2(g["fac_q"], g["fac_k"], g["fac_v"], 12, 12, 4, 2, 16, True)
    assert eq(o, g["fac_o"]), md(o, g["fac_o"])


@pytest.mark.parametrize("Sq,Sk,Hq,Hkv,D,causal,f16", [
    (282, 282, 12, 2, 128, True, True), (1, 283, 12, 2, 128, True, True), (1, 800, 12, 2, 128, True, True), (1, 1, 12, 2, 128, True, True),
    (130, 130, 16, 16, 80, False, False), (7, 40, 4, 4, 64, True, True), (64, 64, 2, 1, 16, True, False), (1, 257, 16, 16, 64, True, True),
    (1024, 1024, 2, 2, 80, False, False), (5, 5, 2, 2, 64, True, True), (6, 7, 2, 1, 64, True, False), (3, 11, 2, 2, 64, True, True),
    (2, 2, 1, 1, 16, False, False), (21, 23, 4, 2, 128, True, True), (300, 300, 4, 2, 128, True, True), (9, 270, 2, 2, 80, False, True)])
def test_fa2_vs_oracle(Sq, Sk, Hq, Hkv, D, causal, f16):
    r = rng(Sq * 3 + Sk + D)
    q = r.standard_normal((Sq, Hq * D)).astype(np.float32)
    k = r.standard_normal((Sk, Hkv * D)).astype(np.float32)
    v = r.standard_normal((Sk, Hkv * D)).astype(np.float32)
    if f16:
        k16, v16 = k.astype(np.float16), v.astype(np.float16)
        o = ops.flash_attention2(q, torch.from_numpy(k16), torch.from_numpy(v16), Sq, Sk, Hq, Hkv, D, causal)
        ref = orc.attention(q, k16.view(np.uint16), v16.view(np.uint16), Sq, Sk, Hq, Hkv, D, causal)
    else:
        o = ops.flash_attention2(q, k, v, Sq, Sk, Hq, Hkv, D, causal)
        ref = orc.attention(q, k, v, Sq, Sk, Hq, Hkv, D, causal)
    assert eq(o, ref), md(o, ref)


@pytest.mark.parametrize("Sq,Sk,Hq,Hkv,D", [(282, 282, 12, 2, 128), (40, 40, 12, 2, 128), (1, 300, 12, 2, 128), (3, 50, 4, 2, 64), (21, 23, 4, 2, 128), (1, 1, 2, 1, 128),
                                            (1, 700, 12, 2, 128), (1, 1500, 4, 2, 128), (1, 513, 2, 1, 64), (600, 600, 4, 2, 128)])
def test_fa2_on_the_engine_kv_layout(Sq, Sk, Hq, Hkv, D):
    """K fp16 rows + V transposed fp16 (the resident slab layout): same bits as the reference layout."""
    r = rng(Sq + 7 * Sk)
    q = r.standard_normal((Sq, Hq * D)).astype(np.float32)
    k16 = r.standard_normal((Sk, Hkv * D)).astype(np.float16)
    v = r.standard_normal((Sk, Hkv * D)).astype(np.float32)
    o = ops.flash_attention2_vt(q, torch.from_numpy(k16), v, Sq, Sk, Hq, Hkv, D, True)
    ref = orc.attention(q, k16.view(np.uint16), v.astype(np.float16).view(np.uint16), Sq, Sk, Hq, Hkv, D, True)
    assert eq(o, ref), md(o, ref)


def test_fa2_batch_is_the_per_set_attention():
    """mllm_hip_fa2_batch (the images of a vision pass in one launch): every set equals its own mllm_hip_fa2 call, for fp32 and fp16 K / V, causal or not, ragged Sq."""
    r = np.random.default_rng(21)
    for (nb, Sq, Sk, Hq, Hkv, D, causal, f16) in ((3, 197, 197, 4, 4, 64, False, False), (2, 70, 90, 4, 2, 128, True, True), (5, 33, 33, 2, 2, 80, False, False)):
        q = r.standard_normal((nb, Sq, Hq * D)).astype(np.float32)
        k = r.standard_normal((nb, Sk, Hkv * D)).astype(np.float32); v = r.standard_normal((nb, Sk, Hkv * D)).astype(np.float32)
        kt, vt = torch.from_numpy(k).cuda(), torch.from_numpy(v).cuda()
        if f16: kt, vt = kt.half(), vt.half()
        got = ops.flash_attention2_batch(q, kt, vt, Sq, Sk, Hq, Hkv, D, causal).cpu().numpy()
        for b in range(nb):
            one = ops.flash_attention2(q[b], kt[b].contiguous(), vt[b].contiguous(), Sq, Sk, Hq, Hkv, D, causal).cpu().numpy()
            assert np.array_equal(got[b], one), (nb, Sq, Sk, D, b)


def test_fa2_online_softmax_rescale_is_forced():
    """A spiked late key forces the running max to jump in the last tile (rescale branch of the online softmax)."""
    Sq = Sk = 96
    r = rng(21)
    q = r.standard_normal((Sq, 128)).astype(np.float32)
    k = r.standard_normal((Sk, 128)).astype(np.float32) * 0.1
    v = r.standard_normal((Sk, 128)).astype(np.float32)
    k[90] = q[95] * 3
    o = ops.flash_attention2(q, k, v, Sq, Sk, 1, 1, 128, True)
    assert eq(o, orc.attention(q, k, v, Sq, Sk, 1, 1, 128, True))


def test_fa2_decode_reads_sk_from_device():
    r = rng(22)
    q = r.standard_normal((1, 12 * 128)).astype(np.float32)
    k = r.standard_normal((800, 256)).astype(np.float16)
    v = r.standard_normal((800, 256)).astype(np.float16)
    sk = torch.tensor([300], dtype=torch.int32, device="cuda")
    o = ops.flash_attention2(q, torch.from_numpy(k), torch.from_numpy(v), 1, 800, 12, 2, 128, True, sk_dev=sk)
    ref = orc.attention(q, k[:300].view(np.uint16), v[:300].view(np.uint16), 1, 300, 12, 2, 128, True)
    assert eq(o, ref), md(o, ref)


def test_transpose_f32_moves_every_element():
    """mllm_hip_transpose_f32 (the data-moving F_TRANSPOSE case, used by the reference-side Conv2D adapter): ragged edges of the 32 x 32 tiles included."""
    import ctypes as C
    import torch
    from mllm_amd import lib
    for rows, cols in ((576, 1024), (196, 768), (33, 65), (1, 7), (31, 32)):
        x = torch.arange(rows * cols, dtype=torch.float32, device="cuda").reshape(rows, cols)
        y = torch.empty((cols, rows), dtype=torch.float32, device="cuda")
        lib.check(lib.load().mllm_hip_transpose_f32(lib.vp(x), lib.vp(y), C.c_int(rows), C.c_int(cols), None), "transpose_f32")
        torch.cuda.synchronize()
        assert torch.equal(y, x.t().contiguous()), (rows, cols)


# ---- the adapter's fused M = 1 launches: one launch = the run of Ops it replaces, output for output, bit for bit ----------------------------------------------------
def _rows_case(K, Ns, seed):
    r = rng(seed)
    Ws = [synth.quantized_blocks(lib.Q4_K, r, N * K, std=0.05, full_range=True) for N in Ns]
    xa = r.standard_normal(K).astype(np.float32)
    xb = r.standard_normal(K).astype(np.float32)
    w = (1.0 + 0.1 * r.standard_normal(K)).astype(np.float32)
    return r, Ws, xa, xb, w


@pytest.mark.parametrize("K,Ns", [(1536, (1536, 256, 256)), (256, (64, 64, 32)), (2048, (2048, 2048, 2048)), (4096, (4096,)), (1536, (151936,)), (768, (768, 128, 128)), (4096, (4096, 4096, 4096)), (1536, (1536, 250))])
@pytest.mark.parametrize("with_add", [False, True])
def test_row_fused_add_norm_linears(K, Ns, with_add):
    """F_TTADD -> RMSNORM -> up to three LINEARs (the q | k | v run): every output equals the separate entry points' and the oracle's."""
    r, Ws, xa, xb, w = _rows_case(K, Ns, K + sum(Ns))
    biases = [(r.standard_normal(N) * 0.1).astype(np.float32) if i != 1 else None for i, N in enumerate(Ns)]
    out = ops.row_fused(xa, [(W, N, b, None) for W, N, b in zip(Ws, Ns, biases)], xb=xb if with_add else None, norm_w=w, eps=1e-6)
    s = ops.add(xa, xb) if with_add else torch.from_numpy(xa).cuda()
    if with_add:
        assert eq(out["sum"], s.cpu().numpy()) and eq(out["sum"], xa + xb)
    n = ops.rmsnorm(s.reshape(1, K), w, 1e-6)
    assert eq(out["norm"], n.reshape(-1).cpu().numpy())
    for W, N, b, y in zip(Ws, Ns, biases, out["y"]):
        want = ops.linear_q4k(W, n, N, bias=b)
        assert eq(y, want.reshape(-1).cpu().numpy()), md(y, want.reshape(-1).cpu().numpy())
        assert eq(y, orc.linear(n.cpu().numpy(), W, orc.Q4_K, N, b).reshape(-1))


@pytest.mark.parametrize("K,N", [(1536, 1536), (8960, 1536), (256, 40), (4096, 4096), (2816, 1024)])
def test_row_fused_linear_then_add(K, N):
    """LINEAR -> F_TTADD (o / down projection and the residual add behind it), no prologue."""
    r, (W,), xa, _, _ = _rows_case(K, (N,), K * 3 + N)
    res = r.standard_normal(N).astype(np.float32)
    out = ops.row_fused(xa, [(W, N, None, res)])
    y = ops.linear_q4k(W, xa.reshape(1, K), N).reshape(-1)
    assert eq(out["y"][0], y.cpu().numpy())
    assert eq(out["post"][0], ops.add(y, res).cpu().numpy())
    assert eq(out["y"][0], orc.linear(xa.reshape(1, K), W, orc.Q4_K, N).reshape(-1))


@pytest.mark.parametrize("K,I", [(1536, 8960), (256, 48), (2048, 5632), (1024, 2816), (4096, 11008), (768, 4864), (512, 1030), (512, 1280), (1280, 3840), (256, 35), (256, 40), (1536, 5)])
def test_row_fused_norm_gate_silu_up_mul(K, I):
    """RMSNORM -> LINEAR gate -> SILU -> LINEAR up -> F_TTMUL (the MLP's first five Ops), with and without the F_TTADD in front."""
    r, (Wg, Wu), xa, xb, w = _rows_case(K, (I, I), K + I)
    for with_add in (False, True):
        out = ops.row_fused(xa, [(Wg, I, None, None), (Wu, I, None, None)], xb=xb if with_add else None, norm_w=w, eps=1e-5, gateup=True)
        s = ops.add(xa, xb) if with_add else torch.from_numpy(xa).cuda()
        n = ops.rmsnorm(s.reshape(1, K), w, 1e-5)
        g = ops.linear_q4k(Wg, n, I).reshape(-1)
        u = ops.linear_q4k(Wu, n, I).reshape(-1)
        sg = ops.silu(g)
        assert eq(out["norm"], n.reshape(-1).cpu().numpy())
        assert eq(out["y"][0], g.cpu().numpy()) and eq(out["y"][1], u.cpu().numpy())
        assert eq(out["silu"], sg.cpu().numpy()) and eq(out["mul"], ops.mul(sg, u).cpu().numpy())
        if I % 8 == 0:
            assert eq(out["silu"], orc.silu(g.cpu().numpy()))


def test_row_fused_refuses_what_it_does_not_cover():
    r, (W,), xa, _, _ = _rows_case(256, (4,), 1)
    with pytest.raises(lib.MllmHipError):
        ops.row_fused(np.zeros(11264, dtype=np.float32), [(np.zeros(144 * 44 * 8, dtype=np.uint8), 8, None, None)])      # K / 256 = 44 super-blocks: beyond the five register steps
    with pytest.raises(lib.MllmHipError):
        ops.row_fused(np.zeros(300, dtype=np.float32), [(W, 4, None, None)])                                              # K % 256 != 0


@pytest.mark.parametrize("S,Hq,Hkv,D", [(1, 12, 2, 128), (7, 4, 4, 64), (282, 12, 2, 128), (1, 32, 32, 128)])
def test_rope2_store2_equals_the_four_ops(S, Hq, Hkv, D):
    r = rng(S + Hq + D)
    q = r.standard_normal((S, Hq * D)).astype(np.float32)
    k = r.standard_normal((S, Hkv * D)).astype(np.float32)
    v = r.standard_normal((S, Hkv * D)).astype(np.float32)
    ang = r.standard_normal((S, D // 2)).astype(np.float32)
    sq, cq = np.sin(ang).astype(np.float32), np.cos(ang).astype(np.float32)
    sk, ck = np.sin(ang * 0.5).astype(np.float32), np.cos(ang * 0.5).astype(np.float32)
    qo, ko, k16, v16 = ops.rope2_store2(q, k, v, S, Hq, Hkv, D, sq, cq, sk, ck)
    assert eq(qo, ops.rope_apply(q, S, Hq, D, sq, cq).cpu().numpy())
    kw = ops.rope_apply(k, S, Hkv, D, sk, ck)
    assert eq(ko, kw.cpu().numpy())
    assert eq(k16.view(torch.int16), kw.to(torch.float16).view(torch.int16).cpu().numpy())
    assert eq(k16.view(torch.int16), ops.rope_apply(k, S, Hkv, D, sk, ck, out_f16=True).view(torch.int16).cpu().numpy())
    assert eq(v16.view(torch.int16), torch.from_numpy(v).to(torch.float16).view(torch.int16).numpy())


@pytest.mark.parametrize("T,Hq,Hkv,D", [(0, 4, 2, 64), (1, 12, 2, 128), (305, 12, 2, 128), (127, 16, 16, 64), (128, 32, 32, 128), (640, 14, 2, 64), (799, 12, 2, 128), (1500, 32, 4, 64)])
def test_fa2_decode_step_equals_the_five_ops(T, Hq, Hkv, D):
    """RoPE(q), RoPE(k), the two fp16 cache appends and F_FA2 of one decode position in one launch = the five entry points one after the other, output for output, bit for bit
    (T = keys already in the cache: 0, across the 128-key V chunks, the 512-key score pass and the three-slot ring)."""
    r = rng(T + Hq + D)
    q = r.standard_normal(Hq * D).astype(np.float32)
    k = r.standard_normal(Hkv * D).astype(np.float32)
    v = r.standard_normal(Hkv * D).astype(np.float32)
    ang = r.standard_normal(D // 2).astype(np.float32)
    sq, cq = np.sin(ang).astype(np.float32), np.cos(ang).astype(np.float32)
    sk, ck = np.sin(ang * 0.5).astype(np.float32), np.cos(ang * 0.5).astype(np.float32)
    past_k = torch.from_numpy(r.standard_normal((T + 1, Hkv * D)).astype(np.float32)).to(torch.float16).cuda()
    past_v = torch.from_numpy(r.standard_normal((T + 1, Hkv * D)).astype(np.float32)).to(torch.float16).cuda()
    ks, vs = past_k.clone(), past_v.clone()
    qo, ko, o = ops.fa2_decode_step(q, k, v, ks, vs, T, Hq, Hkv, D, sq, cq, sk, ck)
    q_want = ops.rope_apply(q.reshape(1, -1), 1, Hq, D, sq.reshape(1, -1), cq.reshape(1, -1))
    k_want = ops.rope_apply(k.reshape(1, -1), 1, Hkv, D, sk.reshape(1, -1), ck.reshape(1, -1))
    kw, vw = past_k.clone(), past_v.clone()
    kw[T] = k_want.reshape(-1).to(torch.float16)
    vw[T] = torch.from_numpy(v).to(torch.float16).cuda()
    o_want = ops.flash_attention2(q_want, kw, vw, 1, T + 1, Hq, Hkv, D, True)
    assert eq(qo, q_want.reshape(-1).cpu().numpy()) and eq(ko, k_want.reshape(-1).cpu().numpy())
    assert eq(ks.view(torch.int16), kw.view(torch.int16).cpu().numpy()) and eq(vs.view(torch.int16), vw.view(torch.int16).cpu().numpy())
    assert eq(o, o_want.reshape(-1).cpu().numpy()), md(o, o_want.reshape(-1).cpu().numpy())
