"""Error behaviour of the C-ABI boundary (include/mllm_hip.h:33-38): every entry point returns MLLM_HIP_OK or a negative MLLM_HIP_ERR_*, validates
its arguments before anything is launched, and treats an empty batch as a no-op.  These calls return before the first HIP call, so they run
without a GPU (no compute); the adapter of INTEGRATION.md relies on exactly these codes to refuse an Op at opCreate / reshape time."""
import ctypes as C
import os

import pytest

from mllm_amd import lib

OK, ERR_HIP, ERR_SHAPE, ERR_DTYPE, ERR_IO, ERR_ARG = 0, -1, -2, -3, -4, -5
F32, F16 = lib.F32, lib.F16
NULL = C.c_void_p(0)
P = C.c_void_p(0x1000)      # a non-null pointer that is never dereferenced on these paths


@pytest.fixture(scope="module")
def L():
    return lib.load()


def test_header_codes_match_the_bindings():
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mllm_hip.h")).read()
    for name, val in (("MLLM_HIP_OK", "0"), ("MLLM_HIP_ERR_HIP", "(-1)"), ("MLLM_HIP_ERR_SHAPE", "(-2)"), ("MLLM_HIP_ERR_DTYPE", "(-3)"),
                      ("MLLM_HIP_ERR_IO", "(-4)"), ("MLLM_HIP_ERR_ARG", "(-5)")):
        assert f"#define {name} {val}" in hdr
    assert lib.OK == OK


def test_gemm_on_packed_operands_validates_before_launch(L):
    f = L.mllm_hip_linear_q4kp_packed
    args = lambda W, xp, ydt, M, N, K: (W, NULL, xp, P, C.c_int(ydt), C.c_int64(N), NULL, C.c_int(M), C.c_int(N), C.c_int(K), NULL)
    assert f(*args(P, P, F32, 32, 64, 300)) == ERR_SHAPE          # K is not a whole number of super-blocks
    assert f(*args(P, P, F32, 32, 0, 256)) == ERR_SHAPE
    assert f(*args(P, P, 99, 32, 64, 256)) == ERR_DTYPE
    assert f(*args(P, P, F32, 0, 64, 256)) == OK                  # empty batch: nothing to do, nothing touched
    assert f(*args(NULL, P, F32, 32, 64, 256)) == ERR_ARG
    assert f(*args(P, NULL, F32, 32, 64, 256)) == ERR_ARG


def test_producers_validate_before_launch(L):
    q = L.mllm_hip_quantize_q8k_packed
    assert q(P, P, C.c_int(4), C.c_int(300), NULL) == ERR_SHAPE
    assert q(P, NULL, C.c_int(4), C.c_int(256), NULL) == ERR_SHAPE
    assert q(P, P, C.c_int(0), C.c_int(256), NULL) == OK
    assert L.mllm_hip_quantize_q8k_packed_act(P, NULL, P, C.c_int(4), C.c_int(256), NULL) == ERR_SHAPE     # no LUT
    assert L.mllm_hip_quantize_q8k_packed_silu_mul(P, P, C.c_int(4), C.c_int(100), NULL) == ERR_SHAPE
    assert L.mllm_hip_quantize_q8k_packed_silu_mul(P, P, C.c_int(0), C.c_int(256), NULL) == OK
    r = L.mllm_hip_rmsnorm_packed
    assert r(P, P, NULL, P, C.c_int(0), C.c_int(256), C.c_float(1e-6), C.c_int(0), NULL) == OK
    assert r(P, P, NULL, NULL, C.c_int(4), C.c_int(256), C.c_float(1e-6), C.c_int(0), NULL) == ERR_SHAPE
    assert r(P, P, NULL, P, C.c_int(4), C.c_int(100), C.c_float(1e-6), C.c_int(0), NULL) == ERR_SHAPE
    assert L.mllm_hip_layernorm_packed(P, P, P, NULL, NULL, C.c_int(4), C.c_int(256), C.c_float(1e-6), NULL) == ERR_ARG
    ln = L.mllm_hip_layernorm
    assert ln(P, P, P, P, NULL, NULL, NULL, C.c_int(0), C.c_int(256), C.c_float(1e-6), NULL) == OK
    assert ln(P, P, P, NULL, P, P, P, C.c_int(4), C.c_int(100), C.c_float(1e-6), NULL) == ERR_SHAPE           # fused Q8_K output needs dim % 256 == 0
    assert ln(P, P, P, NULL, P, NULL, NULL, C.c_int(4), C.c_int(256), C.c_float(1e-6), NULL) == ERR_SHAPE     # qs without d / bsums
    assert ln(P, P, P, NULL, NULL, NULL, NULL, C.c_int(4), C.c_int(256), C.c_float(1e-6), NULL) == ERR_ARG    # no output at all


def test_elementwise_and_attention_validate_before_launch(L):
    assert L.mllm_hip_silu(P, P, C.c_int64(0), NULL) == OK
    assert L.mllm_hip_add(P, P, P, C.c_int64(0), NULL) == OK
    assert L.mllm_hip_silu_mul(P, P, C.c_int(4), C.c_int(6), NULL) == ERR_SHAPE
    assert L.mllm_hip_silu_mul(P, P, C.c_int(0), C.c_int(8), NULL) == OK
    assert L.mllm_hip_act_lut(P, P, C.c_int64(0), P, NULL) == OK
    fa = L.mllm_hip_fa2
    a = lambda kvdt, Sq, Sk, Hq, Hkv, D, ldk=256: (P, C.c_int64(256), P, C.c_int64(ldk), P, C.c_int64(ldk), C.c_int(kvdt), P, C.c_int64(256), C.c_int(Sq),
                                                    C.c_int(Sk), C.c_int(Hq), C.c_int(Hkv), C.c_int(D), C.c_int(1), NULL, NULL, NULL)
    assert fa(*a(F16, 0, 8, 2, 1, 128)) == ERR_SHAPE
    assert fa(*a(F16, 8, 8, 3, 2, 128)) == ERR_SHAPE             # query heads must be a multiple of the K/V heads
    assert fa(*a(7, 8, 8, 2, 1, 128)) == ERR_DTYPE
    assert fa(*a(F16, 8, 8, 2, 1, 128, ldk=250)) == ERR_SHAPE    # rows must allow 16-byte loads
    assert L.mllm_hip_linear(P, C.c_int(F32), NULL, P, P, C.c_int(F16), C.c_int64(8), C.c_int(1), C.c_int(8), C.c_int(8), NULL, NULL) == ERR_DTYPE
    assert L.mllm_hip_linear(P, C.c_int(lib.Q4_K), NULL, P, P, C.c_int(F32), C.c_int64(8), C.c_int(1), C.c_int(8), C.c_int(256), NULL, NULL) == ERR_ARG   # no workspace


def test_engine_create_reports_io_and_argument_errors(L, tmp_path):
    out = C.c_void_p(0)
    assert L.mllm_hip_qwen2vl_create(NULL, b"/nonexistent", C.byref(out)) == ERR_ARG
    from mllm_amd import synth
    cfg = lib.make_config(synth.qwen2vl_tiny()) if hasattr(lib, "make_config") else None
    if cfg is not None:
        assert L.mllm_hip_qwen2vl_create(C.byref(cfg), b"/nonexistent/model.mllm", C.byref(out)) == ERR_IO
        bad = tmp_path / "bad.mllm"
        bad.write_bytes(b"\x00" * 64)                               # wrong magic number
        assert L.mllm_hip_qwen2vl_create(C.byref(cfg), str(bad).encode(), C.byref(out)) == ERR_IO
    assert not out.value


def test_generic_engine_create_validates_config_and_file(L, tmp_path):
    """mllm_hip_model_create: nonsense fields -> ERR_ARG, unsupported geometry -> ERR_SHAPE, all before any file or device work; a truncated or corrupt .mllm
    index -> ERR_IO (nothing in the header is trusted: every length is checked against the mapped size)."""
    import struct
    from mllm_amd import synth
    out = C.c_void_p(0)
    mk = lambda **kw: _cfg(lib.model_config(synth.qwen2vl_tiny()), **kw)

    def _cfg(c, **kw):
        for k, v in kw.items():
            setattr(c, k, v)
        return c
    path = b"/nonexistent/model.mllm"
    assert L.mllm_hip_model_create(NULL, path, C.byref(out)) == ERR_ARG
    assert L.mllm_hip_model_create(C.byref(mk(arch=9)), path, C.byref(out)) == ERR_ARG
    assert L.mllm_hip_model_create(C.byref(mk(heads=0)), path, C.byref(out)) == ERR_ARG              # would divide by zero
    assert L.mllm_hip_model_create(C.byref(mk(kv_heads=0)), path, C.byref(out)) == ERR_ARG
    assert L.mllm_hip_model_create(C.byref(mk(v_heads=0)), path, C.byref(out)) == ERR_ARG
    assert L.mllm_hip_model_create(C.byref(mk(hidden=300)), path, C.byref(out)) == ERR_SHAPE         # not a whole number of super-blocks
    assert L.mllm_hip_model_create(C.byref(mk(hidden=512, heads=16)), path, C.byref(out)) == ERR_SHAPE   # head_dim 32: the decode attention is built for 64 and 128
    assert L.mllm_hip_model_create(C.byref(mk(kv_heads=3, heads=4, hidden=512)), path, C.byref(out)) == ERR_SHAPE
    assert L.mllm_hip_model_create(C.byref(mk()), path, C.byref(out)) == ERR_IO                      # valid config, missing file
    # a header whose index length points past the end of the file; an entry whose name length does; an entry whose data does
    good_cfg = mk()
    for i, blob in enumerate((struct.pack("<iQ", 20012, 1 << 40),
                              struct.pack("<iQ", 20012, 40) + struct.pack("<i", 1 << 20) + b"x" * 36,
                              struct.pack("<iQ", 20012, 4 + 1 + 20) + struct.pack("<i", 1) + b"a" + struct.pack("<QQi", 1 << 30, 12, 0),
                              b"\x2c\x4e\x00")):
        f = tmp_path / f"trunc{i}.mllm"
        f.write_bytes(blob)
        assert L.mllm_hip_model_create(C.byref(good_cfg), str(f).encode(), C.byref(out)) == ERR_IO, i
    assert not out.value


def test_sampling_and_preprocess_entry_points_validate(L):
    assert L.mllm_hip_sort_desc(P, C.c_int(0), P, P, P, C.c_size_t(1 << 20), NULL) == ERR_ARG
    assert L.mllm_hip_sort_desc(P, C.c_int(16), P, P, NULL, C.c_size_t(0), NULL) == ERR_ARG
    assert L.mllm_hip_transpose_f32(P, P, C.c_int(4), C.c_int(4), NULL) == ERR_ARG                   # in place is not supported
    assert L.mllm_hip_transpose_f32(P, C.c_void_p(0x2000), C.c_int(0), C.c_int(4), NULL) == ERR_SHAPE
    import numpy as np
    grid = np.zeros(3, dtype=np.int32)
    assert L.mllm_hip_qwen2vl_preprocess_shape(C.c_int(10), C.c_int(4000), C.c_int(3136), C.c_int(12845056), lib.vp(grid)) == ERR_SHAPE   # aspect ratio above 200
    assert L.mllm_hip_qwen2vl_preprocess_shape(C.c_int(448), C.c_int(448), C.c_int(3136), C.c_int(12845056), lib.vp(grid)) == OK and grid.tolist() == [1, 32, 32]
    assert L.mllm_hip_qwen2vl_preprocess(NULL, C.c_int(8), C.c_int(8), C.c_int(3136), C.c_int(12845056), P, lib.vp(grid), NULL) == ERR_ARG
    assert L.mllm_hip_all_gather_rows(NULL, P, P, C.c_int64(4), C.c_int(8), NULL) == ERR_ARG
    assert L.mllm_hip_comm_create(NULL, C.c_int(1), C.c_int(0), C.byref(C.c_void_p())) == ERR_ARG
    assert L.mllm_hip_rope_table_hf_llama3(C.c_float(1e4), C.c_int(63), C.c_int(8), C.c_float(8), C.c_float(1), C.c_float(4), C.c_float(8192), P, P) == ERR_ARG


def test_fa2_batch_validates_before_any_launch(L):
    i64 = C.c_int64
    def call(Sq=32, Sk=32, Hq=4, Hkv=2, D=64, nb=2, kvdt=0, ldk=128, bk=32 * 128, q=P, k=P):
        return L.mllm_hip_fa2_batch(q, i64(256), k, i64(ldk), P, i64(ldk), C.c_int(kvdt), P, i64(256), C.c_int(Sq), C.c_int(Sk), C.c_int(Hq), C.c_int(Hkv), C.c_int(D),
                                    C.c_int(0), C.c_int(nb), i64(Sq * 256), i64(bk), i64(bk), i64(Sq * 256), NULL)
    assert call(Sq=2) == ERR_SHAPE               # the batched form is the prefill recurrence's (Sq >= 4)
    assert call(Hq=3) == ERR_SHAPE               # query heads must be a multiple of the K/V heads
    assert call(nb=0) == ERR_SHAPE
    assert call(kvdt=12) == ERR_DTYPE            # K / V are fp32 or fp16 rows
    assert call(ldk=130) == ERR_SHAPE            # 16-byte loads of the K rows
    assert call(bk=4100) == ERR_SHAPE
    assert call(q=NULL) == ERR_ARG
    assert call(D=48) == ERR_SHAPE               # head sizes 16, 64, 80, 128


def test_options_live_in_the_library_not_in_the_environment(L, monkeypatch):
    """mllm_hip_set_option: the measurement / bring-up switches of the launch paths are a table in the library; setting the old environment names changes nothing,
    unknown names are refused, -1 means unset."""
    from mllm_amd import lib
    monkeypatch.setenv("MLLM_HIP_VISION_BATCH", "3")
    assert lib.get_option("vision_batch") == -1
    lib.set_option("vision_batch", 2)
    assert lib.get_option("vision_batch") == 2
    lib.set_option("vision_batch", -1)
    assert L.mllm_hip_set_option(b"no_such_switch", C.c_int(1)) == ERR_ARG
    assert L.mllm_hip_set_option(NULL, C.c_int(1)) == ERR_ARG
    v = C.c_int(7)
    assert L.mllm_hip_get_option(b"attn_ds", C.byref(v)) == OK and v.value == -1
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # every option the library knows starts unset (-1), the newest included (a fixed-length initialiser once left the last one at 0)
    names = re.search(r'g_option_names\[OPT_COUNT\] = \{([^}]*)\}', open(os.path.join(root, "mllm_amd", "csrc", "runtime.hip")).read()).group(1)
    names = [n.strip().strip('"') for n in names.split(",")]
    assert "merge_o" in names and "chain_cont" in names
    out = subprocess.run([sys.executable, "-c", "from mllm_amd import lib\nprint([lib.get_option(n) for n in %r])" % names], capture_output=True, text=True,
                         cwd=root, env=dict(os.environ, PYTHONPATH=root))
    assert out.returncode == 0 and out.stdout.strip() == str([-1] * len(names)), (out.stdout, out.stderr[-500:])
    # the one environment variable the library still reads (profiling scripts: replay the decode step as plain launches)
    src = "".join(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mllm_amd", "csrc", f)).read()
                  for f in os.listdir(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mllm_amd", "csrc")) if f.endswith((".hip", ".h")))
    assert set(re.findall(r'getenv\("([A-Z_]+)"\)', src)) == {"MLLM_HIP_NO_GRAPH"}


def test_linear_refuses_a_residual_with_an_fp16_output(L):
    for fn, args in ((L.mllm_hip_linear_q4kp_packed, (P, NULL, P, P, C.c_int(1), C.c_int64(64), P, C.c_int(16), C.c_int(64), C.c_int(256), NULL)),
                     (L.mllm_hip_linear_q4k_q8k, (P, NULL, P, P, P, P, C.c_int(1), C.c_int64(64), P, C.c_int(1), C.c_int(64), C.c_int(256), NULL))):
        assert fn(*args) == ERR_DTYPE


def test_fused_run_entry_points_validate_before_launch(L):
    """The lazy window's fused launches: shapes the kernels do not cover are refused by the launch AND by the host-side *_supported check the adapter asks first."""
    from mllm_amd.ops import _RowFused
    f, sup = L.mllm_hip_row_fused_launch, L.mllm_hip_row_fused_supported
    assert f(NULL, NULL) == ERR_ARG and sup(NULL) == 0

    def args(K=512, nseg=1, mode=0, N=(64, 64, 64), W=True, y=True, mul=True, post=False):
        a = _RowFused()
        a.xa, a.K, a.nseg, a.mode = 0x1000, K, nseg, mode
        for i in range(min(nseg, 3)):
            a.seg[i].W = 0x1000 if W else 0
            a.seg[i].y = 0x1000 if y else 0
            a.seg[i].N = N[i]
            if post:
                a.seg[i].post_out = 0x1000      # ... without post_add
        if mode == 1 and mul:
            a.mul_out = 0x1000
        return a
    for bad in (args(K=300), args(K=0), args(K=11264), args(nseg=0), args(nseg=4), args(W=False), args(y=False), args(post=True), args(mode=2), args(mode=1, nseg=1),
                args(mode=1, nseg=2, N=(64, 32, 0)), args(mode=1, nseg=2, mul=False), args(nseg=2, N=(64, 0, 0))):
        assert sup(C.byref(bad)) == 0 and f(C.byref(bad), NULL) in (ERR_SHAPE, ERR_ARG)
    assert sup(C.byref(args())) == 1 and sup(C.byref(args(nseg=3))) == 1 and sup(C.byref(args(mode=1, nseg=2))) == 1
    st = L.mllm_hip_fa2_decode_step_supported
    assert st(C.c_int(10), C.c_int(12), C.c_int(2), C.c_int(128)) == 1 and st(C.c_int(0), C.c_int(4), C.c_int(4), C.c_int(64)) == 1
    assert st(C.c_int(10), C.c_int(12), C.c_int(5), C.c_int(128)) == 0          # heads not a multiple of the K/V heads
    assert st(C.c_int(10), C.c_int(12), C.c_int(2), C.c_int(80)) == 0           # head size the step kernel is not built for
    assert st(C.c_int(-1), C.c_int(12), C.c_int(2), C.c_int(128)) == 0
    assert st(C.c_int(200000), C.c_int(12), C.c_int(2), C.c_int(128)) == 0      # the score arrays of that many keys do not fit a CU's LDS
    step = L.mllm_hip_fa2_decode_step
    assert step(NULL, P, P, P, P, P, P, P, P, P, P, C.c_int(3), P, C.c_int(4), C.c_int(2), C.c_int(64), NULL) == ERR_ARG
    r2 = L.mllm_hip_rope2_store2
    assert r2(P, P, P, C.c_int(32), P, C.c_int(4), P, P, P, C.c_int(32), P, P, P, P, C.c_int(2), C.c_int(0), C.c_int(64), NULL) == OK        # no positions: nothing to do
    assert r2(P, P, P, C.c_int(32), P, C.c_int(4), P, P, P, C.c_int(32), P, P, P, P, C.c_int(2), C.c_int(1), C.c_int(63), NULL) == ERR_SHAPE
    assert r2(P, P, P, C.c_int(32), P, C.c_int(0), P, P, P, C.c_int(32), P, P, P, P, C.c_int(2), C.c_int(1), C.c_int(64), NULL) == ERR_SHAPE
    assert L.mllm_hip_silu_rows(P, P, C.c_int64(0), C.c_int(13), NULL) == OK
