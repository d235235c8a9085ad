"""the top-k sampling method's candidate set on device and its temperature softmax on the host
(mllm/Generate.cpp:45-90).  The reference's generate() ends in a std::random_device-seeded draw and exposes no intermediate, so
the checker here is the restatement in oracle/oracle.py (parity unpinned for this row; the arithmetic is 5 floats and libm exp)."""
import numpy as np
import pytest

from mllm_amd import lib
from oracle import oracle as orc


def test_topk_probs_host_matches_restatement():
    r = np.random.default_rng(3)
    import ctypes as C
    for k, temp in ((5, 0.7), (1, 0.7), (8, 1.3), (5, 0.05)):
        logits = r.standard_normal(4096).astype(np.float32) * 4
        idx, top, want = orc.topk_sampling_probs(logits, k, temp)
        got = np.empty(k, dtype=np.float32)
        lib.check(lib.load().mllm_hip_topk_probs_host(lib.vp(np.ascontiguousarray(top)), C.c_int(k), C.c_float(temp), lib.vp(got)))
        assert np.array_equal(got, want), (k, temp, got, want)
        assert abs(float(got.sum()) - 1.0) < 1e-5


@pytest.mark.gpu
def test_topk_on_device_matches_partial_sort_order():
    from mllm_amd import ops
    ops.require_gpu()
    r = np.random.default_rng(4)
    for n, k in ((151936, 5), (2048, 1), (1000, 64), (7, 7)):
        x = r.standard_normal(n).astype(np.float32)
        x[r.integers(0, n, size=max(1, n // 50))] = x.max()          # ties at the top: ascending index among equal logits
        val, idx = ops.topk(x, k)
        want_idx, want_val, _ = orc.topk_sampling_probs(x, k, 0.7)
        assert np.array_equal(idx, want_idx) and np.array_equal(val, want_val)
    # end to end: candidate probabilities of the method at the reference's defaults (k = 5, temperature 0.7)
    x = r.standard_normal(151936).astype(np.float32) * 3
    val, idx = ops.topk(x, 5)
    assert np.array_equal(ops.topk_probs(val, 0.7), orc.topk_sampling_probs(x, 5, 0.7)[2])
