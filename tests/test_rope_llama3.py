"""SURVEY §8(f) N4, one representative of the other model families' ops: RoPE with llama3 frequency scaling (rope_scaling {rope_type: llama3}: mllm/Layer.hpp:493-531 ->
_compute_llama3_theta, mllm/backends/cpu/op/CPURoPE.cpp:33-71).  Golden: the reference's own RoPE layer on 40 positions x 2 heads x 64 dims (tests/golden/rope3.npz,
oracle/make_golden.py --rope3); the 32 frequencies fall into all three regimes (kept / interpolated / divided by the factor)."""
import os

import numpy as np
import pytest

from mllm_amd import lib
from oracle import oracle as orc

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rope3.npz"))
P = G["params"]      # pose_type, theta, max_pos, heads, D, factor, low, high, original max_pos
ARGS = (float(P[1]), 64, 40, float(P[5]), float(P[6]), float(P[7]), float(P[8]))


def test_restatement_and_product_table_match_the_reference():
    s, c = orc.rope_table_hf_llama3(*ARGS)
    y = np.asarray(orc.rope_apply(G["x"], 40, 2, 64, s, c)).reshape(40, 128)
    assert np.array_equal(y, G["y"]), float(np.abs(y - G["y"]).max())
    s2, c2 = lib.rope_table_hf_llama3(*ARGS)
    assert np.array_equal(s, s2) and np.array_equal(c, c2)
    s0, _ = orc.rope_table_hf(ARGS[0], 64, 40)
    changed = (np.abs(s - s0).max(axis=0) > 0)[:32]
    assert changed.any() and not changed.all()      # some frequencies scaled, the high ones kept


@pytest.mark.gpu
def test_device_rotate_with_the_scaled_table():
    from mllm_amd import ops
    ops.require_gpu()
    s, c = lib.rope_table_hf_llama3(*ARGS)
    y = ops.rope_apply(G["x"], 40, 2, 64, s, c).cpu().numpy()
    assert np.array_equal(y, G["y"]), float(np.abs(y - G["y"]).max())
