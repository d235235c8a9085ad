"""Qwen2-VL image preprocessing (decoded RGB -> x/255 -> smart_resize -> stb cubic B-spline resample -> normalise -> frame doubled -> convertPatches).

Fixtures: tests/golden/preprocess.npz, the output of the reference's own Qwen2VLImageProcessor::preprocess_images on three synthetic BMPs (oracle/ref_drivers/ref_preprocess.cpp,
oracle/make_golden.py --preprocess): a slight downsample, an upsample through smart_resize's min_pixels branch, and a same-size image (the B-spline still smooths).
The resize is vendored third-party code (stb_image_resize2) whose SIMD summation order is not reproduced: restatement and device path are held to the reference within
TOL = 2e-5 absolute (values of magnitude <= 2.7; observed <= 5e-6), and to EACH OTHER bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as orc

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preprocess.npz"))
TOL = 2e-5


def test_restatement_matches_the_reference_within_tolerance():
    for i in range(3):
        p, grid = orc.qwen2vl_preprocess(G[f"rgb{i}"])
        assert np.array_equal(grid, G[f"grid{i}"])
        assert p.shape == G[f"patches{i}"].shape
        assert float(np.abs(p - G[f"patches{i}"]).max()) <= TOL


def test_smart_resize_branches():
    assert orc.smart_resize(60, 90) == (56, 84)
    assert orc.smart_resize(51, 37) == (84, 56)            # below min_pixels: scaled up by beta, ceil to the factor
    assert orc.smart_resize(112, 112) == (112, 112)
    assert orc.smart_resize(448, 448) == (448, 448)
    h, w = orc.smart_resize(5000, 4000)                    # above max_pixels (12,845,056): scaled down, floor to the factor
    assert h % 28 == 0 and w % 28 == 0 and h * w <= 16384 * 28 * 28 and h * w > 0.97 * 16384 * 28 * 28
    with pytest.raises(ValueError):
        orc.smart_resize(10, 4000)


def test_patchify_is_convert_patches():
    """The vectorised index map against the loop of convertPatches (processing_qwen2_vl.hpp:142-172), literally."""
    r = np.random.default_rng(2)
    chw = r.standard_normal((3, 56, 84)).astype(np.float32)
    got, grid = orc.qwen2vl_patchify(chw)
    ps, ms, tps = 14, 2, 2
    gh, gw = 56 // ps, 84 // ps
    ghm, gwm = gh // ms, gw // ms
    want = np.zeros((gh * gw, 3 * tps * ps * ps), dtype=np.float32)
    for i in range(gh * gw):
        rem = i % (ghm * gwm * ms * ms)
        d1, rem = divmod(rem, gwm * ms * ms)
        d2, rem = divmod(rem, ms * ms)
        d3, d4 = divmod(rem, ms)
        for j in range(want.shape[1]):
            d5, rj = divmod(j, tps * ps * ps)
            d6, rj = divmod(rj, ps * ps)
            d7, d8 = divmod(rj, ps)
            want[i, j] = chw[d5, (d1 * ms + d3) * ps + d7, (d2 * ms + d4) * ps + d8]      # frame d6 of the doubled image = the image
    assert grid.tolist() == [1, gh, gw] and np.array_equal(got, want)


@pytest.mark.gpu
def test_device_preprocess_matches_reference_and_restatement():
    import torch
    from mllm_amd import lib, ops
    ops.require_gpu()
    so = lib.load()
    for i in range(3):
        rgb = np.ascontiguousarray(G[f"rgb{i}"])
        H, W, _ = rgb.shape
        grid = np.zeros(3, dtype=np.int32)
        lib.check(so.mllm_hip_qwen2vl_preprocess_shape(C.c_int(H), C.c_int(W), C.c_int(4 * 28 * 28), C.c_int(16384 * 28 * 28), lib.vp(grid)), "preprocess_shape")
        assert np.array_equal(grid, G[f"grid{i}"])
        out = torch.empty((int(grid.prod()), 1176), dtype=torch.float32, device="cuda")
        g2 = np.zeros(3, dtype=np.int32)
        lib.check(so.mllm_hip_qwen2vl_preprocess(lib.vp(rgb), C.c_int(H), C.c_int(W), C.c_int(4 * 28 * 28), C.c_int(16384 * 28 * 28), lib.vp(out), lib.vp(g2), None), "preprocess")
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert np.array_equal(g2, grid)
        assert float(np.abs(got - G[f"patches{i}"]).max()) <= TOL
        want, _ = orc.qwen2vl_preprocess(rgb)
        assert np.array_equal(got, want), float(np.abs(got - want).max())      # same taps, same order of operations as the restatement


@pytest.mark.gpu
def test_device_pixels_feed_the_tower():
    """The preprocessed patches stay on the device and go straight into the engine's prefill (tiny Qwen2-VL): same logits as the same patches handed over from the host."""
    import torch
    from mllm_amd import lib, synth
    from mllm_amd import synthfile as weights
    cfg = synth.qwen2vl_tiny()
    path = weights.qwen2vl_file(cfg, cache_dir=os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache"))
    rgb = np.ascontiguousarray(G["rgb2"])       # 112 x 112 -> grid 8 x 8 -> 16 visual tokens
    grid = np.zeros(3, dtype=np.int32)
    out = torch.empty((64, 1176), dtype=torch.float32, device="cuda")
    so = lib.load()
    lib.check(so.mllm_hip_qwen2vl_preprocess(lib.vp(rgb), C.c_int(112), C.c_int(112), C.c_int(4 * 28 * 28), C.c_int(16384 * 28 * 28), lib.vp(out), lib.vp(grid), None), "preprocess")
    torch.cuda.synchronize()
    n_tok = 64 // 4
    ids = np.concatenate([[cfg.vision_start_token_id], np.full(n_tok, cfg.image_token_id), [cfg.vision_end_token_id], [5, 6, 7]]).astype(np.int32)
    m = lib.Qwen2VL(cfg, path)
    tok_h, lg_h, _ = m.prefill(ids, out.cpu().numpy(), grid)
    m.clear_kvcache()
    tok = C.c_int32()
    lg_d = np.empty(cfg.vocab, dtype=np.float32)
    lib.check(so.mllm_hip_model_prefill(m._h, lib.vp(ids), C.c_int(ids.size), lib.vp(out), lib.vp(grid), None, C.c_int(0), lib.vp(lg_d), C.byref(tok), None), "prefill(device pixels)")
    assert tok.value == tok_h and np.array_equal(lg_d, lg_h)
    m.close()
