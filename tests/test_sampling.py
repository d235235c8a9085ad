"""the sampled generation methods (mllm/Generate.cpp:45-142).  Candidate selection runs on the device (top-k: mllm_hip_topk; top-p: one
descending radix sort, mllm_hip_sort_desc), the temperature softmax over the candidates and the draw on the host as in the reference.

Pinned by tests/golden/sampling.npz: the candidate ids and the pre-draw probabilities of the reference's own compiled methods on fixed rows
(oracle/ref_drivers/ref_sampling.cpp interposes only the random draw, _sample_element, which is std::random_device-seeded and not comparable)."""
import ctypes as C
import os

import numpy as np
import pytest

from mllm_amd import lib
from oracle import oracle as orc

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampling.npz"))
K, P, TEMP = int(G["k"]), float(G["p"]), float(G["temp"])


def _probs_host(vals, temp):
    vals = np.ascontiguousarray(vals, dtype=np.float32)
    got = np.empty(vals.size, dtype=np.float32)
    lib.check(lib.load().mllm_hip_topk_probs_host(lib.vp(vals), C.c_int(vals.size), C.c_float(temp), lib.vp(got)))
    return got


def test_restatement_and_host_softmax_match_the_reference_topk():
    for name, rows in (("l", G["logits"]), ("s", G["probs"])):
        for i, row in enumerate(rows):
            idx, top, want = orc.topk_sampling_probs(row, K, TEMP)
            assert np.array_equal(idx.astype(np.uint32), G[f"topk_{name}{i}_idx"])
            assert np.array_equal(want, G[f"topk_{name}{i}_prob"])
            assert np.array_equal(_probs_host(top, TEMP), G[f"topk_{name}{i}_prob"])         # the product's host function, same bits


def test_restatement_and_host_softmax_match_the_reference_topp():
    sizes = []
    for i, row in enumerate(G["probs"]):
        idx, vals, want = orc.topp_sampling_probs(row, P, TEMP)
        sizes.append(idx.size)
        assert np.array_equal(idx.astype(np.uint32), G[f"topp_{i}_idx"])
        assert np.array_equal(want, G[f"topp_{i}_prob"])
        if idx.size > 1:
            assert np.array_equal(_probs_host(vals, TEMP), G[f"topp_{i}_prob"])
    assert min(sizes) >= 2 and max(sizes) > 1000      # nuclei from a handful to over a thousand candidates


def test_inverse_cdf_draw():
    so = lib.load()
    p = np.array([0.5, 0.25, 0.125, 0.125], dtype=np.float32)
    pick = lambda u: so.mllm_hip_sample_index_host(lib.vp(p), C.c_int(4), C.c_float(u))
    assert [pick(u) for u in (0.0, 0.49, 0.5, 0.74, 0.75, 0.874, 0.875, 0.999)] == [0, 0, 1, 1, 2, 2, 3, 3]
    u = np.random.default_rng(1).random(20000).astype(np.float32)
    freq = np.bincount([pick(float(v)) for v in u], minlength=4) / u.size
    assert np.abs(freq - p).max() < 0.01


@pytest.mark.gpu
def test_device_candidate_selection_matches_the_reference():
    import torch
    from mllm_amd import ops
    ops.require_gpu()
    so = lib.load()
    for i, row in enumerate(G["logits"]):
        val, idx = ops.topk(row, K)
        assert np.array_equal(idx.astype(np.uint32), G[f"topk_l{i}_idx"])
        assert np.array_equal(ops.topk_probs(val, TEMP), G[f"topk_l{i}_prob"])
    n = G["probs"].shape[1]
    ws_bytes = so.mllm_hip_sort_desc_workspace_bytes(C.c_int(n))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    for i, row in enumerate(G["probs"]):
        x = torch.from_numpy(row).cuda()
        sv = torch.empty(n, dtype=torch.float32, device="cuda")
        si = torch.empty(n, dtype=torch.int32, device="cuda")
        lib.check(so.mllm_hip_sort_desc(lib.vp(x), C.c_int(n), lib.vp(sv), lib.vp(si), lib.vp(ws), C.c_size_t(ws_bytes), None), "sort_desc")
        torch.cuda.synchronize()
        want = G[f"topp_{i}_idx"]
        assert np.array_equal(si.cpu().numpy()[:want.size].astype(np.uint32), want)
        v = sv.cpu().numpy()
        assert np.all(v[:-1] >= v[1:]) and np.array_equal(np.sort(row)[::-1], v)
    # ties: equal keys keep ascending index order (std::sort leaves it unspecified; std::partial_sort of top-k the same way)
    x = np.zeros(1000, dtype=np.float32); x[[5, 700, 42]] = 1.0
    sv = torch.empty(1000, dtype=torch.float32, device="cuda"); si = torch.empty(1000, dtype=torch.int32, device="cuda")
    ws = torch.empty(so.mllm_hip_sort_desc_workspace_bytes(C.c_int(1000)), dtype=torch.uint8, device="cuda")
    lib.check(so.mllm_hip_sort_desc(lib.vp(torch.from_numpy(x).cuda()), C.c_int(1000), lib.vp(sv), lib.vp(si), lib.vp(ws), C.c_size_t(ws.numel()), None))
    torch.cuda.synchronize()
    assert si.cpu().numpy()[:6].tolist() == [5, 42, 700, 0, 1, 2]


@pytest.mark.gpu
def test_topk_on_device_matches_partial_sort_order():
    from mllm_amd import ops
    ops.require_gpu()
    r = np.random.default_rng(4)
    # rows of 16384 values and more take two stages (128 slices leave their k best, one workgroup picks among those): ties that span slices, k = 64, a row length that
    # leaves the last slices short or empty, -inf entries
    for n, k in ((151936, 5), (2048, 1), (1000, 64), (7, 7), (151936, 64), (16384, 50), (16390, 64), (32000, 1)):
        x = r.standard_normal(n).astype(np.float32)
        x[r.integers(0, n, size=max(1, n // 50))] = x.max()          # ties at the top: ascending index among equal logits
        val, idx = ops.topk(x, k)
        want_idx, want_val, _ = orc.topk_sampling_probs(x, k, 0.7)
        assert np.array_equal(idx, want_idx) and np.array_equal(val, want_val), (n, k)
    x = np.full(20000, -np.inf, dtype=np.float32)
    x[[19999, 3, 12000]] = [1.0, 1.0, -2.0]
    val, idx = ops.topk(x, 5)
    assert idx.tolist() == [3, 19999, 12000, 0, 1] and val[:3].tolist() == [1.0, 1.0, -2.0] and np.isneginf(val[3:]).all()


@pytest.mark.gpu
def test_generate_sampled_on_the_engine():
    """Module::generate's method switch on the resident engine (tiny Qwen2-VL): with u = 0 every draw takes the first candidate = the largest score, so top-k and
    top-p must reproduce the greedy ids; with other draws the ids stay inside the candidate set of each step; eos stops the loop."""
    from mllm_amd import synth
    from mllm_amd import synthfile as weights
    cfg = synth.qwen2vl_tiny()
    path = weights.qwen2vl_file(cfg, cache_dir=os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache"))
    pix, grid, ids = synth.qwen2vl_inputs(cfg, (8, 8), 6)
    m = lib.Qwen2VL(cfg, path)
    steps = 12
    tok, _, _ = m.prefill(ids, pix, grid)
    greedy, _ = m.generate(tok, steps)
    runs = {}
    for method in (0, 1, 2):
        m.clear_kvcache()
        tok2, _, _ = m.prefill(ids, pix, grid)
        assert tok2 == tok
        runs[method], _ = m.generate_sampled(tok, steps, method, np.zeros(steps, dtype=np.float32))
        assert runs[method].tolist() == greedy.tolist(), method
    m.clear_kvcache()
    tok2, _, _ = m.prefill(ids, pix, grid)
    u = np.random.default_rng(5).random(steps).astype(np.float32)
    got, _ = m.generate_sampled(tok, steps, 1, u)
    assert got.size == steps and got.min() >= 0 and got.max() < cfg.vocab
    m.clear_kvcache()
    m.prefill(ids, pix, grid)
    cut, _ = m.generate_sampled(tok, steps, 0, np.zeros(steps, dtype=np.float32), eos=int(greedy[3]))
    assert cut.tolist() == greedy[:list(greedy).index(greedy[3]) + 1].tolist()
    # a nucleus mass that keeps no candidate (p <= 0, NaN) is refused before anything runs, and the cache is where it was
    for bad in (0.0, -0.5, float("nan")):
        with pytest.raises(lib.MllmHipError):
            m.generate_sampled(tok, 2, 2, np.zeros(2, dtype=np.float32), top_p=bad)
    again, _ = m.generate_sampled(int(cut[-1]), 3, 0, np.zeros(3, dtype=np.float32))
    assert again.size == 3
    # the tower blocks' on-disk rows were released once packed (a tower pass never has fewer than 16 rows); the towers still give the reference's rows (test_gpu_e2e)
    st = m.memory_stats()
    assert st["released_bytes"] > 0 and st["resident_bytes"] > st["released_bytes"], st
    m.close()
