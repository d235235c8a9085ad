#!/usr/bin/env python3
"""oracle/make_golden.py -- TEST INFRASTRUCTURE.  Generates the committed golden vectors under tests/golden/ by running
the compiled REFERENCE itself (oracle/_ref/ref_ops, ref_qwen2vl, ref_llm, ref_vit, ref_llava_parts, quantize -- built by
oracle/Makefile.ref from /root/reference).  Runs only in the development container; the GPU box sees only the resulting small .npz files.

    make -f oracle/Makefile.ref -j8 && python oracle/make_golden.py [--full]          # op goldens + tiny Qwen2-VL (+ the 2 B runs)
    python oracle/make_golden.py --all [--full]                                          # every golden that depends on a synthetic model file

Op-level cases store their inputs (fp32), the raw weight bytes the reference consumed (written by the reference's own `quantize` tool from seeded fp32 arrays)
and the reference's outputs, so the tests replay them without the reference.  Model-level cases run the reference on the synthetic `.mllm` files of
mllm_amd/synthfile.py (Q4_K / Q4_0 tensors drawn directly in the quantised domain: the reference and the HIP path read the same bytes; no quantiser in between).
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mllm_amd import mllmfile as mf, synth  # noqa: E402
from mllm_amd import synthfile as weights  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")


def run_ops(case, weights, calls, p=(), threads=4):
    """calls: list of list of (array, shape-tuple). Returns list of outputs (flat fp32)."""
    with tempfile.TemporaryDirectory() as td:
        args = [os.path.join(REF, "ref_ops"), f"case={case}", f"weights={weights}", f"out={td}", f"threads={threads}"]
        if len(p):
            args.append("p=" + ",".join(repr(float(v)) if isinstance(v, float) else str(v) for v in p))
        for ci, call in enumerate(calls):
            specs = []
            for ii, (arr, shp) in enumerate(call):
                fn = os.path.join(td, f"in{ci}_{ii}.f32")
                np.ascontiguousarray(arr, dtype=np.float32).tofile(fn)
                specs.append(fn + ":" + ",".join(str(int(s)) for s in shp))
            args.append("call=" + "+".join(specs))
        out = subprocess.run(args, check=True, capture_output=True, text=True)
        res = [np.fromfile(os.path.join(td, f"out{ci}.f32"), dtype=np.float32) for ci in range(len(calls))]
        return res, out.stdout


def quantize_file(specs_arrays, target="Q4_K"):
    """specs_arrays: list of (name, fp32 array). Writes fp32 .mllm, runs the reference `quantize`, returns MllmFile path."""
    td = tempfile.mkdtemp()
    src, dst = os.path.join(td, "w_f32.mllm"), os.path.join(td, "w_q.mllm")
    mf.write_mllm(src, [(n, mf.F32, np.ascontiguousarray(a, dtype=np.float32)) for n, a in specs_arrays])
    if target == "F32":
        return src
    subprocess.run([os.path.join(REF, "quantize"), src, dst, target], check=True, capture_output=True)
    return dst


def raw(path, name):
    f = mf.MllmFile(path)
    r = np.array(f.raw(name))
    dt = f.dtype(name)
    f.close()
    return r, dt


def ops_golden():
    rng = np.random.default_rng(1234)
    G = {}

    def rn(*s, scale=1.0):
        return (rng.standard_normal(s) * scale).astype(np.float32)

    # ---- A1/A2/A5: Linear Q4_K (+bias), M = 5 and M = 1; K = 512, N = 96 ------------------------------------------------
    W, b = rn(96, 512, scale=0.05), rn(96, scale=0.1)
    path = quantize_file([("lin.weight", W), ("lin.bias", b)])
    x5, x1 = rn(5, 512), rn(1, 512)
    (y5, y1), _ = run_ops("linear", path, [[(x5, (1, 1, 5, 512))], [(x1, (1, 1, 1, 512))]], p=(512, 96, 1))
    wq, dt = raw(path, "lin.weight")
    assert dt == mf.Q4_K
    G.update(lin_w=wq, lin_b=b, lin_x5=x5, lin_y5=y5.reshape(5, 96), lin_x1=x1, lin_y1=y1.reshape(1, 96), lin_w_f32=W)
    # Linear with fp32 weights (TinyLlama config / patch-embed path), no bias, K = 64 (not a multiple of 32 lanes*4)
    Wf = rn(24, 72, scale=0.1)
    pathf = quantize_file([("lin.weight", Wf)], target="F32")
    xf = rn(3, 72)
    (yf,), _ = run_ops("linear", pathf, [[(xf, (1, 1, 3, 72))]], p=(72, 24, 0))
    G.update(linf_w=Wf, linf_x=xf, linf_y=yf.reshape(3, 24))

    # ---- A6/A7/A8: tied lm_head through Q4_0 x Q8_0 (Tensor::mm on the embedding Parameter) and the embedding gather ---
    E = rn(160, 512, scale=0.05)
    pathe = quantize_file([("model.embed_tokens.weight", E)])
    eq, dt = raw(pathe, "model.embed_tokens.weight")
    assert dt == mf.Q4_0
    xh = rn(2, 512)
    (lg,), _ = run_ops("mm_tied", pathe, [[(xh, (1, 1, 2, 512))]], p=(160, 512))
    ids = np.array([3, 159, 0, 77, 3], dtype=np.float32)
    (emb,), _ = run_ops("embedding", pathe, [[(ids, (1, 1, 5, 1))]], p=(160, 512))
    G.update(emb_w=eq, mm_x=xh, mm_y=lg.reshape(2, 160), emb_ids=ids, emb_y=emb.reshape(5, 512))

    # ---- A9 / A18 norms ----------------------------------------------------------------------------------------------------
    nw, nb_ = (1.0 + rn(256, scale=0.1)), rn(256, scale=0.1)
    pn = quantize_file([("norm.weight", nw), ("ln.weight", nw), ("ln.bias", nb_)], target="F32")
    xn = rn(3, 256, scale=2.0)
    (yr,), _ = run_ops("rmsnorm", pn, [[(xn, (1, 1, 3, 256))]], p=(256, 1e-6))
    (yl,), _ = run_ops("layernorm", pn, [[(xn, (1, 1, 3, 256))]], p=(256, 1e-6))
    G.update(norm_w=nw, norm_b=nb_, norm_x=xn, rms_y=yr.reshape(3, 256), ln_y=yl.reshape(3, 256))

    # ---- A14 / A18 activations, A15 softmax --------------------------------------------------------------------------------
    xa = np.concatenate([rn(3, 264, scale=3.0).ravel(), np.array([0.0, -0.0, 20.0, -20.0, 90.0, -90.0, 1e-8, 5.5], dtype=np.float32)]).reshape(1, -1)
    n_act = xa.size
    for k in ("silu", "gelu", "quickgelu"):
        (ya,), _ = run_ops(k, pn, [[(xa, (1, 1, 1, n_act))]])
        G[k + "_y"] = ya
    G["act_x"] = xa.ravel()
    xs = rn(2, 3, 24, scale=2.0)
    (ys,), _ = run_ops("softmax", pn, [[(xs, (1, 2, 3, 24))]], p=(0,))
    G.update(sm_x=xs, sm_y=ys.reshape(2, 3, 24))

    # ---- A16 / A17 patch-embed convolutions ---------------------------------------------------------------------------------
    Wc = rn(16, 1176, scale=0.05)
    pc = quantize_file([("proj.weight", Wc)], target="F32")
    px = rn(6, 1176)
    (yc,), _ = run_ops("conv3d", pc, [[(px, (6, 3, 2, 14, 14))]], p=(16,))
    G.update(conv3_w=Wc, conv3_x=px, conv3_y=yc.reshape(6, 16))
    W2, b2 = rn(8, 3, 4, 4, scale=0.2), rn(8, scale=0.1)
    pc2 = quantize_file([("proj.weight", W2), ("proj.bias", b2)], target="F32")
    img = rn(8, 3, 12)  # [H][C][W]
    (y2,), log = run_ops("conv2d", pc2, [[(img, (1, 8, 3, 12))]], p=(8, 4, 1))
    G.update(conv2_w=W2, conv2_b=b2, conv2_x=img, conv2_y=y2, conv2_log=np.frombuffer(log.encode(), dtype=np.uint8))

    # ---- A10 / A11 / A19 rotary ------------------------------------------------------------------------------------------
    q = rn(5, 2 * 128)
    pos = np.array([[0, 1, 1, 1, 2], [0, 1, 1, 2, 2], [0, 1, 2, 1, 2]], dtype=np.float32) * np.float32(3)
    pos[:, 4] = 7
    (ym,), _ = run_ops("mrope", pn, [[(q, (1, 1, 5, 256)), (pos, (3, 1, 1, 5))]], p=(1000000.0, 32768, 2, 128))
    G.update(mrope_x=q, mrope_pos=pos, mrope_y=ym.reshape(5, 256))
    q2 = rn(5, 2 * 64)
    (yh,), _ = run_ops("rope", pn, [[(q2, (1, 1, 5, 128))]], p=(4, 10000.0, 64, 2, 64))  # HFHUBROPE = 4
    G.update(rope_x=q2, rope_y=yh.reshape(5, 128))
    qv = rn(16, 2 * 16)
    grid = np.array([1, 4, 4], dtype=np.float32)
    (yv,), _ = run_ops("vrope", pn, [[(qv, (1, 1, 16, 32)), (grid, (1, 1, 1, 3))]], p=(8, 2, 2, 16))
    G.update(vrope_x=qv, vrope_y=yv.reshape(16, 32))

    # ---- A13 attention, fp32 K/V (vision path): non-causal 40x40 D=16 H=2; causal with GQA (Hq=4, Hkv=2), Sq = Sk = 12 ----
    qa, ka, va = rn(40, 2 * 16), rn(40, 2 * 16), rn(40, 2 * 16)
    (oa,), _ = run_ops("fa2", pn, [[(qa, (1, 1, 40, 32)), (ka, (1, 1, 40, 32)), (va, (1, 1, 40, 32))]], p=(2, 2, 16, 0), threads=2)
    G.update(fa_q=qa, fa_k=ka, fa_v=va, fa_o=oa.reshape(40, 32))
    qb, kb, vb = rn(12, 4 * 16), rn(12, 2 * 16), rn(12, 2 * 16)
    (ob,), _ = run_ops("fa2", pn, [[(qb, (1, 1, 12, 64)), (kb, (1, 1, 12, 32)), (vb, (1, 1, 12, 32))]], p=(4, 2, 16, 1), threads=2)
    G.update(fac_q=qb, fac_k=kb, fac_v=vb, fac_o=ob.reshape(12, 64))

    # ---- A21 blocks: QWen2Attention (A1+A11+A12+A13: prefill 8 tokens, then one decode token), QWen2MLP ---------------------
    H, I, heads, kvh = 256, 512, 2, 1
    names = {}
    pfx = "model.layers.0."
    for n, s, k in synth.qwen2vl_tensors(synth.Qwen2VLConfig(hidden=H, inter=I, layers=1, heads=heads, kv_heads=kvh, vocab=64), vision=False):
        if n.startswith(pfx + "self_attn") or n.startswith(pfx + "mlp"):
            names[n] = synth.tensor_f32(n, s, k) * (np.float32(2.5) if k == "w" else np.float32(1))
    pa = quantize_file(list(names.items()))
    xa8, xa1 = rn(8, H), rn(1, H)
    pos8 = np.tile(np.arange(8, dtype=np.float32), (3, 1))
    pos1 = np.full((3, 1), 8, dtype=np.float32)
    (a8, a1), _ = run_ops("attn", pa, [[(xa8, (1, 1, 8, H)), (pos8, (3, 1, 1, 8))], [(xa1, (1, 1, 1, H)), (pos1, (3, 1, 1, 1))]], p=(H, I, heads, kvh, 32), threads=1)
    (ml,), _ = run_ops("mlp", pa, [[(xa8, (1, 1, 8, H))]], p=(H, I, heads, kvh, 32))
    f = mf.MllmFile(pa)
    for n in f.names():
        if n.startswith(pfx):
            G["blk_" + n[len(pfx):].replace(".", "_")] = np.array(f.raw(n))
    f.close()
    G.update(blk_x8=xa8, blk_x1=xa1, blk_attn8=a8.reshape(8, H), blk_attn1=a1.reshape(1, H), blk_mlp8=ml.reshape(8, H))
    np.savez_compressed(os.path.join(GOLD, "ops.npz"), **G)
    print("ops.npz:", len(G), "arrays,", os.path.getsize(os.path.join(GOLD, "ops.npz")) // 1024, "KiB")


def e2e_tiny():
    c = synth.qwen2vl_tiny()
    td = tempfile.mkdtemp()
    dst = weights.qwen2vl_file(c, cache_dir=td)
    pix, grid, ids = synth.qwen2vl_inputs(c, (8, 8), 6)
    pix.tofile(os.path.join(td, "pix.f32"))
    ids.tofile(os.path.join(td, "ids.i32"))
    cfg = f"{c.hidden},{c.inter},{c.layers},{c.heads},{c.kv_heads},{c.vocab},{c.v_dim},{c.cache_limit},{c.image_token_id},{c.vision_start_token_id},{c.vision_end_token_id},{c.video_token_id}"
    steps = 24
    subprocess.run([os.path.join(REF, "ref_qwen2vl"), "--model", dst, "--ids", os.path.join(td, "ids.i32"), "--pix", os.path.join(td, "pix.f32"),
                    "--grid", "1,8,8", "--steps", str(steps), "--threads", "4", "--out", td, "--cfg", cfg, "--dump-every", "1"], check=True, capture_output=True)
    toks = np.fromfile(os.path.join(td, "tokens.i32"), dtype=np.int32)
    logits = np.stack([np.fromfile(os.path.join(td, f"logits_{s}.f32"), dtype=np.float32) for s in range(steps)])
    # vision tower alone (image_embeds) for stage-wise debugging
    (emb,), _ = run_ops("vision", dst, [[(pix, (64, 3, 2, 14, 14)), (grid.astype(np.float32), (1, 1, 1, 3))]], p=(c.hidden, c.v_dim))
    # text-only prompt (no image): exercises the has_img == false path
    ids_t = np.random.default_rng(5).integers(0, 2000, size=9).astype(np.int32)
    ids_t.tofile(os.path.join(td, "ids_t.i32"))
    td2 = tempfile.mkdtemp()
    subprocess.run([os.path.join(REF, "ref_qwen2vl"), "--model", dst, "--ids", os.path.join(td, "ids_t.i32"), "--steps", "6", "--threads", "4",
                    "--out", td2, "--cfg", cfg, "--dump-every", "1"], check=True, capture_output=True)
    toks_t = np.fromfile(os.path.join(td2, "tokens.i32"), dtype=np.int32)
    logits_t = np.stack([np.fromfile(os.path.join(td2, f"logits_{s}.f32"), dtype=np.float32) for s in range(6)])
    # the untied form (tie_embedding_words = false: a separate Q4_K lm_head Linear, modeling_qwen2_vl.hpp:375-401; what demo_qwen2_vl's larger presets use): image + text, 8 steps
    cu = synth.qwen2vl_tiny()
    cu.tie_embedding = False
    dst_u = weights.qwen2vl_file(cu, cache_dir=td)
    td3 = tempfile.mkdtemp()
    subprocess.run([os.path.join(REF, "ref_qwen2vl"), "--model", dst_u, "--ids", os.path.join(td, "ids.i32"), "--pix", os.path.join(td, "pix.f32"), "--grid", "1,8,8", "--steps", "8",
                    "--threads", "4", "--out", td3, "--cfg", cfg + ",0", "--dump-every", "1"], check=True, capture_output=True)
    toks_u = np.fromfile(os.path.join(td3, "tokens.i32"), dtype=np.int32)
    logits_u = np.stack([np.fromfile(os.path.join(td3, f"logits_{s}.f32"), dtype=np.float32) for s in range(8)])
    np.savez_compressed(os.path.join(GOLD, "qwen2vl_tiny.npz"), ids=ids, grid=grid, tokens=toks, logits=logits, image_embeds=emb.reshape(-1, c.hidden),
                        ids_text=ids_t, tokens_text=toks_t, logits_text=logits_t, tokens_untied=toks_u, logits_untied=logits_u)
    print("qwen2vl_tiny.npz tokens", toks.tolist())


def e2e_ragged():
    """Non-square grids whose patch / token counts are not multiples of the kernels' tiles (the reference's own model on the toy file): grid 6 x 10 (60 patches -> 15 image
    tokens, 24 prompt tokens), 8 x 12 (96 -> 24, 33 tokens), 4 x 18 (72 -> 18, 21 tokens); 6 greedy steps each, every logit, plus the tower's image_embeds."""
    c = synth.qwen2vl_tiny()
    td = tempfile.mkdtemp()
    dst = weights.qwen2vl_file(c, cache_dir=td)
    out = {}
    for k, (gh, gw, nt) in enumerate([(6, 10, 7), (8, 12, 7), (4, 18, 1)]):
        pix, grid, ids = synth.qwen2vl_inputs(c, (gh, gw), nt)
        d = tempfile.mkdtemp()
        pix.tofile(os.path.join(d, "pix.f32"))
        ids.tofile(os.path.join(d, "ids.i32"))
        subprocess.run([os.path.join(REF, "ref_qwen2vl"), "--model", dst, "--ids", os.path.join(d, "ids.i32"), "--pix", os.path.join(d, "pix.f32"), "--grid", f"1,{gh},{gw}",
                        "--steps", "6", "--threads", "4", "--out", d, "--cfg", _cfg_q2vl(c), "--dump-every", "1"], check=True, capture_output=True)
        (emb,), _ = run_ops("vision", dst, [[(pix, (gh * gw, 3, 2, 14, 14)), (grid.astype(np.float32), (1, 1, 1, 3))]], p=(c.hidden, c.v_dim))
        out[f"grid{k}"] = grid
        out[f"ntext{k}"] = np.array(nt)
        out[f"tokens{k}"] = np.fromfile(os.path.join(d, "tokens.i32"), dtype=np.int32)
        out[f"logits{k}"] = np.stack([np.fromfile(os.path.join(d, f"logits_{s}.f32"), dtype=np.float32) for s in range(6)])
        out[f"image_embeds{k}"] = emb.reshape(-1, c.hidden)
        print("ragged", grid.tolist(), out[f"tokens{k}"].tolist(), out[f"image_embeds{k}"].shape)
    # the shortest prompt the reference's model takes: two tokens (its get_position_ids reads S == 1 as a decode step and faults on an empty cache)
    d = tempfile.mkdtemp()
    np.array([17, 23], dtype=np.int32).tofile(os.path.join(d, "ids.i32"))
    subprocess.run([os.path.join(REF, "ref_qwen2vl"), "--model", dst, "--ids", os.path.join(d, "ids.i32"), "--steps", "5", "--threads", "4", "--out", d, "--cfg", _cfg_q2vl(c),
                    "--dump-every", "1"], check=True, capture_output=True)
    out["one_tokens"] = np.fromfile(os.path.join(d, "tokens.i32"), dtype=np.int32)
    out["one_logits"] = np.stack([np.fromfile(os.path.join(d, f"logits_{s}.f32"), dtype=np.float32) for s in range(5)])
    np.savez_compressed(os.path.join(GOLD, "qwen2vl_ragged.npz"), **out)


CACHE = os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache")


def _cfg_q2vl(c):
    return (f"{c.hidden},{c.inter},{c.layers},{c.heads},{c.kv_heads},{c.vocab},{c.v_dim},{c.cache_limit},{c.image_token_id},{c.vision_start_token_id},"
            f"{c.vision_end_token_id},{c.video_token_id}")


def e2e_full():
    """The full-size (Qwen2-VL-2B shaped) reference runs on the synthetic file: (1) 448 x 448 image + 24 tokens, 65 greedy steps: ids, and of steps 0 / 16 / 32 / 48 / 64
    the top-64 logits and every 97th; (2) a 40-token text prompt, 33 steps, steps 0 / 8 / .. / 32 likewise; (3) the vision tower's image_embeds, all 256 x 1536."""
    c = synth.qwen2vl_2b()
    path = weights.qwen2vl_file(c, cache_dir=CACHE)
    pix, grid, ids = synth.qwen2vl_inputs(c, (32, 32), 24)

    def run(ids_, steps, every, with_img):
        td = tempfile.mkdtemp(dir=os.environ.get("MLLM_GOLD_TMP"))
        ids_.astype(np.int32).tofile(os.path.join(td, "ids.i32"))
        cmd = [os.path.join(REF, "ref_qwen2vl"), "--model", path, "--ids", os.path.join(td, "ids.i32"), "--steps", str(steps), "--threads", "8", "--out", td, "--cfg", _cfg_q2vl(c),
               "--dump-every", str(every)]
        if with_img:
            pix.tofile(os.path.join(td, "pix.f32"))
            cmd += ["--pix", os.path.join(td, "pix.f32"), "--grid", "1,32,32"]
        out = subprocess.run(cmd, check=True, capture_output=True, text=True)
        toks = np.fromfile(os.path.join(td, "tokens.i32"), dtype=np.int32)
        dumped = sorted(int(f[7:-4]) for f in os.listdir(td) if f.startswith("logits_") and int(f[7:-4]) % every == 0)
        ti, tv, st = _sampled(np.stack([np.fromfile(os.path.join(td, f"logits_{s_}.f32"), dtype=np.float32) for s_ in dumped]))
        return toks, np.array(dumped, dtype=np.int32), ti, tv, st, out.stdout.strip().splitlines()[0]

    toks, steps, ti, tv, st, timing = run(ids, 65, 16, True)
    np.savez_compressed(os.path.join(GOLD, "qwen2vl_2b_ref.npz"), tokens=toks, steps=steps, top_idx=ti, top_val=tv, strided=st, timing=np.frombuffer(timing.encode(), dtype=np.uint8))
    print("qwen2vl_2b_ref.npz", toks[:8], steps, timing)
    ids_t = np.random.default_rng(5).integers(0, 150000, size=40).astype(np.int32)
    toks, steps, ti, tv, st, timing = run(ids_t, 33, 8, False)
    np.savez_compressed(os.path.join(GOLD, "qwen2vl_2b_ref_text.npz"), ids=ids_t, tokens=toks, steps=steps, top_idx=ti, top_val=tv, strided=st)
    print("qwen2vl_2b_ref_text.npz", toks[:8], steps, timing)
    (emb,), _ = run_ops("vision", path, [[(pix, (1024, 3, 2, 14, 14)), (grid.astype(np.float32), (1, 1, 1, 3))]], p=(c.hidden, c.v_dim), threads=8)
    np.savez_compressed(os.path.join(GOLD, "qwen2vl_2b_ref_vision.npz"), image_embeds=emb.reshape(-1, c.hidden))
    print("qwen2vl_2b_ref_vision.npz", emb.reshape(-1, c.hidden).shape)


def run_ref_llm(c, n_prompt, steps, threads=4):
    td, path = tempfile.mkdtemp(dir=os.environ.get("MLLM_GOLD_TMP")), weights.causal_lm_file(c, CACHE)
    ids = synth.causal_lm_ids(c, n_prompt)
    ids.tofile(os.path.join(td, "ids.i32"))
    cfg = f"{c.hidden},{c.inter},{c.layers},{c.heads},{c.kv_heads},{c.vocab},{c.cache_limit},{int(c.tie_embedding)}"
    out = subprocess.run([os.path.join(REF, "ref_llm"), "--family", c.family, "--model", path, "--ids", os.path.join(td, "ids.i32"), "--steps", str(steps),
                          "--threads", str(threads), "--out", td, "--cfg", cfg], check=True, capture_output=True, text=True)
    toks = np.fromfile(os.path.join(td, "tokens.i32"), dtype=np.int32)
    logits = np.stack([np.fromfile(os.path.join(td, f"logits_{s}.f32"), dtype=np.float32) for s in range(steps)])
    return ids, toks, logits, out.stdout.strip().splitlines()[0]


def run_ref_vit(c, n_img, threads=4):
    td, path = tempfile.mkdtemp(dir=os.environ.get("MLLM_GOLD_TMP")), weights.vit_file(c, CACHE)
    synth.vit_images(c, n_img).tofile(os.path.join(td, "img.f32"))
    cfg = f"{c.hidden},{c.heads},{c.ffn},{c.blocks},{c.patch},{c.img},{c.classes}"
    out = subprocess.run([os.path.join(REF, "ref_vit"), "--model", path, "--img", os.path.join(td, "img.f32"), "--n", str(n_img), "--threads", str(threads),
                          "--out", td, "--cfg", cfg], check=True, capture_output=True, text=True)
    return np.fromfile(os.path.join(td, "vit_logits.f32"), dtype=np.float32).reshape(n_img, c.classes), out.stdout.strip().splitlines()[0]


def run_ref_llava(c, steps, threads=4):
    td, path = tempfile.mkdtemp(dir=os.environ.get("MLLM_GOLD_TMP")), weights.llava_file(c, CACHE)
    ids, img = synth.llava_inputs(c)
    ids.tofile(os.path.join(td, "ids.i32"))
    img.tofile(os.path.join(td, "img.f32"))
    cfg = f"{c.hidden},{c.heads},{c.inter},{c.layers},{c.vocab},{c.cache_limit},{c.v_hidden},{c.v_heads},{c.v_ffn},{c.v_blocks},{c.patch},{c.img}"
    subprocess.run([os.path.join(REF, "ref_llava"), "--model", path, "--ids", os.path.join(td, "ids.i32"), "--img", os.path.join(td, "img.f32"), "--steps",
                    str(steps), "--threads", str(threads), "--out", td, "--cfg", cfg], check=True, capture_output=True, text=True)
    toks = np.fromfile(os.path.join(td, "tokens.i32"), dtype=np.int32)
    logits = np.stack([np.fromfile(os.path.join(td, f"logits_{s}.f32"), dtype=np.float32) for s in range(steps)])
    return ids, toks, logits


def run_ref_llava_parts(c, steps, threads=4, n_text=10):
    """The reference's LLaVA graph composed from its own modules with the position ids as an input (oracle/ref_drivers/ref_llava_parts.cpp): the
    whole-graph run LLaVAModel itself cannot give at this snapshot.  Returns ids, greedy tokens, last-row logits per step, projected visual rows."""
    td, path = tempfile.mkdtemp(dir=os.environ.get("MLLM_GOLD_TMP")), weights.llava_file(c, CACHE)
    ids, img = synth.llava_inputs(c, n_text)
    ids.tofile(os.path.join(td, "ids.i32"))
    img.tofile(os.path.join(td, "img.f32"))
    cfg = f"{c.hidden},{c.heads},{c.inter},{c.layers},{c.vocab},{c.cache_limit},{c.v_hidden},{c.v_heads},{c.v_ffn},{c.v_blocks},{c.patch},{c.img}"
    base = [os.path.join(REF, "ref_llava_parts"), "--model", path, "--ids", os.path.join(td, "ids.i32"), "--img", os.path.join(td, "img.f32"), "--threads",
            str(threads), "--out", td, "--cfg", cfg]
    subprocess.run(base + ["--dump-vision", "1"], check=True, capture_output=True, text=True)
    vis = np.fromfile(os.path.join(td, "vision.f32"), dtype=np.float32).reshape(c.v_tokens, c.v_ffn)
    out = subprocess.run(base + ["--steps", str(steps)], check=True, capture_output=True, text=True)
    toks = np.fromfile(os.path.join(td, "tokens.i32"), dtype=np.int32)
    logits = np.stack([np.fromfile(os.path.join(td, f"logits_{s}.f32"), dtype=np.float32) for s in range(steps)])
    return ids, toks, logits, vis, out.stdout.strip().splitlines()[-1]


def llava_tiny():
    """BASELINE config 5 (demo_llava) at a toy shape with the real head geometry (CLIP head_dim 64, LLaMA head_dim 128, vocab 32064): every logit
    of 6 steps and every projected visual row of the reference's run (ref_llava_parts: see its header for why not LLaVAModel itself)."""
    ids, tok, log, vis, timing = run_ref_llava_parts(synth.llava_tiny(), 6)
    np.savez_compressed(os.path.join(GOLD, "llava_tiny.npz"), ids=ids, tokens=tok, logits=log, vision=vis)
    print("llava_tiny.npz", tok.tolist(), timing)


def llava_full():
    """LLaVA-1.5-7B geometry (4096 / 11008 / 32 heads x 128, 32 layers, vocab 32064) + CLIP-ViT-L/14-336 (1024 / 4096 / 16 heads x 64, 23 blocks,
    577 tokens) on synthetic Q4_K weights: 14-token prompt with one image (S = 589 after the splice) + 5 decode steps.  Stored: greedy ids, top-64
    and every 97th logit of each step, and every 61st projected visual row (all 4096 columns)."""
    c = synth.llava_7b()
    ids, tok, log, vis, timing = run_ref_llava_parts(c, 6, threads=8)
    ti, tv, st = _sampled(log)
    np.savez_compressed(os.path.join(GOLD, "llava_7b.npz"), ids=ids, tokens=tok, top_idx=ti, top_val=tv, strided=st, vision_rows=vis[::61],
                        timing=np.frombuffer(timing.encode(), dtype=np.uint8))
    print("llava_7b.npz", tok.tolist(), timing)


def rope3_golden():
    """SURVEY N4 (one representative of the other families' ops): RoPE with llama3 frequency scaling (Layer.hpp:493-531 -> _compute_llama3_theta, CPURoPE.cpp:33-71) run by
    the reference on 40 positions x 2 heads x 64 dims; theta 500000, factor 8, low / high frequency factors 1 / 4, original context 8192: the 32 frequencies fall into all three
    regimes (kept, interpolated, divided)."""
    r = np.random.default_rng(37)
    q = r.standard_normal((40, 128), dtype=np.float32)
    pn = quantize_file([("dummy.weight", np.zeros(32, dtype=np.float32))], target="F32")
    params = (4, 500000.0, 128, 2, 64, 8.0, 1.0, 4.0, 8192)      # HFHUBROPE = 4
    (y,), _ = run_ops("rope3", pn, [[(q, (1, 1, 40, 128))]], p=params)
    np.savez_compressed(os.path.join(GOLD, "rope3.npz"), x=q, y=y.reshape(40, 128), params=np.array(params, dtype=np.float64))
    print("rope3.npz", y[:4])


def write_bmp(path, rgb):
    """24-bit uncompressed BMP (bottom-up rows, BGR, rows padded to 4 bytes) of an [H][W][3] uint8 array."""
    import struct
    h, w, _ = rgb.shape
    row = (3 * w + 3) & ~3
    body = bytearray()
    for y in range(h - 1, -1, -1):
        line = rgb[y, :, ::-1].tobytes()
        body += line + b"\0" * (row - len(line))
    with open(path, "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", 54 + len(body), 0, 0, 54))
        f.write(struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, len(body), 2835, 2835, 0, 0))
        f.write(bytes(body))


def preprocess_images():
    """the reference's Qwen2VLImageProcessor::preprocess_images (oracle/ref_drivers/ref_preprocess.cpp) on three synthetic BMPs: a 60 x 90 image
    (slight downsample to 56 x 84), a 51 x 37 one (upsample to 84 x 56 through the min_pixels branch of smart_resize) and a 112 x 112 one (same size: the cubic
    B-spline still smooths).  Stored: the RGB bytes, the flattened patches and the grid of each."""
    r = np.random.default_rng(31)
    out = {}
    for i, (h, w) in enumerate(((60, 90), (51, 37), (112, 112))):
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) * 3 % 256)], axis=-1).astype(np.int32)
        rgb = np.clip(base + r.integers(-40, 41, size=base.shape), 0, 255).astype(np.uint8)
        td = tempfile.mkdtemp(dir=os.environ.get("MLLM_GOLD_TMP"))
        write_bmp(os.path.join(td, "img.bmp"), rgb)
        subprocess.run([os.path.join(REF, "ref_preprocess"), "--img", os.path.join(td, "img.bmp"), "--out", td], check=True, capture_output=True, text=True)
        grid = np.fromfile(os.path.join(td, "grid.i32"), dtype=np.int32)
        out[f"rgb{i}"] = rgb
        out[f"grid{i}"] = grid
        out[f"patches{i}"] = np.fromfile(os.path.join(td, "patches.f32"), dtype=np.float32).reshape(int(grid.prod()), 1176)
    np.savez_compressed(os.path.join(GOLD, "preprocess.npz"), **out)
    print("preprocess.npz", [out[f"grid{i}"].tolist() for i in range(3)])


def sampling():
    """candidate ids and pre-draw probabilities of the reference's top-k / top-p methods (oracle/ref_drivers/ref_sampling.cpp: the library's own
    Generate.cpp code with only the random draw interposed) on fixed rows: 4 rows of 2048 logits ~ N(0, 2^2), and their softmax rows sharpened by 3 / 1 / 0.5 / 6
    so that the nuclei at p = 0.92 span one to many hundred candidates."""
    r = np.random.default_rng(29)
    n, rows = 2048, 4
    logits = (r.standard_normal((rows, n)) * 2.0).astype(np.float32)
    probs = []
    for i, sharp in enumerate((3.0, 1.0, 0.5, 6.0)):
        z = logits[i].astype(np.float64) * sharp
        e = np.exp(z - z.max())
        probs.append((e / e.sum()).astype(np.float32))
    probs = np.stack(probs)
    out = {"logits": logits, "probs": probs, "k": 5, "p": np.float32(0.92), "temp": np.float32(0.7)}
    for name, data in (("l", logits), ("s", probs)):
        td = tempfile.mkdtemp(dir=os.environ.get("MLLM_GOLD_TMP"))
        data.tofile(os.path.join(td, "rows.f32"))
        subprocess.run([os.path.join(REF, "ref_sampling"), "--in", os.path.join(td, "rows.f32"), "--n", str(n), "--rows", str(rows), "--k", "5", "--p", "0.92",
                        "--temp", "0.7", "--out", td], check=True, capture_output=True, text=True)
        for i in range(rows):
            out[f"topk_{name}{i}_idx"] = np.fromfile(os.path.join(td, f"topk_{i}.idx"), dtype=np.uint32)
            out[f"topk_{name}{i}_prob"] = np.fromfile(os.path.join(td, f"topk_{i}.prob"), dtype=np.float32)
            if name == "s":
                out[f"topp_{i}_idx"] = np.fromfile(os.path.join(td, f"topp_{i}.idx"), dtype=np.uint32)
                out[f"topp_{i}_prob"] = np.fromfile(os.path.join(td, f"topp_{i}.prob"), dtype=np.float32)
    np.savez_compressed(os.path.join(GOLD, "sampling.npz"), **out)
    print("sampling.npz", [len(out[f"topp_{i}_idx"]) for i in range(rows)], out["topk_l0_idx"], out["topk_l0_prob"])


def _sampled(logits):
    idx = np.stack([np.argsort(-l, kind="stable")[:64] for l in logits]).astype(np.int32)
    return idx, np.take_along_axis(logits, idx, axis=1), np.ascontiguousarray(logits[:, ::97])


def configs_tiny():
    """BASELINE configs 1-3 at toy shapes with the real head geometry (head_dim 64): every logit of the reference's run."""
    q_ids, q_tok, q_log, _ = run_ref_llm(synth.qwen15_tiny(), 20, 8)
    t_ids, t_tok, t_log, _ = run_ref_llm(synth.tinyllama_tiny(), 20, 8)
    k_ids, k_tok, k_log, _ = run_ref_llm(synth.tinyllama_tiny(mf.Q4_K), 20, 8)
    v_log, _ = run_ref_vit(synth.vit_tiny(), 3)
    np.savez_compressed(os.path.join(GOLD, "configs_tiny.npz"), qwen_ids=q_ids, qwen_tokens=q_tok, qwen_logits=q_log, tl_ids=t_ids, tl_tokens=t_tok,
                        tl_logits=t_log, tlq_ids=k_ids, tlq_tokens=k_tok, tlq_logits=k_log, vit_logits=v_log)
    print("configs_tiny.npz", q_tok.tolist(), t_tok.tolist(), k_tok.tolist(), v_log[:, :3])


def configs_full():
    """Qwen1.5-0.5B (Q4_K, 32-token prompt + 16 steps) and ViT-B/16 (Q4_K, 2 images) at their real sizes: greedy ids, top-64 and
    every 97th logit of each step for the LM; all 1000 class logits for the ViT."""
    c = synth.qwen15_05b()
    ids, tok, log, timing = run_ref_llm(c, 32, 16, threads=8)
    ti, tv, st = _sampled(log)
    v_log, v_timing = run_ref_vit(synth.vit_b16(), 2, threads=4)      # FA2 asserts heads % threads == 0 (FlashAttention2.hpp:128)
    np.savez_compressed(os.path.join(GOLD, "configs_full.npz"), qwen_ids=ids, qwen_tokens=tok, qwen_top_idx=ti, qwen_top_val=tv, qwen_strided=st,
                        vit_logits=v_log, qwen_timing=np.frombuffer(timing.encode(), dtype=np.uint8), vit_timing=np.frombuffer(v_timing.encode(), dtype=np.uint8))
    print("configs_full.npz", tok.tolist(), timing, v_timing)


def tinyllama_full():
    """BASELINE config 2 at its real size: TinyLlama-1.1B geometry (22 layers, 2048 hidden, 32 / 4 heads, untied 32000-row head), Q4_K, 24-token prompt + 12 greedy steps run by
    the reference; greedy ids, top-64 and every 97th logit of each step."""
    c = synth.tinyllama_11b(target=mf.Q4_K)
    ids, tok, log, timing = run_ref_llm(c, 24, 12, threads=8)
    ti, tv, st = _sampled(log)
    np.savez_compressed(os.path.join(GOLD, "tinyllama_11b.npz"), ids=ids, tokens=tok, top_idx=ti, top_val=tv, strided=st, timing=np.frombuffer(timing.encode(), dtype=np.uint8))
    print("tinyllama_11b.npz", tok.tolist(), timing)


def n4_golden():
    """the extra ops of the other model families, each run by the reference itself (oracle/ref_drivers/ref_ops.cpp cases swmask, ntkrope, topk, bincount,
    scatter_add, gather_rows, fuyu_gather) on small seeded inputs that exercise their edge cases: window edges on both sides with old keys in the cache, tied scores in
    the top-k, a repeated destination row in scatter_add, negative (= keep the word embedding) indices in the Fuyu gather, a decode step after the prefill for the rotary op."""
    r = np.random.default_rng(41)
    pn = quantize_file([("dummy.weight", np.zeros(32, dtype=np.float32))], target="F32")
    G = {}
    # sliding window: 2 heads, 6 new rows over 10 keys (4 old), window 3
    H, S, KEYS, WIN = 2, 6, 10, 3
    x = r.standard_normal((S, H * KEYS), dtype=np.float32)
    (y,), _ = run_ops("swmask", pn, [[(x, (1, 1, S, H * KEYS))]], p=(WIN, H, KEYS))
    G["swmask_x"], G["swmask_y"], G["swmask_p"] = x, y.reshape(S, H * KEYS), np.array([WIN, H, KEYS])
    # top-k over 16 experts, k = 4, with ties (rows 1 and 3 hold repeated values, row 4 is constant)
    E, K = 16, 4
    sc = r.random((5, E), dtype=np.float32)
    sc[1, [2, 9, 11]] = sc[1].max() + 0.25
    sc[3, [0, 15]] = 0.5; sc[3, [5, 6]] = 0.75
    sc[4, :] = 0.125
    (tv,), _ = run_ops("topk", pn, [[(sc, (1, 1, 5, E))]], p=(K, 0))
    (ti,), _ = run_ops("topk", pn, [[(sc, (1, 1, 5, E))]], p=(K, 1))
    G["topk_x"], G["topk_v"], G["topk_i"] = sc, tv.reshape(5, K), ti.reshape(5, K)
    # the HEAD-axis form of the same function (scores [1, H = 16, S = 5, 1]: in BSHD memory the same [S][H] rows)
    (hv,), _ = run_ops("topk", pn, [[(sc, (1, 1, 5, E))]], p=(K, 0, 1, E))
    (hi,), _ = run_ops("topk", pn, [[(sc, (1, 1, 5, E))]], p=(K, 1, 1, E))
    G["topk_head_v"], G["topk_head_i"] = hv.reshape(5, K), hi.reshape(5, K)
    # bincount of expert ids
    ids = np.array([3, 0, 3, 7, 1, 1, 3], dtype=np.float32)
    (bc,), _ = run_ops("bincount", pn, [[(ids, (1, 1, 1, ids.size))]])
    G["bincount_x"], G["bincount_y"] = ids, bc
    # NTK / LongRoPE rotary (MiniCPM3): theta 10000, 64 positions, original 32 (the table is built for 64 > 32 positions: the long factors), 2 heads x 8 dims
    D, HEADS = 8, 2
    lf = [1.5, 2.25, 3.0, 4.5]; sf = [1.0, 1.0625, 1.125, 1.25]
    q0 = r.standard_normal((6, HEADS * D), dtype=np.float32); q1 = r.standard_normal((1, HEADS * D), dtype=np.float32)
    params = (10000.0, 64, 32, HEADS, D) + tuple(lf) + tuple(sf)
    (y0, y1), _ = run_ops("ntkrope", pn, [[(q0, (1, 1, 6, HEADS * D))], [(q1, (1, 1, 1, HEADS * D))]], p=params)
    G["ntk_x0"], G["ntk_y0"], G["ntk_x1"], G["ntk_y1"], G["ntk_p"] = q0, y0.reshape(6, -1), q1, y1.reshape(1, -1), np.array(params, dtype=np.float64)
    # and with 24 positions < original 32: the short factors
    params_s = (10000.0, 24, 32, HEADS, D) + tuple(lf) + tuple(sf)
    (y2,), _ = run_ops("ntkrope", pn, [[(q0, (1, 1, 6, HEADS * D))]], p=params_s)
    G["ntk_y_short"], G["ntk_p_short"] = y2.reshape(6, -1), np.array(params_s, dtype=np.float64)
    # scatter_add with a repeated destination row; gather rows; Fuyu gather
    dst = r.standard_normal((6, 8), dtype=np.float32); src = r.standard_normal((4, 8), dtype=np.float32)
    idx = np.array([2, 5, 2, 0], dtype=np.float32)
    (sa,), _ = run_ops("scatter_add", pn, [[(dst, (1, 1, 6, 8)), (src, (1, 1, 4, 8)), (idx, (1, 1, 1, 4))]])
    G["sa_dst"], G["sa_src"], G["sa_idx"], G["sa_y"] = dst, src, idx, sa.reshape(6, 8)
    gi = np.array([3, 0, 3], dtype=np.float32)
    (gr,), _ = run_ops("gather_rows", pn, [[(dst, (1, 1, 6, 8)), (gi, (1, 1, 1, 3))]])
    G["gr_idx"], G["gr_y"] = gi, gr.reshape(3, 8)
    patches = r.standard_normal((3, 8), dtype=np.float32)
    fi = np.array([-1, 0, -1, 2, 1, -1], dtype=np.float32)
    (fg,), _ = run_ops("fuyu_gather", pn, [[(dst, (1, 1, 6, 8)), (patches, (1, 1, 3, 8)), (fi, (1, 1, 6, 1))]])
    G["fuyu_patches"], G["fuyu_idx"], G["fuyu_y"] = patches, fi, fg.reshape(6, 8)
    np.savez_compressed(os.path.join(GOLD, "n4_ops.npz"), **G)
    for k in ("topk_v", "topk_i", "bincount_y", "gr_y"):
        print(k, G[k].tolist() if G[k].size < 40 else G[k].shape)


def moe_golden():
    """the reference's own MiniCPMMoE block (ref_ops case moe) on the synthetic Q4_K file of mllm_amd/synthfile.py: 4 experts, 2 per token, 37 tokens (every expert
    receives tokens, some fewer than 16 rows, one more) and a single token (decode: two experts get one row, the others none)."""
    c = synth.moe_tiny()
    path = weights.moe_file(c, CACHE)
    G = {}
    for tag, n in (("p", 37), ("d", 1)):
        x = synth.moe_input(c, n, seed=17 + n)
        (y,), _ = run_ops("moe", path, [[(x, (1, 1, n, c.hidden))]], p=(c.hidden, c.inter, c.experts, c.per_tok))
        G["x_" + tag], G["y_" + tag] = x, y.reshape(n, c.hidden)
    f = mf.MllmFile(path)
    import hashlib
    G["digest"] = np.frombuffer("".join(hashlib.sha256(f.raw(n).tobytes()).hexdigest()[:16] for n in sorted(f.names())).encode(), dtype=np.uint8)
    f.close()
    np.savez_compressed(os.path.join(GOLD, "moe.npz"), **G)
    print("moe.npz", G["y_p"].shape, float(np.abs(G["y_p"]).max()))


def mm_bhsd_golden():
    """The eager-attention form of F_MM (CPUMatmulFunc.hpp:123-172 -> compute/GemmFp.hpp:104-150, x86 path) run by the reference itself on BHSD tensors (ref_ops case
    mm_bhsd): q k^T with rows / columns that do not fill the 8 x 8 micro-kernel, p v, a contraction longer than one 256-wide K block on edge tiles, and all-full tiles."""
    r = np.random.default_rng(53)
    pn = quantize_file([("dummy.weight", np.zeros(32, dtype=np.float32))], target="F32")
    G = {}
    # name: heads, M, N, K, transpose-the-right-operand-first
    # (the reference's own BHSD (SEQUENCE, DIMENSION) transpose asserts on non-square operands -- CPUTransposeFunc.hpp:172 -- so the right operand is handed over as [K][N]; tr stays in the driver for a reference that fixes it)
    cases = {"qk": (3, 21, 21, 64, 0), "pv": (3, 21, 64, 21, 0), "long": (2, 10, 17, 300, 0), "full": (2, 16, 24, 40, 0), "deep": (1, 24, 16, 600, 0)}
    for name, (H, M, N, K, tr) in cases.items():
        a = r.standard_normal((H, M, K), dtype=np.float32)
        b = r.standard_normal((H, N, K) if tr else (H, K, N), dtype=np.float32)
        (y,), _ = run_ops("mm_bhsd", pn, [[(a, (1, H, M, K)), (b, (1, H, N, K) if tr else (1, H, K, N))]], p=(tr,))
        G[name + "_a"], G[name + "_b"], G[name + "_y"], G[name + "_p"] = a, b, y.reshape(H, M, N), np.array([H, M, N, K, tr])
    np.savez_compressed(os.path.join(GOLD, "mm_bhsd.npz"), **G)
    print("mm_bhsd.npz", {k: v.shape for k, v in G.items() if k.endswith("_y")})


if __name__ == "__main__":
    if "--ragged" in sys.argv:
        e2e_ragged()
        sys.exit(0)
    if "--mm-bhsd" in sys.argv:
        mm_bhsd_golden()
        sys.exit(0)
    if "--moe" in sys.argv:
        moe_golden()
        sys.exit(0)
    if "--tinyllama-full" in sys.argv:
        tinyllama_full()
        sys.exit(0)
    if "--n4" in sys.argv:
        n4_golden()
        sys.exit(0)
    if "--llava" in sys.argv:
        llava_tiny()
        sys.exit(0)
    if "--rope3" in sys.argv:
        rope3_golden()
        sys.exit(0)
    if "--preprocess" in sys.argv:
        preprocess_images()
        sys.exit(0)
    if "--sampling" in sys.argv:
        sampling()
        sys.exit(0)
    if "--llava-full" in sys.argv:
        llava_full()
        sys.exit(0)
    if "--all" in sys.argv:      # every golden that depends on a synthetic model file
        e2e_tiny()
        e2e_ragged()
        configs_tiny()
        llava_tiny()
        moe_golden()
        if "--full" in sys.argv:
            e2e_full()
            configs_full()
            tinyllama_full()
            llava_full()
        sys.exit(0)
    if "--configs" in sys.argv:
        configs_tiny()
        if "--full" in sys.argv:
            configs_full()
        sys.exit(0)
    os.makedirs(GOLD, exist_ok=True)
    ops_golden()
    e2e_tiny()
    if "--full" in sys.argv:
        e2e_full()
