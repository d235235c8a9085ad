"""step-by-step comparison of the MoE block's pieces on the device against the restatement (diagnostic)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mllm_amd import mllmfile as mf, ops, synth
from tests.fixtures import weights
from oracle import models as om, oracle as orc
import torch
cfg = synth.moe_tiny()
path = weights.moe_file(cfg)
g = np.load("tests/golden/moe.npz")
x = g["x_p"]
f = mf.MllmFile(path)
raw = lambda n: np.array(f.raw(n))
b = cfg.base
w = om.Weights(path)
y = ops.moe_block(x, raw(b + "gate.weight"), [raw(f"{b}experts.{e}.w1.weight") for e in range(cfg.experts)], [raw(f"{b}experts.{e}.w3.weight") for e in range(cfg.experts)],
                  [raw(f"{b}experts.{e}.w2.weight") for e in range(cfg.experts)], cfg.inter, cfg.per_tok).cpu().numpy()
bad = np.nonzero((y != g["y_p"]).any(axis=1))[0]
print("bad rows", bad.tolist()); print("max abs diff", np.abs(y - g["y_p"]).max(), "nbad elems", int((y != g["y_p"]).sum()))
sc_ref = w.lin(x, b + "gate", cfg.experts, bias=False)
sc = ops.linear_q4k(raw(b + "gate.weight"), x, cfg.experts).cpu().numpy()
print("router scores equal", np.array_equal(sc, sc_ref), np.abs(sc - sc_ref).max())
pr_ref = orc.softmax(sc_ref)
pr = ops.softmax(torch.from_numpy(sc_ref)).cpu().numpy()
print("softmax equal", np.array_equal(pr, pr_ref))
tv_ref, ti_ref = orc.topk_rows(pr_ref, cfg.per_tok)
tv, ti = ops.topk_rows(torch.from_numpy(pr_ref), cfg.per_tok)
print("topk equal", np.array_equal(tv.cpu().numpy(), tv_ref), np.array_equal(ti.cpu().numpy(), ti_ref))
flat = ti_ref.reshape(-1).astype(np.int64)
for e in range(cfg.experts):
    pairs = np.nonzero(flat == e)[0]
    tok = pairs // cfg.per_tok
    print("expert", e, "rows", len(tok), "tokens", tok.tolist())
    xe = x[tok]
    for nm, N, inp in (("w1", cfg.inter, xe), ("w3", cfg.inter, xe)):
        r = w.lin(inp, f"{b}experts.{e}.{nm}", N, bias=False)
        d = ops.linear_q4k(raw(f"{b}experts.{e}.{nm}.weight"), inp, N).cpu().numpy()
        print("  ", nm, np.array_equal(r, d))
    gg = orc.silu(w.lin(xe, f"{b}experts.{e}.w1", cfg.inter, bias=False)); uu = w.lin(xe, f"{b}experts.{e}.w3", cfg.inter, bias=False)
    gu = np.concatenate([w.lin(xe, f"{b}experts.{e}.w1", cfg.inter, bias=False), uu], axis=1)
    act = ops.silu_mul(gu, cfg.inter).cpu().numpy()
    print("   silu_mul", np.array_equal(act, (gg * uu).astype(np.float32)))
    act = (gg * uu).astype(np.float32)
    r = w.lin(act, f"{b}experts.{e}.w2", cfg.hidden, bias=False)
    d = ops.linear_q4k(raw(f"{b}experts.{e}.w2.weight"), act, cfg.hidden).cpu().numpy()
    print("   w2", np.array_equal(r, d))
