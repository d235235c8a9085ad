import sys, os, numpy as np, ctypes as C, torch
sys.path.insert(0, '.')
from mllm_amd import lib, synth, weights, ops
from oracle import oracle as orc
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
g = np.load('tests/golden/qwen2vl_2b_ref_text.npz'); st = np.load('scratch/dbg_stages1.npz'); st0 = np.load('scratch/dbg_stages.npz')
L = lib.load(); L.mllm_hip_qwen2vl_debug_ptr.restype = C.c_void_p; L.mllm_hip_qwen2vl_debug_ptr.argtypes = [C.c_void_p, C.c_int]
S = g['ids'].size
def grab(m, which, rows, cols):
    p = L.mllm_hip_qwen2vl_debug_ptr(m._h, which)
    torch.cuda.synchronize()
    h = np.empty(rows * cols, dtype=np.float32)
    lib.check(L.mllm_hip_d2h(h.ctypes.data_as(C.c_void_p), C.c_void_p(p), C.c_size_t(rows * cols * 4), None))
    return h.reshape(rows, cols)
def cmp(name, a, b):
    d = np.abs(a - b); bad = np.argwhere(a != b)
    print(f'{name:10s} maxdiff {d.max():.3e} ndiff {len(bad)} / {a.size}', 'first', bad[:3].tolist() if len(bad) else '')
os.environ['MLLM_HIP_MAX_LAYERS'] = '2'
m = lib.Qwen2VL(cfg, path)
m.prefill(g['ids'])
qkv = grab(m, 2, S, 2048)
cmp('q_rope', qkv[:, :1536], st['L1_q_rope']); cmp('k_pre', qkv[:, 1536:1792], st['L1_k_pre']); cmp('v_pre', qkv[:, 1792:], st['L1_v_pre'])
cmp('attn', grab(m, 3, S, 1536), st['L1_attn']); cmp('h1', grab(m, 1, S, 1536), st['L1_h1'])
gu = grab(m, 4, S, 17920); cmp('gate', gu[:, :8960], st['L1_gate']); cmp('up', gu[:, 8960:], st['L1_up'])
cmp('act', grab(m, 5, S, 8960), st['L1_act'])
# stand-alone rmsnorm of the exact layer-0 output
x = st0['layer0']
import mllm_amd.mllmfile as mf
f = mf.MllmFile(path); wn = f.f32('model.layers.1.input_layernorm.weight')
cmp('rms op', ops.rmsnorm(x, wn, 1e-6).cpu().numpy(), st['L1_norm'])
