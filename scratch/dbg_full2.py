import sys, os, numpy as np, ctypes as C, torch
sys.path.insert(0, '.')
from mllm_amd import lib, synth, weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
g = np.load('tests/golden/qwen2vl_2b_ref_text.npz'); st = np.load('scratch/dbg_stages.npz')
L = lib.load(); L.mllm_hip_qwen2vl_debug_ptr.restype = C.c_void_p; L.mllm_hip_qwen2vl_debug_ptr.argtypes = [C.c_void_p, C.c_int]
S = g['ids'].size
def grab(m, which, rows, cols, ld=None):
    ld = ld or cols
    p = L.mllm_hip_qwen2vl_debug_ptr(m._h, which)
    buf = torch.empty(rows * ld, dtype=torch.float32, device='cuda')
    import ctypes
    lib.check(L.mllm_hip_d2h(C.c_void_p(buf.data_ptr()), C.c_void_p(p), C.c_size_t(0), None)) if False else None
    torch.cuda.synchronize()
    h = np.empty(rows * ld, dtype=np.float32)
    lib.check(L.mllm_hip_d2h(h.ctypes.data_as(C.c_void_p), C.c_void_p(p), C.c_size_t(rows * ld * 4), None))
    return h.reshape(rows, ld)[:, :cols]
def cmp(name, a, b):
    d = np.abs(a - b); bad = np.argwhere(a != b)
    print(f'{name:10s} maxdiff {d.max():.3e} ndiff {len(bad)} / {a.size}', 'first', bad[:3].tolist() if len(bad) else '')
for nl, key in ((1, 'layer0'), (2, 'layer1'), (3, 'layer2'), (4, 'layer3'), (8, 'layer7'), (16, 'layer15'), (28, 'layer27')):
    os.environ['MLLM_HIP_MAX_LAYERS'] = str(nl)
    m = lib.Qwen2VL(cfg, path) if nl == 1 else m
    m.clear_kvcache()
    m.prefill(g['ids'])
    h = grab(m, 0, S, cfg.hidden)
    cmp(key, h, st[key])
    if nl == 1:
        qkv = grab(m, 2, S, 2048)
        cmp('q_rope', qkv[:, :1536], st['q_rope']); cmp('k_pre', qkv[:, 1536:1792], st['k_pre']); cmp('v_pre', qkv[:, 1792:], st['v_pre'])
        cmp('attn', grab(m, 3, S, 1536), st['attn']); cmp('h1', grab(m, 1, S, 1536), st['h1'])
        gu = grab(m, 4, S, 17920); cmp('gate', gu[:, :8960], st['gate']); cmp('up', gu[:, 8960:], st['up'])
        cmp('act', grab(m, 5, S, 8960), st['act'])
