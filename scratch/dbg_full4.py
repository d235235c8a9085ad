import sys, os, numpy as np, ctypes as C, torch
sys.path.insert(0, '.')
from mllm_amd import lib, synth, weights, ops
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
g = np.load('tests/golden/qwen2vl_2b_ref_text.npz'); z = np.load('scratch/dbg_attn1.npz')
L = lib.load(); L.mllm_hip_qwen2vl_debug_ptr.restype = C.c_void_p; L.mllm_hip_qwen2vl_debug_ptr.argtypes = [C.c_void_p, C.c_int]
S = g['ids'].size
def grab16(m, which, off_rows, rows, cols):
    p = L.mllm_hip_qwen2vl_debug_ptr(m._h, which) + off_rows * cols * 2
    torch.cuda.synchronize()
    h = np.empty(rows * cols, dtype=np.uint16)
    lib.check(L.mllm_hip_d2h(h.ctypes.data_as(C.c_void_p), C.c_void_p(p), C.c_size_t(rows * cols * 2), None))
    return h.reshape(rows, cols)
os.environ['MLLM_HIP_MAX_LAYERS'] = '2'
m = lib.Qwen2VL(cfg, path)
m.prefill(g['ids'])
k = grab16(m, 6, cfg.cache_limit, S, 256); v = grab16(m, 7, cfg.cache_limit, S, 256)
for name, a, b in (('K', k, z['k16']), ('V', v, z['v16'])):
    bad = np.argwhere(a != b)
    print(name, 'ndiff', len(bad), bad[:5].tolist(), [(hex(int(a[tuple(i)])), hex(int(b[tuple(i)]))) for i in bad[:5]])
