import sys, numpy as np, ctypes as C, torch
sys.path.insert(0, '.')
from mllm_amd import ops, lib
ops.require_gpu()
L = lib.load()
r = np.random.default_rng(6)
for (M, dim) in [(9, 1280), (16, 1280), (9, 260), (64, 1280)]:
    x = r.standard_normal((M, dim)).astype(np.float32) * 2 + 0.3
    xd = torch.from_numpy(x).cuda(); st = torch.zeros((M, 2), dtype=torch.float32, device='cuda')
    L.mllm_hip_debug_ln_stats(C.c_void_p(xd.data_ptr()), C.c_void_p(st.data_ptr()), C.c_int(M), C.c_int(dim), C.c_float(1e-6), None)
    torch.cuda.synchronize()
    g = st.cpu().numpy()
    for m in range(M):
        s = np.float32(0)
        for v in x[m]: s = np.float32(s + v)
        mean = np.float32(s / np.float32(dim))
        ssq = np.float32(0)
        for v in x[m]:
            c = np.float32(v - mean)
            ssq = np.float32(np.float64(c) * np.float64(c) + np.float64(ssq))
        rms = np.float32(np.sqrt(np.float32(np.float32(ssq / np.float32(dim)) + np.float32(1e-6))))
        if g[m, 0] != mean or g[m, 1] != rms:
            print(M, dim, 'row', m, 'gpu', g[m], 'cpu', mean, rms, 'sum', s)
print('done')
