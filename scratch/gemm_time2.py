import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from mllm_amd import lib
lib.SO_PATH = os.path.abspath(sys.argv[1])
from mllm_amd import ops
import ctypes as C
ops.require_gpu()
L = lib.load()
r = np.random.default_rng(0)
out = []
for (M, N, K) in ((1024, 5120, 1280), (1024, 5120, 2560), (1024, 5120, 5120), (1024, 2560, 1280), (512, 5120, 1280), (1024, 1024, 1280)):
    W = lib.quantize_host(lib.Q4_K, (r.standard_normal((N, K)) * 0.05).astype(np.float32))
    Wd = torch.from_numpy(W.view(np.uint8)).cuda()
    nbytes = L.mllm_hip_q4k_prepack_bytes(C.c_int(N), C.c_int(K)); xbytes = L.mllm_hip_q4k_prepack_bytes(C.c_int(M), C.c_int(K))
    wp = torch.empty(nbytes, dtype=torch.uint8, device='cuda'); xp = torch.empty(xbytes, dtype=torch.uint8, device='cuda')
    lib.check(L.mllm_hip_q4k_prepack(C.c_void_p(Wd.data_ptr()), C.c_int(N), C.c_int(K), C.c_void_p(wp.data_ptr()), None))
    x = torch.from_numpy(r.standard_normal((M, K)).astype(np.float32)).cuda()
    q = ops.quantize_q8k(x)
    y = torch.empty((M, N), dtype=torch.float32, device='cuda')
    def run():
        lib.check(L.mllm_hip_linear_q4kp_q8k(C.c_void_p(wp.data_ptr()), None, C.c_void_p(q.qs.data_ptr()), C.c_void_p(q.d.data_ptr()), C.c_void_p(q.bsums.data_ptr()), C.c_void_p(xp.data_ptr()),
                  C.c_void_p(y.data_ptr()), C.c_int(lib.F32), C.c_int64(N), None, C.c_int(M), C.c_int(N), C.c_int(K), None))
    for _ in range(3): run()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    out.append('%dx%dx%d %.0fus %.0fTF' % (M, N, K, us, 2.0 * M * N * K / us / 1e6))
print(sys.argv[1], ' | '.join(out))
