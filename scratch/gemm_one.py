import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from mllm_amd import lib, ops
import ctypes as C
ops.require_gpu()
L = lib.load()
r = np.random.default_rng(0)
shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
for (M, N, K) in shapes:
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
    ts = []
    for _ in range(8):
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(round(e0.elapsed_time(e1) * 1e3))
    # row-by-row GEMV (the C-ABI single-row path) on a few rows as the check
    bad = 0
    for m in (0, 1, M // 2, M - 1):
        ref = ops.linear_q4k(W, x[m:m + 1].cpu().numpy(), N)
        bad += int((ref[0] != y[m]).sum().item())
    print(M, N, K, 'us per call', ts, 'mismatches vs GEMV rows', bad, flush=True)
