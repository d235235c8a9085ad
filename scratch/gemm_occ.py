"""occupancy scaling + removal variants of the four-waves-per-tile GEMM: time only"""
import sys, os, numpy as np, torch, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mllm_amd import lib
if len(sys.argv) > 1: lib.SO_PATH = os.path.abspath(sys.argv[1])
from mllm_amd import ops, synth
ops.require_gpu()
L = lib.load()
r = np.random.default_rng(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
full = len(sys.argv) > 2
shapes = [(1024, 512, 1280), (1024, 512, 5120), (1024, 1024, 1280), (1024, 1024, 5120), (1024, 2048, 1280), (1024, 2048, 5120), (1024, 5120, 1280), (1024, 5120, 5120)] if full else [(1024, 5120, 1280), (1024, 5120, 5120)]
out = []
for (M, N, K) in shapes:
    Wd = torch.from_numpy(synth.quantized_blocks(lib.Q4_K, r, N * K)).cuda()
    wp = torch.empty(int(L.mllm_hip_q4k_wpack_bytes(C.c_int(N), C.c_int(K))), dtype=torch.uint8, device="cuda")
    xp = torch.empty(int(L.mllm_hip_q4k_prepack_bytes(C.c_int(M), C.c_int(K))), dtype=torch.uint8, device="cuda")
    lib.check(L.mllm_hip_q4k_prepack(C.c_void_p(Wd.data_ptr()), C.c_int(N), C.c_int(K), C.c_void_p(wp.data_ptr()), st))
    x = torch.from_numpy(r.standard_normal((M, K)).astype(np.float32)).cuda()
    lib.check(L.mllm_hip_quantize_q8k_packed(C.c_void_p(x.data_ptr()), C.c_void_p(xp.data_ptr()), C.c_int(M), C.c_int(K), st))
    line = "%dx%dx%d" % (M, N, K)
    for tag, opt in (("w2", 4), ("w4", -1)):
        lib.set_option("gemm_waves", opt)
        y = torch.zeros((M, N), dtype=torch.float32, device="cuda")
        run = lambda: lib.check(L.mllm_hip_linear_q4kp_packed(C.c_void_p(wp.data_ptr()), None, C.c_void_p(xp.data_ptr()), C.c_void_p(y.data_ptr()), C.c_int(lib.F32), C.c_int64(N), None, C.c_int(M), C.c_int(N), C.c_int(K), st))
        for _ in range(3): run()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        line += "  %s %.1f" % (tag, e0.elapsed_time(e1) * 50)
    out.append(line)
print(os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "shipped", " | ".join(out), flush=True)
