"""timing of the patch-embedding GEMM (1024 x 1176 x 1280 fp32) through the C ABI: median of 50 launches by HIP events; MLLM_SO selects a variant library"""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from mllm_amd import lib
if os.environ.get('MLLM_SO'): lib.SO_PATH = os.environ['MLLM_SO']
from mllm_amd import ops
r = np.random.default_rng(9)
px, W = r.standard_normal((1024, 1176)).astype(np.float32), (r.standard_normal((1280, 1176)) * 0.02).astype(np.float32)
pxd, Wd = torch.from_numpy(px).cuda(), torch.from_numpy(W).cuda()
for _ in range(5): y = ops.patch_gemm(pxd, Wd)
ts = []
for _ in range(50):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); y = ops.patch_gemm(pxd, Wd); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1000)
print(os.environ.get('MLLM_SO', 'default'), 'patch gemm median %.1f us  min %.1f' % (np.median(ts), min(ts)))
