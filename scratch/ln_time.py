import sys, numpy as np, torch
sys.path.insert(0, '.')
from mllm_amd import ops
ops.require_gpu()
r = np.random.default_rng(0)
x = torch.from_numpy(r.standard_normal((1024, 1280)).astype(np.float32)).cuda(); w = torch.ones(1280, device='cuda'); b = torch.zeros(1280, device='cuda')
for _ in range(3): ops.layernorm(x, w, b, 1e-6)
torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.layernorm(x, w, b, 1e-6)
e1.record(); torch.cuda.synchronize()
print('layernorm 1024x1280 us/call %.1f' % (e0.elapsed_time(e1) * 50))
