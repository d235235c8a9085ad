import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from mllm_amd import lib
lib.SO_PATH = os.path.abspath(sys.argv[1])
from mllm_amd import ops
ops.require_gpu()
r = np.random.default_rng(0)
for (S, H, Hkv, D, causal, f16) in ((1024, 16, 16, 80, False, False), (282, 12, 2, 128, True, True)):
    q = torch.from_numpy(r.standard_normal((S, H * D)).astype(np.float32)).cuda()
    k = torch.from_numpy(r.standard_normal((S, Hkv * D)).astype(np.float32)).cuda(); v = torch.from_numpy(r.standard_normal((S, Hkv * D)).astype(np.float32)).cuda()
    if f16: k, v = k.half(), v.half()
    for _ in range(3): ops.flash_attention2(q, k, v, S, S, H, Hkv, D, causal)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.flash_attention2(q, k, v, S, S, H, Hkv, D, causal)
    e1.record(); torch.cuda.synchronize()
    print(sys.argv[1], S, D, 'us/call %.1f' % (e0.elapsed_time(e1) * 100))
