"""prefill attention of the vision block (16 heads, D = 80, 1024 keys) with four and eight waves per workgroup (option fa_waves), results compared bit for bit"""
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from mllm_amd import lib, ops
ops.require_gpu()
r = np.random.default_rng(0)
H, D, Sk = 16, 80, 1024
k = torch.from_numpy(r.standard_normal((Sk, H * D)).astype(np.float32)).cuda(); v = torch.from_numpy(r.standard_normal((Sk, H * D)).astype(np.float32)).cuda()
for Sq in (32, 256, 1024, 1536, 2048, 4096):
    q = torch.from_numpy(r.standard_normal((Sq, H * D)).astype(np.float32)).cuda()
    outs = {}
    for nw in (4, 8, 4, 8):
        lib.set_option("fa_waves", nw)
        for _ in range(3): o = ops.flash_attention2(q, k, v, Sq, Sk, H, H, D, False)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): o = ops.flash_attention2(q, k, v, Sq, Sk, H, H, D, False)
        e1.record(); torch.cuda.synchronize()
        outs[nw] = o.cpu().numpy()
        print('Sq %4d  workgroups %4d  waves %d: %.1f us per launch' % (Sq, H * ((Sq + 31) // 32), nw, e0.elapsed_time(e1) * 100), flush=True)
    print('   bit-equal:', np.array_equal(outs[4], outs[8]))
lib.set_option("fa_waves", -1)
