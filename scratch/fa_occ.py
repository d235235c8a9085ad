"""prefill attention (D = 80, fp32 K/V) timed at query-row counts that put 1, 2, 3, 4 workgroups on a CU (16 heads x S/32 row blocks over 256 CUs), all with 1024 keys:
what one workgroup's 32 chunks cost alone and how it stretches when workgroups share a CU"""
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from mllm_amd import lib
if os.environ.get('MLLM_SO'): lib.SO_PATH = os.environ['MLLM_SO']
from mllm_amd import ops
ops.require_gpu()
r = np.random.default_rng(0)
H, D, Sk = 16, 80, 1024
k = torch.from_numpy(r.standard_normal((Sk, H * D)).astype(np.float32)).cuda(); v = torch.from_numpy(r.standard_normal((Sk, H * D)).astype(np.float32)).cuda()
for Sq in (32, 256, 512, 1024, 1536, 2048):
    q = torch.from_numpy(r.standard_normal((Sq, H * D)).astype(np.float32)).cuda()
    for _ in range(3): ops.flash_attention2(q, k, v, Sq, Sk, H, H, D, False)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.flash_attention2(q, k, v, Sq, Sk, H, H, D, False)
    e1.record(); torch.cuda.synchronize()
    print('Sq %4d  workgroups %4d (%.2f per CU)  %.1f us per launch' % (Sq, H * ((Sq + 31) // 32), H * ((Sq + 31) // 32) / 256, e0.elapsed_time(e1) * 100), flush=True)
