import sys, numpy as np, torch
sys.path.insert(0, '.')
from mllm_amd import ops
from oracle import oracle as orc
ops.require_gpu()
r = np.random.default_rng(3)
Sq = Sk = 40; Hq, Hkv, D = 12, 2, 128
for sc in (1.0, 3.0, 10.0, 30.0, 100.0):
    q = (r.standard_normal((Sq, Hq * D)) * sc).astype(np.float32)
    k = (r.standard_normal((Sk, Hkv * D))).astype(np.float16)
    v = (r.standard_normal((Sk, Hkv * D))).astype(np.float16)
    o = ops.flash_attention2(q, torch.from_numpy(k), torch.from_numpy(v), Sq, Sk, Hq, Hkv, D, True).cpu().numpy()
    ref = orc.attention(q, k.view(np.uint16), v.view(np.uint16), Sq, Sk, Hq, Hkv, D, True)
    bad = np.argwhere(o != ref)
    print('scale', sc, 'ndiff', len(bad), 'maxdiff', np.abs(o - ref).max(), 'first', bad[:2].tolist())
    # decode path
    o1 = ops.flash_attention2(q[-1:], torch.from_numpy(k), torch.from_numpy(v), 1, Sk, Hq, Hkv, D, True).cpu().numpy()
    r1 = orc.attention(q[-1:], k.view(np.uint16), v.view(np.uint16), 1, Sk, Hq, Hkv, D, True)
    print('   decode ndiff', int((o1 != r1).sum()), np.abs(o1 - r1).max())
