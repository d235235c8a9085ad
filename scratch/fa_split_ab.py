"""(needs the role-split kernel of scratch/fa2_prefill_split_kernel.hip.txt built in and an option "fa_split" selecting it: not in the tree)
 prefill attention: the four-phase kernel against the role-split one (option fa_split), bit for bit and timed; vision-block shape, ragged and causal shapes"""
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from mllm_amd import lib, ops
ops.require_gpu()
r = np.random.default_rng(0)
def run(Sq, Sk, H, Hkv, D, causal, reps=10):
    q = torch.from_numpy(r.standard_normal((Sq, H * D)).astype(np.float32)).cuda()
    k = torch.from_numpy(r.standard_normal((Sk, Hkv * D)).astype(np.float32)).cuda(); v = torch.from_numpy(r.standard_normal((Sk, Hkv * D)).astype(np.float32)).cuda()
    outs = {}; ts = {}
    for mode in (0, 1, 0, 1):
        lib.set_option("fa_split", mode)
        for _ in range(3): o = ops.flash_attention2(q, k, v, Sq, Sk, H, Hkv, D, causal)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): o = ops.flash_attention2(q, k, v, Sq, Sk, H, Hkv, D, causal)
        e1.record(); torch.cuda.synchronize()
        outs[mode] = o.cpu().numpy(); ts.setdefault(mode, []).append(e0.elapsed_time(e1) * 1000 / reps)
    print('Sq %4d Sk %4d H %2d/%2d D %3d causal %d: four-phase %s us, split %s us, bit-equal %s' % (Sq, Sk, H, Hkv, D, causal, ['%.1f' % t for t in ts[0]], ['%.1f' % t for t in ts[1]],
          np.array_equal(outs[0], outs[1])), flush=True)
for shape in [(32, 1024, 16, 16, 80, False), (1024, 1024, 16, 16, 80, False), (2048, 1024, 16, 16, 80, False), (4096, 1024, 16, 16, 80, False), (282, 282, 12, 2, 128, True), (197, 197, 12, 12, 64, False),
              (577, 577, 16, 16, 64, False), (37, 101, 4, 2, 16, True), (5, 7, 2, 1, 64, True), (64, 33, 2, 2, 80, False), (100, 260, 4, 4, 128, True)]:
    run(*shape)
lib.set_option("fa_split", -1)
