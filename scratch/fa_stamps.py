"""phase boundaries of the prefill attention (diagnostic build -DFA_STAMPS via scratch/variant.sh): cycles between the stamps of chunks 4..7, workgroup 0 wave 0, for a launch with
one workgroup per CU (Sq 512) and with two (Sq 1024)"""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from mllm_amd import lib
lib.SO_PATH = os.environ['MLLM_SO']
from mllm_amd import ops
ops.require_gpu()
r = np.random.default_rng(0)
H, D, Sk = 16, 80, 1024
k = torch.from_numpy(r.standard_normal((Sk, H * D)).astype(np.float32)).cuda(); v = torch.from_numpy(r.standard_normal((Sk, H * D)).astype(np.float32)).cuda()
names = ['wait prev PV (barrier)', 'park + barrier', 'fetch issue + scores + Part', 'barrier', 'softmax', 'barrier', 'logsum + P V']
for Sq in (512, 1024):
    q = torch.from_numpy(r.standard_normal((Sq, H * D)).astype(np.float32)).cuda()
    for _ in range(3): ops.flash_attention2(q, k, v, Sq, Sk, H, H, D, False)
    torch.cuda.synchronize()
    buf = np.zeros(64, dtype=np.uint64)
    assert lib.load().mllm_hip_debug_read_fa_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
    full = buf.reshape(4, 16).astype(np.int64)
    st = full[:, :8]
    d = np.diff(st, axis=1)
    print('Sq', Sq, 'cycles per phase (median of chunks 4..7):')
    for i, n in enumerate(names): print('   %-30s %6d' % (n, int(np.median(d[:, i]))))
    print('   chunk total                    %6d' % int(np.median(st[1:, 0] - st[:-1, 0])))
    print('   inside the softmax: fold of the partials %d, tile / prefix maximum %d, expf x 5 %d, stores %d' % tuple(int(np.median(x)) for x in (full[:, 8] - full[:, 4], full[:, 9] - full[:, 8], full[:, 10] - full[:, 9], full[:, 5] - full[:, 10])))
