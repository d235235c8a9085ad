"""Per-wave timeline of gemm_q4k_kernel from the stamp build (scratch/stamps.sh): python scratch/stamps_gemm.py M N K"""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from mllm_amd import lib
lib.SO_PATH = os.path.abspath('scratch/libmllm_hip_stamps.so')
from mllm_amd import ops
ops.require_gpu()
L = lib.load()
M, N, K = (int(a) for a in sys.argv[1:4])
r = np.random.default_rng(0)
W = lib.quantize_host(lib.Q4_K, (r.standard_normal((N, K)) * 0.05).astype(np.float32))
Wd = torch.from_numpy(W.view(np.uint8)).cuda()
L.mllm_hip_q4k_wpack_bytes.restype = C.c_size_t
wp = torch.empty(L.mllm_hip_q4k_wpack_bytes(C.c_int(N), C.c_int(K)), dtype=torch.uint8, device='cuda')
xp = torch.empty(L.mllm_hip_q4k_prepack_bytes(C.c_int(M), C.c_int(K)), dtype=torch.uint8, device='cuda')
lib.check(L.mllm_hip_q4k_prepack(C.c_void_p(Wd.data_ptr()), C.c_int(N), C.c_int(K), C.c_void_p(wp.data_ptr()), None))
x = torch.from_numpy(r.standard_normal((M, K)).astype(np.float32)).cuda()
q = ops.quantize_q8k(x)
y = torch.empty((M, N), dtype=torch.float32, device='cuda')
for _ in range(3):
    lib.check(L.mllm_hip_linear_q4kp_q8k(C.c_void_p(wp.data_ptr()), None, C.c_void_p(q.qs.data_ptr()), C.c_void_p(q.d.data_ptr()), C.c_void_p(q.bsums.data_ptr()), C.c_void_p(xp.data_ptr()),
              C.c_void_p(y.data_ptr()), C.c_int(lib.F32), C.c_int64(N), None, C.c_int(M), C.c_int(N), C.c_int(K), None))
torch.cuda.synchronize()
HS, NS, NWG = 12, 8, 64
rec = 4 * HS * NS + 8
buf = np.zeros(NWG * rec, dtype=np.uint64)
assert L.mllm_hip_debug_read_gemm_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size)) == 0
buf = buf.reshape(NWG, rec)
nwg_total = ((N + 63) // 64) * ((M + 31) // 32)
nrec = min(NWG, (nwg_total + 36) // 37)
t00 = int(buf[:nrec, rec - 8].min())
names = ['top', 'vmcnt', 'barrier', 'dma issued', 'lds+retire', 'expand', 'mfma issued']
seg = np.zeros((nrec, 4, HS, NS - 1))
for g in range(nrec):
    st = buf[g, :4 * HS * NS].reshape(4, HS, NS).astype(np.int64)
    rt = [int(v) for v in buf[g, rec - 8:rec - 2]]
    hwid, xcc = int(buf[g, rec - 2]), int(buf[g, rec - 1])
    nh = min(HS, 2 * (K // 256))
    print('wg %4d  start %7.2f us | prologue issued +%5.2f  first slot +%5.2f  loop %6.2f  exchange +%5.2f  stores acked +%5.2f | life %6.2f us  xcc %d se %d cu %2d' % (
        g * 37, (rt[0] - t00) / 100.0, (rt[1] - rt[0]) / 100.0, (rt[2] - rt[1]) / 100.0, (rt[3] - rt[2]) / 100.0, (rt[4] - rt[3]) / 100.0, (rt[5] - rt[4]) / 100.0,
        (rt[5] - rt[0]) / 100.0, xcc & 15, (hwid >> 13) & 7, (hwid >> 8) & 15))
    for w in range(4):
        for hs in range(nh):
            for i in range(6):
                seg[g, w, hs, i] = st[w, hs, i + 1] - st[w, hs, i]
            if hs + 1 < nh: seg[g, w, hs, 6] = st[w, hs + 1, 0] - st[w, hs, 0]
nh = min(HS, 2 * (K // 256))
print('cycles per segment, median over recorded workgroups and waves (half-steps 3..%d):' % (nh - 2))
ss = seg[:, :, 3:nh - 1, :]
for i, n in enumerate(['wait vmcnt', 'wait barrier', 'dma issue', 'lds req + retire (+dd)', 'expand', 'mfma issue', 'whole half-step']):
    v = ss[..., i].ravel()
    print('  %-24s median %6.0f  p10 %6.0f  p90 %6.0f' % (n, np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
for hb in (0, 1):
    v = seg[:, :, 4 + hb:nh - 1:2, :]
    print('  hb=%d: ' % hb + '  '.join('%s %.0f' % (n, np.median(v[..., i])) for i, n in enumerate(['vm', 'bar', 'dma', 'lds+ret', 'exp', 'mfma', 'step'])))
