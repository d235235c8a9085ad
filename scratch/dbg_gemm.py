import sys, numpy as np
sys.path.insert(0, '.')
import torch
from mllm_amd import lib, ops
from oracle import oracle as orc
ops.require_gpu()
for (M,K,N) in [(100,8960,192),(282,1536,2048),(64,1536,256),(128,1536,256)]:
    r = np.random.default_rng(M+K+N)
    W = (r.standard_normal((N, K)) * 0.05).astype(np.float32)
    x = r.standard_normal((M, K)).astype(np.float32)
    Wq = lib.quantize_host(lib.Q4_K, W)
    y = ops.linear_q4k(Wq, x, N).cpu().numpy()
    ref = orc.linear(x, Wq, orc.Q4_K, N)
    err = np.abs(y-ref)
    print((M,K,N), 'max', err.max())
    print(' per 16-row max:', [float('%.1e'%err[i:i+16].max()) for i in range(0,M,16)])
    print(' per 32-col max:', [float('%.1e'%err[:,i:i+32].max()) for i in range(0,N,32)])
    bad = np.argwhere(err > 1e-4)
    print(' n bad', len(bad), bad[:10].tolist())
M,K,N=128,1536,256
r = np.random.default_rng(M+K+N)
W = (r.standard_normal((N, K)) * 0.05).astype(np.float32)
x = r.standard_normal((M, K)).astype(np.float32)
q = ops.quantize_q8k(x)
blocks = orc.quantize_q8_K(x).reshape(M, K // 256, 292)
d = blocks[:, :, :4].copy().view(np.float32).reshape(M, K // 256)
qs = blocks[:, :, 4:260].reshape(M, K).view(np.int8)
gq = q.qs.cpu().numpy(); gd=q.d.cpu().numpy()
print('qs diff', np.argwhere(gq!=qs)[:10].tolist(), 'd diff', np.argwhere(gd!=d)[:10].tolist())
row=44
xq8 = ops.Q8K(1,K); xq8.qs=q.qs[row:row+1].contiguous(); xq8.d=q.d[row:row+1].contiguous(); xq8.bsums=q.bsums[row:row+1].contiguous()
Wq = lib.quantize_host(lib.Q4_K, W)
yv = ops.linear_q4k(Wq, None, N, xq=xq8).cpu().numpy()
ref = orc.linear(x[row:row+1], Wq, orc.Q4_K, N)
print('gemv row err', np.abs(yv-ref).max())
print('row qs min/max', gq[row].min(), gq[row].max(), 'count -128:', (gq[row]==-128).sum(), 'other rows count -128 per row (first 50):', [(gq[i]==-128).sum() for i in range(40,50)])
