import sys, numpy as np, torch
sys.path.insert(0, '.')
from mllm_amd import ops
from oracle import oracle as orc
ops.require_gpu()
z = np.load('scratch/dbg_attn1.npz')
q, k16, v16, o_ref = z['q'], z['k16'], z['v16'], z['o']
S = q.shape[0]
def run(q, k16, v16, tag):
    o = ops.flash_attention2(q, torch.from_numpy(k16.view(np.float16)), torch.from_numpy(v16.view(np.float16)), S, S, 12, 2, 128, True).cpu().numpy()
    ref = orc.attention(q, k16, v16, S, S, 12, 2, 128, True)
    bad = np.argwhere(o != ref)
    print(tag, 'ndiff', len(bad), 'max', np.abs(o - ref).max(), 'first', bad[:3].tolist(), 'ref==stage', np.array_equal(ref, o_ref))
    return o, ref
o, ref = run(q, k16, v16, 'orig')
sub = np.argwhere((k16 & 0x7c00) == 0); print('subnormal at', sub.tolist(), hex(int(k16[tuple(sub[0])])))
k2 = k16.copy(); k2[(k2 & 0x7c00) == 0] = 0
run(q, k2, v16, 'flushed')
# per-head check
bad = np.argwhere(o != ref)
heads = sorted(set((bad[:, 1] // 128).tolist())); rows = sorted(set(bad[:, 0].tolist()))
print('bad heads', heads, 'bad rows', rows[:10], '...')
