import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '.')
from mllm_amd import lib
lib.SO_PATH = os.path.abspath(os.environ.get('STAMPSO', 'scratch/libmllm_hip_stamps.so'))
from mllm_amd import synth, weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
m = lib.Qwen2VL(cfg, path)
names = 'head,entry,scores,softmax,parked,walk0,walk1,summer'.split(',')
for n in (60, 124, 252, 380, 508):
    m.clear_kvcache()
    ids = (np.arange(n) * 7919 % 150000).astype(np.int32)
    tok, _, _ = m.prefill(ids, want_logits=False)
    gen, _ = m.generate(tok, 4)
    ms, nb = m.time_gemv(11, 56)
    buf = np.zeros(8192*8, dtype=np.uint64)
    assert lib.load().mllm_hip_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size)) == 0
    st = buf.reshape(-1,8)[:12].astype(np.int64)
    rel = (st - st[:,1:2]) * 10.0 / 1000.0
    print('T', n + 4, 'us %.2f ' % (ms*1000), ' '.join('%s %.2f' % (nm, np.median(rel[:, i])) for i, nm in enumerate(names)))
