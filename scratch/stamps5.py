import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '.')
from mllm_amd import lib
lib.SO_PATH = os.path.abspath(os.environ.get('STAMPSO', 'scratch/libmllm_hip_stamps.so'))
from mllm_amd import synth, weights
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
m = lib.Qwen2VL(cfg, path)
for n in (124, 380):
    m.clear_kvcache()
    ids = (np.arange(n) * 7919 % 150000).astype(np.int32)
    tok, _, _ = m.prefill(ids, want_logits=False)
    gen, _ = m.generate(tok, 4)
    ms, nb = m.time_gemv(11, 56)
    buf = np.zeros(8192*8, dtype=np.uint64)
    assert lib.load().mllm_hip_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size)) == 0
    st = buf.reshape(-1,8)[:12]
    print('T', n + 4, 'groups with a moved max per head:', (st[:, 0] & 0xffffffff).tolist(), 'moved keys:', (st[:, 0] >> 32).tolist())
