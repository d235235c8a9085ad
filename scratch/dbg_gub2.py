import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '.')
import torch
from mllm_amd import lib, synth, weights, mllmfile as mf
lib.SO_PATH = os.path.abspath(os.environ.get('DBGSO', 'scratch/lib_dbg.so'))
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
m = lib.Qwen2VL(cfg, path)
ids = (np.arange(40) * 7919 % 150000).astype(np.int32)
tok, _, _ = m.prefill(ids, want_logits=False)
os.environ['MLLM_HIP_TIME_LAYERS'] = '1'
m.time_gemv(13, 1)
torch.cuda.synchronize()
lg = np.zeros(2048, dtype=np.float32)
# logits buffer: read through a decode-free path: debug ptr index? use hipMemcpy from the ctx pointer exposed as which=8
fn = lib.load().mllm_hip_qwen2vl_debug_ptr; fn.restype = C.c_void_p
p = fn(m._h, C.c_int(8))
hip = C.CDLL('libamdhip64.so')
hip.hipMemcpy(lg.ctypes.data_as(C.c_void_p), C.c_void_p(p), C.c_size_t(lg.nbytes), C.c_int(2))
d = lg[:1024].reshape(64, 16)
np.set_printoptions(linewidth=200, suppress=True)
print(d[:14])
print('outv', lg[1024:1034])
f = mf.MllmFile(path)
g = np.frombuffer(f.raw('model.layers.0.mlp.gate_proj.weight').tobytes(), dtype=np.uint8)[:144*6]
print('gate row0 block0 hdr d,dmin fp16:', np.frombuffer(g[:4].tobytes(), dtype=np.float16), 'block1:', np.frombuffer(g[144:148].tobytes(), dtype=np.float16))
act = np.zeros(16, dtype=np.float32)
p5 = fn(m._h, C.c_int(5))
hip.hipMemcpy(act.ctypes.data_as(C.c_void_p), C.c_void_p(p5), C.c_size_t(act.nbytes), C.c_int(2))
print('act[:16]', act)
