import sys, os, ctypes as C, numpy as np, subprocess
sys.path.insert(0, '.')
if len(sys.argv) > 1:
    import torch
    from mllm_amd import lib, synth, weights
    if os.environ.get('DBGSO'): lib.SO_PATH = os.path.abspath(os.environ['DBGSO'])
    cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
    m = lib.Qwen2VL(cfg, path)
    ids = (np.arange(40) * 7919 % 150000).astype(np.int32)
    tok, _, _ = m.prefill(ids, want_logits=False)
    os.environ['MLLM_HIP_TIME_LAYERS'] = '1'
    m.time_gemv(13, 1)
    torch.cuda.synchronize()
    fn = lib.load().mllm_hip_qwen2vl_debug_ptr; fn.restype = C.c_void_p
    p = fn(m._h, C.c_int(5))
    out = torch.empty(cfg.inter, dtype=torch.float32, device='cuda')
    C.cast(0, C.c_void_p)
    torch.cuda.synchronize()
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    hip.hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(p), C.c_size_t(cfg.inter * 4), C.c_int(3))
    np.save(sys.argv[1], out.cpu().numpy())
else:
    e = dict(os.environ); e['MLLM_HIP_NO_GUB'] = '1'
    subprocess.run([sys.executable, __file__, '/tmp/act_old.npy'], env=e, check=True)
    subprocess.run([sys.executable, __file__, '/tmp/act_new.npy'], check=True)
    a, b = np.load('/tmp/act_old.npy'), np.load('/tmp/act_new.npy')
    bad = np.nonzero(a != b)[0]
    print('mismatches', bad.size, 'of', a.size, 'first', bad[:20].tolist())
    for i in bad[:8]: print(i, a[i], b[i])
    if bad.size: print('bad mod 5 hist', np.bincount(bad % 5, minlength=5).tolist(), 'bad//5 %7 hist', np.bincount((bad // 5) % 7, minlength=7).tolist())
