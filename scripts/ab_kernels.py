"""A/B timing of one C-ABI entry point across several builds of the library, interleaved in one process
(cdna_hip_programming.md rule 24).  usage: ab_kernels.py logmel|cnn|cnn3|head libA.so libB.so ..."""
import ctypes as C, sys, time
import numpy as np, torch
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import ops
what, libs = sys.argv[1], sys.argv[2:]
dev = torch.device("cuda", 0)
B = 4096
pcm = torch.from_numpy(pkg.synth.make_clips_tiled(0, B, unique=64)).to(dev)
mel = ops.logmel(pcm, True)
packed = torch.from_numpy(ops.pack_state_dict(pkg.synth.make_state_dict("simple"))).to(dev)
pooled = ops.cnn_pool(mel, packed, 2); out = torch.empty_like(mel); lg = torch.empty(B, 2, device=dev)
if what == "cnn3":       # the 3-conv WakewordModel's conv stack
    from wakeword_jupyterlab_amd import _native as nat
    packed3 = torch.from_numpy(ops.pack_state_dict(pkg.synth.make_state_dict("full"))).to(dev)
    scratch = torch.empty(nat.lib.ww_cnn_scratch_bytes(B, 3), dtype=torch.uint8, device=dev)
    pooled3 = torch.empty(B, 128, device=dev)
hs = []
for path in libs:
    h = C.CDLL(path); h.ww_init(); hs.append(h)
def run(h):
    if what == "logmel":
        rc = h.ww_logmel_f32(C.c_void_p(pcm.data_ptr()), C.c_int64(B), C.c_int64(16000), C.c_int64(16000), 1, C.c_void_p(out.data_ptr()), None)
    elif what == "head":
        rc = h.ww_lstm_fc_f32(C.c_void_p(pooled.data_ptr()), C.c_int64(B), C.c_void_p(packed.data_ptr()), 2, C.c_void_p(lg.data_ptr()), None)
    elif what == "cnn3":
        rc = h.ww_cnn_pool_f32(C.c_void_p(mel.data_ptr()), C.c_int64(B), 32, C.c_void_p(packed3.data_ptr()), 3, C.c_void_p(scratch.data_ptr()), C.c_void_p(pooled3.data_ptr()), None)
    else:
        rc = h.ww_cnn_pool_f32(C.c_void_p(mel.data_ptr()), C.c_int64(B), 32, C.c_void_p(packed.data_ptr()), 2, None, C.c_void_p(pooled.data_ptr()), None)
    assert rc == 0, rc
res = {p: [] for p in libs}
for path, h in zip(libs, hs):
    for _ in range(3): run(h)
    torch.cuda.synchronize()
    if what == "logmel":      # every build must reproduce the shipped library's result
        print("%-40s max |diff| vs shipped %.3g" % (path.split("/")[-1], float((out - mel.view_as(out)).abs().max())))
        out.zero_()
for rnd in range(12):
    for p, h in zip(libs, hs):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): run(h)
        torch.cuda.synchronize(); res[p].append((time.perf_counter() - t) * 100)
for p in libs:
    v = np.array(res[p]); print("%-40s median %.4f ms  min %.4f ms" % (p.split("/")[-1], np.median(v), v.min()))
