import ctypes as C, numpy as np, torch
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import ops, _native as nat
pcm = torch.from_numpy(pkg.synth.make_clips_tiled(0, 4096, unique=64)).cuda()
lib = C.CDLL(nat.LIB_PATH)
out = (C.c_ulonglong * 16)()
mel = ops.logmel(pcm, True); torch.cuda.synchronize(); lib.ww_debug_stamps(out)
ops.logmel(pcm, True); torch.cuda.synchronize(); lib.ww_debug_stamps(out)
names = ["load+win+pass1", "pass2", "pass3", "split+power(+prefetch issue)", "pieces", "combine", "wait-all-frames", "final log/store", "loop top"]
tot = sum(out[i] for i in range(9))
print("K1 (wave 1 of block 7): total", tot)
for i, n in enumerate(names): print("  %-30s %10d  %5.1f%%" % (n, out[i], 100.0 * out[i] / tot))
packed = torch.from_numpy(ops.pack_state_dict(pkg.synth.make_state_dict("simple"))).cuda()
ops.cnn_pool(mel, packed, 2); torch.cuda.synchronize(); lib.ww_debug_cnn_stamps(out)
ops.cnn_pool(mel, packed, 2); torch.cuda.synchronize(); lib.ww_debug_cnn_stamps(out)
cn = ["loop top", "MFMA stream", "epilogue", "(producer work)", "barrier wait"]
for who, off in (("consumer wave 1", 0), ("producer wave 9", 8)):
    tot = sum(out[off + i] for i in range(5))
    print("K2", who, "total", tot)
    for i, n in enumerate(cn): print("  %-30s %10d  %5.1f%%" % (n, out[off + i], 100.0 * out[off + i] / max(1, tot)))
