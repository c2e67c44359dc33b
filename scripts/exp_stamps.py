import ctypes as C, numpy as np, torch
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import ops, _native as nat
pcm = torch.from_numpy(pkg.synth.make_clips_tiled(0, 4096, unique=64)).cuda()
ops.logmel(pcm, True); torch.cuda.synchronize()
out = (C.c_ulonglong * 16)(); nat.lib._handle  # noqa
lib = C.CDLL(nat.LIB_PATH); lib.ww_debug_stamps(out)
ops.logmel(pcm, True); torch.cuda.synchronize(); lib.ww_debug_stamps(out)
names = ["load+win+pass1", "pass2", "pass3", "split+power", "pieces", "combine", "wait-all-frames", "final log/store", "loop top", "prologue"]
tot = sum(out[i] for i in range(9))
for i, n in enumerate(names): print("%-18s %10d  %5.1f%%" % (n, out[i], 100.0 * out[i] / tot))
print("total cycles (one wave, 8 clips)", tot)
