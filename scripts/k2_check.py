"""Quick GPU check of the conv kernel under the default arithmetic: pooled features against the float64 oracle for a few clips and both
widths, a whole batch against the per-clip results, bounded-wait expiries, then the time per 4096 clips.
usage: k2_check.py [n_small] [n_big]"""
import ctypes as C, sys, time
import numpy as np, torch
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import _native as nat, ops
from oracle import model_oracle

n_small = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_big = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda", 0)
sd = pkg.synth.make_state_dict("simple", seed=1234)
model = pkg.SimpleWakewordModel()
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
model = model.to(dev).eval()
packed = model.packed_weights()


def timeouts():
    c = C.c_uint(0)
    nat.lib.ww_sync_timeouts(C.byref(c))
    return c.value


pcm = torch.from_numpy(pkg.synth.make_clips(0, n_small)).to(dev)
mel = ops.logmel(pcm, True)
for width in (32, 31, 7):
    m = mel[..., :width].contiguous()
    pooled = ops.cnn_pool(m, packed, 2)
    torch.cuda.synchronize()
    ref = model_oracle.pooled_features_np(m.cpu().numpy().astype(np.float64), sd)
    err = np.abs(pooled.cpu().numpy() - ref).max() / np.abs(ref).max()
    print(f"width {width}: n {n_small} pooled rel err {err:.3e}; expired waits {timeouts()}", flush=True)
pcm = torch.from_numpy(pkg.synth.make_clips_tiled(0, n_big, unique=64)).to(dev)
mel = ops.logmel(pcm, True)
a = ops.cnn_pool(mel, packed, 2)
b = torch.cat([ops.cnn_pool(mel[i:i + 37].contiguous(), packed, 2) for i in range(0, 37 * 4, 37)])
torch.cuda.synchronize()
print("batch vs small launches bitwise:", bool((a[:37 * 4] == b).all()), "tiled clips repeat bitwise:", bool((a[:64] == a[64:128]).all()),
      "expired waits", timeouts(), flush=True)
for _ in range(3):
    ops.cnn_pool(mel, packed, 2)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    t = time.perf_counter()
    for _ in range(10):
        ops.cnn_pool(mel, packed, 2)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t) * 100)
print(f"cnn_pool {n_big} clips: median {np.median(ts):.4f} ms, min {min(ts):.4f} ms", flush=True)
