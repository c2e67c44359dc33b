import ctypes as C, numpy as np, torch, time
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import ops
dev = torch.device("cuda", 0)
pcm = torch.from_numpy(pkg.synth.make_clips_tiled(0, 4608, unique=64)).to(dev)
for B in (768, 1536, 3072, 3840, 4096, 4608):
    x = pcm[:B]
    for _ in range(3): ops.logmel(x, True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(30): ops.logmel(x, True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 30
    print(B, "%.4f ms  %.2f M clips/s  (%.2f us per block-round)" % (dt * 1e3, B / dt / 1e6, dt * 1e6 / -(-B // 768)))
