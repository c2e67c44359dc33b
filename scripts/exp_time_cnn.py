import numpy as np, torch, time, os
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import ops
dev = torch.device("cuda", 0)
sd = pkg.synth.make_state_dict("simple", seed=1234)
packed = torch.from_numpy(ops.pack_state_dict(sd)).to(dev)
big = (torch.rand(4096, 1, 80, 32, device=dev) * -60.0)
for mode in ("f32", "f16x3"):
    ops.set_conv_math(mode)
    for _ in range(3): ops.cnn_pool(big, packed, 2)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): ops.cnn_pool(big, packed, 2)
    torch.cuda.synchronize(); print(os.environ.get("WW_LIB_OVERRIDE","default"), mode, "cnn_pool 4096 clips: %.3f ms" % ((time.perf_counter()-t)*50))
