import ctypes as C, numpy as np, torch, time
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import ops
dev = torch.device("cuda", 0)
packed = torch.from_numpy(ops.pack_state_dict(pkg.synth.make_state_dict("simple"))).to(dev)
for B in (16, 256, 1024, 4096):
    pooled = torch.randn(B, 64, device=dev)
    for _ in range(5): ops.lstm_fc(pooled, packed, 2)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(50): ops.lstm_fc(pooled, packed, 2)
    ev[1].record(); torch.cuda.synchronize()
    print(B, "%.2f us" % (ev[0].elapsed_time(ev[1]) * 1000 / 50))
