"""Launch one stage repeatedly (for rocprofv3 --pmc / --kernel-trace passes).  usage: prof_kernel.py logmel|cnn|head [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import ops

what = sys.argv[1] if len(sys.argv) > 1 else "logmel"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
B = 4096
pcm = torch.from_numpy(pkg.synth.make_clips_tiled(0, B, unique=64)).to(dev)
packed = torch.from_numpy(ops.pack_state_dict(pkg.synth.make_state_dict("simple"))).to(dev)
mel = ops.logmel(pcm, True)
pooled = ops.cnn_pool(mel, packed, 2)
torch.cuda.synchronize()
for _ in range(reps):
    if what == "logmel":
        ops.logmel(pcm, True)
    elif what == "cnn":
        ops.cnn_pool(mel, packed, 2)
    else:
        ops.lstm_fc(pooled, packed, 2)
torch.cuda.synchronize()
