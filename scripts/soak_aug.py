import random, sys, os, time
import numpy as np, torch
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import ops
from oracle import augment_oracle as ao
dev = torch.device("cuda", 0)
B = 1024
x = pkg.synth.make_clips_tiled(0, B, unique=64); x = x / np.abs(x).max(axis=1, keepdims=True)
pcm = torch.from_numpy(x).to(dev)
rng = random.Random(1); plans = [ao.draw_plan(rng) for _ in range(B)]
ref = ops.augment(pcm, plans).clone()
bad = 0; t0 = time.time(); n = 0
while time.time() - t0 < 60:
    outs = [ops.augment(pcm, plans) for _ in range(10)]
    bad += sum(not torch.equal(o, ref) for o in outs); n += 10
print("augment: runs", n, "mismatches", bad)
# K1 small batches (8-wave form) and odd sizes
for Bn in (1, 37, 255, 256):
    p = pcm[:Bn].contiguous(); r = ops.logmel(p, True).clone(); bad = 0
    for _ in range(400):
        outs = [ops.logmel(p, True) for _ in range(50)]
        bad += sum(not torch.equal(o, r) for o in outs)
    print("K1 B=%d: 20000 runs, mismatches %d" % (Bn, bad))
