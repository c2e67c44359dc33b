#!/usr/bin/env python3
"""sha256 of the augmentation's output for fixed clips and plans: run under two builds (WW_LIB_OVERRIDE) to see whether a kernel change is bit-neutral.

    PYTHONPATH=. python scripts/aug_hash.py [--batch 512]"""
import argparse
import hashlib
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import wakeword_jupyterlab_amd as pkg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--repeat", type=int, default=1, help="run the same augmentation this many times; every run must give the same bits")
args = ap.parse_args()
x = pkg.synth.make_clips_tiled(0, args.batch, unique=64)
x = x / np.abs(x).max(axis=1, keepdims=True)
pcm = torch.from_numpy(x).cuda()
random.seed(0)
proc = pkg.AudioProcessor()
plans = [proc.draw_augment_plan() for _ in range(args.batch)]
out = proc.augment_batch(pcm, plans)
torch.cuda.synchronize()
bad = 0
for _ in range(args.repeat - 1):
    bad += int(not torch.equal(proc.augment_batch(pcm, plans), out))
if args.repeat > 1:
    print(f"{args.repeat} runs, {bad} differing from the first")
print(hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest(), os.environ.get("WW_LIB_OVERRIDE", "shipped"))
