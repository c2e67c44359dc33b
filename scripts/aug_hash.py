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
args = ap.parse_args()
x = pkg.synth.make_clips_tiled(0, args.batch, unique=64)
x = x / np.abs(x).max(axis=1, keepdims=True)
pcm = torch.from_numpy(x).cuda()
random.seed(0)
proc = pkg.AudioProcessor()
plans = [proc.draw_augment_plan() for _ in range(args.batch)]
out = proc.augment_batch(pcm, plans)
torch.cuda.synchronize()
print(hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest(), os.environ.get("WW_LIB_OVERRIDE", "shipped"))
