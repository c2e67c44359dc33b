#!/usr/bin/env python3
"""Per-clip error of the augmentation against oracle/augment_oracle.py, with each clip's plan: which transform the error of a clip belongs to.

    PYTHONPATH=. python scripts/aug_errors.py [--clips 8]           (WW_LIB_OVERRIDE selects a build)"""
import argparse
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import wakeword_jupyterlab_amd as pkg  # noqa: E402
from oracle import augment_oracle as ao  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=8)
args = ap.parse_args()
x = pkg.synth.make_clips_tiled(0, args.clips, unique=64)
x = x / np.abs(x).max(axis=1, keepdims=True)
random.seed(0)
proc = pkg.AudioProcessor()
plans = [proc.draw_augment_plan() for _ in range(args.clips)]
OFF = {"shift": 0, "n_steps": None, "rate": None, "crop": 0, "sigma": 0.0, "seed": 0}
rows = []
for name, sel in (("all", plans), ("pitch only", [dict(OFF, n_steps=p["n_steps"]) for p in plans]),
                  ("stretch only", [dict(OFF, rate=p["rate"], crop=p["crop"]) for p in plans])):
    got = proc.augment_batch(torch.from_numpy(x).cuda(), sel).cpu().numpy()
    for i in range(args.clips):
        want = ao.augment(x[i], sel[i])
        err = got[i].astype(np.float64) - want
        rows.append({"case": name, "clip": i, "n_steps": sel[i]["n_steps"], "rate": sel[i]["rate"],
                     "max_over_peak": float(np.abs(err).max() / np.abs(want).max()),
                     "rms_over_rms": float(np.sqrt((err ** 2).mean()) / np.sqrt((want ** 2).mean()))})
for r in rows:
    print(json.dumps(r))
