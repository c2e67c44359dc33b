#!/usr/bin/env python3
"""KA: throughput of AudioProcessor.augment_audio on the GPU, and its error against the oracle.

    PYTHONPATH=. python scripts/bench_augment.py [--batch 4096] [--steps 10]
Plans are drawn like the reference does (each transform with probability 0.8); the same plans every step."""
import argparse
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import wakeword_jupyterlab_amd as pkg  # noqa: E402
from wakeword_jupyterlab_amd import ops  # noqa: E402


def measure(batch=4096, steps=10, check=8, device=0):
    """Run the augmentation benchmark and return its result dict (bench.py embeds it in its JSON line at N=1)."""
    args = argparse.Namespace(batch=batch, steps=steps, check=check)
    from oracle import augment_oracle as ao
    dev = torch.device("cuda", device)
    B = args.batch
    x = pkg.synth.make_clips_tiled(0, B, unique=64)
    x = x / np.abs(x).max(axis=1, keepdims=True)
    pcm = torch.from_numpy(x).to(dev)
    random.seed(0)
    proc = pkg.AudioProcessor()
    plans = [proc.draw_augment_plan() for _ in range(B)]          # the product's own draws; the oracle below only CHECKS a few clips
    import ctypes as C
    from wakeword_jupyterlab_amd import _native as nat
    arr = (nat.AugmentPlan * B)()
    for a, p in zip(arr, plans):
        a.shift, a.crop_start = p["shift"], p["crop"]
        a.pitch_rate = 2.0 ** (-p["n_steps"] / 12.0) if p["n_steps"] is not None else 0.0
        a.stretch_rate = p["rate"] or 0.0
        a.noise_sigma, a.noise_seed = p["sigma"], p["seed"]
    out = ops.augment(pcm, arr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = ops.augment(pcm, arr)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    # + log-mel, the stage it feeds
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mel = ops.logmel(ops.augment(pcm, arr), False)
    torch.cuda.synchronize()
    dt_mel = (time.perf_counter() - t0) / args.steps
    got = out[:args.check].cpu().numpy()
    t0 = time.perf_counter()
    want = np.stack([ao.augment(x[i], plans[i]) for i in range(args.check)])
    cpu_s = (time.perf_counter() - t0) / max(1, args.check)
    err = np.abs(got.astype(np.float64) - want)
    res = {"workload": f"augment_audio, {B} clips of 1 s, plans drawn with p=0.8 per transform "
                       f"({sum(p['n_steps'] is not None for p in plans)} pitch, {sum(p['rate'] is not None for p in plans)} stretch)",
           "ms_per_batch": dt * 1e3, "clips_per_s": B / dt, "clips_per_s_with_logmel": B / dt_mel,
           "oracle_clips_per_s_1_thread": 1.0 / cpu_s, "checked_clips": args.check,
           "max_abs_err_over_peak": float((err.max(axis=1) / np.abs(want).max(axis=1)).max()),
           "rms_err_over_rms": float((np.sqrt((err ** 2).mean(axis=1)) / np.sqrt((want.astype(np.float64) ** 2).mean(axis=1))).max())}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--check", type=int, default=8, help="clips compared with oracle/augment_oracle.py")
    args = ap.parse_args()
    print(json.dumps(measure(args.batch, args.steps, args.check)))


if __name__ == "__main__":
    main()
