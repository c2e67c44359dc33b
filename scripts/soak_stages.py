#!/usr/bin/env python3
"""Localise a rare nondeterminism: each stage alone, many launches on fixed inputs, bitwise against the first result.
    PYTHONPATH=. python scripts/soak_stages.py [--seconds 40]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import _native as nat, ops

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=40.0); ap.add_argument("--only", default="")
args = ap.parse_args()
dev = torch.device("cuda", 0)
res = {}
for arch, B in (("simple", 4096), ("simple", 256), ("full", 777)):
    n_conv = 2 if arch == "simple" else 3
    packed = torch.from_numpy(ops.pack_state_dict(pkg.synth.make_state_dict(arch, seed=3))).to(dev)
    pcm = torch.from_numpy(pkg.synth.make_clips_tiled(0, B, unique=64)).to(dev)
    mel = ops.logmel(pcm, True).clone()
    pooled = ops.cnn_pool(mel, packed, n_conv).clone()
    logits = ops.lstm_fc(pooled, packed, n_conv).clone()
    torch.cuda.synchronize()
    stages = {"K1": (lambda: ops.logmel(pcm, True), mel), "K2": (lambda: ops.cnn_pool(mel, packed, n_conv), pooled),
              "K3": (lambda: ops.lstm_fc(pooled, packed, n_conv), logits)}
    for name, (fn, ref) in stages.items():
        if args.only and name not in args.only: continue
        t0, n, bad, worst, where = time.time(), 0, 0, 0.0, []
        while time.time() - t0 < args.seconds:
            ys = [fn() for _ in range(50)]
            for y in ys:
                if not torch.equal(y, ref):
                    bad += 1
                    d = (y - ref).abs().reshape(B, -1).amax(dim=1)
                    idx = torch.nonzero(d > 0).flatten()
                    worst = max(worst, float(d.max()))
                    if len(where) < 6: where.append((idx[:8].tolist(), int(idx.numel())))
            n += len(ys)
        res[f"{arch}-{B}-{name}"] = {"launches": n, "mismatches": bad, "max_abs_diff": worst, "clips": where}
        print(f"{arch}-{B}-{name}", res[f"{arch}-{B}-{name}"], flush=True)
print(json.dumps({"sync_timeouts": int(nat.lib.ww_sync_timeouts())}))
