#!/usr/bin/env python3
"""Soak test of the counter-synchronised conv kernel (and the whole forward): many launches on rotating inputs, every
result compared BITWISE with the first result for that input (the kernels are deterministic), plus ww_sync_timeouts().
    PYTHONPATH=. python scripts/soak.py [--seconds 120]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import wakeword_jupyterlab_amd as pkg  # noqa: E402
from wakeword_jupyterlab_amd import _native as nat, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cases = []
    for arch, B in (("simple", 4096), ("simple", 257), ("simple", 1000), ("full", 777), ("simple", 256)):
        sd = pkg.synth.make_state_dict(arch, seed=3)
        m = (pkg.SimpleWakewordModel() if arch == "simple" else pkg.WakewordModel())
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m = m.to(dev).eval()
        for s in range(2):
            pcm = torch.from_numpy(pkg.synth.make_clips_tiled(100 * s, B, unique=64)).to(dev)
            with torch.no_grad():
                ref = m.forward_pcm(pcm).clone()
            cases.append((arch, B, m, pcm, ref))
    torch.cuda.synchronize()
    t0, launches, bad = time.time(), 0, 0
    per_case = [0] * len(cases)
    while time.time() - t0 < args.seconds:
        for i, (arch, B, m, pcm, ref) in enumerate(cases):
            with torch.no_grad():
                ys = [m.forward_pcm(pcm) for _ in range(20)]
            for y in ys:
                if not torch.equal(y, ref):
                    bad += 1
            launches += len(ys)
            per_case[i] += len(ys)
        print("t=%.0fs forwards=%d mismatches=%d timeouts=%d" % (time.time() - t0, launches, bad, nat.lib.ww_sync_timeouts()), flush=True)
    print(json.dumps({"seconds": time.time() - t0, "forwards": launches, "mismatches": bad, "sync_timeouts": int(nat.lib.ww_sync_timeouts()),
                      "cases": [(a, b) for a, b, *_ in cases]}))
    sys.exit(1 if bad or nat.lib.ww_sync_timeouts() else 0)


if __name__ == "__main__":
    main()
