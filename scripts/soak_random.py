#!/usr/bin/env python3
"""Randomised soak: random architecture / batch size / mel width / conv math, every forward run three times (bitwise equal)
and, for small batches, checked against the CPU oracle.  PYTHONPATH=. python scripts/soak_random.py [--seconds 240]"""
import argparse, json, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import wakeword_jupyterlab_amd as pkg
from oracle import model_oracle
from wakeword_jupyterlab_amd import _native as nat, ops

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=240.0); args = ap.parse_args()
dev = torch.device("cuda", 0)
rng = random.Random(12345)
models = {}
for arch in ("simple", "full"):
    sd = pkg.synth.make_state_dict(arch, seed=9)
    m = (pkg.SimpleWakewordModel() if arch == "simple" else pkg.WakewordModel())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); models[arch] = (m.to(dev).eval(), sd)
t0, trials, bad_rep, bad_par, worst = time.time(), 0, 0, 0, 0.0
while time.time() - t0 < args.seconds:
    arch = rng.choice(["simple", "simple", "full"])
    n = rng.choice([1, 2, 3, 17, 63, 64, 255, 256, 257, 300, 511, 777, 1024, 1500, 4096 if arch == "simple" else 900])
    width = rng.choice([32, 32, 32, 31, 17, 5])
    ops.set_conv_math(rng.choice(["f16x3", "f16x3", "f16x3d", "f32"]))
    m, sd = models[arch]
    x = torch.from_numpy((pkg.synth.normal(rng.randrange(1 << 20), n * 80 * width).astype(np.float32).reshape(n, 1, 80, width) * 15 - 35)).to(dev)
    with torch.no_grad():
        ys = [m(x).clone() for _ in range(3)]
    if not (torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])):
        bad_rep += 1
        print("NOT REPEATABLE", arch, n, width, ops.get_conv_math(), flush=True)
    if n <= 64:
        with torch.no_grad():
            ref = model_oracle.torch_module_from_state_dict(sd)(x.cpu()).numpy()
        err = float(np.abs(ys[0].cpu().numpy() - ref).max()); worst = max(worst, err)
        if err > 1e-3:
            bad_par += 1
            print("PARITY", arch, n, width, ops.get_conv_math(), err, flush=True)
    trials += 1
ops.set_conv_math("f16x3")
out = {"seconds": time.time() - t0, "trials": trials, "not_repeatable": bad_rep, "parity_failures": bad_par, "worst_logit_err": worst,
       "sync_timeouts": int(nat.lib.ww_sync_timeouts())}
print(json.dumps(out))
sys.exit(1 if bad_rep or bad_par or out["sync_timeouts"] else 0)
