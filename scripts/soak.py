#!/usr/bin/env python3
"""Soak test of the counter-synchronised kernels: many launches on fixed inputs, every result compared BITWISE with the first
result for that input (the kernels are deterministic), plus ww_sync_timeouts().  One script, three views:

    --what forward   the whole PCM -> logits forward on five (arch, batch) cases                     (default)
    --what stages    K1, K2, K3 each alone: localises a rare difference to a stage and to clips
    --what train     the training step of both models (split-precision kernels), logits + every gradient
    --what conv3     the 3-conv stack with its HBM intermediate poisoned between runs (0x00 / 0xff / 0x7b): separates
                     "conv2 wrote something else" from "conv3 read something stale", back-to-back launches included

    PYTHONPATH=. python scripts/soak.py [--what forward] [--seconds 120] [--conv-math f16x3|f16x3d|f32]
How the consumer-counter race of round 1 was found."""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import wakeword_jupyterlab_amd as pkg  # noqa: E402
from wakeword_jupyterlab_amd import _native as nat, ops  # noqa: E402

dev = torch.device("cuda", 0)


def soak_forward(seconds):
    cases = []
    for arch, B in (("simple", 4096), ("simple", 257), ("simple", 1000), ("full", 777), ("simple", 256)):
        sd = pkg.synth.make_state_dict(arch, seed=3)
        m = (pkg.SimpleWakewordModel() if arch == "simple" else pkg.WakewordModel())
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m = m.to(dev).eval()
        for s in range(2):
            pcm = torch.from_numpy(pkg.synth.make_clips_tiled(100 * s, B, unique=64)).to(dev)
            with torch.no_grad():
                cases.append((arch, B, m, pcm, m.forward_pcm(pcm).clone()))
    torch.cuda.synchronize()
    t0, launches, bad = time.time(), 0, 0
    while time.time() - t0 < seconds:
        for arch, B, m, pcm, ref in cases:
            with torch.no_grad():
                ys = [m.forward_pcm(pcm) for _ in range(20)]
            bad += sum(not torch.equal(y, ref) for y in ys)
            launches += len(ys)
        print("t=%.0fs forwards=%d mismatches=%d timeouts=%d" % (time.time() - t0, launches, bad, nat.lib.ww_sync_timeouts()), flush=True)
    return {"forwards": launches, "mismatches": bad, "cases": [(a, b) for a, b, *_ in cases]}


def soak_stages(seconds):
    res, bad_total = {}, 0
    for arch, B in (("simple", 4096), ("simple", 256), ("full", 777)):
        n_conv = 2 if arch == "simple" else 3
        packed = torch.from_numpy(ops.pack_state_dict(pkg.synth.make_state_dict(arch, seed=3))).to(dev)
        pcm = torch.from_numpy(pkg.synth.make_clips_tiled(0, B, unique=64)).to(dev)
        mel = ops.logmel(pcm, True).clone()
        pooled = ops.cnn_pool(mel, packed, n_conv).clone()
        logits = ops.lstm_fc(pooled, packed, n_conv).clone()
        stages = {"K1": (lambda: ops.logmel(pcm, True), mel), "K2": (lambda: ops.cnn_pool(mel, packed, n_conv), pooled),
                  "K3": (lambda: ops.lstm_fc(pooled, packed, n_conv), logits)}
        for name, (fn, ref) in stages.items():
            t0, n, bad, worst, where = time.time(), 0, 0, 0.0, []
            while time.time() - t0 < seconds / 9:
                for y in [fn() for _ in range(50)]:
                    if not torch.equal(y, ref):
                        bad += 1
                        d = (y - ref).abs().reshape(B, -1).amax(dim=1)
                        worst = max(worst, float(d.max()))
                        if len(where) < 6:
                            where.append(torch.nonzero(d > 0).flatten()[:8].tolist())
                n += 50
            res[f"{arch}-{B}-{name}"] = {"launches": n, "mismatches": bad, "max_abs_diff": worst, "clips": where}
            bad_total += bad
            print(f"{arch}-{B}-{name}", res[f"{arch}-{B}-{name}"], flush=True)
    return {"stages": res, "mismatches": bad_total}


def soak_conv3(seconds):
    B = 777
    packed = torch.from_numpy(ops.pack_state_dict(pkg.synth.make_state_dict("full", seed=3))).to(dev)
    mel = ops.logmel(torch.from_numpy(pkg.synth.make_clips_tiled(0, B, unique=64)).to(dev), True).clone()
    scratch = torch.zeros(nat.lib.ww_cnn_scratch_bytes(B, 3), dtype=torch.uint8, device=dev)
    pooled = torch.empty(B, 128, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

    def run(fill=None, sync=True):
        if fill is not None:
            scratch.fill_(fill)
        nat.check(nat.lib.ww_cnn_pool_f32(p(mel), B, 32, p(packed), 3, p(scratch), p(pooled), None))
        if sync:
            torch.cuda.synchronize()
    written = B * 655360 + B * 4                     # relu(conv2) + one scale float per clip; the 256-byte padding behind is never written
    run(0)
    ref_s, ref_p = scratch[:written].clone(), pooled.clone()
    out = {"poisoned_runs": 0, "scratch_mismatch": 0, "pooled_mismatch": 0, "back_to_back": 0, "back_to_back_mismatch": 0}
    fills, t0 = [0, 255, 0x7B], time.time()
    while time.time() - t0 < seconds / 2:
        run(fills[out["poisoned_runs"] % 3])
        out["poisoned_runs"] += 1
        out["scratch_mismatch"] += int(not torch.equal(scratch[:written], ref_s))
        out["pooled_mismatch"] += int(not torch.equal(pooled, ref_p))
    t0 = time.time()
    while time.time() - t0 < seconds / 2:
        outs = []
        for _ in range(25):
            run(sync=False)
            outs.append(pooled.clone())
        torch.cuda.synchronize()
        out["back_to_back"] += 25
        out["back_to_back_mismatch"] += sum(not torch.equal(o, ref_p) for o in outs)
    out["mismatches"] = out["scratch_mismatch"] + out["pooled_mismatch"] + out["back_to_back_mismatch"]
    return out


def soak_train(seconds):
    """The training step (split-precision kernels: barrier-per-step rings, wave-pair exchanges, global read-modify-write partials):
    the same step with the same dropout seed over and over, logits and every gradient compared bitwise with the first run."""
    import torch.nn.functional as F
    cases = []
    for arch, B, T in (("simple", 300, 32), ("simple", 4096, 32), ("full", 260, 31), ("full", 1024, 32), ("simple", 37, 31)):
        sd = pkg.synth.make_state_dict(arch, seed=7)
        m = (pkg.SimpleWakewordModel() if arch == "simple" else pkg.WakewordModel())
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m = m.to(dev).train()
        x = (torch.randn(B, 1, 80, T, device=dev) * 15 - 35).clamp_(-80, 0)
        y = torch.randint(0, 2, (B,), device=dev)

        def step(m=m, x=x, y=y):
            m.zero_grad()
            torch.manual_seed(5)
            out = m(x)
            F.cross_entropy(out, y).backward()
            return [out.detach().clone()] + [p.grad.clone() for p in m.parameters()]
        cases.append((arch, B, step, step()))
    torch.cuda.synchronize()
    t0, steps, bad = time.time(), 0, 0
    while time.time() - t0 < seconds:
        for arch, B, step, ref in cases:
            for _ in range(5):
                got = step()
                bad += any(not torch.equal(a, b) for a, b in zip(got, ref))
                steps += 1
        print("t=%.0fs steps=%d mismatches=%d" % (time.time() - t0, steps, bad), flush=True)
    return {"train_steps": steps, "mismatches": bad, "cases": [(a, b) for a, b, *_ in cases], "train_math": ops.get_train_math()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="forward", choices=["forward", "stages", "conv3", "train"])
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--conv-math", default=None, choices=["f32", "f16x3", "f16x3d"])
    args = ap.parse_args()
    if args.conv_math:
        ops.set_conv_math(args.conv_math)
    t0 = time.time()
    res = {"forward": soak_forward, "stages": soak_stages, "conv3": soak_conv3, "train": soak_train}[args.what](args.seconds)
    res.update(what=args.what, conv_math=ops.get_conv_math(), seconds=time.time() - t0, sync_timeouts=int(nat.lib.ww_sync_timeouts()))
    print(json.dumps(res))
    sys.exit(1 if res["mismatches"] or res["sync_timeouts"] else 0)


if __name__ == "__main__":
    main()
