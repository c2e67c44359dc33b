"""Calibration of the auto mode's rounding-floor test (ww_logmel.hip: kFloorRatio) on the GPU.

For a family of noise-free signals (tones, tone pairs, chirps, impulses, square waves, harmonic stacks with digital silence,
tones over a very low noise floor) run the float32 front end, compare with the oracle (float64 FFT) and bin the error of
every LIVE mel band by r = (P_b / wmax_b) / E_frame, E_frame = sum_b P_b / wmax_b -- the quantity the kernel tests.
Prints, per decade of r, the largest error; the ratio below which errors exceed 1e-4 dB (with margin) is kFloorRatio.
Also reports how many clips each mode leaves above 1e-4 dB and how many clips auto mode redid.

    python scripts/diag_floor.py            (on the GPU box)
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wakeword_jupyterlab_amd as pkg  # noqa: E402
from oracle import mel_oracle  # noqa: E402
from wakeword_jupyterlab_amd import _native as nat  # noqa: E402
from wakeword_jupyterlab_amd import ops  # noqa: E402


def family(seed=0):
    r = np.random.default_rng(seed)
    t = np.arange(16000) / 16000.0
    sig, names = [], []
    def add(name, x):
        names.append(name); sig.append(np.asarray(x, np.float64))
    for f in [60, 100, 220, 440, 997.3, 1500, 2500, 3999.5, 5000, 6500, 7400, 7900]:
        add(f"tone{f}", np.sin(2 * np.pi * f * t))
        add(f"tone{f}_a1e-3", 1e-3 * np.sin(2 * np.pi * f * t + 1.0))
    for _ in range(24):
        f1, f2 = r.uniform(80, 7800, 2)
        add("pair", np.sin(2 * np.pi * f1 * t) + 10 ** r.uniform(-3, 0) * np.sin(2 * np.pi * f2 * t + r.uniform(0, 6)))
    for _ in range(12):
        f0, f1 = sorted(r.uniform(100, 7500, 2))
        add("chirp", np.sin(2 * np.pi * (f0 * t + 0.5 * (f1 - f0) * t * t)))
    for pos in [0, 1, 511, 8000, 15999]:
        add(f"impulse{pos}", (np.arange(16000) == pos).astype(np.float64))
    for f in [50, 100, 333, 1000]:
        add(f"square{f}", np.sign(np.sin(2 * np.pi * f * t)))
    for _ in range(24):                                     # harmonic stacks ("clean TTS") with digital silence around
        f0 = r.uniform(90, 300)
        x = sum((0.7 ** h) * np.sin(2 * np.pi * f0 * (h + 1) * t + r.uniform(0, 6)) for h in range(int(r.integers(3, 25))))
        env = np.zeros(16000)
        a, b = sorted(r.integers(0, 16000, 2))
        env[a:b] = np.hanning(max(2, b - a))
        add("stack", x * env)
    for db in [-40, -50, -60, -70, -80, -90, -100, -120]:    # tone over white noise at a given level
        for f in [300, 3000]:
            add(f"tone{f}_noise{db}", np.sin(2 * np.pi * f * t) + 10 ** (db / 20) * r.standard_normal(16000))
    for _ in range(8):                                      # decaying resonances
        x = sum(np.exp(-t * r.uniform(3, 40)) * np.sin(2 * np.pi * r.uniform(100, 7000) * t) for _ in range(4))
        add("bell", x)
    for i in range(8):
        add("synth", pkg.synth.make_clip(i))
    return names, np.stack(sig).astype(np.float32)


def main():
    dev = torch.device("cuda", 0)
    names, x = family()
    pcm = torch.from_numpy(x).to(dev)
    ref = mel_oracle.logmel_batch(x, normalize=True)[:, 0]                     # [N, 80, 32] dB
    # the kernel's test quantity from the float64 mel powers
    M = mel_oracle.mel_filterbank().astype(np.float64)
    wmax = M.max(axis=1)
    out = {}
    for mode in ("f32", "f64", "auto"):
        ops.set_logmel_math(mode)
        out[mode] = ops.logmel(pcm, True).cpu().numpy()[:, 0]
    ops.set_logmel_math("auto")
    err = {m: np.abs(out[m] - ref) for m in out}
    ratios = np.full(ref.shape, np.nan)
    for i, clip in enumerate(x):
        y = clip.astype(np.float64) / np.abs(clip).max()
        spec = np.fft.rfft(mel_oracle.hann_window()[:, None] * mel_oracle.frame_signal(y), axis=0)
        P = M @ (spec.real ** 2 + spec.imag ** 2)                             # [80, 32]
        u = P / wmax[:, None]
        ratios[i] = u / np.maximum(u.sum(axis=0, keepdims=True), 1e-300)
    live = ref > -80.0
    print("%d clips; per decade of r = (P_b/wmax_b)/E: max |err| dB of live bands in f32 mode" % len(x))
    lr = np.log10(np.maximum(ratios, 1e-30))
    rows = []
    for d in range(-14, 0):
        sel = live & (lr >= d) & (lr < d + 1)
        if sel.any():
            rows.append((d, int(sel.sum()), float(err["f32"][sel].max()), float(err["f64"][sel].max()), float(err["auto"][sel].max())))
            print("  1e%+03d..1e%+03d  n=%7d  f32 %.2e   f64 %.2e   auto %.2e" % (d, d + 1, *rows[-1][1:]))
    per_clip = {m: err[m].max(axis=(1, 2)) for m in err}
    redone = int(sum(not np.array_equal(out["auto"][i], out["f32"][i]) for i in range(len(x))))
    print("clips above 1e-4 dB:  f32 %d   f64 %d   auto %d   (auto redid %d of %d clips)" % (
        (per_clip["f32"] > 1e-4).sum(), (per_clip["f64"] > 1e-4).sum(), (per_clip["auto"] > 1e-4).sum(), redone, len(x)))
    worst = np.argsort(-per_clip["auto"])[:5]
    for i in worst:
        print("  worst auto: %-18s f32 %.2e  f64 %.2e  auto %.2e" % (names[i], per_clip["f32"][i], per_clip["f64"][i], per_clip["auto"][i]))
    # smallest ratio at which f32 is still within 1e-4 everywhere above it
    order = np.argsort(lr[live])
    e_sorted = err["f32"][live][order]
    r_sorted = lr[live][order]
    tail_max = np.maximum.accumulate(e_sorted[::-1])[::-1]                    # max error among bands with ratio >= r
    ok = np.nonzero(tail_max <= 1e-4)[0]
    print("f32 errors stay <= 1e-4 dB for every live band with r >= 1e%.2f" % (r_sorted[ok[0]] if len(ok) else 0.0))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"rows": rows, "redone": redone, "clips": len(x), "above_1e-4": {m: int((per_clip[m] > 1e-4).sum()) for m in per_clip},
               "max_err": {m: float(per_clip[m].max()) for m in per_clip}}, open(os.path.join(ROOT, "gpurun_out", "diag_floor.json"), "w"))


if __name__ == "__main__":
    main()
