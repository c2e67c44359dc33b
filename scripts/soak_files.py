#!/usr/bin/env python3
"""Soak of the file-fed path: WakewordDataset.loader(shuffle=True) epochs over written WAV files for N seconds; every item's log-mel must
equal, bit for bit, what the first (unshuffled) epoch produced for that file (reader threads, staging slots, the read-ahead thread, K0 and
K1 are deterministic).  --mixed: every third file at 48 / 44.1 / 8 / 22.05 kHz (both K0 kernels inside one batch).
PYTHONPATH=. python scripts/soak_files.py [--seconds 60] [--files 1500] [--batch 96] [--mixed]"""
import argparse, json, os, shutil, struct, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd.audio import AudioProcessor
from wakeword_jupyterlab_amd.dataset import WakewordDataset


def wav(x, sr=16000):
    raw = np.clip(np.round(np.asarray(x, np.float64) * 32767), -32768, 32767).astype("<i2").tobytes()
    return b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, sr, sr * 2, 2, 16) + b"data" + struct.pack("<I", len(raw)) + raw


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--files", type=int, default=1500)
    ap.add_argument("--batch", type=int, default=96)
    ap.add_argument("--mixed", action="store_true", help="mixed sample rates: the filter kernel and the 16 kHz kernel share batches")
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="ww_soak_files_")
    try:
        base = pkg.synth.make_clips(0, 50)
        paths = []
        for i in range(a.files):
            n = 16000 - 97 * (i % 23)                               # never longer than 1 s: no random crop, so items are deterministic
            p = os.path.join(tmp, f"f{i:05d}.wav")
            sr = (48000, 44100, 8000, 22050)[(i // 3) % 4] if a.mixed and i % 3 == 0 else 16000
            x = base[i % 50][:n] * (0.2 + 0.0005 * i)
            if sr != 16000:                                          # the same content "recorded" at another rate (<= 1 s): linear interpolation is enough here
                m = int(n * sr / 16000) - 8
                x = np.interp(np.arange(m) * (16000.0 / sr), np.arange(n), x)
            with open(p, "wb") as f:
                f.write(wav(x, sr))
            paths.append(p)
        ds = WakewordDataset(paths[: a.files // 3], paths[a.files // 3:], AudioProcessor(), verbose=False)
        ref = torch.cat([d for d, _ in ds.loader(a.batch)])          # file order
        # which file is which: a fingerprint per item (items differ in gain and length)
        key = {tuple(ref[i, 0, :2, :3].flatten().tolist()): i for i in range(a.files)}
        assert len(key) == a.files
        torch.manual_seed(0)
        ld = ds.loader(a.batch, shuffle=True)
        t0, epochs, items, bad = time.time(), 0, 0, 0
        while time.time() - t0 < a.seconds:
            seen = set()
            for data, target in ld:
                for row, lab in zip(data, target[:, 0].tolist()):
                    i = key.get(tuple(row[0, :2, :3].flatten().tolist()), -1)
                    if i < 0 or not torch.equal(row, ref[i]) or lab != ds.labels[i] or i in seen:
                        bad += 1
                    seen.add(i)
                items += data.shape[0]
            bad += len(seen) != a.files
            epochs += 1
        print(json.dumps({"what": "files, mixed rates" if a.mixed else "files", "epochs": epochs, "items": items, "mismatches": bad, "seconds": time.time() - t0, "unreadable": ds.unreadable}))
        return 1 if bad else 0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
