#!/usr/bin/env python3
"""sha256 of K0's output for files at several rates / formats (run under two builds, WW_LIB_OVERRIDE, to see whether a kernel change is bit-neutral)
and the decode leg's times.    PYTHONPATH=. python scripts/k0_hash.py"""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import wakeword_jupyterlab_amd as pkg  # noqa: E402
import bench_files  # noqa: E402

import tempfile  # noqa: E402
import wave  # noqa: E402

h = hashlib.sha256()
rng = np.random.default_rng(5)
proc = pkg.AudioProcessor()
with tempfile.TemporaryDirectory() as tmp:
    paths = []
    for k, (sr, ch, secs) in enumerate(((48000, 1, 1.0), (44100, 2, 1.3), (22050, 1, 0.7), (8000, 1, 1.0), (32000, 2, 2.5), (96000, 1, 1.0),
                                       (24000, 1, 0.2), (16000, 1, 1.0), (11025, 1, 1.0), (16000, 1, 5923 / 16000), (16000, 1, 3 / 16000), (16000, 2, 1.0))):
        n = int(sr * secs)
        x = (rng.standard_normal((n, ch)) * 6000).astype("<i2")
        for rep in range(3):
            path = os.path.join(tmp, f"f{k}_{rep}.wav")
            with wave.open(path, "wb") as w:
                w.setnchannels(ch); w.setsampwidth(2); w.setframerate(sr); w.writeframes(x.tobytes())
            paths.append(path)
    import random
    random.seed(3)
    out, ok = proc.load_clips_gpu(paths)
    assert bool(ok.all())
    h.update(out.cpu().numpy().tobytes())
print(h.hexdigest(), os.environ.get("WW_LIB_OVERRIDE", "shipped"))
print(json.dumps({k: (round(v["ms"], 4) if isinstance(v, dict) else v) for k, v in bench_files.measure_decode(steps=5).items()}))
