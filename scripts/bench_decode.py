#!/usr/bin/env python3
"""K0 alone: WAV sample bytes already in HBM -> normalised 1 s clips [B, 16000] (decode, mono mix, polyphase
resample, whole-file peak normalise, crop / zero-pad).  PYTHONPATH=. python scripts/bench_decode.py"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import wakeword_jupyterlab_amd as pkg  # noqa: E402
from wakeword_jupyterlab_amd import _native as nat  # noqa: E402


def run(B, sr, channels, seconds, steps=10):
    dev = torch.device("cuda", 0)
    frames = int(sr * seconds)
    base = (pkg.synth.make_clip(3)[: min(16000, frames)] * 20000).astype("<i2")
    one = np.resize(base, frames * channels).astype("<i2").tobytes()
    one += b"\0" * (-len(one) % 16)
    raw = torch.frombuffer(bytearray(one * B), dtype=torch.uint8).to(dev)
    proto = nat.ClipDesc()
    nat.check(nat.lib.ww_resampler_prepare(sr, C.byref(proto)))
    descs = (nat.ClipDesc * B)()
    for i, d in enumerate(descs):
        d.byte_offset, d.n_frames, d.channels, d.sample_rate, d.format = i * len(one), frames, channels, sr, nat.FMT_S16
        d.up, d.down, d.half_len, d.taps_dev, d.crop_start = proto.up, proto.down, proto.half_len, proto.taps_dev, 0
    desc_t = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
    out = torch.empty((B, 16000), device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: nat.check(nat.lib.ww_decode_resample(C.c_void_p(raw.data_ptr()), C.c_void_p(desc_t.data_ptr()), B, 1,  # noqa: E731
                                                        C.c_void_p(out.data_ptr()), st))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"sample_rate": sr, "channels": channels, "seconds": seconds, "clips": B, "ms": dt * 1e3, "clips_per_s": B / dt,
            "input_GBps": len(one) * B / dt / 1e9, "finite": bool(torch.isfinite(out).all())}


if __name__ == "__main__":
    res = [run(4096, 16000, 1, 1.0), run(4096, 48000, 1, 1.0), run(4096, 44100, 2, 1.5), run(2048, 8000, 1, 1.0)]
    print(json.dumps(res))
