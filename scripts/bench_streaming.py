#!/usr/bin/env python3
"""BASELINE configs[4]: streaming 10 ms-hop sliding-window inference, 256 concurrent microphones, one hipGraph replay
per hop on 1 x MI355X.  Reports per-hop latency (host enqueue -> results ready) p50/p99 and sustained hops/s.

    PYTHONPATH=. python scripts/bench_streaming.py [--mics 256] [--hop 160] [--hops 1000]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import wakeword_jupyterlab_amd as pkg  # noqa: E402


def measure(mics=256, hop=160, hops=1000, device=0):
    """Run the streaming config and return its result dict (bench.py embeds it in its JSON line at N=1)."""
    args = argparse.Namespace(mics=mics, hop=hop, hops=hops)
    dev = torch.device("cuda", device)
    sd = pkg.synth.make_state_dict("simple", seed=1234)
    m = pkg.SimpleWakewordModel()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    det = pkg.StreamingDetector(m, n_mics=args.mics, hop_samples=args.hop)
    # pre-fill the rings with one second of audio, as the config asks
    audio = torch.from_numpy(pkg.synth.make_clips_tiled(0, args.mics, unique=min(64, args.mics))).to(dev)
    for k in range(16000 // args.hop):
        det.step(audio[:, k * args.hop:(k + 1) * args.hop])
    det.stream.synchronize()
    hops = [audio[:, (k % (16000 // args.hop)) * args.hop:(k % (16000 // args.hop) + 1) * args.hop].contiguous() for k in range(64)]
    torch.cuda.synchronize()

    # (a) latency: one hop at a time, wait for the result
    lat = np.empty(args.hops)
    for k in range(args.hops):
        t0 = time.perf_counter()
        det.step(hops[k % 64])
        det.stream.synchronize()
        lat[k] = time.perf_counter() - t0
    # (b) throughput: enqueue back to back, synchronise once
    t0 = time.perf_counter()
    for k in range(args.hops):
        det.step(hops[k % 64])
    det.stream.synchronize()
    thr = args.hops / (time.perf_counter() - t0)
    # (c) device time of one replay
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    with torch.cuda.stream(det.stream):
        ev[0].record()
        for k in range(100):
            det.step()
        ev[1].record()
    det.stream.synchronize()
    dev_us = ev[0].elapsed_time(ev[1]) * 10.0
    out = {"workload": f"streaming, {args.mics} mics, hop {args.hop} samples ({1000 * args.hop / 16000:.1f} ms), window 1 s, hipGraph replay per hop",
           "hops": args.hops, "latency_us_p50": float(np.percentile(lat, 50) * 1e6), "latency_us_p99": float(np.percentile(lat, 99) * 1e6),
           "latency_us_max": float(lat.max() * 1e6), "hops_per_s_back_to_back": thr, "windows_per_s": thr * args.mics,
           "device_us_per_replay": dev_us, "realtime_factor": thr * args.hop / 16000.0,
           "finite_probs": int(torch.isfinite(det.prob).sum().item())}
    det.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mics", type=int, default=256)
    ap.add_argument("--hop", type=int, default=160)
    ap.add_argument("--hops", type=int, default=1000)
    args = ap.parse_args()
    print(json.dumps(measure(args.mics, args.hop, args.hops)))


if __name__ == "__main__":
    main()
