#!/usr/bin/env python3
"""PCIe-inclusive rates (NOT the headline value, which is measured with inputs resident in HBM):
  (a) float32 PCM in pinned host memory -> H2D copy -> K1 -> K2 -> K3
  (b) int16 PCM in pinned host memory   -> H2D copy -> K0 (decode/normalise) -> K1 -> K2 -> K3   (half the bytes)
Copy and compute overlap across batches on two streams (double buffering)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import _native as nat, ops

dev = torch.device("cuda", 0)
B, iters = 4096, 12
sd = pkg.synth.make_state_dict("simple", seed=1234)
m = pkg.SimpleWakewordModel(); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.to(dev).eval()
clips = pkg.synth.make_clips_tiled(0, B, unique=64)
f32_host = torch.from_numpy(clips).pin_memory()
i16_host = torch.from_numpy(np.clip(np.round(clips / np.abs(clips).max() * 32767), -32768, 32767).astype(np.int16)).pin_memory()
proto = nat.ClipDesc(); nat.check(nat.lib.ww_resampler_prepare(16000, C.byref(proto)))
descs = (nat.ClipDesc * B)()
for i in range(B):
    d = descs[i]; d.byte_offset, d.n_frames, d.channels, d.sample_rate, d.format, d.crop_start = i * 32000, 16000, 1, 16000, nat.FMT_S16, 0
    d.up, d.down, d.half_len, d.taps_dev = 1, 1, 0, None
desc_dev = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
copy_s, comp_s = torch.cuda.Stream(), torch.cuda.Stream()

def run(kind):
    bufs = [torch.empty((B, 16000), device=dev, dtype=torch.float32 if kind == "f32" else torch.int16) for _ in range(2)]
    pcm = torch.empty((B, 16000), device=dev)
    evs = [torch.cuda.Event() for _ in range(2)]; done = [torch.cuda.Event() for _ in range(2)]
    src = f32_host if kind == "f32" else i16_host
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(iters):
        b = it & 1
        with torch.cuda.stream(copy_s):
            copy_s.wait_event(done[b])
            bufs[b].copy_(src, non_blocking=True); evs[b].record(copy_s)
        with torch.cuda.stream(comp_s), torch.no_grad():
            comp_s.wait_event(evs[b])
            if kind == "f32":
                m.forward_pcm(bufs[b])
            else:
                nat.check(nat.lib.ww_decode_resample(C.c_void_p(bufs[b].data_ptr()), C.c_void_p(desc_dev.data_ptr()), B, 1,
                                                     C.c_void_p(pcm.data_ptr()), C.c_void_p(comp_s.cuda_stream)))
                m.forward_pcm(pcm, normalize=False)
            done[b].record(comp_s)
    torch.cuda.synchronize()
    return B * iters / (time.perf_counter() - t0)

run("f32"); run("i16")
out = {"clips_per_s_f32_over_pcie": run("f32"), "clips_per_s_int16_over_pcie_with_K0": run("i16"), "batch": B,
       "bytes_per_clip": {"f32": 64000, "int16": 32000}}
print(json.dumps(out))
