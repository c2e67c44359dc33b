"""The 3-conv WakewordModel's inference step at the headline batch (what the notebook's eval loop runs,
/root/reference/wakeword_training_script.py:141-184; wakeword_training.ipynb:727+): K1 -> conv stack (two launches) -> K3.
measure() is embedded in the bench line (`forward_3conv`); alone: PYTHONPATH=. python scripts/bench_forward.py"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FLOPS_CONV = 1_474_560 + 94_371_840 + 377_487_360          # conv1 + conv2 + conv3 per clip, T = 32
ALGO_BYTES = 80 * 32 * 4 + 128 * 4                         # log-mel in, pooled out: what the conv stack must move per clip


def measure(batch=4096, steps=10, device=0, pcm=None, arch="full"):
    import wakeword_jupyterlab_amd as pkg
    from wakeword_jupyterlab_amd import _native as nat
    from wakeword_jupyterlab_amd import ops
    dev = torch.device("cuda", device)
    n_conv = 3 if arch == "full" else 2
    sd = pkg.synth.make_state_dict(arch, seed=1234)
    m = (pkg.WakewordModel() if arch == "full" else pkg.SimpleWakewordModel())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    packed = m.packed_weights()
    if pcm is None:
        pcm = torch.from_numpy(pkg.synth.make_clips_tiled(0, batch, unique=256)).to(dev)
    B = pcm.shape[0]
    mel = torch.empty((B, 1, 80, 32), device=dev)
    pooled = torch.empty((B, ops.c_last(n_conv)), device=dev)
    logits = torch.empty((B, 2), device=dev)
    nbytes = nat.check(nat.lib.ww_cnn_scratch_bytes(B, n_conv))
    scratch = torch.empty(max(1, nbytes), device=dev, dtype=torch.uint8)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    stream = torch.cuda.current_stream()
    st = C.c_void_p(stream.cuda_stream)

    def step(ev=None):
        if ev: ev[0].record(stream)
        nat.check(nat.lib.ww_logmel_f32(p(pcm), B, 16000, 16000, 1, p(mel), st))
        if ev: ev[1].record(stream)
        nat.check(nat.lib.ww_cnn_pool_f32(p(mel), B, 32, p(packed), n_conv, p(scratch) if nbytes else None, p(pooled), st))
        if ev: ev[2].record(stream)
        nat.check(nat.lib.ww_lstm_fc_f32(p(pooled), B, p(packed), n_conv, p(logits), st))
        if ev: ev[3].record(stream)
    for _ in range(3):
        step()
    torch.cuda.synchronize(dev)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
    t0 = time.perf_counter()
    for k in range(steps):
        step(evs[k])
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    ms = np.array([[e[i].elapsed_time(e[i + 1]) for i in range(3)] for e in evs]).mean(axis=0)
    conv_math = ops.get_conv_math()
    peak = 2.5e15 if conv_math != "f32" else 157.3e12
    # HBM traffic of the stack from the newest committed PMC pass of this arch (cannot be read in-process)
    traffic, src = None, None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_full_pmc_traffic.json")), reverse=True):
        try:
            k = json.load(open(path))["kernels"]
            hits = [d["hbm_bytes_per_launch_corrected"] * d.get("launches_per_step", 1) for name, d in k.items() if "cnn2w_kernel" in name or "cnn3w_kernel" in name]
            if hits and B == 4096:
                traffic, src = float(sum(hits)), os.path.basename(path)
                break
        except Exception:
            continue
    return {"workload": f"3-conv WakewordModel (wakeword_training_script.py:141-184) inference step, batch {B}: log-mel (K1) -> conv1+conv2 "
                        f"(cnn2w_kernel<false>) -> conv3+pool (cnn3w_kernel) -> LSTM+fc (K3), PCM resident in HBM, random-init weights seed 1234",
            "conv_math": conv_math, "subbatch_clips": int(os.environ.get("WW_CNN3_SUBBATCH", "0") or 0),
            "ms_per_step": dt * 1e3, "clips_per_s": B / dt,
            "stages_ms": {"K1_logmel": float(ms[0]), "K2_conv_stack": float(ms[1]), "K3_lstm_fc": float(ms[2])},
            "roofline": {"kernel": "cnn2w_kernel<false> + cnn3w_kernel (both convs as 1-D Winograd F(2,3), split precision, float32 intermediate)",
                         "bound": "mfma", "achieved": FLOPS_CONV * B / (ms[1] * 1e-3) / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
                         "frac": FLOPS_CONV * B / (ms[1] * 1e-3) / peak, "flops_per_launch_pair": FLOPS_CONV * B,
                         "traffic": traffic, "traffic_source": src, "algorithmic_bytes": ALGO_BYTES * B},
            "finite_logits": bool(torch.isfinite(logits).all())}


if __name__ == "__main__":
    print(json.dumps(measure()))
