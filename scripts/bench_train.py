"""one training step (train-mode forward + CrossEntropyLoss + backward) of SimpleWakewordModel on the HIP kernels.
measure() is embedded in the bench line; run alone for a per-kernel split: python scripts/bench_train.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(batch=4096, steps=5, device=0, cpu_sample=64, arch="simple", math=None):
    import wakeword_jupyterlab_amd as pkg
    from oracle import model_oracle
    from wakeword_jupyterlab_amd import ops
    if math:
        ops.set_train_math(math)
    math = ops.get_train_math()
    dev = torch.device("cuda", device)
    sd = pkg.synth.make_state_dict(arch, seed=1234)
    m = pkg.SimpleWakewordModel() if arch == "simple" else pkg.WakewordModel()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    crit = torch.nn.CrossEntropyLoss()
    x = (torch.randn(batch, 1, 80, 32, device=dev) * 15 - 35).clamp_(-80, 0)
    y = torch.randint(0, 2, (batch,), device=dev)

    def step():
        opt.zero_grad()
        loss = crit(m(x), y)
        loss.backward()
        opt.step()
        return loss
    for _ in range(5):           # allocator, clocks, lazily created optimizer state: the third to fifth step still run 5-8 % slow (scripts/proto/train_steps.py)
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    # the same step in torch on the host CPU, on a small sample (context only)
    ref = model_oracle.torch_module_from_state_dict(sd).train()
    opt_r = torch.optim.Adam(ref.parameters(), lr=1e-3)
    xc, yc = x[:cpu_sample].cpu(), y[:cpu_sample].cpu()
    t1 = time.perf_counter()
    for _ in range(2):
        opt_r.zero_grad()
        crit(ref(xc), yc).backward()
        opt_r.step()
    dt_cpu = (time.perf_counter() - t1) / 2
    return {"workload": f"{'SimpleWakewordModel' if arch == 'simple' else '3-conv WakewordModel'} training step (train-mode forward with dropout + CrossEntropyLoss + backward + Adam), "
                        f"batch {batch}, log-mel inputs resident in HBM; " +
                        ("conv stack in split precision on the f16 matrix cores, conv1 weight gradient on the f64 matrix cores, head exact fp32 "
                         "(csrc/ww_train_h.hip, ww_train.hip)" if math == "f16x3" else "exact fp32 kernels (csrc/ww_train.hip)"),
            "train_math": math,
            "ms_per_step": dt * 1e3, "clips_per_s": batch / dt, "final_loss": float(loss.item()),
            # forward + data gradient + weight gradient = 3 x the forward's algorithmic flops (96.5 / 474 MFLOP per clip); the
            # step also holds the optimiser and torch's loss, so this is a floor for the kernels' own rate
            "algorithmic_TFLOPs": 3 * (96.5e6 if arch == "simple" else 474.0e6) * batch / dt / 1e12,
            "cpu_torch_clips_per_s": cpu_sample / dt_cpu, "cpu_threads": int(torch.get_num_threads()), "cpu_sample": cpu_sample}


def measure_pipeline(batch=4096, steps=4, device=0, arch="simple"):
    """The training data path and the step together, as WakewordDataset(augment=True) -> model.train() chain them
    (wakeword_training_script.py:103-138, 241-267): PCM resident in HBM -> augment_audio (plans drawn with p = 0.8 per transform on the
    host, as the reference does) -> normalise + log-mel -> forward + CrossEntropyLoss + backward + Adam."""
    import random
    import wakeword_jupyterlab_amd as pkg
    from wakeword_jupyterlab_amd import _native as nat
    from wakeword_jupyterlab_amd import ops
    dev = torch.device("cuda", device)
    sd = pkg.synth.make_state_dict(arch, seed=1234)
    m = pkg.SimpleWakewordModel() if arch == "simple" else pkg.WakewordModel()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    crit = torch.nn.CrossEntropyLoss()
    x = pkg.synth.make_clips_tiled(0, batch, unique=64)
    pcm = torch.from_numpy(x / np.abs(x).max(axis=1, keepdims=True)).float().to(dev)
    y = torch.randint(0, 2, (batch,), device=dev)
    random.seed(0)
    proc = pkg.AudioProcessor()

    def draw():
        arr = (nat.AugmentPlan * batch)()
        for a in arr:
            p = proc.draw_augment_plan()                          # the product's own draws (reference order, python `random`)
            a.shift, a.crop_start = p["shift"], p["crop"]
            a.pitch_rate = 2.0 ** (-p["n_steps"] / 12.0) if p["n_steps"] is not None else 0.0
            a.stretch_rate = p["rate"] or 0.0
            a.noise_sigma, a.noise_seed = p["sigma"], p["seed"]
        return arr
    plans = [draw() for _ in range(steps + 1)]        # the host-side draws are not timed: python `random`, ~10 us per clip, overlappable

    def step(arr):
        mel = ops.logmel(ops.augment(pcm, arr), True)
        opt.zero_grad()
        loss = crit(m(mel), y)
        loss.backward()
        opt.step()
        return loss
    step(plans[0])
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(plans[i + 1])
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    return {"workload": f"training data path + step, {'2' if arch == 'simple' else '3'}-conv model, batch {batch}: PCM in HBM -> augment_audio (p = 0.8 per transform) -> "
                        "log-mel -> train-mode forward + CrossEntropyLoss + backward + Adam; new plans every step (drawn on the host, not timed)",
            "ms_per_batch": dt * 1e3, "clips_per_s": batch / dt, "final_loss": float(loss.item()), "train_math": ops.get_train_math()}


def ops_reset_train_math():
    from wakeword_jupyterlab_amd import ops
    ops.set_train_math("f16x3")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "pipeline":
        print(measure_pipeline(arch=sys.argv[2] if len(sys.argv) > 2 else "simple", batch=int(sys.argv[3]) if len(sys.argv) > 3 else 4096))
    else:
        print(measure(arch=sys.argv[1] if len(sys.argv) > 1 else "simple", batch=int(sys.argv[2]) if len(sys.argv) > 2 else 4096,
                      math=sys.argv[3] if len(sys.argv) > 3 else None))
