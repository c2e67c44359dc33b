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
    for _ in range(2):
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
            "cpu_torch_clips_per_s": cpu_sample / dt_cpu, "cpu_threads": int(torch.get_num_threads()), "cpu_sample": cpu_sample}


def ops_reset_train_math():
    from wakeword_jupyterlab_amd import ops
    ops.set_train_math("f16x3")


if __name__ == "__main__":
    print(measure(arch=sys.argv[1] if len(sys.argv) > 1 else "simple", batch=int(sys.argv[2]) if len(sys.argv) > 2 else 4096,
                  math=sys.argv[3] if len(sys.argv) > 3 else None))
