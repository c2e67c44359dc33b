"""Phase stamps of conv2_dgrad_h_kernel (diagnostic build with -DWW_DG_STAMPS, loaded through WW_LIB_OVERRIDE): shader cycles per step and wave of
workgroup 7 in each phase -- 0 rebuild + issue of the next rows' loads, 1 the MFMA loop, 2 the dW1 / db1 block, 3 the next rows into LDS, 4 the barrier.
usage: WW_LIB_OVERRIDE=.../lib_dgstamps.so PYTHONPATH=. python scripts/dg_stamps.py"""
import ctypes as C, json, os
import numpy as np, torch
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import _native as nat

dev = torch.device("cuda", 0)
m = pkg.SimpleWakewordModel()
m.load_state_dict({k: torch.from_numpy(v) for k, v in pkg.synth.make_state_dict("simple", seed=1234).items()})
m = m.to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=1e-3); crit = torch.nn.CrossEntropyLoss()
x = (torch.randn(4096, 1, 80, 32, device=dev) * 15 - 35).clamp_(-80, 0); y = torch.randint(0, 2, (4096,), device=dev)
def step():
    opt.zero_grad(); loss = crit(m(x), y); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
z = (C.c_ulonglong * 40)()
assert nat.lib.ww_debug_dg_stamps(z) == 0
N = 20
for _ in range(N): step()
torch.cuda.synchronize()
assert nat.lib.ww_debug_dg_stamps(z) == 0
v = np.array(list(z), dtype=np.float64).reshape(8, 5) / N / (16 * 20)          # per step: 16 clips x 20 steps per workgroup and launch
print(json.dumps({"cycles_per_step_by_wave_and_phase": np.round(v, 1).tolist(), "phases": ["top", "mfma", "dW1", "fill", "barrier"],
                  "per_wave_total": np.round(v.sum(1), 1).tolist(), "mean": np.round(v.mean(0), 1).tolist()}))
