"""K1 / K2 time per step over the first steps of a process (what the driver's 5 warm-up + 20 timed steps see) and after 2 s of load.
usage: PYTHONPATH=. python scripts/proto/cold_ramp.py"""
import ctypes as C, json, time
import numpy as np, torch
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import _native as nat, ops

dev = torch.device("cuda", 0)
B = 4096
pcm = torch.from_numpy(pkg.synth.make_clips_tiled(0, B, unique=64)).to(dev)
m = pkg.SimpleWakewordModel()
m.load_state_dict({k: torch.from_numpy(v) for k, v in pkg.synth.make_state_dict("simple", seed=1234).items()})
m = m.to(dev).eval()
mel = torch.empty(B, 80, 32, device=dev)
st = torch.cuda.current_stream()
def one(ev):
    ev[0].record(st)
    mel_ = ops.logmel(pcm, True)
    ev[1].record(st)
    with torch.no_grad():
        m(mel_ if mel_.dim() == 4 else mel_.unsqueeze(1))
    ev[2].record(st)
def run(n):
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
    for e in evs: one(e)
    torch.cuda.synchronize()
    return [(round(e[0].elapsed_time(e[1]), 4), round(e[1].elapsed_time(e[2]), 4)) for e in evs]
first = run(60)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 2.0:
    run(50)
late = run(20)
k = lambda rows, a, b: [round(float(np.mean([r[i] for r in rows[a:b]])), 4) for i in (0, 1)]
print(json.dumps({"what": "ms per step (K1, K2+K3) by step index after process start, HIP events", "steps_0_4": k(first, 0, 5), "steps_5_24": k(first, 5, 25),
                  "steps_25_59": k(first, 25, 60), "after_2s_of_load": k(late, 0, 20), "first_60": first}))
