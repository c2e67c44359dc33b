"""Training step time vs the number of timed steps (the bench line times 5 after 2 warm-up steps)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import wakeword_jupyterlab_amd as pkg
dev = torch.device("cuda", 0)
sd = pkg.synth.make_state_dict("simple", seed=1234)
m = pkg.SimpleWakewordModel(); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=1e-3); crit = torch.nn.CrossEntropyLoss()
x = (torch.randn(4096, 1, 80, 32, device=dev) * 15 - 35).clamp_(-80, 0); y = torch.randint(0, 2, (4096,), device=dev)
def step():
    opt.zero_grad(); loss = crit(m(x), y); loss.backward(); opt.step(); return loss
for warm, steps in ((2, 5), (0, 5), (0, 20), (0, 50), (0, 5)):
    for _ in range(warm): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = 0.0
    for _ in range(steps):
        a = time.perf_counter(); step(); th += time.perf_counter() - a
    host = th / steps
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"warm {warm} steps {steps}: {dt*1e3:.3f} ms per step (host enqueue {host*1e3:.3f} ms per step)", flush=True)
