"""debug: conv1 gradients of the split-precision training kernels against the exact-fp32 ones, element by element"""
import sys, os
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import ops
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 70
sd = pkg.synth.make_state_dict("simple", seed=21)
if len(sys.argv) > 2:      # swap conv1 channels 0 and 5
    for k in ("conv1.weight", "conv1.bias"):
        v = sd[k].copy(); v[[0, 5]] = v[[5, 0]]; sd[k] = v
    v = sd["conv2.weight"].copy(); v[:, [0, 5]] = v[:, [5, 0]]; sd["conv2.weight"] = v
x = (pkg.synth.normal(5, B * 80 * 32).astype(np.float32).reshape(B, 1, 80, 32) * 15 - 35)
labels = torch.from_numpy((np.arange(B) % 2).astype(np.int64)).to(dev)
g = {}
for math in ("f32", "f16x3"):
    ops.set_train_math(math)
    m = pkg.SimpleWakewordModel(); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.to(dev).train()
    m.lstm.dropout = 0.0; m.dropout.p = 0.0
    F.cross_entropy(m(torch.from_numpy(x).to(dev)), labels).backward()
    g[math] = {k: p.grad.detach().cpu().numpy().astype(np.float64) for k, p in m.named_parameters()}
for k in ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias"):
    a, b = g["f32"][k], g["f16x3"][k]
    d = (b - a) / np.abs(a).max()
    print(k, "max rel diff %.2e" % np.abs(d).max())
d = ((g["f16x3"]["conv1.weight"] - g["f32"]["conv1.weight"]) / np.abs(g["f32"]["conv1.weight"]).max()).reshape(32, 9)
np.set_printoptions(linewidth=200, precision=1)
print((d * 1e6)[:8])
print("bias diff x1e6:", ((g["f16x3"]["conv1.bias"] - g["f32"]["conv1.bias"]) / np.abs(g["f32"]["conv1.bias"]).max()) * 1e6)
z = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(sd["conv1.weight"]).double(), torch.from_numpy(sd["conv1.bias"]).double(), padding=1)
for c in (0, 1, 2, 16):
    zc = z[:, c].abs()
    print("channel", c, "mean|z| %.3g" % float(zc.mean()), "positions with |z| < 1e-3: %d, < 1e-4: %d, < 1e-5: %d" % (int((zc < 1e-3).sum()), int((zc < 1e-4).sum()), int((zc < 1e-5).sum())), "frac positive %.3f" % float((z[:, c] > 0).double().mean()))
P = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in sd.items()}
a = torch.tensor(x, dtype=torch.float64)
for li in (1, 2):
    a = F.relu(F.conv2d(a, P[f"conv{li}.weight"], P[f"conv{li}.bias"], padding=1))
h = a.mean(dim=(2, 3))
for layer in (0, 1):
    gg = h @ P[f"lstm.weight_ih_l{layer}"].T + P[f"lstm.bias_ih_l{layer}"] + P[f"lstm.bias_hh_l{layer}"]
    c = torch.sigmoid(gg[:, :256]) * torch.tanh(gg[:, 512:768]); h = torch.sigmoid(gg[:, 768:]) * torch.tanh(c)
F.cross_entropy(h @ P["fc.weight"].T + P["fc.bias"], labels.cpu()).backward()
ex = P["conv1.weight"].grad.numpy().reshape(32, 9)
np.set_printoptions(linewidth=220, precision=4)
c = int(np.argmax(np.abs(g["f16x3"]["conv1.weight"].reshape(32, 9) - g["f32"]["conv1.weight"].reshape(32, 9)).max(axis=1)))
print("worst channel", c, "max|grad| over tensor %.4e" % np.abs(ex).max())
print("f64  ", ex[c]); print("f32  ", g["f32"]["conv1.weight"].reshape(32, 9)[c]); print("split", g["f16x3"]["conv1.weight"].reshape(32, 9)[c])
print("per-channel max|grad|:", np.abs(ex).max(axis=1))
