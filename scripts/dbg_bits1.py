"""debug: the conv1 sign bits the split-precision training forward leaves in its workspace, against torch"""
import sys, os
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import ops, _native as nat
dev = torch.device("cuda", 0)
n = 6
sd = pkg.synth.make_state_dict("simple", seed=21)
x = (pkg.synth.normal(5, n * 80 * 32).astype(np.float32).reshape(n, 1, 80, 32) * 15 - 35)
ops.set_train_math("f16x3")
m = pkg.SimpleWakewordModel(); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.to(dev).train()
m(torch.from_numpy(x).to(dev))
ws = ops._TrainStep.last_workspace.cpu().numpy().view(np.uint32)
def a256(f): return (f * 4 + 255) // 256 * 64
o = 0
def take(f):
    global o
    at = o; o += a256(f); return at
take(n * 64)                     # (relu(conv2) is not stored by the split-precision forward of the 2-conv model)
for _ in range(2): take(4 * n * 256); take(n * 256); take(n * 256)
take((64 + 256) * 768 + 2 * 768); take(2 * 144 * 64); take(2 * 144 * 64)
take(n * 256); take(n * 1024); take(n * 256); take(n * 1024); take(n * 64); take(n * 64)
kp = 64 * 32 * 9 + 64
take(256 * kp)
mb = take(n * 80 * 32 * 2); take(int(nat.lib.ww_packed_weights_floats(2))); b1 = take(n * 80 * 32)
bits1 = ws[b1:b1 + n * 80 * 32].reshape(n, 80, 32)
# conv2's mask image = the accumulator ballots: [clip][40 tile rows][4 N-tiles][2 column halves][2 rows][4 j] x 64 bits,
# bit 16 kq + pi <-> column 16 c + 4 kq + j, channel 16 nt + pi  ->  words [clip][row][col][2]
raw = ws[mb:mb + n * 80 * 32 * 2].view(np.uint8).reshape(n, 40, 4, 2, 2, 4, 4, 2)       # [.., t, nt, c, r, j, kq, half] bytes
mask2 = np.zeros((n, 80, 32, 8), np.uint8)
for t in range(40):
    for nt in range(4):
        for c in range(2):
            for r in range(2):
                for j in range(4):
                    for kq in range(4):
                        mask2[:, 2 * t + r, 16 * c + 4 * kq + j, 2 * nt:2 * nt + 2] = raw[:, t, nt, c, r, j, kq]
mask2 = np.ascontiguousarray(mask2).view(np.uint32).reshape(n, 80, 32, 2)
X = torch.from_numpy(x).double()
z1 = F.conv2d(X, torch.from_numpy(sd["conv1.weight"]).double(), torch.from_numpy(sd["conv1.bias"]).double(), padding=1)
z2 = F.conv2d(F.relu(z1), torch.from_numpy(sd["conv2.weight"]).double(), torch.from_numpy(sd["conv2.bias"]).double(), padding=1)
want1 = np.zeros((n, 80, 32), np.uint32)
for c in range(32): want1 |= ((z1[:, c] > 0).numpy().astype(np.uint32) << np.uint32(c))
diff = bits1 ^ want1
print("conv1 sign words differing:", int((diff != 0).sum()), "of", diff.size)
for c in range(32):
    k = int(((diff >> np.uint32(c)) & 1).sum())
    if k: print("  channel", c, "bits differing", k, "near-zero |z| there:", float(z1[:, c].abs().numpy()[((diff >> np.uint32(c)) & 1) == 1].max()))
want2 = np.zeros((n, 80, 32, 2), np.uint32)
for c in range(64): want2[..., c // 32] |= ((z2[:, c] > 0).numpy().astype(np.uint32) << np.uint32(c % 32))
d2 = mask2 ^ want2
print("conv2 mask words differing:", int((d2 != 0).sum()), "of", d2.size)
