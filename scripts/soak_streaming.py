"""Two StreamingDetectors fed the same hop sequence must produce bitwise identical probabilities at every hop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import wakeword_jupyterlab_amd as pkg
dev = torch.device("cuda", 0)
sd = pkg.synth.make_state_dict("simple", seed=1234)
m = pkg.SimpleWakewordModel(); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.to(dev).eval()
mics, hop = 256, 160
audio = torch.from_numpy(pkg.synth.make_clips_tiled(0, mics, unique=64)).to(dev)
hops = [audio[:, (k % 100) * hop:(k % 100 + 1) * hop].contiguous() for k in range(100)]
seqs = []
for rep in range(2):
    det = pkg.StreamingDetector(m, n_mics=mics, hop_samples=hop)
    out = []
    for k in range(3000):
        det.step(hops[k % 100]); det.stream.synchronize(); out.append(det.prob.clone())
    seqs.append(torch.stack(out)); det.close()
same = torch.equal(seqs[0], seqs[1])
fin = torch.isfinite(seqs[0]).all().item()
print("streaming: 3000 hops x 2 detectors identical:", same, "finite:", fin)
sys.exit(0 if same else 1)
