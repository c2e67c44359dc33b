"""Temporary: dump the split-precision training step's bit images / gradients for the failing case (run under WW_LIB_OVERRIDE twice)."""
import os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import ops
out = sys.argv[1]
dev = torch.device("cuda", 0)
ops.keep_train_workspace(True)
sd = pkg.synth.make_state_dict("simple", seed=21)
x = (pkg.synth.normal(5, 70 * 80 * 32).astype(np.float32).reshape(70, 1, 80, 32) * 15 - 35)
labels = torch.from_numpy((np.arange(70) % 2).astype(np.int64)).to(dev)
res = {}
for math in ("f32", "f16x3"):
    ops.set_train_math(math)
    m = pkg.SimpleWakewordModel()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).train()
    m.lstm.dropout = 0.0; m.dropout.p = 0.0
    logits = m(torch.from_numpy(x).to(dev))
    if math == "f16x3":
        mask, sign1 = ops.train_last_bit_images()
        res["mask"], res["sign1"] = mask.cpu().numpy(), sign1.cpu().numpy()
    loss = F.cross_entropy(logits, labels)
    loss.backward()
    res["logits_" + math] = logits.detach().cpu().numpy()
    for k, p in m.named_parameters():
        res[f"g_{math}_{k}"] = p.grad.detach().cpu().numpy()
with torch.no_grad():
    packed = torch.from_numpy(ops.pack_state_dict(sd)).to(dev)
    for cm in ("f16x3", "f32"):
        ops.set_conv_math(cm)
        res["pooled_" + cm] = ops.cnn_pool(torch.from_numpy(x).to(dev), packed, 2).cpu().numpy()
np.savez(out, **res)
print("saved", out)
