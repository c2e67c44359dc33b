"""First contact with the 1-D Winograd conv2 kernel: agreement with the direct split-precision kernel and the f64 oracle on small
batches (bounded: a protocol bug shows up as NaN + ww_sync_timeouts, not as a hang), then timing at 4096 clips."""
import sys, time
import numpy as np, torch
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import ops, _native as nat
from oracle import model_oracle
dev = torch.device("cuda", 0)
sd = pkg.synth.make_state_dict("simple", seed=7)
packed = torch.from_numpy(ops.pack_state_dict(sd)).to(dev)
for batch, width in [(1, 32), (3, 32), (5, 17), (300, 31), (2, 1), (777, 32)]:
    x = (pkg.synth.normal(11, batch * 80 * width).astype(np.float32).reshape(batch, 1, 80, width) * 15 - 35)
    xt = torch.from_numpy(x).to(dev)
    res = {}
    for mode in ("f16x3", "f16x3d"):
        ops.set_conv_math(mode)
        res[mode] = ops.cnn_pool(xt, packed, 2).cpu().numpy()
        torch.cuda.synchronize()
    ref = model_oracle.pooled_features_np(x[:16], sd)
    print(batch, width, "wino vs direct %.3e" % np.abs(res["f16x3"] - res["f16x3d"]).max(), "wino vs f64 %.3e" % np.abs(res["f16x3"][:16] - ref).max(),
          "direct vs f64 %.3e" % np.abs(res["f16x3d"][:16] - ref).max(), "timeouts", nat.lib.ww_sync_timeouts(), flush=True)
    if nat.lib.ww_sync_timeouts() or not np.isfinite(res["f16x3"]).all():
        sys.exit("protocol problem")
B = 4096
mel = torch.from_numpy(np.tile((pkg.synth.normal(3, 64 * 2560).astype(np.float32).reshape(64, 1, 80, 32) * 15 - 35), (64, 1, 1, 1))).to(dev)
for mode in ("f16x3d", "f16x3", "f16x3d", "f16x3"):
    ops.set_conv_math(mode)
    for _ in range(3): ops.cnn_pool(mel, packed, 2)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): ops.cnn_pool(mel, packed, 2)
    torch.cuda.synchronize(); print(mode, "%.4f ms" % ((time.perf_counter() - t) * 50), flush=True)
a = ops.cnn_pool(mel, packed, 2); b = ops.cnn_pool(mel, packed, 2)
print("repeatable", torch.equal(a, b), "timeouts", nat.lib.ww_sync_timeouts())
